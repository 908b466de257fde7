#!/usr/bin/env python3
"""
bench.py — frames/sec of the DP3 hot path (640x480 depth+RGB → backproject → crop → 1024-pt
FPS → colour gather) on N MI355X, one process per GPU.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one pass of the hot path over ONE synthetic 300-frame episode that is already
resident in HBM (BASELINE.json configs[1]).  Every rank owns its own episode (weak scaling,
no data-path collective: frames are independent); the only collectives are the
timing barrier and the max-over-ranks reduction.  Rank 0 prints ONE JSON line.

The FPS distances are computed the way the reference computes them: DP3:47 samples with
use_cuda=True, i.e. pytorch3d's CUDA kernel with nvcc's fused multiply-adds (dist_mode 1).  The
un-fused mode (pytorch3d's CPU build) is timed after the headline and reported in `config`.

After the timed region the outputs of the last step are checked against the CPU oracle on a few
frames (the checker is never inside the timed region); a mismatch makes the run fail.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "point-cloud-processing_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np

H, W, K_FPS, FRAMES = 480, 640, 1024, 300
BYTES_PER_FRAME = H * W * (4 + 3) + K_FPS * (12 + 12 + 8)      # 2,183,168 B
HBM_PEAK_GBS = 8000.0                                          # MI355X_MICROARCH.md: 8 TB/s spec
MODE_NAME = {0: "nofma", 1: "fma"}


def measured_traffic():
    """HBM-side bytes per FPS launch from the newest committed rocprofv3 PMC summary
    (profiles/rNN_*_traffic.json: separate FETCH_SIZE / WRITE_SIZE passes, gfx950 corrections)."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json")))
    if not files:
        return None, None
    with open(files[-1]) as fh:
        t = json.load(fh)
    return t.get("hbm_bytes_per_launch"), os.path.basename(files[-1])


# ---------------------------------------------------------------------------------------- CPU side
def _host_cores():
    try:
        return len(os.sched_getaffinity(0))
    except AttributeError:
        return os.cpu_count() or 1


def _oracle_frames(seed0, budget_s, max_frames, scene, dist_mode):
    """C oracle over frames seed0, seed0+1, ... for about budget_s seconds → (frames, seconds)."""
    from oracle import cbind
    from pcp import synth
    intr = synth.intrinsics(H, W)
    po = cbind.make_params(intr, intr, apply_crop=1)
    make = synth.frame if scene == "random" else synth.scene_frame
    cbind.frame_pipeline(*make(10_000, H, W), po, K_FPS, dist_mode)          # warm (page in, build)
    n, t_total = 0, 0.0
    while n < max_frames and t_total < budget_s:
        d, c = make(seed0 + n, H, W)
        t0 = time.perf_counter()
        cbind.frame_pipeline(d, c, po, K_FPS, dist_mode)
        t_total += time.perf_counter() - t0
        n += 1
    return n, t_total


def cpu_worker(seed0, budget_s, scene, dist_mode):
    """Child-process entry (`bench.py --cpu-worker ...`): prints "<frames> <seconds>"."""
    n, t = _oracle_frames(seed0, budget_s, 10_000, scene, dist_mode)
    print(n, t, flush=True)


def cpu_baseline(scene="random", dist_mode=1, budget_s=10.0, numpy_s=3.0, multi_s=4.0):
    """The CPU side of the comparison on this box's host cores, all on bounded samples of the same episode:
      value                  the C oracle ("port" of the reference's per-frame path, -O2 -ffp-contract=off), ONE
                             thread like the reference's serial loops (XPC:96-125, DP3:68-90);
      numpy_restatement_fps  the op-for-op numpy restatement (what the reference's numpy lines cost; its FPS is a
                             numpy loop, pytorch3d being absent), one process;
      multi_process_fps      the C oracle in `multi_process_count` independent processes (one per core of this
                             process's CPU share, at most 16) — the reference has no such mode; it bounds what
                             process-level parallelism could give the CPU path.  Frames over the slowest child's
                             ORACLE time (its frame synthesis excluded); multi_process_fps_wall is the same run
                             against the wall clock."""
    n, t = _oracle_frames(0, budget_s, 600, scene, dist_mode)
    out = {"value": n / t, "unit": "frames/s", "cores": 1, "kind": "port",
           "sample": f"first {n} frames of the same synthetic episode (seeds 0..{n - 1}), C oracle -O2 "
                     f"-ffp-contract=off, stage1+crop+FPS({MODE_NAME[dist_mode]})+gather, {t:.1f} s",
           "host_cores": _host_cores(), "host_cpu_count": os.cpu_count()}
    try:
        from oracle import numpy_ref as R
        from pcp import synth
        intr = synth.intrinsics(H, W)
        make = synth.frame if scene == "random" else synth.scene_frame
        m, tn = 0, 0.0
        while tn < numpy_s and m < 100:
            d, c = make(m, H, W)
            t0 = time.perf_counter()
            s1 = R.stage1_frame(d, c, intr, intr)
            R.process_point_cloud(s1["pts_g"], s1["clrs_v"], K=K_FPS)
            tn += time.perf_counter() - t0
            m += 1
        out["numpy_restatement_fps"] = m / tn
        out["numpy_restatement_sample"] = f"{m} frames, {tn:.1f} s, 1 process (un-fused FPS distances)"
    except Exception as e:                                         # never fatal for the bench line
        out["numpy_restatement_error"] = repr(e)
    try:
        procs = max(1, min(16, _host_cores()))
        cmd = [sys.executable, os.path.abspath(__file__), "--cpu-worker", "--scene", scene,
               "--dist-mode", str(dist_mode), "--cpu-seconds", str(multi_s)]
        t0 = time.perf_counter()
        kids = [subprocess.Popen(cmd + ["--cpu-seed0", str(1000 * (r + 1))], stdout=subprocess.PIPE, text=True)
                for r in range(procs)]
        done = [k.communicate(timeout=120)[0].split() for k in kids]
        wall = time.perf_counter() - t0
        frames = sum(int(d[0]) for d in done)
        busy = max(float(d[1]) for d in done)
        out["multi_process_fps"] = frames / busy                   # oracle-only busy time: frame synthesis excluded
        out["multi_process_fps_wall"] = frames / wall              # the same run against the wall clock (synthesis included)
        out["multi_process_count"] = procs
        out["multi_process_sample"] = f"{frames} frames over {procs} processes, {busy:.1f} s busy ({wall:.1f} s wall)"
    except Exception as e:
        out["multi_process_error"] = repr(e)
    return out


# ---------------------------------------------------------------------------------------- checker
def verify_against_oracle(pipe, frames_idx, seed0, scene, dist_mode):
    """Last step's outputs of `pipe` vs the C oracle on the given frames.  Runs AFTER the timed region."""
    import torch
    from oracle import cbind
    from pcp import synth
    intr = synth.intrinsics(H, W)
    po = cbind.make_params(intr, intr, apply_crop=1)
    make = synth.frame if scene == "random" else synth.scene_frame
    off = pipe.stage1.offsets.cpu().numpy()
    bad = []
    for b in frames_idx:
        d, c = make(seed0 + b, H, W)
        ref = cbind.frame_pipeline(d, c, po, K_FPS, dist_mode)
        ok = (int(off[b + 1] - off[b]) == ref["M"]
              and np.array_equal(pipe.fps.idx[b].cpu().numpy(), ref["idx"])
              and np.array_equal(pipe.fps.xyz[b].cpu().numpy(), ref["xyz"])
              and np.array_equal(pipe.rgb_out[b].cpu().numpy(), ref["rgb"]))
        if not ok:
            bad.append(b)
    return bad


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--frames", type=int, default=FRAMES, help="frames per episode (default 300)")
    ap.add_argument("--algo", type=int, default=0, help="FPS algorithm (0 auto, 1 brute, 2 bucket)")
    ap.add_argument("--dist-mode", type=int, default=1, choices=(0, 1),
                    help="1 (default): fused f32 distance = pytorch3d's CUDA build, what the reference runs (DP3:47); "
                         "0: un-fused = pytorch3d's CPU build")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-verify", action="store_true", help="skip the post-timing check against the oracle")
    ap.add_argument("--no-other-mode", action="store_true", help="skip timing the other distance mode")
    ap.add_argument("--no-secondary", action="store_true", help="skip the K = 2500 and two-episode secondary figures")
    ap.add_argument("--scene", choices=["random", "tabletop"], default="random",
                    help="random: BASELINE's uniform-depth frames (default); tabletop: ray-cast table + boxes (SURVEY 8d)")
    ap.add_argument("--streams", type=int, default=1,
                    help="opt-in: N buffer sets + HIP streams, step i on stream i %% N (consecutive episodes overlap)")
    ap.add_argument("--cpu-worker", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--cpu-seed0", type=int, default=0, help=argparse.SUPPRESS)
    ap.add_argument("--cpu-seconds", type=float, default=4.0, help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.cpu_worker:                                  # child of cpu_baseline(): never touches torch or the GPU
        cpu_worker(args.cpu_seed0, args.cpu_seconds, args.scene, args.dist_mode)
        return

    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    from pcp import synth
    make_episode = synth.episode if args.scene == "random" else synth.scene_episode
    seed0 = rank * 1000
    depth_h, rgb_h = make_episode(args.frames, seed0=seed0, H=H, W=W)          # rank r = episode r; made before the
    ndev = torch.cuda.device_count()                                            # GPU is touched (scene_episode forks)
    local_dev = local % max(ndev, 1)                 # rehearsals with more ranks than GPUs share devices
    torch.cuda.set_device(local_dev)
    dev = torch.device("cuda", local_dev)
    rccl = ndev >= world                             # one GPU per rank: RCCL; otherwise (rehearsal) gloo on the host
    if world > 1:
        import torch.distributed as dist
        if rccl:
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")

    from pcp import _ffi, ops
    intr = synth.intrinsics(H, W)
    params = _ffi.make_params(intr, intr, apply_crop=True)
    B = args.frames
    depth = torch.from_numpy(depth_h).to(dev)
    rgb = torch.from_numpy(rgb_h).to(dev)
    del depth_h, rgb_h
    # --streams N (default 1): N independent buffer sets and HIP streams; step i runs whole on stream i % N,
    # so consecutive episodes overlap on the chip the way a dataset run (BASELINE configs[3]) would overlap
    # them.  Every step still does all of its work inside the timed region.  The default is the strict
    # one-episode-at-a-time step; the multi-stream figures are quoted in DESIGN.md only.
    nbuf = max(1, args.streams)
    pipes = [ops.FramePipeline(B, H, W, K_FPS, params, dev, dist_mode=args.dist_mode, algo=args.algo) for _ in range(nbuf)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(nbuf)] if nbuf > 1 else [torch.cuda.current_stream(dev)]
    for s_ in streams:
        s_.wait_stream(torch.cuda.current_stream(dev))

    step_no = [0]

    def step(ev=None, use=None):
        j = step_no[0] % nbuf
        step_no[0] += 1
        pipe = use or pipes[j]
        f = pipe.fps
        with torch.cuda.stream(streams[j]):
            out, offsets = pipe.stage1.run(depth, rgb, params)
            if ev is not None:
                ev[0].record(streams[j])
            f.enqueue(out["xyz32"], offsets, pipe.stage1.cap)                    # pcp_fps_batched
            if ev is not None:
                ev[1].record(streams[j])
            pipe.gather(depth, rgb)                                              # pcp_gather_rgb_pixels

    def barrier():
        if world > 1:
            torch.distributed.barrier()

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    events = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
              for _ in range(args.steps)]
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(events[i])
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    last_pipe = pipes[(step_no[0] - 1) % nbuf]

    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if rccl else "cpu")
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())

    fps_ms = sum(a.elapsed_time(b) for a, b in events) / max(args.steps, 1)   # dominant kernel(s)
    off = last_pipe.stage1.offsets.cpu().numpy()
    m_mean = float(np.diff(off).mean())
    last_pipe.check()                                                          # stage-1 capacity / FPS bounds flags
    assert int(last_pipe.fps.idx.min().item()) >= 0, "FPS produced padding on a bench frame"

    # ---- after the timed region: checker, the other distance mode, the CPU side -------------------------------
    failed = 0
    if not args.no_verify:
        picks = sorted({0, B // 3, (2 * B) // 3, B - 1})
        bad = verify_against_oracle(last_pipe, picks, seed0, args.scene, args.dist_mode)
        if bad:
            print(f"[bench] rank {rank}: GPU results differ from the oracle on frames {bad}", file=sys.stderr, flush=True)
            failed = 1
    other = None
    if not args.no_other_mode:
        om = 1 - args.dist_mode
        del pipes[1:]
        opipe = ops.FramePipeline(B, H, W, K_FPS, params, dev, dist_mode=om, algo=args.algo)
        n_other = max(1, min(args.steps, 5))
        step_no[0] = 0
        step(use=opipe)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(n_other):
            step_no[0] = 0
            step(use=opipe)
        torch.cuda.synchronize()
        other = {"dist_mode": MODE_NAME[om], "ms_per_step": (time.perf_counter() - t1) / n_other * 1e3,
                 "steps": n_other}
        if not args.no_verify:
            bad = verify_against_oracle(opipe, [0, B - 1], seed0, args.scene, om)
            if bad:
                print(f"[bench] rank {rank}: {MODE_NAME[om]} results differ from the oracle on frames {bad}",
                      file=sys.stderr, flush=True)
                failed = 1

    # Secondary figures (never `value`): the reference's own K = 2500 (DP3:47) on the same episode, and the dataset-mode
    # launch shape — TWO episodes (600 clouds) per launch, which puts two workgroups on (almost) every CU instead of on
    # 44 of the 256.  Same kernels, all work inside the timed loops, N = 1 only.
    secondary = None
    if not args.no_secondary and world == 1:
        secondary = {}
        del pipes[:]
        last_pipe = opipe = None
        torch.cuda.empty_cache()
        n_sec = max(1, min(args.steps, 5))

        def timed(pipe, d, c):
            def one():
                out, offsets = pipe.stage1.run(d, c, params)
                pipe.fps.enqueue(out["xyz32"], offsets, pipe.stage1.cap)
                pipe.gather(d, c)
            one()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(n_sec):
                one()
            torch.cuda.synchronize()
            pipe.check()
            return (time.perf_counter() - t1) / n_sec

        dt = timed(ops.FramePipeline(B, H, W, 2500, params, dev, dist_mode=args.dist_mode, algo=args.algo), depth, rgb)
        secondary["k2500_ms_per_step"] = dt * 1e3
        secondary["k2500_frames_per_s"] = B / dt
        torch.cuda.empty_cache()
        d2, c2 = torch.cat([depth, depth]), torch.cat([rgb, rgb])
        dt = timed(ops.FramePipeline(2 * B, H, W, K_FPS, params, dev, dist_mode=args.dist_mode, algo=args.algo), d2, c2)
        secondary["two_episode_ms_per_step"] = dt * 1e3
        secondary["two_episode_frames_per_s"] = 2 * B / dt
        secondary["note"] = (f"{n_sec} steps each after one warm-up; two_episode = one launch sequence over {2 * B} frames "
                             "(the episode twice); neither is `value`")
        del d2, c2

    if rank == 0:
        total_frames = B * world * args.steps
        value = total_frames / elapsed
        achieved = (B * BYTES_PER_FRAME) / (fps_ms * 1e-3) / 1e9 if fps_ms > 0 else 0.0
        line = {
            "metric": "frames/sec (640x480 depth -> 1024-pt FPS)", "value": value, "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64 backprojection/crop, f32 FPS distances",
            "data": "synthetic",
            "config": {"workload": f"BASELINE configs[1]: one synthetic {B}-frame episode per GPU, "
                                   f"640x480 depth+RGB -> crop -> {K_FPS}-pt FPS, resident in HBM",
                       "frames_per_step_per_gpu": B, "K": K_FPS, "mean_points_after_crop": m_mean,
                       "fps_algo": args.algo, "dist_mode": MODE_NAME[args.dist_mode],
                       "dist_mode_note": "fma = pytorch3d CUDA build (the reference's use_cuda=True, DP3:47); "
                                         "nofma = pytorch3d CPU build",
                       "other_dist_mode": other,
                       "verified_vs_oracle": None if args.no_verify else (failed == 0),
                       "streams": nbuf, "scene": args.scene, "secondary": secondary},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS,
                         # the committed PMC passes were taken on the default (random-depth) workload
                         "traffic": measured_traffic()[0] if args.scene == "random" else None,
                         "traffic_source": measured_traffic()[1] if args.scene == "random" else None,
                         "kernel": "fps (pcp_fps_batched launches)", "kernel_ms_per_launch": fps_ms,
                         "algorithmic_bytes_per_launch": B * BYTES_PER_FRAME},
        }
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline(scene=args.scene, dist_mode=args.dist_mode)
        print(json.dumps(line), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()
    if failed:
        sys.exit(1)


if __name__ == "__main__":
    main()
