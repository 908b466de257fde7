#!/usr/bin/env python3
"""CPU baselines beside the GPU number: the op-for-op numpy restatement and the C oracle of the
per-frame path (640x480 -> crop -> 1024-pt FPS) on this box's host cores — one process, and one process per
core of the box's CPU share — compute-only and with the PLY round trip the reference's stages perform.
Test infrastructure (it imports oracle/, which is why it lives under tests/; pytest does not collect it).    python3 tests/cpu_baselines.py [--procs 16] [--frames 8]"""
import argparse, json, os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "point-cloud-processing_amd")]
import numpy as np

H, W, K = 480, 640, 1024


def _frames(n, seed0):
    from pcp import synth
    return [synth.frame(seed0 + i, H, W) for i in range(n)]


def c_oracle(n, seed0=0, with_ply=False):
    from oracle import cbind
    from pcp import ply, synth
    intr = synth.intrinsics(H, W)
    po = cbind.make_params(intr, intr, apply_crop=1)
    pu = cbind.make_params(intr, intr, apply_crop=0)
    fr = _frames(n, seed0)
    cbind.frame_pipeline(*fr[0], po, K, 0)
    tmp = tempfile.mkdtemp()
    t0 = time.perf_counter()
    for i, (d, c) in enumerate(fr):
        if not with_ply:
            cbind.frame_pipeline(d, c, po, K, 0)
        else:                                   # stage 1 -> global_frame PLY -> stage 2 -> filtered PLY, as S1/S2 do
            r = cbind.backproject_align_crop(d, c, pu)
            p = os.path.join(tmp, f"pc_{i:05d}.ply")
            ply.write_point_cloud(p, r["xyz64"], r["rgb32"])
            xyz, rgb = ply.read_point_cloud(p)
            m = cbind.crop_mask(xyz, [-0.05, -0.12, -0.008], [0.4, 0.35, 0.6]).astype(bool)
            sx, idx = cbind.fps(xyz[m].astype(np.float32), K, 0)
            ply.write_point_cloud(os.path.join(tmp, f"pc_{i:05d}_filtered.ply"), sx, rgb[m][idx])
    return n / (time.perf_counter() - t0)


def numpy_ref(n):
    from oracle import numpy_ref as R
    from pcp import synth
    intr = synth.intrinsics(H, W)
    fr = _frames(n, 0)
    t0 = time.perf_counter()
    for d, c in fr:
        s1 = R.stage1_frame(d, c, intr, intr)
        R.process_point_cloud(s1["pts_g"], s1["clrs_v"], K=K)
    return n / (time.perf_counter() - t0)


def _worker(args):
    n, seed0 = args
    return c_oracle(n, seed0)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--procs", type=int, default=16)
    ap.add_argument("--frames", type=int, default=40)
    a = ap.parse_args()
    out = {"workload": "640x480 depth+RGB -> crop -> 1024-pt FPS, synthetic frames"}
    out["c_oracle_1proc_frames_per_s"] = round(c_oracle(a.frames), 2)
    out["c_oracle_1proc_with_ply_io_frames_per_s"] = round(c_oracle(max(a.frames // 2, 4), with_ply=True), 2)
    try:
        out["numpy_restatement_1proc_frames_per_s"] = round(numpy_ref(3), 3)
    except Exception as e:                      # the numpy FPS loop is slow; never fatal for the table
        out["numpy_restatement_error"] = repr(e)
    import multiprocessing as mp
    with mp.get_context("spawn").Pool(a.procs) as pool:
        t0 = time.perf_counter()
        pool.map(_worker, [(a.frames, 1000 * r) for r in range(a.procs)])
        dt = time.perf_counter() - t0
    out[f"c_oracle_{a.procs}procs_frames_per_s"] = round(a.procs * a.frames / dt, 1)
    out["procs"] = a.procs
    print(json.dumps(out), flush=True)
