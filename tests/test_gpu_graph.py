"""GPU (-m gpu): the C ABI only enqueues on the caller's stream — no allocation, no synchronisation — so a whole step
(stage 1, both scans, scatter, FPS, colour gather) can be recorded into a HIP graph and replayed on new input, checked
bit for bit against the same step run eagerly.  The frame-pipeline test records the 64-row wide FPS shape only (its
clouds are small); test_every_fps_launch_shape_records_into_a_hip_graph records the halved / quartered-bucket and the
deep shape too, including the hipFuncSetAttribute call the deep shape makes under capture."""
import numpy as np
import pytest
import torch

from pcp import _ffi, ops, synth

pytestmark = pytest.mark.gpu


def test_frame_pipeline_records_into_a_hip_graph_and_replays():
    B, H, W, K = 6, 120, 160, 256
    intr = synth.intrinsics(H, W)
    p = _ffi.make_params(intr, intr, apply_crop=True)
    d0, c0 = synth.episode(B, seed0=40, H=H, W=W)
    d1, c1 = synth.episode(B, seed0=90, H=H, W=W)
    depth = torch.from_numpy(d0).cuda()
    rgb = torch.from_numpy(c0).cuda()
    pipe = ops.FramePipeline(B, H, W, K, p, dist_mode=1)

    def snapshot():
        torch.cuda.synchronize()
        pipe.check()
        n = int(pipe.stage1.offsets[-1])
        return [t.clone() for t in (pipe.fps.xyz, pipe.rgb_out, pipe.fps.idx, pipe.stage1.offsets,
                                    pipe.stage1.out["xyz32"][:n])]

    # eager references for both inputs
    pipe.run(depth, rgb)
    ref0 = snapshot()
    depth.copy_(torch.from_numpy(d1)); rgb.copy_(torch.from_numpy(c1))
    pipe.run(depth, rgb)
    ref1 = snapshot()
    assert not torch.equal(ref0[2], ref1[2])

    # record one step (warm-up on a side stream first, as graph capture wants), then replay it on both inputs
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        pipe.run(depth, rgb)
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        pipe.run(depth, rgb)
    for it, (d, c, ref) in enumerate(((d0, c0, ref0), (d1, c1, ref1), (d0, c0, ref0))):
        depth.copy_(torch.from_numpy(d)); rgb.copy_(torch.from_numpy(c))
        for t in (pipe.fps.xyz, pipe.rgb_out, pipe.fps.idx):
            t.fill_(-7)                                    # stale outputs must not pass
        g.replay()
        torch.cuda.synchronize()
        flags = (int(pipe.stage1.overflow[0]), int(pipe.fps.overflow[0]))
        assert flags == (0, 0), f"replay {it}: overflow flags {flags}, offsets {pipe.stage1.offsets.tolist()}"
        got = snapshot()
        for a, b in zip(got, ref):
            assert torch.equal(a, b)


def test_static_entry_point_records_into_a_hip_graph():
    """pcp_fps_batched_static is meant for callers that cannot allocate — code under graph capture first of all: record it,
    replay it on two inputs, compare with pcp_fps_batched."""
    import ctypes as C
    L = _ffi.load_library()
    rng = np.random.default_rng(3)
    M, K = 30000, 200
    clouds = [rng.uniform(-1, 1, (M, 3)).astype(np.float32), rng.normal(0, 0.3, (M, 3)).astype(np.float32)]
    xyz = torch.from_numpy(clouds[0]).cuda()
    off = torch.tensor([0, M], dtype=torch.int64, device="cuda")
    idx = torch.full((1, K), -7, dtype=torch.int64, device="cuda")
    out = torch.zeros((1, K, 3), dtype=torch.float32, device="cuda")
    flag = torch.zeros(1, dtype=torch.int32, device="cuda")
    p = lambda t: C.c_void_p(t.data_ptr())
    assert L.pcp_fps_static_prepare() == 0                  # resolve the static array's address OUTSIDE the capture (pcp.h)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        rc = L.pcp_fps_batched_static(p(xyz), p(off), 1, M, M, K, 1, 0, p(idx), p(out), p(flag),
                                      C.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0
    ref = ops.Fps(1, M, M, K, "cuda", 1, 0)
    for c in (clouds[1], clouds[0], clouds[1]):
        xyz.copy_(torch.from_numpy(c))
        idx.fill_(-7); flag.fill_(9)
        g.replay()
        ref.run(xyz, off)
        torch.cuda.synchronize()
        assert int(flag) == 0
        assert torch.equal(idx, ref.idx) and torch.equal(out, ref.xyz)


@pytest.mark.parametrize("algo", [0, 10])
def test_every_fps_launch_shape_records_into_a_hip_graph(algo):
    """max_pts > 262,144 makes pcp_fps_batched launch all four bucket shapes (wide, halved, quartered buckets, deep — the
    deep one raises its dynamic-LDS limit with hipFuncSetAttribute on every launch): record them, replay on other clouds
    of the same sizes, compare with the eager call.  The batch mixes the three wide size classes, so since the end of round 3
    the per-shape launches exit at once and the ANY-shape launch runs those clouds, under capture too (algo 0: a workgroup per
    cloud; algo 10: as a quantum-sliced launch — slot prep kernel, 96 KB of dynamic LDS set under capture, clouds changing
    workgroups during the replay)."""
    rng = np.random.default_rng(9)
    sizes = [3000, 70000, 140000, 270000, 64, 60000, 61000, 100000]
    K = 96 if algo == 0 else 700
    off = torch.from_numpy(np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)).cuda()
    total = int(sum(sizes))
    sets = [np.concatenate([rng.uniform(-1, 1, (m, 3)) * rng.uniform(0.2, 1.0, 3) for m in sizes]).astype(np.float32)
            for _ in range(2)]
    xyz = torch.from_numpy(sets[0]).cuda()
    plan = ops.Fps(len(sizes), total, 300000, K, "cuda", 1, algo)
    refs = []
    for s_ in sets:
        xyz.copy_(torch.from_numpy(s_))
        plan.run(xyz, off)
        torch.cuda.synchronize()
        assert int(plan.overflow) == 0
        refs.append((plan.idx.clone(), plan.xyz.clone()))
    assert not torch.equal(refs[0][0], refs[1][0])
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        plan.run(xyz, off)
    for it in (0, 1, 0):
        xyz.copy_(torch.from_numpy(sets[it]))
        plan.idx.fill_(-7); plan.xyz.fill_(-7.0)
        g.replay()
        torch.cuda.synchronize()
        assert int(plan.overflow) == 0
        assert torch.equal(plan.idx, refs[it][0]) and torch.equal(plan.xyz, refs[it][1]), f"replay on set {it}"
