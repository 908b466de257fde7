"""GPU (-m gpu): the C ABI only enqueues on the caller's stream — no allocation, no synchronisation — so a whole step
(stage 1, both scans, scatter, every FPS launch shape, colour gather) can be recorded into a HIP graph and replayed on
new input.  Checked bit for bit against the same step run eagerly."""
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
