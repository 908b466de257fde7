"""GPU (-m gpu): the HEADLINE configuration at full size — BASELINE.json configs[1], exactly the step bench.py times:
ops.FramePipeline(300, 480, 640, 1024, algo=0) on synth.episode(300, seed0=0) — against the C oracle, in both FPS
distance modes and on the ray-cast tabletop episode (72 k-point clouds → the split-bucket launch shape at batch 300).
Per variant: >= 24 frames spread over the batch bit-exact against the oracle (idx, xyz, rgb, M), and size-independent
properties on all 300 frames."""
import numpy as np
import pytest
import torch

from oracle import cbind
from pcp import _ffi, ops, synth
from helpers import bench_intr, scene_episode_cached

pytestmark = pytest.mark.gpu
H, W, K, B = 480, 640, 1024, 300


def _check_episode(depth_h, rgb_h, dist_mode, m_range):
    intr = bench_intr()
    params = _ffi.make_params(intr, intr, apply_crop=True)
    po = cbind.make_params(intr, intr, apply_crop=1)
    pipe = ops.FramePipeline(B, H, W, K, params, "cuda", dist_mode=dist_mode, algo=0)
    depth = torch.from_numpy(depth_h).cuda()
    rgb = torch.from_numpy(rgb_h).cuda()
    for _ in range(2):                                            # buffers are reused step after step in the bench
        xyz, col, idx, off = pipe.run(depth, rgb)
    torch.cuda.synchronize()
    pipe.check()
    off_h = off.cpu().numpy()
    idx_h, xyz_h, col_h = idx.cpu().numpy(), xyz.cpu().numpy(), col.cpu().numpy()
    M = np.diff(off_h)
    assert M.min() >= m_range[0] and M.max() <= m_range[1], (M.min(), M.max())
    # (i) frames spread over the batch (first, last, every 13th), bit-exact against the oracle
    picks = sorted(set(range(0, B, 13)) | {B - 1})
    assert len(picks) >= 24
    for b in picks:
        ref = cbind.frame_pipeline(depth_h[b], rgb_h[b], po, K, dist_mode)
        assert int(M[b]) == ref["M"], f"frame {b}"
        np.testing.assert_array_equal(idx_h[b], ref["idx"], err_msg=f"frame {b} idx")
        np.testing.assert_array_equal(xyz_h[b], ref["xyz"], err_msg=f"frame {b} xyz")
        np.testing.assert_array_equal(col_h[b], ref["rgb"], err_msg=f"frame {b} rgb")
    # (ii) every frame: start index 0, distinct in-range picks, xyz/rgb are the gathered rows
    assert (idx_h[:, 0] == 0).all() and idx_h.min() >= 0
    assert (idx_h < M[:, None]).all()
    srt = np.sort(idx_h, axis=1)
    assert (np.diff(srt, axis=1) > 0).all(), "a frame picked the same point twice"
    rows = torch.from_numpy(idx_h + off_h[:-1, None]).cuda().reshape(-1)
    np.testing.assert_array_equal(pipe.stage1.out["xyz32"].index_select(0, rows).reshape(B, K, 3).cpu().numpy(), xyz_h)
    # the pipeline fetches colours from the sampled points' source pixels; a stage-1 run that stores every colour
    # (rgb32) and a plain row gather must give the same values
    s1 = ops.Stage1(B, H, W, "cuda", cap=int(off_h[-1]), want=("rgb32",))
    full, off2 = s1.run(depth, rgb, params)
    assert torch.equal(off2, off)
    np.testing.assert_array_equal(full["rgb32"].index_select(0, rows).reshape(B, K, 3).cpu().numpy(), col_h)


@pytest.mark.parametrize("dist_mode", [1, 0])
def test_headline_episode_300_frames_vs_oracle(dist_mode):
    depth_h, rgb_h = synth.episode(B, seed0=0, H=H, W=W)
    _check_episode(depth_h, rgb_h, dist_mode, (50_000, 65_536))


def test_headline_tabletop_episode_300_frames_vs_oracle():
    depth_h, rgb_h = scene_episode_cached(B, 0)
    _check_episode(depth_h, rgb_h, 1, (40_000, 131_072))          # ~72 k points: most clouds run the two-part split-bucket shape
