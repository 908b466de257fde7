#!/usr/bin/env python3
"""
Generates tests/golden/stage2_*.npz by IMPORTING the reference's own stage-2 module
(/root/reference/data_processing_pipeline/dp3_point_cloud_processing.py) and calling its
`process_point_cloud` (DP3:23-52) on seeded clouds.  Run in the build container only; the
.npz outputs are committed, this script is the record of how they were made.

    python tests/golden/make_golden_stage2.py

What the reference's own lines pin here:
  * the workspace crop DP3:35-42 (mask, boolean gather, row order),
  * the float32 cast DP3:10 (`torch.from_numpy(points).float()`),
  * the colour gather DP3:48 (`pc_rgb[idx]`, numpy negative indexing for pytorch3d's -1 padding),
  * K = 2500 (DP3:47).
What they do NOT pin: the sampling itself.  pytorch3d is not installed here (and there is no
CUDA device), so `pytorch3d.ops.sample_farthest_points` is replaced by a RECORDER that
  (1) captures the tensor the reference hands to pytorch3d (`fps_in`), and
  (2) answers with indices computed by oracle/numpy_ref.fps under pytorch3d's published
      contract (start 0, -1 padding beyond M, sampled = points[idx] with zero rows for -1).
`torch.Tensor.cuda` is patched to the identity inside this process only (DP3:12 would need a
device).  `open3d` is an empty stand-in (only save_ply / process touch it; neither is called).
FPS parity therefore stays "unpinned by the reference"; crop, cast and gather are pinned.
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference/data_processing_pipeline"
sys.path.insert(0, ROOT)
from oracle import numpy_ref as R  # noqa: E402  (the scripted sampler; test infrastructure)

RECORD = {}


def _sample_farthest_points(points, K, **kw):
    """Recorder standing in for pytorch3d.ops.sample_farthest_points(points=(1,M,3), K=[k])."""
    assert points.dim() == 3 and points.shape[0] == 1 and points.dtype == torch.float32
    k = int(K[0])
    pts = points[0].numpy().copy()
    RECORD["fps_in"] = pts
    sampled, idx = R.fps(pts, k)
    RECORD["idx"] = idx.copy()
    return torch.from_numpy(sampled)[None], torch.from_numpy(idx)[None]


def load_reference():
    sys.dont_write_bytecode = True
    sys.modules.setdefault("open3d", types.ModuleType("open3d"))
    p3d = types.ModuleType("pytorch3d")
    ops = types.ModuleType("pytorch3d.ops")
    ops.sample_farthest_points = _sample_farthest_points
    p3d.ops = ops
    sys.modules["pytorch3d"] = p3d
    sys.modules["pytorch3d.ops"] = ops
    torch.Tensor.cuda = lambda self, *a, **k: self          # DP3:12, this process only
    sys.path.insert(0, REF)
    import dp3_point_cloud_processing as dp3
    return dp3


def run(dp3, pc_xyz, pc_rgb):
    RECORD.clear()
    with np.errstate(all="ignore"):
        xyz_f, rgb_f = dp3.process_point_cloud(pc_xyz, pc_rgb)
    return dict(pc_xyz=pc_xyz, pc_rgb=pc_rgb, fps_in=RECORD["fps_in"], idx=RECORD["idx"],
                out_xyz=np.asarray(xyz_f), out_rgb=np.asarray(rgb_f))


def stage1_cloud(seed, H, W, f):
    """A base-frame cloud the way stage 2 sees it after the PLY round trip:
    f64 points from the stage-1 restatement, colours k/255 in f64."""
    rng = np.random.default_rng(seed)
    depth = (rng.integers(300, 1500, (H, W)).astype(np.uint16).astype(np.float32) / 1000.0).astype(np.float32)
    rgb = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    intr = dict(fx=f, fy=f, cx=W / 2, cy=H / 2)
    out = R.stage1_frame(depth, rgb, intr, intr)
    colours = np.rint(out["clrs_v"].astype(np.float64) * 255.0) / 255.0
    return np.ascontiguousarray(out["pts_g"]), colours


def main():
    dp3 = load_reference()

    # 1. a realistic cloud: more than 2500 points survive the crop (no padding)
    xyz, rgb = stage1_cloud(21, 120, 160, 151.25)
    g = run(dp3, xyz, rgb)
    assert g["fps_in"].shape[0] > 2500 and g["idx"].min() >= 0
    np.savez_compressed(os.path.join(HERE, "stage2_frame_120x160.npz"), **g)

    # 2. fewer than 2500 survivors: pytorch3d pads idx with -1, DP3:48 then picks the LAST cropped row
    xyz, rgb = stage1_cloud(22, 48, 64, 60.5)
    g = run(dp3, xyz, rgb)
    assert 0 < g["fps_in"].shape[0] < 2500 and g["idx"][-1] == -1
    np.savez_compressed(os.path.join(HERE, "stage2_padded_48x64.npz"), **g)

    # 3. hand-made boundary rows: on a face (strict compare → out), one ulp inside, NaN / inf, and values
    #    whose float32 cast rounds (ties-to-even, denormal, large) — all inside a random cloud
    rng = np.random.default_rng(23)
    xyz = rng.uniform(-0.3, 0.8, (4000, 3))
    special = [
        [0.1, 0.1, 0.1], [-0.05, 0.1, 0.1], [0.4, 0.1, 0.1], [np.nextafter(-0.05, 1), 0.1, 0.1],
        [np.nextafter(0.4, 0), 0.1, 0.1], [0.1, -0.12, 0.1], [0.1, np.nextafter(-0.12, 1), 0.1],
        [0.1, 0.35, 0.1], [0.1, np.nextafter(0.35, 0), 0.1], [0.1, 0.1, -0.008],
        [0.1, 0.1, np.nextafter(-0.008, 1)], [0.1, 0.1, 0.6], [0.1, 0.1, np.nextafter(0.6, 0)],
        [np.nan, 0.1, 0.1], [0.1, np.inf, 0.1], [0.1, 0.1, -np.inf],
        [1.0 + 2.0 ** -24, 0.1, 0.1],                          # outside; (kept for the mask only)
        [0.25 + 2.0 ** -26, 0.25 + 3 * 2.0 ** -26, 0.5 + 2.0 ** -25],   # f32 ties: to even down / up / down
        [1e-46, 2.0 ** -140, 0.3],                             # underflow to 0 / f32 denormal
        [0.3999999999999999, 0.3499999999999999, 0.5999999999999999],   # inside in f64, ON the face once cast to f32
    ]
    xyz[5:5 + len(special)] = special
    rgb = rng.integers(0, 256, (4000, 3)) / 255.0
    g = run(dp3, xyz, rgb)
    np.savez_compressed(os.path.join(HERE, "stage2_boundary.npz"), **g)
    print("stage-2 golden vectors written to", HERE,
          {k: v.shape for k, v in g.items()})


if __name__ == "__main__":
    main()
