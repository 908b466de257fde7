#!/usr/bin/env python3
"""GPU drop-in for the reference's dp3_point_cloud_processing.py (same function names, argument
meaning, prints, CLI and file naming): workspace crop → farthest-point sampling → colour gather,
all on the MI355X through libpcp_hip.so.  Citations are lines of the reference file (DP3)."""
import argparse
import os
import sys

import numpy as np
import torch

# the stage driver starts this file as a child process with PYTHONPATH removed (S2:9-18)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pcp import ops, ply  # noqa: E402

WORK_SPACE = ((-0.05, 0.4), (-0.12, 0.35), (-0.008, 0.6))      # DP3:29-33
NUM_POINTS = 2500                                              # DP3:47
DIST_MODE = int(os.environ.get("PCP_DIST_MODE", "0"))          # 0 un-fused (default), 1 fused (CUDA pytorch3d)


def _device():
    if not torch.cuda.is_available():
        raise RuntimeError("pcp needs a ROCm device (MI355X); there is no CPU fallback")
    return torch.device("cuda")


def farthest_point_sampling(points, num_points=2500, use_cuda=False):
    """DP3:7-21.  points (M,3) ndarray → (sampled (num_points,3) float32, idx (num_points,) int64).
    `use_cuda` is accepted for signature compatibility; the sampling always runs on the GPU."""
    pts = torch.from_numpy(np.ascontiguousarray(points)).float().to(_device())        # DP3:10-12
    m = pts.shape[0]
    if m == 0:
        raise ValueError("farthest_point_sampling: empty point cloud")              # reference dies in pytorch3d
    off = torch.tensor([0, m], dtype=torch.int64, device=pts.device)
    xyz, idx = ops.fps_batched(pts.contiguous(), off, int(num_points), dist_mode=DIST_MODE, max_pts=m)
    return xyz[0].cpu().numpy(), idx[0].cpu().numpy()                                 # DP3:18-19


def process_point_cloud(pc_xyz, pc_rgb, num_points=NUM_POINTS, verbose=True):
    """DP3:23-52: crop (six strict f64 compares) → FPS → pc_rgb[idx] (negative index = last row)."""
    dev = _device()
    xyz = torch.from_numpy(np.ascontiguousarray(pc_xyz, dtype=np.float64)).to(dev)
    rgb = torch.from_numpy(np.ascontiguousarray(pc_rgb, dtype=np.float64)).to(dev)
    lo = [w[0] for w in WORK_SPACE]
    hi = [w[1] for w in WORK_SPACE]
    _, x32, c64, m = ops.crop_compact(xyz, rgb, lo, hi, want_xyz64=False)             # DP3:35-42
    if verbose:
        print(f" → After crop: {m} points")
    if m == 0:
        raise ValueError("process_point_cloud: no point inside the workspace")
    off = torch.tensor([0, m], dtype=torch.int64, device=dev)
    fxyz, idx = ops.fps_batched(x32.contiguous(), off, int(num_points), dist_mode=DIST_MODE, max_pts=m)
    sel = idx[0]
    sel = torch.where(sel < 0, sel + m, sel)                                           # DP3:48 numpy negative index
    pc_rgb_fps = c64.index_select(0, sel).cpu().numpy()
    pc_xyz_fps = fxyz[0].cpu().numpy()
    if verbose:
        print(f" → After FPS: {pc_xyz_fps.shape[0]} points")
    return pc_xyz_fps, pc_rgb_fps


def save_ply(path, xyz, rgb):
    ply.write_point_cloud(path, xyz, rgb)                                              # DP3:54-58


def process(input_root, output_root, verbose=True):
    """DP3:61-92: every *.ply of input_root → <name>_filtered.ply in output_root."""
    os.makedirs(output_root, exist_ok=True)
    names = sorted(f for f in os.listdir(input_root) if f.endswith(".ply"))
    print(f"\n=== Processing folder: {input_root} ({len(names)} clouds) ===")
    for fn in names:
        src = os.path.join(input_root, fn)
        dst = os.path.join(output_root, fn.replace(".ply", "_filtered.ply"))
        if verbose:
            print(f"\nProcessing {fn} ...")
        pc_xyz, pc_rgb = ply.read_point_cloud(src)
        if pc_xyz.shape[0] == 0:
            print(f"Warning: Empty point cloud at {src}")                              # DP3:76-78
            continue
        if pc_rgb is None:
            pc_rgb = np.zeros_like(pc_xyz)
        if verbose:
            print(f"Loaded XYZ: {pc_xyz.shape} | RGB: {pc_rgb.shape}")
        xyz_f, rgb_f = process_point_cloud(pc_xyz, pc_rgb, verbose=verbose)
        save_ply(dst, xyz_f, rgb_f)
        if verbose:
            print(f"Saved → {dst}")
    print(f"\nDone. All filtered clouds saved in {output_root}")


def main():
    ap = argparse.ArgumentParser(description="Process point clouds with FPS and workspace cropping.")
    ap.add_argument("--input_root", type=str, required=True)
    ap.add_argument("--output_root", type=str, required=True)
    args = ap.parse_args()
    process(args.input_root, args.output_root)


if __name__ == "__main__":
    main()
