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

dist_mode_for = ops.dist_mode_for
REFERENCE_USE_CUDA = True                                      # DP3:47 hard-codes use_cuda=True


def _device():
    if not torch.cuda.is_available():
        raise RuntimeError("pcp needs a ROCm device (MI355X); there is no CPU fallback")
    return torch.device("cuda")


def farthest_point_sampling(points, num_points=2500, use_cuda=False):
    """DP3:7-21.  points (M,3) ndarray → (sampled (num_points,3) float32, idx (num_points,) int64).
    The sampling always runs on the MI355X; `use_cuda` selects the distance arithmetic of the pytorch3d build the
    reference would have used (True: CUDA kernel, fused; False: CPU build, un-fused) — see dist_mode_for()."""
    pts = torch.from_numpy(np.ascontiguousarray(points)).float().to(_device())        # DP3:10-12
    m = pts.shape[0]
    if m == 0:
        raise ValueError("farthest_point_sampling: empty point cloud")              # reference dies in pytorch3d
    off = torch.tensor([0, m], dtype=torch.int64, device=pts.device)
    xyz, idx = ops.fps_batched(pts.contiguous(), off, int(num_points), dist_mode=dist_mode_for(use_cuda), max_pts=m)
    return xyz[0].cpu().numpy(), idx[0].cpu().numpy()                                 # DP3:18-19


def process_point_cloud(pc_xyz, pc_rgb, num_points=NUM_POINTS, verbose=True, use_cuda=REFERENCE_USE_CUDA):
    """DP3:23-52: crop (six strict f64 compares) → FPS → pc_rgb[idx] (negative index = last row).
    The reference samples with use_cuda=True (DP3:47), hence the fused distance mode by default."""
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
    fxyz, idx = ops.fps_batched(x32.contiguous(), off, int(num_points), dist_mode=dist_mode_for(use_cuda), max_pts=m)
    sel = idx[0]
    sel = torch.where(sel < 0, sel + m, sel)                                           # DP3:48 numpy negative index
    pc_rgb_fps = c64.index_select(0, sel).cpu().numpy()
    pc_xyz_fps = fxyz[0].cpu().numpy()
    if verbose:
        print(f" → After FPS: {pc_xyz_fps.shape[0]} points")
    return pc_xyz_fps, pc_rgb_fps


def save_ply(path, xyz, rgb):
    ply.write_point_cloud(path, xyz, rgb)                                              # DP3:54-58


BATCH = int(os.environ.get("PCP_STAGE2_BATCH", "64"))          # clouds per launch sequence in process()


def process_point_clouds_batched(clouds, num_points=NUM_POINTS, voxel_size=None, sor=None,
                                 use_cuda=REFERENCE_USE_CUDA):
    """process_point_cloud (DP3:23-52) for a list of (pc_xyz, pc_rgb) pairs in ONE launch sequence: batched
    crop, batched FPS, one colour gather.  Returns [(xyz (K,3) f32, rgb (K,3) f64, cropped_count)]; a cloud
    with no point inside the workspace yields None (the serial loop raises for it, see process()).
    voxel_size (metres) and sor = (nb_neighbors, std_ratio) insert the BASELINE config-3 filters between crop
    and FPS (Open3D semantics, no counterpart in the reference; cropped_count is then the filtered count)."""
    dev = _device()
    sizes = [int(c[0].shape[0]) for c in clouds]
    off_h = np.zeros(len(clouds) + 1, np.int64)
    off_h[1:] = np.cumsum(sizes)
    xyz = torch.from_numpy(np.ascontiguousarray(np.concatenate([c[0] for c in clouds]), dtype=np.float64)).to(dev)
    rgb = torch.from_numpy(np.ascontiguousarray(np.concatenate([c[1] for c in clouds]), dtype=np.float64)).to(dev)
    lo = [w[0] for w in WORK_SPACE]
    hi = [w[1] for w in WORK_SPACE]
    filtered = voxel_size is not None or sor is not None
    x64, x32, c64, out_off = ops.crop_compact_batched(xyz, rgb, torch.from_numpy(off_h).to(dev), lo, hi,
                                                      want_xyz64=filtered)                     # DP3:35-42
    total = int(out_off[-1].item())
    if total == 0:
        return [None] * len(clouds)
    if filtered:
        c32 = c64[:total].float().contiguous()
        x64 = x64[:total].contiguous()
        ext = max(h - l for l, h in zip(lo, hi))          # cropped clouds (and their voxel means) lie inside the box
        if voxel_size is not None:
            v, out_off = ops.voxel_downsample(x64, c32, out_off, float(voxel_size), max_extent=ext)
            total = int(out_off[-1].item())
            x64, c32 = v["xyz64"][:total].contiguous(), v["rgb32"][:total].contiguous()
        if sor is not None and total > 0:
            s, out_off = ops.statistical_outlier_removal(x64, c32, out_off, int(sor[0]), float(sor[1]), max_extent=ext)
            total = int(out_off[-1].item())
            x64, c32 = s["xyz64"][:total].contiguous(), s["rgb32"][:total].contiguous()
        if total == 0:
            return [None] * len(clouds)
        x32, c64 = x64.float().contiguous(), c32.double()
    counts = np.diff(out_off.cpu().numpy())
    fxyz, idx = ops.fps_batched(x32[:total].contiguous(), out_off, int(num_points),
                                dist_mode=dist_mode_for(use_cuda), max_pts=int(counts.max()))
    m = torch.from_numpy(counts).to(dev).unsqueeze(1)
    sel = torch.where(idx < 0, idx + m, idx) + out_off[:-1].unsqueeze(1)                   # DP3:48 numpy negative index
    sel = torch.where(m > 0, sel, torch.zeros_like(sel))       # an empty crop has no row to gather (reported as None below)
    col = c64[:max(total, 1)].index_select(0, sel.reshape(-1)).reshape(len(clouds), int(num_points), 3).cpu().numpy()
    pts = fxyz.cpu().numpy()
    return [None if counts[b] == 0 else (pts[b], col[b], int(counts[b])) for b in range(len(clouds))]


def process(input_root, output_root, verbose=True, voxel_size=None, sor=None, num_points=NUM_POINTS):
    """DP3:61-92: every *.ply of input_root → <name>_filtered.ply in output_root.  Files are read, processed
    and written in chunks of BATCH clouds (same outputs, prints and error behaviour as the serial loop).
    voxel_size / sor / num_points are the extensions filter_point_clouds.py exposes (defaults = the reference)."""
    from concurrent.futures import ThreadPoolExecutor
    os.makedirs(output_root, exist_ok=True)
    names = sorted(f for f in os.listdir(input_root) if f.endswith(".ply"))
    print(f"\n=== Processing folder: {input_root} ({len(names)} clouds) ===")
    with ThreadPoolExecutor(max_workers=8) as pool:
        for c0 in range(0, len(names), BATCH):
            chunk = names[c0:c0 + BATCH]
            loaded = list(pool.map(lambda fn: ply.read_point_cloud(os.path.join(input_root, fn)), chunk))
            keep = [i for i, (xyz, _) in enumerate(loaded) if xyz.shape[0] > 0]
            clouds = [(loaded[i][0], loaded[i][1] if loaded[i][1] is not None else np.zeros_like(loaded[i][0]))
                      for i in keep]
            results = process_point_clouds_batched(clouds, num_points, voxel_size, sor) if clouds else []
            by_file = dict(zip(keep, results))
            writes = []
            for i, fn in enumerate(chunk):
                src = os.path.join(input_root, fn)
                dst = os.path.join(output_root, fn.replace(".ply", "_filtered.ply"))
                if verbose:
                    print(f"\nProcessing {fn} ...")
                if i not in by_file:
                    print(f"Warning: Empty point cloud at {src}")                      # DP3:76-78
                    continue
                if verbose:
                    print(f"Loaded XYZ: {clouds[keep.index(i)][0].shape} | RGB: {clouds[keep.index(i)][1].shape}")
                res = by_file[i]
                if res is None:
                    for w in writes:
                        w.result()
                    raise ValueError("process_point_cloud: no point inside the workspace")   # reference dies in pytorch3d
                xyz_f, rgb_f, m = res
                if verbose:
                    print(f" → After crop: {m} points")
                    print(f" → After FPS: {xyz_f.shape[0]} points")
                writes.append(pool.submit(save_ply, dst, xyz_f, rgb_f))
                if verbose:
                    print(f"Saved → {dst}")
            for w in writes:
                w.result()
    print(f"\nDone. All filtered clouds saved in {output_root}")


def main():
    ap = argparse.ArgumentParser(description="Process point clouds with FPS and workspace cropping.")
    ap.add_argument("--input_root", type=str, required=True)
    ap.add_argument("--output_root", type=str, required=True)
    args = ap.parse_args()
    process(args.input_root, args.output_root)


if __name__ == "__main__":
    main()
