#!/usr/bin/env python3
"""GPU drop-in for the reference's extract_point_clouds_one_cam.py (XPC): same function names,
CLI and output tree (`camera_frame/` and `global_frame/` PLYs), with deproject + align_colors +
validity mask + base-frame transform fused into libpcp_hip.so kernels.

Camera intrinsics: the reference re-opens the ROS bag (XPC:37-56).  ROS 2 is CPU-side and often
absent where the GPU is, so `extract_intrinsics` first looks for `<extracted_data_root>/
intrinsics.json` ({"color": {fx,fy,cx,cy}, "depth": {...}}) and only then tries rosbag2_py."""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pcp import _ffi, ops, ply  # noqa: E402

T_DEPTH_TO_COLOR = np.eye(4, dtype=np.float32)                     # XPC:17-22
T_DEPTH_TO_COLOR[0, 3] = np.float32(-0.059)
BASE_TO_EXT_CAM = np.array([[-0.0056, 0.9985, -0.055, 0.0744],     # XPC:24-29 (fallback only)
                            [-0.9998, -0.0047, 0.0176, 0.0993],
                            [0.0173, 0.0551, 0.9983, 0.012],
                            [0.0, 0.0, 0.0, 1.0]], dtype=np.float32)
CAMERA_INFO_TOPICS = {"color": "/camera/camera/color/camera_info",   # XPC:32-35
                      "depth": "/camera/camera/depth/camera_info"}
BATCH = 32


def _device():
    if not torch.cuda.is_available():
        raise RuntimeError("pcp needs a ROCm device (MI355X); there is no CPU fallback")
    return torch.device("cuda")


def extract_intrinsics(bag_path, extracted_data_root=None):
    """XPC:37-56 — {'color': {fx,fy,cx,cy}, 'depth': {...}}."""
    for root in (extracted_data_root, bag_path):
        if root and os.path.exists(os.path.join(root, "intrinsics.json")):
            with open(os.path.join(root, "intrinsics.json")) as fh:
                raw = json.load(fh)
            return {k: {n: np.float64(raw[k][n]) for n in ("fx", "fy", "cx", "cy")} for k in ("color", "depth")}
    try:
        import rosbag2_py
        from rclpy.serialization import deserialize_message
        from rosidl_runtime_py.utilities import get_message
    except ImportError as exc:
        raise RuntimeError("no intrinsics.json and ROS 2 (rosbag2_py) is not importable") from exc
    reader = rosbag2_py.SequentialReader()
    reader.open(rosbag2_py.StorageOptions(uri=bag_path, storage_id="sqlite3"), rosbag2_py.ConverterOptions("", ""))
    types = {t.name: t.type for t in reader.get_all_topics_and_types()}
    found = {}
    while reader.has_next() and len(found) < len(CAMERA_INFO_TOPICS):
        topic, data, _ = reader.read_next()
        for key, name in CAMERA_INFO_TOPICS.items():
            if topic == name and key not in found and name in types:
                k = np.array(deserialize_message(data, get_message(types[name])).k).reshape(3, 3)
                found[key] = {"fx": k[0, 0], "fy": k[1, 1], "cx": k[0, 2], "cy": k[1, 2]}
    return found


def _params(depth_intr, color_intr, T, ext, crop=False):
    return _ffi.make_params(depth_intr, color_intr, T=T, ext=ext, apply_crop=crop)


def deproject(depth, params):
    """XPC:58-64: (H,W) depth → (H*W,3) float64 camera-frame points."""
    d = torch.from_numpy(np.ascontiguousarray(depth, dtype=np.float32))[None].to(_device())
    rgb = torch.zeros((1, 1, 1, 3), dtype=torch.uint8, device=d.device)
    out = ops.backproject_dense(d, rgb, _params(params, params, T_DEPTH_TO_COLOR, np.eye(4)), want=("pts_cam",))
    return out["pts_cam"][0].cpu().numpy()


def align_colors(pts_cam, rgb, color_params, T_depth_to_color):
    """XPC:66-78: (colors (N,3) float32, mask (N,) bool) for camera-frame points."""
    dev = _device()
    pts = torch.from_numpy(np.ascontiguousarray(pts_cam, dtype=np.float64)).to(dev)
    img = torch.from_numpy(np.ascontiguousarray(rgb, dtype=np.uint8)).to(dev)
    colors, mask = ops.align_colors(pts, img, color_params, T_depth_to_color)
    return colors.cpu().numpy(), mask.cpu().numpy().astype(bool)


def _read_rgb(path):
    from PIL import Image
    return np.asarray(Image.open(path).convert("RGB"))             # == cv2.imread + BGR2RGB (XPC:103)


def process(bag_path, extracted_data_root, output_root, base_to_ext_cam, io_threads=8):
    """XPC:80-125."""
    intr = extract_intrinsics(bag_path, extracted_data_root)
    for sub in ("camera_frame", "global_frame"):
        os.makedirs(os.path.join(output_root, sub), exist_ok=True)
    rgb_path = os.path.join(extracted_data_root, "camera", "rgb")
    depth_path = os.path.join(extracted_data_root, "camera", "depth")
    if not os.path.exists(rgb_path):
        print(f"ERROR: RGB path does not exist: {rgb_path}")
        return
    n_files = len([f for f in os.listdir(rgb_path) if f.endswith(".png")])
    frames = [i for i in range(n_files)
              if os.path.exists(os.path.join(rgb_path, f"rgb_{i:05d}.png"))
              and os.path.exists(os.path.join(depth_path, f"depth_{i:05d}.npy"))]       # XPC:100-101
    dev = _device()
    p = _params(intr["depth"], intr["color"], T_DEPTH_TO_COLOR, base_to_ext_cam)
    plan, done = None, 0

    def load(i):                                                   # XPC:103-104
        return (np.load(os.path.join(depth_path, f"depth_{i:05d}.npy")).astype(np.float32),
                _read_rgb(os.path.join(rgb_path, f"rgb_{i:05d}.png")))

    def write_pair(i, cam, glob, col):                             # XPC:113-123
        ply.write_point_cloud(os.path.join(output_root, "camera_frame", f"pc_{i:05d}.ply"), cam, col)
        ply.write_point_cloud(os.path.join(output_root, "global_frame", f"pc_{i:05d}.ply"), glob, col)

    # The reference's loop is serial (XPC:96-125).  Here the files of batch b+1 are decoded and the PLYs of batch b-1
    # written by a thread pool while the GPU handles batch b; outputs are the same files, byte for byte.
    from concurrent.futures import ThreadPoolExecutor
    batches = [frames[b0:b0 + BATCH] for b0 in range(0, len(frames), BATCH)]
    with ThreadPoolExecutor(max_workers=max(1, int(io_threads))) as pool:
        loads = [pool.submit(load, i) for i in batches[0]] if batches else []
        writes = []
        for bi, ids in enumerate(batches):
            got = [f.result() for f in loads]
            loads = [pool.submit(load, i) for i in batches[bi + 1]] if bi + 1 < len(batches) else []
            depth = np.stack([g[0] for g in got]); rgb = np.stack([g[1] for g in got])
            B, H, W = depth.shape
            if plan is None or (plan.B, plan.H, plan.W) != (B, H, W):
                plan = ops.Stage1(B, H, W, dev, want=("xyz_cam64", "xyz64", "rgb32"))
            out, off = plan.run(torch.from_numpy(depth).to(dev), torch.from_numpy(rgb).to(dev), p)
            off = off.cpu().numpy()
            cam, glob, col = (out[k][:off[-1]].cpu().numpy() for k in ("xyz_cam64", "xyz64", "rgb32"))
            for f in writes:
                f.result()                                         # the previous batch is on disk (and its errors raised)
            writes = [pool.submit(write_pair, i, cam[off[j]:off[j + 1]], glob[off[j]:off[j + 1]], col[off[j]:off[j + 1]])
                      for j, i in enumerate(ids)]
            for i in ids:
                done += 1
                if (i + 1) % 50 == 0:
                    print(f"  Processed {i + 1} frames...")
        for f in writes:
            f.result()
    return done


def main():
    ap = argparse.ArgumentParser(description="Extract point clouds from RGB-D data.")
    ap.add_argument("--bag_path", type=str, required=True)
    ap.add_argument("--extracted_data_root", type=str, required=True)
    ap.add_argument("--output_root", type=str, required=True)
    ap.add_argument("--extrinsics", type=str, help="Comma-separated 16 floats for BASE_TO_EXT_CAM matrix")
    args = ap.parse_args()
    if args.extrinsics:
        ext = np.array([float(v) for v in args.extrinsics.split(",")], dtype=np.float64).reshape(4, 4)   # XPC:138
    else:
        ext = np.array(BASE_TO_EXT_CAM, dtype=np.float32)                                              # XPC:141
    process(args.bag_path, args.extracted_data_root, args.output_root, ext)


if __name__ == "__main__":
    main()
