"""Seeded synthetic RGB-D frames of the benchmark's shape: depth is u16
millimetres uniform in [300,1500) converted to f32 metres exactly as the topic extractor does
(extract_topics_one_camera.py:68-72), rgb is uniform u8."""
import numpy as np


def frame(seed, H=480, W=640, Hc=None, Wc=None):
    rng = np.random.default_rng(seed)
    depth_mm = rng.integers(300, 1500, (H, W)).astype(np.uint16)
    depth = (depth_mm.astype(np.float32) / 1000.0).astype(np.float32)
    rgb = rng.integers(0, 256, (Hc or H, Wc or W, 3), dtype=np.uint8)
    return depth, rgb


def episode(n_frames, seed0=0, H=480, W=640):
    depth = np.empty((n_frames, H, W), np.float32)
    rgb = np.empty((n_frames, H, W, 3), np.uint8)
    for i in range(n_frames):
        depth[i], rgb[i] = frame(seed0 + i, H, W)
    return depth, rgb


def intrinsics(H=480, W=640, f=605.0):
    return dict(fx=f, fy=f, cx=W / 2, cy=H / 2)
