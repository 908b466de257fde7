"""Shared test helpers: golden loading, synthetic frames."""
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_npz(name):
    z = np.load(os.path.join(GOLDEN, name), allow_pickle=False)
    return {k: z[k] for k in z.files}


def load_digests():
    with open(os.path.join(GOLDEN, "stage1_digests.json")) as fh:
        return json.load(fh)


def intr_dict(a):
    return dict(fx=float(a[0]), fy=float(a[1]), cx=float(a[2]), cy=float(a[3]))


def synth_frame(seed, H=480, W=640, Hc=None, Wc=None):
    rng = np.random.default_rng(seed)
    depth_mm = rng.integers(300, 1500, (H, W)).astype(np.uint16)
    depth = (depth_mm.astype(np.float32) / 1000.0).astype(np.float32)
    rgb = rng.integers(0, 256, (Hc or H, Wc or W, 3), dtype=np.uint8)
    return depth, rgb


def bench_intr(H=480, W=640, f=605.0):
    return dict(fx=f, fy=f, cx=W / 2, cy=H / 2)
