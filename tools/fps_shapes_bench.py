#!/usr/bin/env python3
"""Times pcp_fps_batched (K=1024) on 300-cloud batches of differently shaped synthetic clouds.
    python3 tools/fps_shapes_bench.py            (on the GPU box)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "point-cloud-processing_amd")]
import numpy as np, torch
from pcp import ops

def main():
    B, n, K = 300, 60000, 1024
    rng = np.random.default_rng(0)
    u = lambda lo, hi: rng.uniform(lo, hi, n)
    shapes = {
        "cube": lambda: np.c_[u(0, 1), u(0, 1), u(0, 1)],
        "tabletop": lambda: np.c_[u(-0.05, 0.4), u(-0.12, 0.35), 0.02 + 0.002 * rng.standard_normal(n) + (rng.random(n) < 0.15) * u(0, 0.12)],
        "plane": lambda: np.c_[u(-0.05, 0.4), u(-0.12, 0.35), 0.02 + 0.002 * rng.standard_normal(n)],
        "slab_8_8_1": lambda: np.c_[u(0, 0.8), u(0, 0.8), u(0, 0.1)],
    }
    off = torch.arange(B + 1, dtype=torch.int64, device="cuda") * n
    for name, gen in shapes.items():
        base = gen().astype(np.float32)
        pts = torch.from_numpy(np.concatenate([base[rng.permutation(n)] for _ in range(4)] * (B // 4))).cuda()
        plan = ops.Fps(B, B * n, n, K, pts.device)
        for _ in range(2):
            plan.run(pts, off)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(5):
            plan.run(pts, off)
        b.record(); torch.cuda.synchronize()
        print(f"{name:12s} {a.elapsed_time(b) / 5:7.3f} ms / {B} clouds of {n} points", flush=True)

if __name__ == "__main__":
    main()
