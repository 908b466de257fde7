#!/usr/bin/env python3
"""Randomised FPS parity soak (not collected by pytest; run by hand on a GPU box):
   python tests/stress_fps.py [--seconds 150] [--seed 0]
Random batches of clouds (sizes on both sides of every launch-shape bound, volumes, sheets, lattices with exact
ties, duplicates), random K, distance mode and algo 2/3/4/5/6/9/10 (deferred rounds on / off, multi-pick rounds on / off / with exact-key deferral, quantum-sliced launch), each cloud checked bit-exact against the C oracle."""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "point-cloud-processing_amd")]
import numpy as np, torch
from oracle import cbind
from pcp import ops


def make_cloud(rng, m):
    kind = rng.integers(0, 6)
    if kind == 0:
        return rng.uniform(-0.5, 0.5, (m, 3)).astype(np.float32)
    if kind == 1:                                    # sheet with noise
        p = rng.uniform(0, 1, (m, 3)); p[:, rng.integers(0, 3)] = 0.3 + 0.002 * rng.standard_normal(m)
        return p.astype(np.float32)
    if kind == 2:                                    # lattice: many exact ties
        g = rng.integers(0, 12, (m, 3)).astype(np.float32)
        return g
    if kind == 3:                                    # heavy duplicates
        base = rng.uniform(-1, 1, (max(1, m // 7), 3)).astype(np.float32)
        return base[rng.integers(0, len(base), m)]
    if kind == 4:                                    # clusters
        c = rng.uniform(-1, 1, (8, 3)); return (c[rng.integers(0, 8, m)] + 0.02 * rng.standard_normal((m, 3))).astype(np.float32)
    return (rng.standard_normal((m, 3)) * np.array([1.0, 0.2, 0.01])).astype(np.float32)   # rod-ish


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=150)
    ap.add_argument("--seed", type=int, default=0)
    a = ap.parse_args(argv)
    rng = np.random.default_rng(a.seed)
    sizes = [1, 2, 63, 64, 65, 127, 128, 129, 1000, 2047, 2048, 2049, 5000, 20000, 65535, 65536, 65537, 90000,
             131072, 131073, 200000, 262144, 262145, 300000]
    t_end, n_batches, n_clouds = time.time() + a.seconds, 0, 0
    while time.time() < t_end:
        nb = int(rng.integers(1, 6))
        big = rng.random() < 0.25
        ms = [int(rng.choice(sizes if big else sizes[:16])) + int(rng.integers(0, 3)) for _ in range(nb)]
        if rng.random() < 0.2:                       # a batch large enough for the deferred rounds by default
            ms += [int(rng.integers(2048, 6000)) for _ in range(200)]
        clouds = [make_cloud(rng, m) for m in ms]
        K = int(rng.choice([1, 2, 5, 64, 255, 256, 257, 300, 700, 1024, 1500]))
        dm, algo = int(rng.integers(0, 2)), int(rng.choice([2, 3, 4, 5, 6, 9, 10]))
        off = np.zeros(len(clouds) + 1, np.int64); off[1:] = np.cumsum([len(c) for c in clouds])
        xyz, idx = ops.fps_batched(torch.from_numpy(np.concatenate(clouds)).cuda(), torch.from_numpy(off).cuda(), K,
                                   dist_mode=dm, algo=algo, max_pts=max(ms))
        xyz, idx = xyz.cpu().numpy(), idx.cpu().numpy()
        check = range(len(clouds)) if len(clouds) < 20 else list(range(nb)) + list(rng.integers(nb, len(clouds), 6))
        for b in check:
            rx, ri = cbind.fps(clouds[b], K, dm)
            if not (np.array_equal(idx[b], ri) and np.array_equal(xyz[b], rx)):
                bad = int(np.argmax(idx[b] != ri))
                print(f"MISMATCH seed={a.seed} batch={n_batches} cloud={b} M={len(clouds[b])} K={K} dm={dm} algo={algo} first bad pick {bad}", flush=True)
                sys.exit(1)
            n_clouds += 1
        n_batches += 1
        if n_batches % 10 == 0:
            print(f"{n_batches} batches, {n_clouds} clouds ok", flush=True)
    print(f"done: {n_batches} batches, {n_clouds} clouds, all bit-exact", flush=True)


if __name__ == "__main__":
    main()
