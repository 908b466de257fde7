#!/usr/bin/env python3
"""Developer check: quantum-sliced FPS launches (algo 10 forced / algo 0 by batch size) against the resident launches
(algo 5), bit for bit, on stage-1 output and on ragged random batches; then timing at 300 clouds."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "point-cloud-processing_amd"))
import numpy as np, torch
from pcp import _ffi, ops, synth
dev = "cuda"

def run(xyz, off, K, algo, mode=1, max_pts=None):
    B = off.numel() - 1
    if max_pts is None:
        max_pts = int((off[1:] - off[:-1]).max())
    plan = ops.Fps(B, xyz.shape[0], max_pts, K, dev, mode, algo)
    out, idx = plan.run(xyz, off)
    torch.cuda.synchronize()
    return idx.clone(), out.clone(), plan

def cmp(name, xyz, off, K, mode=1, ref_algo=5, test_algo=10):
    i0, o0, _ = run(xyz, off, K, ref_algo, mode)
    bad = 0
    for rep in range(3):
        i1, o1, _ = run(xyz, off, K, test_algo, mode)
        same = bool((i0 == i1).all()) and bool((o0.view(torch.int32) == o1.view(torch.int32)).all())
        if not same:
            bad += 1
            rows = (i0 != i1).any(dim=1).nonzero().flatten().tolist()
            print(f"  MISMATCH {name} rep {rep}: clouds {rows[:10]} ... first diff col",
                  int((i0[rows[0]] != i1[rows[0]]).nonzero()[0]))
    print(("ok  " if not bad else "FAIL"), name, "B", off.numel() - 1, "K", K, "mode", mode, flush=True)
    return bad == 0

ok = True
g = torch.Generator().manual_seed(5)
# ragged random batches: sizes across the wide shape, a few tiny / empty clouds
for B, K, top in ((24, 256, 30000), (40, 512, 65536), (17, 1024, 9000), (12, 300, 131072), (33, 128, 70000)):
    sizes = torch.randint(2048, top, (B,), generator=g)
    sizes[0] = 0; sizes[1] = 5; sizes[2] = K - 1; sizes[-1] = top
    off = torch.zeros(B + 1, dtype=torch.int64); off[1:] = torch.cumsum(sizes, 0)
    xyz = torch.rand(int(off[-1]), 3, generator=g).to(dev)
    xyz[:, 2] *= 0.2
    for mode in (0, 1):
        ok &= cmp(f"random top {top}", xyz, off.to(dev), K, mode)
# stage-1 output
H, W = 480, 640
intr = synth.intrinsics(H, W, 605.0)
p = _ffi.make_params(intr, intr, apply_crop=True)
for frames in (48, 300):
    d, c = synth.episode(frames, 0, H, W)
    s1 = ops.Stage1(frames, H, W, dev, want=("xyz32",))
    out, off = s1.run(torch.from_numpy(d).to(dev), torch.from_numpy(c).to(dev), p)
    torch.cuda.synchronize()
    total = int(off[-1]); xyz = out["xyz32"][:total].clone()
    for K in (1024, 2500):
        ok &= cmp(f"stage-1 {frames} frames, forced", xyz, off, K)
        if frames == 300:
            ok &= cmp("stage-1 300 frames, by batch size", xyz, off, K, test_algo=0)
    if frames == 300:
        for algo in (5, 0, 10):
            for K in (1024, 2500):
                plan = ops.Fps(frames, total, H * W, K, dev, 1, algo)
                plan.run(xyz, off); torch.cuda.synchronize()
                ts = []
                for _ in range(7):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(); plan.run(xyz, off); e1.record(); torch.cuda.synchronize()
                    ts.append(e0.elapsed_time(e1))
                print(f"timing 300 clouds K={K} algo={algo}: min {min(ts):.3f} median {sorted(ts)[3]:.3f} ms", flush=True)
print("ALL OK" if ok else "FAILED")
sys.exit(0 if ok else 1)
