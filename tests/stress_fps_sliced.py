#!/usr/bin/env python3
"""Soak for the quantum-sliced FPS launches (run by hand on the MI355X; tests/test_gpu_soak_slice.py runs a slice):
batches of 257-336 clouds (the sizes pcp_fps_batched slices by itself) and forced slicing (algo 10) on smaller ones,
random K and distance mode, cloud sizes within one launch shape or across two or three of them (one any-shape launch), every launch repeated while a second stream keeps some CUs busy with unrelated work (uneven
load moves the quantum ends and with them who hands which cloud to whom).  Every cloud of every repeat must equal the
resident launch's result (algo 5), which is itself checked against the C oracle on a sample of clouds."""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "point-cloud-processing_amd"))
import numpy as np
import torch
from oracle import cbind
from pcp import ops


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=120)
    ap.add_argument("--seed", type=int, default=0)
    a = ap.parse_args(argv)
    rng = np.random.default_rng(a.seed)
    g = torch.Generator(device="cuda").manual_seed(a.seed)
    side = torch.cuda.Stream()
    noise = torch.rand(2048, 2048, device="cuda")
    t_end, n_batches, n_launches, n_clouds = time.time() + a.seconds, 0, 0, 0
    while time.time() < t_end:
        auto = rng.random() < 0.6
        B = int(rng.integers(257, 337)) if auto else int(rng.integers(2, 120))
        if auto and rng.random() < 0.15:
            B = int(rng.integers(513, 625))                             # two launches over the halves of the batch
        # tops beyond 65,536 make mixed-size batches: those go as ONE any-shape sliced launch
        top = int(rng.choice([6000, 30000, 65536, 100000, 150000])) if not auto else int(rng.choice([8000, 30000, 65536, 90000, 140000]))
        if auto and top > 131072 and rng.random() < 0.5:
            B = int(rng.integers(337, 513))                             # mixed batch of 337-512 clouds: one any-shape sliced launch
        sizes = rng.integers(2048, top + 1, B)
        sizes[rng.integers(0, B)] = top
        if rng.random() < 0.3:
            sizes[rng.integers(0, B)] = int(rng.choice([0, 1, 7, 63]))
        off = np.zeros(B + 1, np.int64); off[1:] = np.cumsum(sizes)
        total = int(off[-1])
        xyz = torch.rand(total, 3, device="cuda", generator=g)
        xyz[:, 2] *= float(rng.choice([0.02, 0.3, 1.0]))                # sheets and volumes
        if rng.random() < 0.3:
            xyz = (xyz * 64).floor() / 64                               # lattice: exact ties
        K = int(rng.choice([64, 256, 300, 1024, 1500, 2500]))
        dm = int(rng.integers(0, 2))
        offd = torch.from_numpy(off).cuda()
        ref = ops.Fps(B, total, top, K, "cuda", dm, 5)
        rx, ri = ref.run(xyz, offd)
        torch.cuda.synchronize()
        rx, ri = rx.clone(), ri.clone()
        xh = xyz.cpu().numpy()
        for b in rng.integers(0, B, 3):
            if sizes[b] == 0:
                continue
            ox, oi = cbind.fps(xh[off[b]:off[b + 1]], K, dm)
            if not (np.array_equal(ri[b].cpu().numpy(), oi) and np.array_equal(rx[b].cpu().numpy(), ox)):
                print(f"MISMATCH (resident launch vs oracle) seed={a.seed} batch={n_batches} cloud={b}", flush=True)
                sys.exit(1)
        plan = ops.Fps(B, total, top, K, "cuda", dm, 0 if auto else 10)
        for rep in range(6):
            if rep % 2:
                with torch.cuda.stream(side):                           # uneven load on some CUs
                    for _ in range(int(rng.integers(1, 6))):
                        noise = torch.tanh(noise @ noise * 1e-3)
            plan.idx.fill_(-7)
            sx, si = plan.run(xyz, offd)
            torch.cuda.synchronize()
            if not (bool((si == ri).all()) and bool((sx.view(torch.int32) == rx.view(torch.int32)).all())):
                rows = (si != ri).any(dim=1).nonzero().flatten().tolist()
                print(f"MISMATCH seed={a.seed} batch={n_batches} rep={rep} B={B} K={K} dm={dm} auto={auto} top={top} clouds {rows[:8]}",
                      flush=True)
                sys.exit(1)
            n_launches += 1
            n_clouds += B
        n_batches += 1
        if n_batches % 5 == 0:
            print(f"{n_batches} batches, {n_launches} sliced launches, {n_clouds} clouds ok", flush=True)
    print(f"done: {n_batches} batches, {n_launches} sliced launches, {n_clouds} clouds, all bit-exact", flush=True)


if __name__ == "__main__":
    main()
