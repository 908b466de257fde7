#!/usr/bin/env python3
"""Developer probe: pcp_fps_batched on every cloud of an episode ALONE (B = 1), to see how the per-cloud cost spreads.
   python tools/fps_per_cloud.py --scene tabletop --frames 300 --K 1024"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "point-cloud-processing_amd"))
import numpy as np, torch
from pcp import _ffi, ops, synth

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=300)
ap.add_argument("--K", type=int, default=1024)
ap.add_argument("--scene", choices=["random", "tabletop"], default="tabletop")
a = ap.parse_args()
dev = "cuda"
H, W = 480, 640
intr = synth.intrinsics(H, W)
p = _ffi.make_params(intr, intr, apply_crop=True)
d, c = (synth.episode if a.scene == "random" else synth.scene_episode)(a.frames, 0, H, W)
s1 = ops.Stage1(a.frames, H, W, dev, want=("xyz32",))
out, off = s1.run(torch.from_numpy(d).to(dev), torch.from_numpy(c).to(dev), p)
torch.cuda.synchronize()
offh = off.cpu().numpy()
xyz = out["xyz32"][:int(offh[-1])].clone()
M = np.diff(offh)
ts = []
for b in range(a.frames):
    x = xyz[offh[b]:offh[b + 1]].contiguous()
    o = torch.tensor([0, int(M[b])], dtype=torch.int64, device=dev)
    plan = ops.Fps(1, int(M[b]), H * W, a.K, dev, 1, 0)
    plan.run(x, o); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); plan.run(x, o); e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1))
ts = np.array(ts)
print("M: min %d mean %.0f max %d" % (M.min(), M.mean(), M.max()))
print("ms per cloud alone: min %.3f mean %.3f p90 %.3f max %.3f  sum/256 %.3f" % (ts.min(), ts.mean(), np.percentile(ts, 90), ts.max(), ts.sum() / 256))
order = np.argsort(ts)
print("slowest:", [(int(i), int(M[i]), round(float(ts[i]), 3)) for i in order[-5:]])
print("fastest:", [(int(i), int(M[i]), round(float(ts[i]), 3)) for i in order[:5]])
