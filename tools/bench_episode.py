#!/usr/bin/env python3
"""PCIe-inclusive rate of the hot path: a 300-frame synthetic episode that starts in HOST memory, streamed
through pcp.episode.EpisodeRunner (pinned staging, H2D on a copy stream under the previous chunk's kernels,
(T,K,6) rows back on the host).  Never the headline `value` (bench.py times HBM-resident input)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "point-cloud-processing_amd")]
import numpy as np, torch
from pcp import _ffi, synth
from pcp.episode import EpisodeRunner

H, W, K, T = 480, 640, 1024, 300
intr = synth.intrinsics(H, W)
p = _ffi.make_params(intr, intr, apply_crop=True)
depth, rgb = synth.episode(T, seed0=0, H=H, W=W)
mm = np.rint(depth * 1000).astype(np.uint16)
out = {}
for name, kw, d in (("f32_depth_rgb_uploaded", {"rgb_on_host": False}, depth), ("u16_depth_rgb_uploaded", {"depth_mm": True, "rgb_on_host": False}, mm),
                    ("f32_depth", {}, depth), ("u16_depth", {"depth_mm": True}, mm)):
    for chunk in (50, 100, 150, 300):
        r = EpisodeRunner(H, W, K, p, chunk=chunk, **kw)
        r.run_arrays(d, rgb)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            rows, _ = r.run_arrays(d, rgb)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 3
        out[f"{name}_chunk{chunk}"] = {"ms_per_episode": round(dt * 1e3, 2), "frames_per_s": round(T / dt, 1)}
        del r
# steady state: four episodes back to back as ONE chunk stream (dataset mode), no drain at the episode boundaries
for name, kw, d in (("u16_depth", {"depth_mm": True}, mm), ("f32_depth", {}, depth)):
    for chunk in (100, 150):
        r = EpisodeRunner(H, W, K, p, chunk=chunk, **kw)
        gen = lambda: ((d[i:i + chunk], rgb[i:i + chunk]) for _ in range(4) for i in range(0, T, chunk))
        r.run_chunks(gen())
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r.run_chunks(gen())
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        out[f"{name}_chunk{chunk}_stream_of_4_episodes"] = {"ms_per_episode": round(dt * 1e3 / 4, 2), "frames_per_s": round(4 * T / dt, 1)}
        del r
print(json.dumps(out), flush=True)
