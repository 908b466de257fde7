#!/usr/bin/env python3
"""BASELINE configs[3] on the GPUs at hand: E synthetic episodes x T frames of 640x480 (seed = episode*1000 + frame,
made in memory) → sharded over the ranks → EpisodeRunner → DP3 zarr store, with a host / PCIe / kernel / zarr breakdown.
   python tools/bench_dataset.py --episodes 6 --frames 300 --out /tmp/ds.zarr [--json profiles/r03_dataset.json]
   torchrun --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 tools/bench_dataset.py ...   (one rank per GPU; ranks
   beyond the GPU count share devices — a rehearsal, not a measurement)
Prints ONE JSON line on rank 0.  "source_s" is the synthesis that stands in for file decode; it is excluded from
frames_per_s_pipeline and included in frames_per_s_wall."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "point-cloud-processing_amd")]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--episodes", type=int, default=6)
    ap.add_argument("--frames", type=int, default=300)
    ap.add_argument("--K", type=int, default=1024)
    ap.add_argument("--chunk", type=int, default=150)
    ap.add_argument("--H", type=int, default=480)
    ap.add_argument("--W", type=int, default=640)
    ap.add_argument("--out", required=True)
    ap.add_argument("--json", default=None)
    ap.add_argument("--f32-depth", action="store_true", help="upload float32 metres instead of raw u16 millimetres")
    ap.add_argument("--device-rgb", action="store_true", help="upload rgb and gather colours on the device")
    ap.add_argument("--compressor", default="blosc", choices=["blosc", "zlib", "none"])
    a = ap.parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("gloo")                 # host-side metadata only; no data-path collective
    from pcp import dataset
    src = dataset.SyntheticEpisodes(a.episodes, a.frames, a.H, a.W, depth_mm=not a.f32_depth)
    res = dataset.run(src, a.out, K=a.K, chunk=a.chunk, compressor=None if a.compressor == "none" else a.compressor,
                      rgb_on_host=not a.device_rgb)
    if world > 1:
        parts = [None] * world
        dist.all_gather_object(parts, res)
    else:
        parts = [res]
    if rank == 0:
        import torch
        wall = max(p["wall_s"] for p in parts)
        pipe = max(p["wall_s"] - p["source_s"] for p in parts)
        line = dict(metric="frames/sec, dataset mode (episodes -> zarr)", frames=res["frames"], episodes=res["episodes"],
                    n_ranks=world, n_gpus=min(world, torch.cuda.device_count()), K=a.K, chunk=a.chunk, compressor=a.compressor,
                    depth="u16 mm" if not a.f32_depth else "f32 m", rgb="host" if not a.device_rgb else "device",
                    frames_per_s_wall=res["frames"] / wall, frames_per_s_pipeline=res["frames"] / pipe,
                    breakdown_s_max_over_ranks={k: max(p[k] for p in parts) for k in
                                                ("source_s", "staging_s", "h2d_s", "kernels_s", "d2h_s", "gpu_wait_s",
                                                 "host_rows_s", "runner_s", "zarr_write_s", "wall_s")},
                    per_rank=[{k: (round(v, 4) if isinstance(v, float) else v) for k, v in p.items()} for p in parts])
        print(json.dumps(line))
        if a.json:
            with open(a.json, "w") as fh:
                json.dump(line, fh, indent=1)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
