"""Dataset-level run of the hot path — BASELINE configs[3]: many episodes, sharded over one process per GPU, streamed
through EpisodeRunner (the loops XPC:96-125 + DP3:68-90 per trajectory, S1:55 / S2:34 over trajectories) and written as
the DP3 zarr store (CHK:179-200) by build_zarr.assemble — every rank compresses and writes the chunks its own rows
fill.  The episode source is pluggable: a directory tree (build_zarr.build) or, here, seeded synthetic episodes made in
memory (seed = episode * 1000 + frame), so a 30 k-frame run needs no 65 GB tree."""
import os
import time

import numpy as np

from . import _ffi, build_zarr, synth


class SyntheticEpisodes:
    """n_episodes x frames seeded frames of H x W (depth as raw u16 millimetres when depth_mm), with 7-D state /
    action rows in the extractor's format."""

    def __init__(self, n_episodes, frames=300, H=480, W=640, depth_mm=True, f=None):
        self.n, self.frames, self.H, self.W, self.depth_mm = int(n_episodes), int(frames), H, W, bool(depth_mm)
        self.intr = synth.intrinsics(H, W, f if f is not None else 605.0 * W / 640.0)
        self.lengths = [self.frames] * self.n

    def load(self, ep):
        mm, rgb = synth.episode_mm(self.frames, ep * 1000, self.H, self.W)
        depth = mm if self.depth_mm else (mm.astype(np.float32) / 1000.0).astype(np.float32)
        state = np.arange(self.frames * 7, dtype=np.float64).reshape(self.frames, 7) * 0.001 + ep
        action = state * 0.5
        action[:, 6] = 30.0 * np.sin(np.arange(self.frames) + ep)
        return depth, rgb, state, action


def run(source, out_zarr, K=1024, chunk=150, device=None, compressor="blosc", threads=None, rgb_on_host=True, params=None,
        prefetch=4, warmup=True):
    """Stream every episode of `source` this rank owns through the GPU path and assemble the store.
    Returns dict(frames, episodes, seconds by phase, rates) — this rank's view; frames / episodes are global."""
    import torch
    from .episode import EpisodeRunner
    if not torch.cuda.is_available():
        raise RuntimeError("pcp needs a ROCm device (MI355X); there is no CPU fallback")
    ndev = torch.cuda.device_count()
    if device is None:
        device = f"cuda:{int(os.environ.get('LOCAL_RANK', '0')) % ndev}"     # rehearsals with more ranks than GPUs share devices
    torch.cuda.set_device(torch.device(device))
    if threads is None:
        threads = max(1, min(16, len(os.sched_getaffinity(0))))       # the GPU box gives one rank 16 CPUs
    p = params if params is not None else _ffi.make_params(source.intr, source.intr, apply_crop=True)
    runner = EpisodeRunner(source.H, source.W, K, p, device=device, chunk=chunk, depth_mm=source.depth_mm,
                           rgb_on_host=rgb_on_host, profile=True)
    if warmup:                                                  # first-touch of the pinned buffers, lazy module loads
        z = np.zeros((min(chunk, 8), source.H, source.W), np.uint16 if source.depth_mm else np.float32)
        runner.run_arrays(z, np.zeros(z.shape + (3,), np.uint8))
        for k in runner.stats:
            runner.stats[k] = 0 if isinstance(runner.stats[k], int) else 0.0
    t = dict(source_s=0.0, runner_s=0.0)
    from . import shard
    _, rank, world = shard._dist()
    mine = shard.assign_contiguous(source.lengths, world)[rank]        # the same deal assemble() makes
    ddt = np.uint16 if source.depth_mm else np.float32
    ready = {}
    # Chunks are ENCODED (Blosc-zstd or zlib) in the background while the GPU works on the next episodes, under the assumption that no
    # episode is skipped — then episode i starts at row sum(lengths[:i]) and a group's rows fill known chunks; assemble()
    # uses them only if the assumption held for every episode of every rank, and encodes the rest (group seams) itself.
    from concurrent.futures import ThreadPoolExecutor
    from . import zarr_store
    starts = np.concatenate([[0], np.cumsum(source.lengths)]).astype(np.int64)
    R = build_zarr.CHUNK_SIZE
    enc_pool = ThreadPoolExecutor(max_workers=max(1, threads))
    enc_futs = {}
    # Episodes are loaded a few at a time and streamed through the runner BACK TO BACK (one chunk stream, no drain at
    # episode boundaries: the upload of an episode's first chunk overlaps the kernels of the previous episode's last).
    for g0 in range(0, len(mine), max(1, int(prefetch))):
        group = mine[g0:g0 + max(1, int(prefetch))]
        t0 = time.perf_counter()
        loaded = [source.load(ep) for ep in group]
        t1 = time.perf_counter()

        def chunks():
            for depth, rgb, _, _ in loaded:
                for i in range(0, depth.shape[0], chunk):
                    yield (np.ascontiguousarray(depth[i:i + chunk], dtype=ddt), np.ascontiguousarray(rgb[i:i + chunk], dtype=np.uint8))

        rows, counts = runner.run_chunks(chunks())
        t["source_s"] += t1 - t0
        t["runner_s"] += time.perf_counter() - t1
        at = 0
        for ep, (depth, _, state, action) in zip(group, loaded):
            n = depth.shape[0]
            ready[ep] = (rows[at:at + n], counts[at:at + n], state, action)
            at += n
        del loaded
        if compressor is not None and (np.asarray(counts) > 0).all():
            g0 = int(starts[group[0]])                            # the group's rows are [g0, g0 + len(rows)) if nobody skips
            for c in range(-(-g0 // R), (g0 + rows.shape[0]) // R):
                enc_futs[c] = enc_pool.submit(zarr_store.encode_chunk, rows[c * R - g0:(c + 1) * R - g0], compressor)

    def produce(ep):
        rows, counts, state, action = ready.pop(ep)
        if (np.asarray(counts) == 0).any():                       # the reference's filter step dies on an empty crop (S2:15-18)
            print(f"Warning: episode {ep} has a frame with no point inside the workspace. Skipping.")
            return None
        a = action.copy()
        a[:, 6] = np.clip(a[:, 6] / 30.0, -1.0, 1.0)              # CHK:100-103
        return rows, a, state

    timing = {}
    t0 = time.perf_counter()
    frames, episodes = build_zarr.assemble(source.lengths, produce, out_zarr, compressor, threads, timing,
                                           encoded=(lambda: {c: f.result() for c, f in enc_futs.items()}, source.lengths))
    enc_pool.shutdown(wait=True)
    wall = time.perf_counter() - t0 + t["source_s"] + t["runner_s"]
    runner.close()
    st = runner.stats
    mine = st["frames"]
    out = dict(frames=frames, episodes=episodes, frames_this_rank=mine, wall_s=wall, source_s=t["source_s"],
               runner_s=t["runner_s"], zarr_write_s=timing.get("write_s", 0.0),
               staging_s=st["stage_s"], h2d_s=st["h2d_ms"] / 1e3, kernels_s=st["kernels_ms"] / 1e3,
               d2h_s=st["d2h_ms"] / 1e3, gpu_wait_s=st["wait_s"], host_rows_s=st["host_rows_s"],
               chunks_written_here=timing.get("chunks_written_here", 0), boundary_pieces_sent=timing.get("boundary_pieces_sent", 0),
               chunks_encoded_ahead=timing.get("chunks_encoded_ahead", 0))
    out["frames_per_s_runner_this_rank"] = mine / t["runner_s"] if t["runner_s"] > 0 else 0.0
    out["frames_per_s_wall"] = frames / wall if wall > 0 else 0.0
    return out
