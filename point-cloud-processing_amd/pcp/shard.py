"""Episode sharding for multi-GPU runs.  Frames and episodes are independent, so
there is no data-path collective: each rank processes whole episodes on its own GPU and writes the zarr
chunks its own rows fill; only the few rows of chunks that straddle two ranks' episodes (and the small
state / action arrays) travel to rank 0, host-side.  Rows land in sorted-episode order, which keeps
`meta/episode_ends` (CHK:151-152,200) identical to a single-process run."""
import os

import numpy as np


def assign_episodes(lengths, world_size):
    """Longest-first greedy: returns a list (per rank) of episode indices, each sorted ascending."""
    order = sorted(range(len(lengths)), key=lambda i: (-int(lengths[i]), i))
    load = [0] * world_size
    mine = [[] for _ in range(world_size)]
    for i in order:
        r = min(range(world_size), key=lambda k: (load[k], k))
        mine[r].append(i)
        load[r] += int(lengths[i])
    return [sorted(m) for m in mine]


def assign_contiguous(lengths, world_size):
    """Contiguous blocks of episodes with balanced frame counts: rank r gets episodes [cut[r], cut[r+1]).
    Neighbouring episodes then share a rank, so almost every 100-frame zarr chunk is filled by ONE rank and can be
    compressed and written where its rows were computed (write_rows_sharded)."""
    n = len(lengths)
    csum = np.concatenate([[0], np.cumsum([int(x) for x in lengths])]).astype(np.int64)
    total = int(csum[-1])
    cuts = [0]
    for r in range(1, world_size):
        target = total * r / world_size
        c = int(np.searchsorted(csum, target, side="left"))
        # the boundary nearest to the target, never before the previous cut
        if c > 0 and abs(csum[c - 1] - target) <= abs(csum[min(c, n)] - target):
            c -= 1
        cuts.append(min(max(c, cuts[-1]), n))
    cuts.append(n)
    return [list(range(cuts[r], cuts[r + 1])) for r in range(world_size)]


def _dist(group=None):
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        return dist, dist.get_rank(group), dist.get_world_size(group)
    return None, 0, 1


def gather_to_root(obj, group=None, dst=0):
    """Python object of every rank → list on rank `dst` (None elsewhere).  For SMALL host-side metadata only."""
    dist, rank, world = _dist(group)
    if dist is None:
        return [obj]
    out = [None] * world if rank == dst else None
    dist.gather_object(obj, out, dst=dst, group=group)
    return out


def all_gather_small(obj, group=None):
    """Small Python object of every rank → list on every rank (row counts, shapes)."""
    dist, rank, world = _dist(group)
    if dist is None:
        return [obj]
    out = [None] * world
    dist.all_gather_object(out, obj, group=group)
    return out


def barrier(group=None):
    dist, _, _ = _dist(group)
    if dist is not None:
        dist.barrier(group=group)


def gather_episodes(local, n_episodes, group=None, dst=0):
    """local: {episode_index: ndarray (T_e,K,6)} held by this rank.  Rank `dst` returns the list of all episodes'
    arrays in episode order, the other ranks None (a gather to ONE rank; write_rows_sharded avoids even that for
    the point-cloud rows of a dataset build)."""
    parts = gather_to_root(local, group, dst)
    if parts is None:
        return None
    merged = {}
    for p in parts:
        merged.update(p)
    missing = [i for i in range(n_episodes) if i not in merged]
    if missing:
        raise RuntimeError(f"episodes {missing} were processed by no rank")
    return [merged[i] for i in range(n_episodes)]


def stack_for_zarr(episodes):
    """CHK:151-162: concatenate per-episode blocks and derive cumulative episode_ends (int64)."""
    ends = np.cumsum([e.shape[0] for e in episodes]).astype(np.int64)
    return np.concatenate(episodes, axis=0).astype(np.float32), ends


def write_rows_sharded(path, local, counts, tail_shape, chunk_rows=100, dtype="<f4", compressor="blosc", threads=8,
                       group=None, encoded=None):
    """Write ONE zarr array of shape (sum(counts),) + tail_shape, chunked (chunk_rows,) + tail_shape (CHK:188), whose
    rows are the concatenation of the episodes' blocks in episode order, from blocks that live on different ranks.
      local  {episode: ndarray (counts[episode],) + tail_shape} — this rank's episodes
      counts rows of EVERY episode (0 = skipped), the same list on every rank
    Every rank compresses and writes the chunks that its own rows fill completely (thread pool; the codecs release the
    GIL).  Pieces of chunks shared with another rank go to rank 0, which assembles those few chunks and writes
    .zarray.  The directory must exist on a file system all ranks see (one node, BASELINE configs[3]).
    encoded: {chunk index: bytes} — chunks of this rank that the caller has ALREADY encoded (same rows, same chunk grid,
    same compressor: e.g. compressed in the background while the GPU worked on later episodes); they are only written.
    Returns (chunks written here, boundary pieces sent)."""
    from concurrent.futures import ThreadPoolExecutor
    from . import zarr_store
    dist, rank, world = _dist(group)
    counts = [int(c) for c in counts]
    starts = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
    total = int(starts[-1])
    dt = np.dtype(dtype)
    tail_shape = tuple(int(t) for t in tail_shape)
    if rank == 0:
        zarr_store.write_zarray(path, (total,) + tail_shape, (chunk_rows,) + tail_shape, dtype, compressor)
    barrier(group)
    R = max(1, min(int(chunk_rows), total)) if total > 0 else 1          # zarr clamps the chunk edge to the array edge
    # overlaps of my episodes with the chunk grid
    per_chunk = {}
    for ep, block in local.items():
        n = counts[ep]
        if n == 0:
            continue
        if block.shape[0] != n or tuple(block.shape[1:]) != tail_shape:
            raise ValueError(f"episode {ep}: block {block.shape} does not match count {n} / tail {tail_shape}")
        g0 = int(starts[ep])
        for c in range(g0 // R, (g0 + n - 1) // R + 1):
            lo, hi = max(g0, c * R), min(g0 + n, (c + 1) * R)
            per_chunk.setdefault(c, []).append((lo - c * R, block[lo - g0:hi - g0]))
    full, pieces = [], []
    for c, parts in per_chunk.items():
        need = min(R, total - c * R)
        if sum(p.shape[0] for _, p in parts) == need:
            full.append((c, parts))
        else:
            pieces.extend((c, off, np.ascontiguousarray(p, dtype=dt)) for off, p in parts)
    zeros = (0,) * len(tail_shape)

    def write_one(item):
        c, parts = item
        if encoded is not None and c in encoded:
            zarr_store.write_chunk_bytes(path, (c,) + zeros, encoded[c])
            return
        if len(parts) == 1 and parts[0][1].shape[0] == R and parts[0][1].dtype == dt:
            blockc = parts[0][1]
        else:
            blockc = np.zeros((R,) + tail_shape, dtype=dt)                # an edge chunk keeps zero padding
            for off, p in parts:
                blockc[off:off + p.shape[0]] = p
        zarr_store.write_chunk(path, (c,) + zeros, blockc, compressor)

    if threads > 1 and len(full) > 1:
        with ThreadPoolExecutor(max_workers=threads) as pool:
            list(pool.map(write_one, full))
    else:
        for item in full:
            write_one(item)
    got = gather_to_root(pieces, group)
    if got is not None:
        shared = {}
        for plist in got:
            for c, off, p in plist:
                shared.setdefault(c, []).append((off, p))
        for c, parts in shared.items():
            need = min(R, total - c * R)
            if sum(p.shape[0] for _, p in parts) != need:
                raise RuntimeError(f"chunk {c} of {os.path.basename(path)} is covered by {sum(p.shape[0] for _, p in parts)} "
                                   f"of {need} rows: an episode was processed by no rank")
            write_one((c, parts))
    barrier(group)
    return len(full), len(pieces)
