"""Episode sharding for multi-GPU runs.  Frames and episodes are independent, so
there is no data-path collective: each rank processes whole episodes and the (T_e,K,6) blocks are
gathered host-side in sorted-episode order, which keeps `meta/episode_ends` (CHK:151-152,200)
identical to a single-process run."""
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


def gather_episodes(local, n_episodes, group=None):
    """local: {episode_index: ndarray (T_e,K,6)} held by this rank.  Every rank returns the list of all
    episodes' arrays in episode order (torch.distributed object gather; gloo or nccl/RCCL)."""
    import torch.distributed as dist
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        merged = dict(local)
    else:
        parts = [None] * dist.get_world_size(group)
        dist.all_gather_object(parts, local, group=group)
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
