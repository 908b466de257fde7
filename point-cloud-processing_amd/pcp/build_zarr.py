#!/usr/bin/env python3
"""Direct DP3 zarr assembly: what chunking_data.py produces, built either from the
filtered PLYs (the reference's route, CHK:121-144) or straight from GPU results, skipping the PLY
round trip.  State/action handling restates CHK:73-103; the zarr layout is CHK:179-200."""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pcp import ply, zarr_store  # noqa: E402

CHUNK_SIZE = 100            # CHK:16
FRAME_SKIP = 1              # CHK:18
NORMALIZE_GRIPPER = True    # CHK:21
USE_JOINT_DELTAS = False    # CHK:25


def load_states_actions(traj_extracted, frame_skip=FRAME_SKIP, normalize_gripper=NORMALIZE_GRIPPER,
                        use_joint_deltas=USE_JOINT_DELTAS):
    """CHK:73-103 → (state (n,7), action (n,7), state_indices)."""
    action_raw = np.atleast_2d(np.loadtxt(os.path.join(traj_extracted, "actions.txt")))
    state_raw = np.atleast_2d(np.loadtxt(os.path.join(traj_extracted, "states.txt")))
    s_idx = np.arange(0, len(state_raw), frame_skip)
    a_idx = np.minimum(s_idx + frame_skip - 1, len(action_raw) - 1)                  # CHK:86-87
    state, action = state_raw[s_idx], action_raw[a_idx].copy()
    if use_joint_deltas and action.shape[1] >= 7 and state.shape[1] >= 6:
        action[:, :6] = action[:, :6] - state[:, :6]                                 # CHK:93-97
    if normalize_gripper and action.shape[1] >= 7:
        action[:, 6] = np.clip(action[:, 6] / 30.0, -1.0, 1.0)                       # CHK:100-103
    return state, action, s_idx


def rows_from_filtered_ply(traj_pc_path, state_indices):
    """CHK:123-144: (n,K,6) float64 rows [xyz, colours*255] from pc_*_filtered.ply."""
    files = sorted(f for f in os.listdir(traj_pc_path) if f.endswith("_filtered.ply"))
    files = [files[i] for i in state_indices if i < len(files)]
    out = []
    for f in files:
        pts, col = ply.read_point_cloud(os.path.join(traj_pc_path, f))
        out.append(np.concatenate([pts, (col * 255) if col is not None else np.zeros_like(pts)], axis=1))
    return out


def _count_states(extracted):
    with open(os.path.join(extracted, "states.txt")) as fh:
        return sum(1 for line in fh if line.strip())


def _process_traj(data_root, traj, source, K, frame_skip, intr, ext, runner_box):
    """One trajectory → (rows list, action, state) or None when it is skipped (CHK:67-70, 127-132)."""
    extracted = os.path.join(data_root, traj, "extracted_data", traj)
    pc_path = os.path.join(data_root, traj, "filtered_point_clouds")
    state, action, s_idx = load_states_actions(extracted, frame_skip)
    if source == "ply":
        rows = rows_from_filtered_ply(pc_path, s_idx)
    else:
        from pcp import _ffi, extract_point_clouds_one_cam as xpc
        from pcp.episode import EpisodeRunner
        intr_t = intr or xpc.extract_intrinsics(extracted, extracted)
        d0 = np.load(os.path.join(extracted, "camera", "depth", "depth_00000.npy"))
        runner = runner_box.get("runner")
        if runner is None or (runner.H, runner.W) != d0.shape:
            p = _ffi.make_params(intr_t["depth"], intr_t["color"], T=xpc.T_DEPTH_TO_COLOR, ext=ext, apply_crop=True)
            runner = runner_box["runner"] = EpisodeRunner(d0.shape[0], d0.shape[1], K, p, device=runner_box.get("device", "cuda"))
        all_rows, counts, ids = runner.run_tree(extracted)
        # The reference's route writes one filtered PLY per frame that has both input files (XPC:100-101) until a
        # frame crops to zero points: pytorch3d then raises inside DP3:47, the bag's child process dies
        # (S2:15-18,45) and the later frames never get a file.  CHK:123-125 then pairs states with the sorted
        # file list BY POSITION, and CHK:127-132 skips the trajectory when the counts differ.  Same here: the
        # rows that "exist" are those before the first empty crop, paired by position.
        empty = np.nonzero(np.asarray(counts) == 0)[0]
        n_ok = int(empty[0]) if len(empty) else len(all_rows)
        if len(empty):
            print(f"Error: frame {ids[n_ok]} of {traj} has no point inside the workspace "
                  f"(the reference's filter step fails there, S2:15-18)")
        rows = [all_rows[i] for i in s_idx if i < n_ok]
    if len(rows) != state.shape[0]:
        print(f"Warning: {traj} has {state.shape[0]} states but {len(rows)} point clouds. Skipping.")   # CHK:127-132
        return None
    return rows, action, state


def assemble(lengths, produce, out_zarr, compressor="blosc", threads=8, timing=None, encoded=None):
    """The dataset-level loop of BASELINE configs[3] (S2:34-45 + CHK:47-200) for any episode source.
      lengths[i]  expected frame count of episode i (only used to balance the ranks)
      produce(i)  → None (episode skipped, CHK:67-70 / 127-132) or (rows (n,K,6), action (n,7), state (n,7))
    Under torch.distributed (one process per GPU) the episodes are dealt to the ranks in contiguous, frame-balanced
    blocks; every rank runs `produce` for its own, then compresses and writes the point-cloud chunks its rows fill
    (shard.write_rows_sharded) — no rank ever holds another rank's rows, nothing is stacked on rank 0, and the only
    data that travels is the rows of the few chunks straddling two ranks plus the (n,7) state / action arrays.
    Rank 0 writes .zarray / action / state / episode_ends.  The store equals a single-process run's byte for byte.
    encoded: optional ({chunk index: bytes} or a callable returning it, chunk_rows_assumed_counts) — point-cloud chunks
    this rank encoded ahead of time under the assumption that every episode i yields exactly assumed_counts[i] rows;
    used only if that turns out to be true for ALL episodes (the common case: no trajectory skipped).
    Returns (total frames, episodes kept)."""
    import time
    from pcp import shard
    dist, rank, world = shard._dist()
    t0 = time.perf_counter()
    mine = shard.assign_contiguous(lengths, world)[rank]
    local_rows, local_meta = {}, {}
    for i in mine:
        res = produce(i)
        if res is None:
            local_meta[i] = None
            continue
        rows, action, state = res
        rows = np.ascontiguousarray(np.asarray(rows, dtype=np.float32))      # CHK:195 stores float32
        local_rows[i] = rows
        local_meta[i] = (rows.shape, np.asarray(action), np.asarray(state))
    t1 = time.perf_counter()
    shapes = {}
    for part in shard.all_gather_small({i: (None if m is None else m[0]) for i, m in local_meta.items()}):
        shapes.update(part)
    missing = [i for i in range(len(lengths)) if i not in shapes]
    if missing:
        raise RuntimeError(f"episodes {missing} were processed by no rank")
    kept = [i for i in range(len(lengths)) if shapes[i] is not None]
    if not kept:
        raise RuntimeError("no complete trajectory found")
    tails = {tuple(shapes[i][1:]) for i in kept if shapes[i][0] > 0} or {tuple(shapes[kept[0]][1:])}
    if len(tails) != 1:
        raise ValueError(f"episodes disagree on the cloud shape: {sorted(tails)}")     # np.stack would fail, CHK:158
    counts = [0 if shapes[i] is None else int(shapes[i][0]) for i in range(len(lengths))]
    if rank == 0:
        zarr_store.create_dp3_layout(out_zarr)
    shard.barrier()
    pre = None
    if encoded is not None:
        get, assumed = encoded
        if [int(c) for c in assumed] == counts:
            pre = get() if callable(get) else get
    wrote = shard.write_rows_sharded(os.path.join(out_zarr, "data", "point_cloud"), local_rows, counts, tails.pop(),
                                     CHUNK_SIZE, "<f4", compressor, threads, encoded=pre)
    metas = shard.gather_to_root({i: (m[1], m[2]) for i, m in local_meta.items() if m is not None})
    if metas is not None:
        merged = {}
        for part in metas:
            merged.update(part)
        ends = np.cumsum([counts[i] for i in kept]).astype(np.int64)                    # CHK:151-152
        zarr_store.write_dp3_small_arrays(out_zarr, np.vstack([merged[i][0] for i in kept]),
                                          np.vstack([merged[i][1] for i in kept]), ends, CHUNK_SIZE, compressor)
    shard.barrier()
    if timing is not None:
        timing.update(produce_s=t1 - t0, write_s=time.perf_counter() - t1, chunks_written_here=wrote[0],
                      chunks_encoded_ahead=0 if pre is None else len(pre),
                      boundary_pieces_sent=wrote[1], episodes_here=len(mine))
    return int(sum(counts)), len(kept)


def build(data_root, out_zarr, source="ply", K=2500, frame_skip=FRAME_SKIP, compressor="blosc", intr=None,
          ext=None):
    """source='ply': read <traj>/filtered_point_clouds (the reference's route).
    source='gpu': run depth+rgb of <traj>/extracted_data/<traj> through the GPU path directly.
    Under torch.distributed (one process per GPU, BASELINE configs[3]) see assemble()."""
    import torch.distributed as dist
    distributed = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
    rank = dist.get_rank() if distributed else 0
    trajs = []
    for traj in sorted(d for d in os.listdir(data_root) if os.path.isdir(os.path.join(data_root, d))):
        extracted = os.path.join(data_root, traj, "extracted_data", traj)
        need = [os.path.join(extracted, "states.txt"), os.path.join(extracted, "actions.txt"),
                os.path.join(extracted, "camera", "rgb")]
        if source == "ply":
            need.append(os.path.join(data_root, traj, "filtered_point_clouds"))
        if not all(os.path.exists(p) for p in need):
            if rank == 0:
                print(f"Skipping {traj}: Missing one or more data components.")        # CHK:67-70
            continue
        trajs.append(traj)
    lengths = [_count_states(os.path.join(data_root, t, "extracted_data", t)) for t in trajs]
    runner_box = {}
    if source == "gpu":
        import torch
        ndev = torch.cuda.device_count()
        if ndev == 0:
            raise RuntimeError("pcp needs a ROCm device (MI355X); there is no CPU fallback")
        local_dev = int(os.environ.get("LOCAL_RANK", "0")) % ndev       # rehearsals with more ranks than GPUs share devices
        torch.cuda.set_device(local_dev)                                # libpcp launches on the current device
        runner_box["device"] = f"cuda:{local_dev}"

    def produce(i):
        res = _process_traj(data_root, trajs[i], source, K, frame_skip, intr, ext, runner_box)
        if res is None:
            return None
        rows, action, state = res
        rows = np.asarray(rows, dtype=np.float32) if len(rows) else np.zeros((0, K, 6), np.float32)
        return rows, action, state

    return assemble(lengths, produce, out_zarr, compressor)


def main():
    ap = argparse.ArgumentParser(description="Assemble the DP3 zarr dataset")
    ap.add_argument("--data_root", required=True)
    ap.add_argument("--out_zarr", required=True)
    ap.add_argument("--source", choices=("ply", "gpu"), default="ply")
    ap.add_argument("--num_points", type=int, default=2500)
    ap.add_argument("--compressor", choices=("blosc", "zlib", "none"), default="blosc",
                    help="blosc = the reference's Blosc-zstd (chunking_data.py:180); zlib; none")
    a = ap.parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:                                   # launched by torch.distributed.run: one process per GPU
        import torch.distributed as dist
        dist.init_process_group("gloo")             # host-side object gather only; no data-path collective
    n, e = build(a.data_root, a.out_zarr, a.source, a.num_points,
                 compressor=None if a.compressor == "none" else a.compressor)
    if int(os.environ.get("RANK", "0")) == 0:
        print(f"Saved Zarr file: {a.out_zarr}  ({n} frames, {e} episodes)")
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
