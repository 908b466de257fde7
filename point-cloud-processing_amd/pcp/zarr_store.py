"""Minimal zarr-v2 DirectoryStore writer/reader for the DP3 dataset layout that the reference's
chunking_data.py creates (CHK:179-200):

    <root>/data/point_cloud  (T,K,6) float32   chunks (100,K,6)
    <root>/data/action       (T,7)   float32   chunks (100,7)
    <root>/data/state        (T,7)   float32   chunks (100,7)
    <root>/meta/episode_ends (E,)    int64     chunks (E,)

zarr / numcodecs are not installed here, so the store is written by hand.  Compressors: "blosc" = the
reference's `zarr.Blosc(cname='zstd', clevel=3, shuffle=1)` (CHK:180) — Blosc-1 frames written by
`blosc_frame.py` around pyarrow's zstd codec, the default of the dataset builders; "zlib" = the zarr-v2 built-in
codec (no dependency beyond the standard library); None = raw chunks.  Array names, shapes, dtypes,
chunking, C order and fill value are the reference's."""
import json
import os
import shutil
import zlib

import numpy as np

from . import blosc_frame

BLOSC = {"id": "blosc", "cname": "zstd", "clevel": 3, "shuffle": 1, "blocksize": 0}     # CHK:180 as numcodecs writes it


def _compressor_meta(compressor):
    if compressor in (None, "none"):
        return None
    if compressor == "zlib":
        # level 1: the rows are float32 coordinates (high-entropy mantissas), level 3 gains 1-2 % of size for twice the
        # CPU time, and compression is what bounds a dataset-mode run once the GPU path is in place
        return {"id": "zlib", "level": 1}
    if compressor == "blosc":
        return dict(BLOSC)
    raise ValueError("compressor must be None, 'zlib' or 'blosc'")


def _write_json(path, obj):
    with open(path, "w") as fh:
        json.dump(obj, fh, indent=4, sort_keys=True)


def create_group(path):
    os.makedirs(path, exist_ok=True)
    _write_json(os.path.join(path, ".zgroup"), {"zarr_format": 2})


def chunk_grid(shape, chunks):
    """Effective chunk shape (zarr clamps a chunk edge to the array edge, never below 1) and chunks per axis."""
    chunks = tuple(int(min(c, s)) if s > 0 else int(max(c, 1)) for c, s in zip(chunks, shape))
    chunks = tuple(max(c, 1) for c in chunks)
    return chunks, [-(-s // c) for s, c in zip(shape, chunks)]


def write_zarray(path, shape, chunks, dtype, compressor="blosc"):
    """Create the array directory and its .zarray; returns (meta, effective chunk shape)."""
    chunks, _ = chunk_grid(shape, chunks)
    os.makedirs(path, exist_ok=True)
    dt = np.dtype(dtype)
    meta = {"chunks": list(chunks), "compressor": _compressor_meta(compressor), "dtype": dt.str,
            "fill_value": 0 if dt.kind in "iu" else 0.0, "filters": None, "order": "C",
            "shape": [int(n) for n in shape], "zarr_format": 2}
    _write_json(os.path.join(path, ".zarray"), meta)
    return meta, chunks


def encode_chunk(block, compressor="blosc"):
    """The bytes of one chunk file from a FULL-chunk-shaped block (edge chunks zero-padded by the caller)."""
    block = np.ascontiguousarray(block)
    raw = block.tobytes()
    comp = _compressor_meta(compressor)
    if comp and comp["id"] == "blosc":                          # numcodecs hands Blosc the array's item size as type size
        raw = blosc_frame.compress(raw, block.dtype.itemsize, comp["clevel"], comp["shuffle"])
    elif comp:
        raw = zlib.compress(raw, comp["level"])                 # both release the GIL: chunks compress in parallel threads
    return raw


def write_chunk_bytes(path, cidx, raw):
    with open(os.path.join(path, ".".join(str(int(i)) for i in cidx)), "wb") as fh:
        fh.write(raw)


def write_chunk(path, cidx, block, compressor="blosc"):
    """One chunk file "i.j.k" from a FULL-chunk-shaped C-contiguous block."""
    write_chunk_bytes(path, cidx, encode_chunk(block, compressor))


def write_array(path, data, chunks, dtype, compressor="blosc", threads=1):
    """One zarr-v2 array: .zarray + one file per chunk ("i.j.k"), edge chunks zero-padded.  threads > 1 compresses and
    writes the chunks on a thread pool."""
    data = np.asarray(data)
    if data.dtype != np.dtype(dtype) or not data.flags.c_contiguous:
        data = np.ascontiguousarray(data.astype(dtype))
    meta, chunks = write_zarray(path, data.shape, chunks, dtype, compressor)
    dt = np.dtype(dtype)
    _, grid = chunk_grid(data.shape, chunks)

    def one(cidx):
        sl = tuple(slice(i * c, min((i + 1) * c, s)) for i, c, s in zip(cidx, chunks, data.shape))
        part = data[sl]
        if part.shape == chunks:
            block = part
        else:
            block = np.zeros(chunks, dtype=dt)
            block[tuple(slice(0, n) for n in part.shape)] = part
        write_chunk(path, cidx, block, compressor)

    cells = list(np.ndindex(*grid))
    if threads > 1 and len(cells) > 1:
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=threads) as pool:
            list(pool.map(one, cells))
    else:
        for c in cells:
            one(c)
    return meta


def read_array(path):
    with open(os.path.join(path, ".zarray")) as fh:
        meta = json.load(fh)
    dt, shape, chunks = np.dtype(meta["dtype"]), tuple(meta["shape"]), tuple(meta["chunks"])
    out = np.zeros(shape, dtype=dt)
    grid = [-(-s // c) for s, c in zip(shape, chunks)]
    for cidx in np.ndindex(*grid):
        fn = os.path.join(path, ".".join(str(i) for i in cidx))
        if not os.path.exists(fn):
            continue
        raw = open(fn, "rb").read()
        comp = meta["compressor"]
        if comp:
            if comp["id"] == "blosc":
                raw = blosc_frame.decompress(raw)
            elif comp["id"] == "zlib":
                raw = zlib.decompress(raw)
            else:
                raise ValueError(f"unsupported compressor {comp['id']}")
        block = np.frombuffer(raw, dtype=dt).reshape(chunks)
        sl = tuple(slice(i * c, min((i + 1) * c, s)) for i, c, s in zip(cidx, chunks, shape))
        out[sl] = block[tuple(slice(0, s.stop - s.start) for s in sl)]
    return out, meta


def create_dp3_layout(out_zarr):
    """CHK:175-187: wipe an existing store, create the root, `data` and `meta` groups."""
    if os.path.exists(out_zarr):
        print(f"Overwriting existing Zarr file: {out_zarr}")           # CHK:175-177
        shutil.rmtree(out_zarr)
    create_group(out_zarr)
    create_group(os.path.join(out_zarr, "data"))
    create_group(os.path.join(out_zarr, "meta"))


def write_dp3_small_arrays(out_zarr, action, state, episode_ends, chunk_size=100, compressor="blosc"):
    """data/action, data/state, meta/episode_ends (CHK:196-200)."""
    action, state = np.asarray(action), np.asarray(state)
    episode_ends = np.asarray(episode_ends, dtype=np.int64)
    write_array(os.path.join(out_zarr, "data", "action"), action, (chunk_size, action.shape[1]), "<f4", compressor)
    write_array(os.path.join(out_zarr, "data", "state"), state, (chunk_size, state.shape[1]), "<f4", compressor)
    write_array(os.path.join(out_zarr, "meta", "episode_ends"), episode_ends, (max(len(episode_ends), 1),),
                "<i8", compressor)                                                     # CHK:200


def write_dp3_dataset(out_zarr, point_cloud, action, state, episode_ends, chunk_size=100, compressor="blosc", threads=1):
    """CHK:175-200: overwrite <out_zarr> with the DP3 layout."""
    create_dp3_layout(out_zarr)
    point_cloud = np.asarray(point_cloud)
    write_array(os.path.join(out_zarr, "data", "point_cloud"), point_cloud,
                (chunk_size,) + tuple(point_cloud.shape[1:]), "<f4", compressor, threads)   # CHK:188,195
    write_dp3_small_arrays(out_zarr, action, state, episode_ends, chunk_size, compressor)


def read_dp3_dataset(out_zarr):
    g = lambda *p: read_array(os.path.join(out_zarr, *p))[0]
    return dict(point_cloud=g("data", "point_cloud"), action=g("data", "action"), state=g("data", "state"),
                episode_ends=g("meta", "episode_ends"))
