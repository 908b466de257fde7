"""GPU (-m gpu): the streamed episode runner and the direct zarr assembly against
the oracle's per-frame pipeline and against the reference's PLY route rows."""
import json
import os

import numpy as np
import pytest

from oracle import cbind, numpy_ref as R
from helpers import synth_frame

pytestmark = pytest.mark.gpu


def _oracle_rows(frames, intr, K, dist_mode=1):
    """dist_mode 1: the runner samples the way the loop it replaces does (use_cuda=True, DP3:47)."""
    po = cbind.make_params(intr, intr, apply_crop=1)
    rows, counts = [], []
    for d, c in frames:
        r = cbind.frame_pipeline(d, c, po, K, dist_mode)
        rows.append(cbind.pack_dp3(r["xyz"], r["rgb"]))
        counts.append(r["M"])
    return np.stack(rows), np.array(counts)


def test_episode_runner_chunks_and_tail_match_oracle():
    from pcp import _ffi
    from pcp.episode import EpisodeRunner
    H, W, K, T = 60, 80, 128, 23
    intr = dict(fx=75.6, fy=75.6, cx=W / 2, cy=H / 2)
    frames = [synth_frame(300 + i, H, W) for i in range(T)]
    depth = np.stack([f[0] for f in frames]); rgb = np.stack([f[1] for f in frames])
    runner = EpisodeRunner(H, W, K, _ffi.make_params(intr, intr, apply_crop=True), chunk=8)   # 8+8+7
    rows, counts = runner.run_arrays(depth, rgb)
    ref_rows, ref_counts = _oracle_rows(frames, intr, K)
    assert rows.shape == (T, K, 6) and rows.dtype == np.float32
    np.testing.assert_array_equal(counts, ref_counts)
    np.testing.assert_array_equal(rows, ref_rows)
    rows2, _ = runner.run_arrays(depth[:3], rgb[:3])                 # reuse: shorter episode, one partial chunk
    np.testing.assert_array_equal(rows2, ref_rows[:3])
    e, c = runner.run_arrays(depth[:0], rgb[:0])
    assert e.shape == (0, K, 6) and c.shape == (0,)


def test_direct_zarr_equals_ply_route_rows(tmp_path):
    from PIL import Image
    from pcp import build_zarr, zarr_store
    H, W, K = 60, 80, 200
    intr = dict(fx=75.6, fy=75.6, cx=W / 2, cy=H / 2)
    root = str(tmp_path / "processed")
    all_frames, n_per = [], [5, 3]
    for b, n in enumerate(n_per):
        ext = os.path.join(root, f"traj{b}", "extracted_data", f"traj{b}")
        os.makedirs(os.path.join(ext, "camera", "rgb")); os.makedirs(os.path.join(ext, "camera", "depth"))
        json.dump({"color": intr, "depth": intr}, open(os.path.join(ext, "intrinsics.json"), "w"))
        for i in range(n):
            d, c = synth_frame(500 + 10 * b + i, H, W)
            np.save(os.path.join(ext, "camera", "depth", f"depth_{i:05d}.npy"), d)
            Image.fromarray(c).save(os.path.join(ext, "camera", "rgb", f"rgb_{i:05d}.png"))
            all_frames.append((d, c))
        st = np.arange(n * 7, dtype=float).reshape(n, 7) + b
        ac = st * 0.5; ac[:, 6] = 45.0
        np.savetxt(os.path.join(ext, "states.txt"), st); np.savetxt(os.path.join(ext, "actions.txt"), ac)
    os.makedirs(os.path.join(root, "traj_incomplete"))
    out = str(tmp_path / "dp3.zarr")
    total, eps = build_zarr.build(root, out, source="gpu", K=K, ext=R.BASE_TO_EXT_CAM_S1.reshape(4, 4))
    assert (total, eps) == (8, 2)
    got = zarr_store.read_dp3_dataset(out)
    np.testing.assert_array_equal(got["episode_ends"], [5, 8])
    assert got["point_cloud"].shape == (8, K, 6) and got["action"].shape == (8, 7)
    assert (got["action"][:, 6] == 1.0).all()
    # rows == what the PLY route (XPC → PLY → DP3 → PLY → CHK:134-144) yields for the same frames
    po = cbind.make_params(intr, intr, apply_crop=1)
    for t, (d, c) in enumerate(all_frames):
        r = cbind.frame_pipeline(d, c, po, K, 1)
        np.testing.assert_array_equal(got["point_cloud"][t], R.dp3_rows(r["xyz"], r["rgb"]))


def test_depth_mm_upload_path_equals_float_path_and_numpy_division():
    import torch
    from pcp import _ffi, ops
    from pcp.episode import EpisodeRunner
    allv = np.arange(65536, dtype=np.uint16)                           # every representable raw depth
    got = ops.depth_mm_to_m(torch.from_numpy(allv).cuda()).cpu().numpy()
    np.testing.assert_array_equal(got, allv.astype(np.float32) / 1000.0)          # XTP:72
    H, W, K, T = 60, 80, 64, 11
    intr = dict(fx=75.6, fy=75.6, cx=W / 2, cy=H / 2)
    rng = np.random.default_rng(8)
    mm = rng.integers(0, 6000, (T, H, W)).astype(np.uint16)
    rgb = rng.integers(0, 256, (T, H, W, 3), dtype=np.uint8)
    p = _ffi.make_params(intr, intr, apply_crop=True)
    a, ca = EpisodeRunner(H, W, K, p, chunk=4, depth_mm=True).run_arrays(mm, rgb)
    b, cb = EpisodeRunner(H, W, K, p, chunk=4).run_arrays(mm.astype(np.float32) / 1000.0, rgb)
    np.testing.assert_array_equal(a, b); np.testing.assert_array_equal(ca, cb)


def test_rgb_stays_on_the_host_and_the_rows_are_the_same():
    """XPC:72-77: the projection mask needs rgb.shape only and colours are wanted for K points, so the runner's
    default keeps rgb in host memory (pcp_sampled_color_pixels + a host gather).  Rows must equal the device-side
    colour route bit for bit and the oracle's — also with a colour image of another size than the depth image, with
    raw u16 depth, and for clouds with fewer than K points (-1 padding → the LAST cropped row's colour, DP3:48)."""
    import torch
    from pcp import _ffi, ops
    from pcp.episode import EpisodeRunner
    H, W, Hc, Wc, K, T = 60, 80, 45, 70, 160, 13
    di = dict(fx=75.6, fy=75.6, cx=W / 2, cy=H / 2)
    ci = dict(fx=66.0, fy=56.0, cx=Wc / 2 - 1.5, cy=Hc / 2 + 0.75)
    rng = np.random.default_rng(21)
    mm = rng.integers(300, 1500, (T, H, W)).astype(np.uint16)
    mm[3, 5:, :] = 0                                                   # a sparse frame: fewer than K survivors
    mm[7] = 0                                                          # an empty one
    rgb = rng.integers(0, 256, (T, Hc, Wc, 3), dtype=np.uint8)
    depth = mm.astype(np.float32) / 1000.0
    p = _ffi.make_params(di, ci, apply_crop=True)
    host = EpisodeRunner(H, W, K, p, chunk=5, Hc=Hc, Wc=Wc, rgb_on_host=True, profile=True)
    devc = EpisodeRunner(H, W, K, p, chunk=5, Hc=Hc, Wc=Wc, rgb_on_host=False)
    a, ca = host.run_arrays(depth, rgb)
    b, cb = devc.run_arrays(depth, rgb)
    np.testing.assert_array_equal(a, b); np.testing.assert_array_equal(ca, cb)
    assert host.d_rgb is None and host.stats["frames"] == T and host.stats["kernels_ms"] > 0
    assert 0 < ca[3] < K and ca[7] == 0 and (a[7] == 0).all()
    po = cbind.make_params(di, ci, apply_crop=1)
    for t in range(T):
        r = cbind.frame_pipeline(depth[t], rgb[t], po, K, 1)
        np.testing.assert_array_equal(a[t], cbind.pack_dp3(r["xyz"], r["rgb"]))
    c, cc = EpisodeRunner(H, W, K, p, chunk=4, Hc=Hc, Wc=Wc, depth_mm=True).run_arrays(mm, rgb)
    np.testing.assert_array_equal(c, a); np.testing.assert_array_equal(cc, ca)
    # the operator itself: pixel indices reproduce the device gather
    pipe = ops.FramePipeline(T, H, W, K, p)
    d_dev, c_dev = torch.from_numpy(depth).cuda(), torch.from_numpy(rgb).cuda()
    xyz, col, idx, off = pipe.run(d_dev, c_dev)
    cpix = ops.sampled_color_pixels(d_dev, Hc, Wc, p, pipe.stage1.out["pix"], off, idx).cpu().numpy()
    col = col.cpu().numpy()
    flat = rgb.reshape(T, Hc * Wc, 3)
    want = np.take_along_axis(flat, np.maximum(cpix, 0)[:, :, None].astype(np.int64), axis=1).astype(np.float64) / 255.0
    want[cpix < 0] = 0.0
    np.testing.assert_array_equal(col, want.astype(np.float32))        # XPC:77: float64 divide, float32 store
    assert (cpix < Hc * Wc).all() and (cpix[7] == -1).all()
