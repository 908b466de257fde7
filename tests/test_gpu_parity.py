"""GPU (-m gpu): the HIP path, called through the C ABI, against the oracle and the committed
golden vectors.  Bit-exact for masks, indices, f64 coordinates and colours (the stated bar for
xyz/rgb is 1e-6 relative; we assert equality because the operation order is reproduced)."""
import hashlib

import numpy as np
import pytest
import torch

from oracle import cbind, numpy_ref as R
from pcp import _ffi, ops, synth
from helpers import load_npz, load_digests, intr_dict, synth_frame, bench_intr

pytestmark = pytest.mark.gpu
DEV = "cuda"
FIXTURES = ["stage1_48x64.npz", "stage1_64x96_c40x60.npz", "stage1_edge_32x32.npz"]


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def test_native_library_is_loaded():
    L = _ffi.load_library()
    assert L.pcp_device_count() >= 1
    assert "libpcp_hip.so" in open("/proc/self/maps").read()


# ---------------------------------------------------------------- stage 1 vs reference goldens
@pytest.mark.parametrize("name", FIXTURES)
def test_dense_matches_reference_golden(name):
    g = load_npz(name)
    p = _ffi.make_params(intr_dict(g["dintr"]), intr_dict(g["cintr"]), T=g["T"], ext=g["ext"],
                         apply_crop=False)
    out = ops.backproject_dense(dev(g["depth"][None]), dev(g["rgb"][None]), p)
    np.testing.assert_array_equal(out["pts_cam"][0].cpu().numpy(), g["pts_cam"])
    np.testing.assert_array_equal(out["colors"][0].cpu().numpy(), g["colors"])
    np.testing.assert_array_equal(out["proj_mask"][0].cpu().numpy().astype(bool), g["proj_mask"])
    np.testing.assert_array_equal(out["valid"][0].cpu().numpy().astype(bool), g["mask"])
    np.testing.assert_array_equal(out["pts_base"][0].cpu().numpy()[g["mask"]], g["pts_g"])


@pytest.mark.parametrize("name", FIXTURES)
def test_compaction_matches_reference_golden(name):
    g = load_npz(name)
    p = _ffi.make_params(intr_dict(g["dintr"]), intr_dict(g["cintr"]), T=g["T"], ext=g["ext"],
                         apply_crop=False)
    H, W = g["depth"].shape
    # batch of 3 identical frames: exercises CSR offsets across frames
    depth = dev(np.stack([g["depth"]] * 3))
    rgb = dev(np.stack([g["rgb"]] * 3))
    s1 = ops.Stage1(3, H, W, DEV, want=("xyz_cam64", "xyz64", "xyz32", "rgb32", "pix"))
    out, off = s1.run(depth, rgb, p)
    off = off.cpu().numpy()
    m1 = int(g["mask"].sum())
    assert off.tolist() == [0, m1, 2 * m1, 3 * m1]
    assert int(s1.overflow.item()) == 0
    for b in range(3):
        sl = slice(off[b], off[b + 1])
        np.testing.assert_array_equal(out["xyz_cam64"][sl].cpu().numpy(), g["pts_cam"][g["mask"]])
        np.testing.assert_array_equal(out["xyz64"][sl].cpu().numpy(), g["pts_g"])
        np.testing.assert_array_equal(out["xyz32"][sl].cpu().numpy(), g["pts_g"].astype(np.float32))
        np.testing.assert_array_equal(out["rgb32"][sl].cpu().numpy(), g["colors"][g["mask"]])
        np.testing.assert_array_equal(out["pix"][sl].cpu().numpy(), np.nonzero(g["mask"])[0])


def test_full_frame_digest_seed0():
    dg = load_digests()["seed0_480x640"]
    depth, rgb = synth_frame(0)
    intr = bench_intr()
    p = _ffi.make_params(intr, intr, apply_crop=False)
    out = ops.backproject_dense(dev(depth[None]), dev(rgb[None]), p,
                                want=("pts_cam", "colors", "proj_mask", "valid"))
    assert sha(out["pts_cam"][0].cpu().numpy()) == dg["pts_cam_sha"]
    assert sha(out["colors"][0].cpu().numpy()) == dg["colors_sha"]
    assert sha(out["valid"][0].cpu().numpy()) == dg["mask_sha"]
    s1 = ops.Stage1(1, 480, 640, DEV, want=("xyz64",))
    o, off = s1.run(dev(depth[None]), dev(rgb[None]), p)
    assert int(off[1]) == dg["M1"]
    assert sha(o["xyz64"][:dg["M1"]].cpu().numpy()) == dg["pts_g_sha"]


def test_fused_crop_matches_oracle_and_order():
    intr = bench_intr()
    frames = [synth_frame(s) for s in (0, 1, 2)]
    depth = dev(np.stack([f[0] for f in frames]))
    rgb = dev(np.stack([f[1] for f in frames]))
    p = _ffi.make_params(intr, intr, apply_crop=True)
    po = cbind.make_params(intr, intr, apply_crop=1)
    s1 = ops.Stage1(3, 480, 640, DEV, want=("xyz64", "xyz32", "rgb32", "pix"))
    out, off = s1.run(depth, rgb, p)
    off = off.cpu().numpy()
    dense = ops.backproject_dense(depth, rgb, p, want=("crop_mask",))["crop_mask"].cpu().numpy()
    for b, (d, c) in enumerate(frames):
        ref = cbind.backproject_align_crop(d, c, po)
        sl = slice(off[b], off[b + 1])
        assert off[b + 1] - off[b] == ref["xyz64"].shape[0]
        np.testing.assert_array_equal(out["pix"][sl].cpu().numpy(), ref["pix"])       # crop mask, bit-exact
        np.testing.assert_array_equal(out["xyz64"][sl].cpu().numpy(), ref["xyz64"])
        np.testing.assert_array_equal(out["xyz32"][sl].cpu().numpy(), ref["xyz32"])
        np.testing.assert_array_equal(out["rgb32"][sl].cpu().numpy(), ref["rgb32"])
        np.testing.assert_array_equal(np.nonzero(dense[b])[0], ref["pix"])
        assert np.all(np.diff(ref["pix"]) > 0)                                        # row-major order kept
        # the numpy restatement of DP3:35-39 on the reference-order points gives the same mask
        full = R.stage1_frame(d, c, intr, intr)
        keep = R.crop_mask(full["pts_g"])
        np.testing.assert_array_equal(np.nonzero(full["mask"])[0][keep], ref["pix"])


@pytest.mark.parametrize("dintr", [
    dict(fx=87.3, fy=91.1, cx=32.0, cy=24.0),                       # integer principal point: zero numerators on a column / row
    dict(fx=87.3, fy=91.1, cx=0.0, cy=0.0),
    dict(fx=87.3, fy=91.1, cx=1e-310, cy=31.7),                     # denormal principal point: library must divide the plain way
    dict(fx=87.3, fy=91.1, cx=32.3, cy=3e120),                      # huge principal point: plain divisions again
    dict(fx=2.5e-250, fy=91.1, cx=32.3, cy=24.6),                   # focal length outside the reciprocal's safe range
    dict(fx=-87.3, fy=91.1, cx=32.3, cy=24.6),                      # negative focal length (mirrored image)
    dict(fx=87.3 * (1 + 2.0 ** -52), fy=91.1 * (1 - 2.0 ** -53), cx=31.999999999999996, cy=24.000000000000004),
])
def test_unusual_intrinsics_keep_the_exact_quotients(dintr):
    """The deprojection quotients go through a reciprocal + FMA residual steps when the intrinsics are tame and through true
    divisions otherwise (pcp_to_dev decides per launch): both must equal the reference's numpy division bit for bit."""
    rng = np.random.default_rng(11)
    H, W = 48, 64
    depth = (rng.integers(1, 4000, (2, H, W)).astype(np.float32) / 1000).astype(np.float32)
    depth[0, ::7, ::5] = 0.0
    depth[1, 3, :] = np.float32(1e-40)                              # float denormals pass z > 0 when z_min allows it
    rgb = rng.integers(0, 256, (2, H, W, 3), dtype=np.uint8)
    cintr = dict(fx=90.0, fy=90.0, cx=W / 2, cy=H / 2)
    for crop in (False, True):
        p = _ffi.make_params(dintr, cintr, apply_crop=crop)
        po = cbind.make_params(dintr, cintr, apply_crop=int(crop))
        p.z_min = po.z_min = 0.0
        s1 = ops.Stage1(2, H, W, DEV, want=("xyz_cam64", "xyz64", "pix"))
        with np.errstate(all="ignore"):
            out, off = s1.run(dev(depth), dev(rgb), p)
            off = off.cpu().numpy()
            for b in range(2):
                ref = cbind.backproject_align_crop(depth[b], rgb[b], po)
                sl = slice(off[b], off[b + 1])
                np.testing.assert_array_equal(out["pix"][sl].cpu().numpy(), ref["pix"])
                np.testing.assert_array_equal(out["xyz64"][sl].cpu().numpy(), ref["xyz64"])
                full = R.stage1_frame(depth[b], rgb[b], dintr, cintr)     # the numpy expression itself, XPC:58-64
                cam = full["pts_cam"][ref["pix"]]
                np.testing.assert_array_equal(out["xyz_cam64"][sl].cpu().numpy(), cam)


def test_stage1_capacity_overflow_flag_and_ragged_size():
    # H*W not a multiple of 64 / 1024, tiny capacity
    rng = np.random.default_rng(4)
    H, W = 37, 53
    depth = (rng.integers(300, 1500, (2, H, W)).astype(np.float32) / 1000).astype(np.float32)
    rgb = rng.integers(0, 256, (2, H, W, 3), dtype=np.uint8)
    intr = dict(fx=50.0, fy=50.0, cx=W / 2, cy=H / 2)
    p = _ffi.make_params(intr, intr, apply_crop=False)
    po = cbind.make_params(intr, intr, apply_crop=0)
    s1 = ops.Stage1(2, H, W, DEV, want=("xyz64", "pix"))
    out, off = s1.run(dev(depth), dev(rgb), p)
    off = off.cpu().numpy()
    for b in range(2):
        ref = cbind.backproject_align_crop(depth[b], rgb[b], po)
        np.testing.assert_array_equal(out["pix"][off[b]:off[b + 1]].cpu().numpy(), ref["pix"])
        np.testing.assert_array_equal(out["xyz64"][off[b]:off[b + 1]].cpu().numpy(), ref["xyz64"])
    small = ops.Stage1(2, H, W, DEV, cap=100, want=("pix",))
    o2, off2 = small.run(dev(depth), dev(rgb), p)
    assert int(small.overflow.item()) == 1
    np.testing.assert_array_equal(off2.cpu().numpy(), off)                 # true counts survive
    np.testing.assert_array_equal(o2["pix"][:100].cpu().numpy(), out["pix"][:100].cpu().numpy())


def test_all_invalid_frame_gives_empty_rows():
    depth = np.zeros((2, 16, 16), np.float32)
    depth[1] = 0.8
    rgb = np.zeros((2, 16, 16, 3), np.uint8)
    intr = dict(fx=20.0, fy=20.0, cx=8.0, cy=8.0)
    p = _ffi.make_params(intr, intr, apply_crop=False)
    s1 = ops.Stage1(2, 16, 16, DEV, want=("pix",))
    _, off = s1.run(dev(depth), dev(rgb), p)
    off = off.cpu().numpy()
    assert off[0] == 0 and off[1] == 0 and off[2] > 0


# ---------------------------------------------------------------- stage-2 crop on PLY-style input
def test_crop_mask_and_compact_vs_oracle():
    rng = np.random.default_rng(9)
    xyz = rng.uniform(-0.3, 0.8, (70001, 3))
    xyz[:9] = [[0.1, 0.1, 0.1], [-0.05, 0.1, 0.1], [0.4, 0.1, 0.1], [np.nextafter(-0.05, 1), 0.1, 0.1],
               [0.1, np.nextafter(0.35, 0), 0.1], [0.1, 0.35, 0.1], [0.1, 0.1, -0.008],
               [0.1, 0.1, np.nextafter(0.6, 0)], [np.nan, 0.1, 0.1]]
    rgb = rng.integers(0, 256, (70001, 3)) / 255.0
    lo = [w[0] for w in R.WORK_SPACE]
    hi = [w[1] for w in R.WORK_SPACE]
    ref = R.crop_mask(xyz)
    assert ref[:9].tolist() == [True, False, False, True, True, False, False, True, False]
    got = ops.crop_mask(dev(xyz), lo, hi).cpu().numpy().astype(bool)
    np.testing.assert_array_equal(got, ref)
    o64, o32, orgb, m = ops.crop_compact(dev(xyz), dev(rgb), lo, hi)
    assert m == int(ref.sum())
    np.testing.assert_array_equal(o64.cpu().numpy(), xyz[ref])
    np.testing.assert_array_equal(o32.cpu().numpy(), xyz[ref].astype(np.float32))
    np.testing.assert_array_equal(orgb.cpu().numpy(), rgb[ref])
    e64, _, _, m0 = ops.crop_compact(dev(np.zeros((0, 3))), None, lo, hi)
    assert m0 == 0 and e64.shape[0] == 0


# ---------------------------------------------------------------- FPS
def run_fps(clouds, K, dist_mode=0, algo=0):
    sizes = [c.shape[0] for c in clouds]
    off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    cat = np.concatenate(clouds, axis=0).astype(np.float32) if sum(sizes) else np.zeros((0, 3), np.float32)
    xyz, idx = ops.fps_batched(dev(cat), dev(off), K, dist_mode=dist_mode, algo=algo,
                               max_pts=max(sizes + [1]))
    return xyz.cpu().numpy(), idx.cpu().numpy()


ALGOS = [1, 2, 3, 4]      # brute force, bucket (auto), bucket with deferred rounds forced on / off


@pytest.mark.parametrize("algo", ALGOS)
def test_fps_known_answers(algo):
    collinear = np.array([[0, 0, 0], [1, 0, 0], [2, 0, 0], [3, 0, 0], [10, 0, 0]], np.float32)
    corners = np.array([[x, y, z] for x in (0, 1) for y in (0, 1) for z in (0, 1)] + [[.5, .5, .5]], np.float32)
    dup = np.zeros((6, 3), np.float32)
    tiny = np.array([[0, 0, 0], [3, 0, 0], [1, 0, 0]], np.float32)
    one = np.array([[1, 2, 3]], np.float32)
    empty = np.zeros((0, 3), np.float32)
    xyz, idx = run_fps([collinear, corners, dup, tiny, one, empty], 5, algo=algo)
    assert idx[0].tolist() == [0, 4, 3, 1, 2]
    assert idx[1][:3].tolist() == [0, 7, 1]
    assert idx[2].tolist() == [0, 0, 0, 0, 0]
    assert idx[3].tolist() == [0, 1, 2, -1, -1]
    assert idx[4].tolist() == [0, -1, -1, -1, -1]
    assert idx[5].tolist() == [-1] * 5
    np.testing.assert_array_equal(xyz[3][3:], 0)
    np.testing.assert_array_equal(xyz[3][:3], tiny[[0, 1, 2]])
    np.testing.assert_array_equal(xyz[5], 0)


@pytest.mark.parametrize("algo", ALGOS)
@pytest.mark.parametrize("dist_mode", [0, 1])
def test_fps_sizes_vs_oracle(dist_mode, algo):
    rng = np.random.default_rng(21)
    for M in (1, 2, 63, 64, 65, 1000, 1025, 4099):
        pts = rng.uniform(-0.5, 0.5, (M, 3)).astype(np.float32)
        for K in sorted({1, 2, 17, min(M, 1024), M, M + 5}):
            ref_xyz, ref_idx = cbind.fps(pts, K, dist_mode)
            xyz, idx = run_fps([pts], K, dist_mode, algo)
            np.testing.assert_array_equal(idx[0], ref_idx, err_msg=f"M={M} K={K}")
            np.testing.assert_array_equal(xyz[0], ref_xyz)


@pytest.mark.parametrize("algo", ALGOS)
def test_fps_ragged_batch_vs_oracle(algo):
    rng = np.random.default_rng(22)
    sizes = [5000, 0, 1, 777, 20000, 64, 3]
    clouds = [rng.normal(0, 0.2, (m, 3)).astype(np.float32) for m in sizes]
    xyz, idx = run_fps(clouds, 300, 0, algo)
    for b, c in enumerate(clouds):
        ref_xyz, ref_idx = cbind.fps(c, 300, 0)
        np.testing.assert_array_equal(idx[b], ref_idx)
        np.testing.assert_array_equal(xyz[b], ref_xyz)


@pytest.mark.parametrize("algo", ALGOS)
def test_fps_ties_on_a_lattice(algo):
    # integer lattice: many exactly equal distances → exercises lowest-index tie-breaking
    g = np.stack(np.meshgrid(np.arange(12), np.arange(12), np.arange(6), indexing="ij"), -1).reshape(-1, 3)
    rng = np.random.default_rng(23)
    pts = g[rng.permutation(len(g))].astype(np.float32)
    ref_xyz, ref_idx = cbind.fps(pts, 400, 0)
    xyz, idx = run_fps([pts], 400, 0, algo)
    np.testing.assert_array_equal(idx[0], ref_idx)


@pytest.mark.parametrize("algo", ALGOS)
def test_fps_non_finite_points_behave_like_the_oracle(algo):
    # a NaN distance never lowers min_d (d < min_d is false), so a NaN point keeps the initial maximum and is
    # picked again and again — degenerate, but brute force, bucket pruning and the oracle must agree on it
    rng = np.random.default_rng(25)
    pts = rng.uniform(-0.5, 0.5, (5000, 3)).astype(np.float32)
    pts[17] = [np.nan, 0.1, 0.2]; pts[400, 2] = np.inf; pts[4000] = [np.nan] * 3; pts[77, 0] = -np.inf
    ref_xyz, ref_idx = cbind.fps(pts, 64, 0)
    xyz, idx = run_fps([pts], 64, 0, algo)
    np.testing.assert_array_equal(idx[0], ref_idx)
    clean = pts.copy(); clean[400, 2] = np.inf                       # only infinities: distances inf, no NaN
    clean[17] = [0.0, 0.1, 0.2]; clean[4000] = [0.3, 0.3, 0.3]
    ref_xyz, ref_idx = cbind.fps(clean, 64, 0)
    xyz, idx = run_fps([clean], 64, 0, algo)
    np.testing.assert_array_equal(idx[0], ref_idx)


@pytest.mark.parametrize("dist_mode", [0, 1])
def test_fps_batch_of_300_ragged_clouds(dist_mode):
    """A batch of more clouds than CUs (two workgroups share some CUs) with ragged sizes — empty, single point,
    fewer points than K, not a multiple of 64, exactly 64 k, one cloud for the deep shape, duplicates and a
    lattice with ties — must give the oracle's picks for every cloud."""
    rng = np.random.default_rng(26)
    K = 96
    sizes = [0, 1, 2, 63, 64, 65, 95, 96, 97, 1000, 2047, 2048, 4099, 6400]
    clouds = []
    for b in range(300):
        m = sizes[b % len(sizes)] if b < 280 else int(rng.integers(2048, 9000))
        clouds.append(rng.uniform(-0.5, 0.5, (m, 3)).astype(np.float32))
    g = np.stack(np.meshgrid(np.arange(10), np.arange(10), np.arange(5), indexing="ij"), -1).reshape(-1, 3)
    clouds[7] = g[rng.permutation(len(g))].astype(np.float32)             # exact ties
    clouds[263] = np.zeros((3000, 3), np.float32)                          # all duplicates
    clouds[290] = rng.uniform(-0.5, 0.5, (65536, 3)).astype(np.float32)    # largest size of the wide shape
    clouds[5] = rng.uniform(-0.5, 0.5, (70000, 3)).astype(np.float32)      # belongs to the deep launch
    xyz, idx = run_fps(clouds, K, dist_mode, 2)
    for b, c in enumerate(clouds):
        if b % 9 == 0 or b in (5, 7, 263, 290) or b >= 256:
            ref_xyz, ref_idx = cbind.fps(c, K, dist_mode)
            np.testing.assert_array_equal(idx[b], ref_idx, err_msg=f"cloud {b} M={len(c)}")
            np.testing.assert_array_equal(xyz[b], ref_xyz, err_msg=f"cloud {b}")


@pytest.mark.parametrize("algo", [2, 3])
def test_fps_cloud_shapes_cover_every_sort_plan(algo):
    """Cubic clouds sort along the 3-D Hilbert curve, flat ones along the 2-D curve over their two long axes
    (any axis may be the thin one), the rest along the adaptive Morton order: all must give the oracle's picks."""
    rng = np.random.default_rng(24)
    n = 30000
    u = lambda lo, hi: rng.uniform(lo, hi, n)
    shapes = {
        "cube": np.c_[u(0, 1), u(0, 1), u(0, 1)],
        "flat_z": np.c_[u(0, 1), u(0, 0.9), 0.2 + 0.002 * rng.standard_normal(n)],
        "flat_x": np.c_[0.001 * rng.standard_normal(n), u(-1, 1), u(0, 1.7)],
        "flat_y": np.c_[u(0, 3), np.full(n, 0.25), u(0, 2.5)],
        "slab_8_8_1": np.c_[u(0, 0.8), u(0, 0.8), u(0, 0.1)],
        "rod": np.c_[u(0, 2), u(0, 0.05), u(0, 0.04)],
        "line": np.c_[u(0, 1), np.zeros(n), np.zeros(n)],
        "tabletop": np.c_[u(-0.05, 0.4), u(-0.12, 0.35), 0.02 + 0.002 * rng.standard_normal(n) + (rng.random(n) < 0.15) * u(0, 0.12)],
    }
    clouds = [v.astype(np.float32) for v in shapes.values()]
    xyz, idx = run_fps(clouds, 512, 0, algo)
    for b, (name, c) in enumerate(zip(shapes, clouds)):
        ref_xyz, ref_idx = cbind.fps(c, 512, 0)
        np.testing.assert_array_equal(idx[b], ref_idx, err_msg=name)
        np.testing.assert_array_equal(xyz[b], ref_xyz, err_msg=name)


@pytest.mark.parametrize("algo", ALGOS)
def test_fps_bench_frame_60k_points(algo):
    intr = bench_intr()
    depth, rgb = synth_frame(0)
    ref = cbind.backproject_align_crop(depth, rgb, cbind.make_params(intr, intr, apply_crop=1))
    pts = ref["xyz32"]
    assert 50000 < pts.shape[0] < 70000
    for K in (1024, 2500):
        ref_xyz, ref_idx = cbind.fps(pts, K, 0)
        xyz, idx = run_fps([pts], K, 0, algo)
        np.testing.assert_array_equal(idx[0], ref_idx)
        np.testing.assert_array_equal(xyz[0], ref_xyz)


@pytest.mark.parametrize("dist_mode", [0, 1])
def test_fps_deferred_rounds_hand_over_to_the_eager_loop(dist_mode):
    """algo 3 runs the first 256 rounds with deferred bucket updates, drains the pending lists and continues with
    the eager loop: K just below, at and above the hand-over, and clouds whose size ends the rounds before it."""
    rng = np.random.default_rng(27)
    sizes = [60000, 255, 256, 257, 258, 300, 5000, 64, 4096, 70001, 131072, 131073, 262144, 262145]   # incl. the split-bucket shapes' bounds and one deep cloud
    clouds = [(rng.normal(0, 0.3, (m, 3)) * np.array([1.0, 0.7, 0.1])).astype(np.float32) for m in sizes]
    for K in (255, 256, 257, 258, 700):
        xyz, idx = run_fps(clouds, K, dist_mode, 3)
        for b, c in enumerate(clouds):
            ref_xyz, ref_idx = cbind.fps(c, K, dist_mode)
            np.testing.assert_array_equal(idx[b], ref_idx, err_msg=f"K={K} M={len(c)}")
            np.testing.assert_array_equal(xyz[b], ref_xyz, err_msg=f"K={K} M={len(c)}")


# ---------------------------------------------------------------- gather / pack / whole pipeline
def test_gather_and_pack():
    rng = np.random.default_rng(31)
    rgb = (rng.integers(0, 256, (10, 3)) / 255.0).astype(np.float32)
    off = np.array([0, 4, 10], np.int64)
    idx = np.array([[0, 3, -1], [5, 0, -1]], np.int64)
    got = ops.gather_rgb(dev(rgb), dev(off), dev(idx), True).cpu().numpy()
    np.testing.assert_array_equal(got[0], rgb[:4][idx[0]])        # numpy negative index → last row
    np.testing.assert_array_equal(got[1], rgb[4:][idx[1]])
    got0 = ops.gather_rgb(dev(rgb), dev(off), dev(idx), False).cpu().numpy()
    np.testing.assert_array_equal(got0[:, 2], 0)
    xyz = rng.uniform(-1, 1, (2, 3, 3)).astype(np.float32)
    packed = ops.pack_dp3(dev(xyz), dev(got)).cpu().numpy()
    np.testing.assert_array_equal(packed.reshape(-1, 6), R.dp3_rows(xyz.reshape(-1, 3), got.reshape(-1, 3)))


def test_gather_from_source_pixels_equals_gather_of_stored_colours():
    """pcp_gather_rgb_pixels (colour lookup for the K sampled points only) vs pcp_gather_rgb over stage 1's rgb32, with
    a colour image of another resolution, -1 padding (numpy negative index → last row) and an empty cloud."""
    g = load_npz("stage1_64x96_c40x60.npz")
    p = _ffi.make_params(intr_dict(g["dintr"]), intr_dict(g["cintr"]), T=g["T"], ext=g["ext"], apply_crop=False)
    H, W = g["depth"].shape
    depth = np.stack([g["depth"], np.zeros_like(g["depth"]), g["depth"][::-1].copy()])
    rgb = np.stack([g["rgb"], g["rgb"], g["rgb"][::-1].copy()])
    s1 = ops.Stage1(3, H, W, DEV, want=("rgb32", "pix"))
    out, off = s1.run(dev(depth), dev(rgb), p)
    m = np.diff(off.cpu().numpy())
    assert m[1] == 0 and m[0] > 100
    rng = np.random.default_rng(33)
    idx = np.stack([np.r_[rng.integers(0, max(mb, 1), 60), [-1, -1, -1, -1]] for mb in m]).astype(np.int64)
    idx[1] = -1
    for compat in (True, False):
        a = ops.gather_rgb(out["rgb32"], off, dev(idx), compat).cpu().numpy()
        b = ops.gather_rgb_pixels(dev(depth), dev(rgb), p, out["pix"], off, dev(idx), compat).cpu().numpy()
        np.testing.assert_array_equal(a, b)
    assert (b[1] == 0).all() and a[0].any()


@pytest.mark.parametrize("dist_mode", [None, 0, 1])
def test_frame_pipeline_vs_oracle_config1(dist_mode):
    """BASELINE.json configs[0]: one 640x480 frame → crop → 1024-pt FPS (plus 2 more seeds).  dist_mode None = the
    pipeline's default, which is the reference's use_cuda=True arithmetic (mode 1, DP3:47)."""
    intr = bench_intr()
    seeds = (0, 1, 2)
    frames = [synth_frame(s) for s in seeds]
    p = _ffi.make_params(intr, intr, apply_crop=True)
    po = cbind.make_params(intr, intr, apply_crop=1)
    pipe = ops.FramePipeline(len(seeds), 480, 640, 1024, p, DEV, dist_mode=dist_mode)
    want_mode = 1 if dist_mode is None else dist_mode
    assert pipe.fps.dist_mode == want_mode
    depth = dev(np.stack([f[0] for f in frames]))
    rgb = dev(np.stack([f[1] for f in frames]))
    for _ in range(2):           # buffers are reused: the second run must give the same answer
        xyz, col, idx, off = pipe.run(depth, rgb)
        torch.cuda.synchronize()
        for b, (d, c) in enumerate(frames):
            ref = cbind.frame_pipeline(d, c, po, 1024, want_mode)
            assert int(off[b + 1] - off[b]) == ref["M"]
            np.testing.assert_array_equal(idx[b].cpu().numpy(), ref["idx"])
            np.testing.assert_array_equal(xyz[b].cpu().numpy(), ref["xyz"])
            np.testing.assert_array_equal(col[b].cpu().numpy(), ref["rgb"])


def test_frame_pipeline_vs_oracle_on_tabletop_scenes():
    """SURVEY.md 8d's second input distribution: a ray-cast table with boxes.  The cropped cloud is a smooth sheet
    of ~72 k points (more than the wide FPS shape takes, so these frames run the deep shape), with dropped pixels."""
    intr = bench_intr()
    frames = [synth.scene_frame(s) for s in (0, 5)]
    p = _ffi.make_params(intr, intr, apply_crop=True)
    po = cbind.make_params(intr, intr, apply_crop=1)
    pipe = ops.FramePipeline(len(frames), 480, 640, 1024, p, DEV, dist_mode=0)
    xyz, col, idx, off = pipe.run(dev(np.stack([f[0] for f in frames])), dev(np.stack([f[1] for f in frames])))
    torch.cuda.synchronize()
    for b, (d, c) in enumerate(frames):
        ref = cbind.frame_pipeline(d, c, po, 1024, 0)
        assert 60000 < ref["M"] < 110000 and int(off[b + 1] - off[b]) == ref["M"]
        np.testing.assert_array_equal(idx[b].cpu().numpy(), ref["idx"])
        np.testing.assert_array_equal(xyz[b].cpu().numpy(), ref["xyz"])
        np.testing.assert_array_equal(col[b].cpu().numpy(), ref["rgb"])


def test_fps_properties_at_full_size():
    """Size-independent properties on a full bench frame at the reference's K = 2500."""
    intr = bench_intr()
    depth, rgb = synth_frame(7)
    p = _ffi.make_params(intr, intr, apply_crop=True)
    pipe = ops.FramePipeline(1, 480, 640, 2500, p, DEV)
    xyz, col, idx, off = pipe.run(dev(depth[None]), dev(rgb[None]))
    idx = idx[0].cpu().numpy()
    xyz = xyz[0].cpu().numpy()
    M = int(off[1])
    assert idx[0] == 0 and idx.min() >= 0 and idx.max() < M
    assert len(np.unique(idx)) == 2500                         # distinct points ⇒ distinct picks
    pts = pipe.stage1.out["xyz32"][:M].cpu().numpy()
    np.testing.assert_array_equal(pts[idx], xyz)
    # greedy property: the pick's distance to the earlier picks never increases
    prev = np.inf
    for k in range(1, 2500, 50):
        d = ((xyz[:k] - xyz[k]) ** 2).sum(1).min()
        assert d <= prev * (1 + 1e-5)
        prev = d
