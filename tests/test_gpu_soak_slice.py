"""GPU (-m gpu): a seeded slice of the randomised soaks (tests/stress_fps.py, stress_stage1.py, stress_filters.py — run by
hand for minutes on the MI355X) inside the driver's own run: a few seconds each, fixed seeds, every case bit-exact
against the C oracle, plus a deterministic sweep over the FPS launch-shape bounds."""
import numpy as np
import pytest
import torch

from oracle import cbind
from pcp import ops

pytestmark = pytest.mark.gpu


def _no_mismatch(mod, seconds, seed, capsys):
    try:
        mod.main(["--seconds", str(seconds), "--seed", str(seed)])
    except SystemExit as e:                                # the soak exits 1 on the first mismatch, after printing it
        pytest.fail(f"{mod.__name__} seed {seed}: {capsys.readouterr().out[-400:]} (exit {e.code})")
    out = capsys.readouterr().out
    assert "all bit-exact" in out or "done" in out, out[-300:]
    return out


def test_fps_launch_shape_bounds_both_modes_every_algo():
    """Clouds one point below, on and above every size bound of the FPS launch shapes (2,048 brute/bucket; 65,536 /
    131,072 / 262,144 wide / halved / quartered / deep buckets), every cloud kind of the soak, both distance modes, the
    deferred rounds forced on and off (algo 3 / 4) and the multi-pick rounds forced on and off (algo 5 / 6)."""
    import stress_fps
    rng = np.random.default_rng(20261111)
    bounds = [2048, 65536, 131072, 262144]
    case = 0
    for bound in bounds:
        for m in (bound - 1, bound, bound + 1):
            cloud = stress_fps.make_cloud(rng, m)
            for K in (33, 150):
                dm, algo = case & 1, (2, 3, 4, 5, 6, 9)[case % 6]
                case += 1
                off = torch.tensor([0, m], dtype=torch.int64, device="cuda")
                xyz, idx = ops.fps_batched(torch.from_numpy(cloud).cuda(), off, K, dist_mode=dm, algo=algo, max_pts=m)
                rx, ri = cbind.fps(cloud, K, dm)
                np.testing.assert_array_equal(idx[0].cpu().numpy(), ri, err_msg=f"M={m} K={K} mode={dm} algo={algo}")
                np.testing.assert_array_equal(xyz[0].cpu().numpy(), rx)
    assert case == 24


def test_fps_multi_pick_rounds_equal_the_sequential_picks_on_batches():
    """algo 5 (up to four certified picks per round) against the oracle on ragged batches: volumes, sheets, lattices with
    exact ties, heavy duplicates, clusters, rods; K up to the reference's 2,500."""
    import stress_fps
    rng = np.random.default_rng(5)
    for trial in range(6):
        ms = [int(rng.choice([1, 2, 63, 64, 65, 1000, 2049, 5000, 20000, 61000])) for _ in range(5)]
        clouds = [stress_fps.make_cloud(rng, m) for m in ms]
        K = (64, 300, 1024, 2500, 257, 700)[trial]
        dm = trial & 1
        off = np.zeros(len(ms) + 1, np.int64); off[1:] = np.cumsum(ms)
        xyz, idx = ops.fps_batched(torch.from_numpy(np.concatenate(clouds)).cuda(), torch.from_numpy(off).cuda(), K,
                                   dist_mode=dm, algo=5, max_pts=max(ms))
        for b, c in enumerate(clouds):
            rx, ri = cbind.fps(c, K, dm)
            np.testing.assert_array_equal(idx[b].cpu().numpy(), ri, err_msg=f"trial {trial} cloud {b} M={ms[b]} K={K}")
            np.testing.assert_array_equal(xyz[b].cpu().numpy(), rx)


def test_fps_deferred_multi_pick_rounds_incl_ring_wrap():
    """algo 9: multi-pick rounds with exact-key deferred updates forced on for small batches too, at K beyond the 1,024-entry
    sample ring (a pending pick must be flushed before the ring wraps over it) and on clouds of every launch shape."""
    import stress_fps
    rng = np.random.default_rng(9)
    for M, K in ((61000, 2500), (61000, 4096), (20000, 3000), (140000, 2600), (270000, 1500), (3000, 3000)):
        for kind in range(2):
            cloud = stress_fps.make_cloud(rng, M) if kind else rng.uniform(-0.5, 0.5, (M, 3)).astype(np.float32)
            dm = kind
            off = torch.tensor([0, M], dtype=torch.int64, device="cuda")
            xyz, idx = ops.fps_batched(torch.from_numpy(cloud).cuda(), off, K, dist_mode=dm, algo=9, max_pts=M)
            rx, ri = cbind.fps(cloud, K, dm)
            np.testing.assert_array_equal(idx[0].cpu().numpy(), ri, err_msg=f"M={M} K={K} mode={dm}")
            np.testing.assert_array_equal(xyz[0].cpu().numpy(), rx)


def test_fps_quantum_sliced_launch_equals_the_sequential_picks():
    """algo 10: fewer resident workgroups than clouds, clouds change workgroups every 150 us (round state parked in the
    workspace, agent-scope release / acquire hand-off): ragged batches over the three wide launch shapes, empty and tiny clouds,
    K beyond the sample ring, both distance modes — every cloud against the oracle, and twice in a row (the second call
    finds the first call's parked state and slot words in the workspace)."""
    import stress_fps
    rng = np.random.default_rng(10)
    for trial in range(6):
        sizes = [[0, 1, 5, 2049, 61000, 30000, 8000, 61000, 45000, 3000, 64, 65536],
                 [61000] * 10, [70000, 131072, 3000, 90000, 65537, 20000, 100000], [20000] * 16 + [3],
                 [61000, 0, 61000, 61000, 61000, 2, 61000],
                 [150000, 262144, 70000, 200000, 131073, 5000, 240000, 180000]][trial]      # quartered buckets too
        clouds = [stress_fps.make_cloud(rng, m) if m else np.zeros((0, 3), np.float32) for m in sizes]
        K = (300, 2500, 700, 1500, 1024, 1200)[trial]
        dm = trial & 1
        off = np.zeros(len(sizes) + 1, np.int64); off[1:] = np.cumsum(sizes)
        cat = torch.from_numpy(np.concatenate(clouds)).cuda()
        plan = ops.Fps(len(sizes), cat.shape[0], max(sizes), K, "cuda", dm, 10)
        ref = [cbind.fps(c, K, dm) if len(c) else None for c in clouds]
        for rep in range(2):
            xyz, idx = plan.run(cat, torch.from_numpy(off).cuda())
            torch.cuda.synchronize()
            for b, c in enumerate(clouds):
                if ref[b] is None:
                    assert (idx[b] == -1).all()
                    continue
                rx, ri = ref[b]
                np.testing.assert_array_equal(idx[b].cpu().numpy(), ri, err_msg=f"trial {trial} rep {rep} cloud {b} M={sizes[b]} K={K}")
                np.testing.assert_array_equal(xyz[b].cpu().numpy(), rx)


def test_fps_mixed_size_episode_goes_as_one_any_shape_sliced_launch():
    """257-336 clouds per call whose sizes straddle 65,536 (and 131,072) points, algo 0: fps_slice_prep finds more than one wide
    shape in the batch, the per-shape sliced launches exit at once and the ANY-shape launch runs every cloud with the bucket
    split its size asks for — equal to the resident launches (algo 5, shape after shape) on every cloud and to the oracle on a
    sample; a batch of one shape only still goes through its own per-shape launch (same check)."""
    import stress_fps
    rng = np.random.default_rng(300)
    for case, (B, K, dm) in enumerate(((300, 300, 1), (272, 257, 0), (300, 256, 1), (400, 256, 1))):
        if case == 0:
            sizes = rng.integers(60000, 72000, B)                  # whole and halved buckets, about half each
            sizes[7], sizes[150], sizes[299] = 0, 140000, 65536
        elif case == 1:
            sizes = rng.integers(20000, 30000, B)
            sizes[5], sizes[100], sizes[271] = 70000, 135000, 65537   # three shapes, two of them with one cloud only
        elif case == 3:
            # 337-512 clouds with a size bound beyond 131,072 points: the per-shape launches are resident (two workgroups per CU)
            # for whole / halved buckets and sliced for quartered ones; the mixed batch goes as one any-shape sliced launch
            sizes = rng.integers(30000, 70000, B)
            sizes[3], sizes[399] = 135000, 0
        else:
            sizes = rng.integers(66000, 70000, B)                  # one shape (halved buckets) only
        base = stress_fps.make_cloud(rng, int(sizes.max()))
        off = np.zeros(B + 1, np.int64); off[1:] = np.cumsum(sizes)
        pts = np.concatenate([base[rng.permutation(len(base))[:m]] + np.float32(0.001 * (b % 7)) for b, m in enumerate(sizes)])
        cat, offd = torch.from_numpy(pts).cuda(), torch.from_numpy(off).cuda()
        mx = int(sizes.max())
        ref = ops.Fps(B, len(pts), mx, K, "cuda", dm, 5)
        rx, ri = ref.run(cat, offd)
        got = ops.Fps(B, len(pts), mx, K, "cuda", dm, 0)
        got.idx.fill_(-7)
        for rep in range(2):                                       # the second call finds the first one's slot words and state
            gx, gi = got.run(cat, offd)
            torch.cuda.synchronize()
            assert int(got.overflow.item()) == 0
            assert torch.equal(gi, ri) and torch.equal(gx.view(torch.int32), rx.view(torch.int32)), (case, rep)
        for b in (0, 3, 5, 7, 100, 150, B - 1):
            if sizes[b] == 0:
                assert (gi[b] == -1).all()
                continue
            ox, oi = cbind.fps(pts[off[b]:off[b + 1]], K, dm)
            np.testing.assert_array_equal(gi[b].cpu().numpy(), oi, err_msg=f"case {case} cloud {b} M={sizes[b]}")
            np.testing.assert_array_equal(gx[b].cpu().numpy(), ox)


def test_fps_mixed_size_batches_of_resident_size_go_as_one_any_shape_launch():
    """2-256 clouds per call (algo 0, the 64-cloud chunks of the stage-2 mirror) with sizes on both sides of 65,536 / 131,072
    points: one any-shape launch with a workgroup per cloud instead of one resident launch per shape — every cloud against the
    oracle; a workspace without the control words (an older caller's size) still gives the same picks."""
    import stress_fps
    from pcp import _ffi
    rng = np.random.default_rng(64)
    for case, sizes in enumerate(([61000, 70000, 64000, 66000, 131072, 140000, 5, 0, 65536, 65537, 3000, 90000],
                                  [70000, 60000], [20000] * 3 + [200000], [66000] * 5)):
        K, dm = (300, 1024, 257, 500)[case], case & 1
        clouds = [stress_fps.make_cloud(rng, m) if m else np.zeros((0, 3), np.float32) for m in sizes]
        off = np.zeros(len(sizes) + 1, np.int64); off[1:] = np.cumsum(sizes)
        cat, offd = torch.from_numpy(np.concatenate(clouds)).cuda(), torch.from_numpy(off).cuda()
        plan = ops.Fps(len(sizes), cat.shape[0], max(sizes), K, "cuda", dm, 0)
        base = ops.Fps(len(sizes), cat.shape[0], max(sizes), K, "cuda", dm, 5).ws_bytes
        assert 0 < plan.ws_bytes - base <= 8192                     # the control words
        ref = [cbind.fps(c, K, dm) if len(c) else None for c in clouds]
        for rep in range(2):
            plan.idx.fill_(-7)
            xyz, idx = plan.run(cat, offd)
            torch.cuda.synchronize()
            assert int(plan.overflow.item()) == 0
            for b, c in enumerate(clouds):
                if ref[b] is None:
                    assert (idx[b] == -1).all()
                    continue
                np.testing.assert_array_equal(idx[b].cpu().numpy(), ref[b][1], err_msg=f"case {case} rep {rep} cloud {b} M={sizes[b]}")
                np.testing.assert_array_equal(xyz[b].cpu().numpy(), ref[b][0])
        # the same call through the C ABI with the smaller workspace of a caller that did not ask again: shape after shape
        L = _ffi.load_library()
        ws = torch.empty(base + 256, dtype=torch.uint8, device="cuda")
        wp = (ws.data_ptr() + 255) & ~255
        idx2 = torch.full((len(sizes), K), -7, dtype=torch.int64, device="cuda")
        rc = L.pcp_fps_batched(cat.data_ptr(), offd.data_ptr(), len(sizes), cat.shape[0], max(sizes), K, dm, 0, idx2.data_ptr(), None,
                               None, wp, base, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        assert rc == 0 and torch.equal(idx2, idx)


def test_fps_batch_of_a_little_more_than_two_clouds_per_cu_goes_as_two_sliced_launches():
    """513-624 clouds per call (algo 0): two launches over the halves of the batch, each quantum-sliced — equal to the
    resident launch (algo 5) on every cloud and to the oracle on a sample, with and without out_xyz rows of the second half
    starting mid-array."""
    rng = np.random.default_rng(600)
    for B, K in ((513, 256), (600, 300), (624, 257)):
        sizes = rng.integers(2048, 5000, B)
        sizes[B // 2] = 0
        off = np.zeros(B + 1, np.int64); off[1:] = np.cumsum(sizes)
        pts = rng.uniform(-0.5, 0.5, (int(off[-1]), 3)).astype(np.float32)
        cat, offd = torch.from_numpy(pts).cuda(), torch.from_numpy(off).cuda()
        ref = ops.Fps(B, len(pts), 5000, K, "cuda", 1, 5)
        rx, ri = ref.run(cat, offd)
        got = ops.Fps(B, len(pts), 5000, K, "cuda", 1, 0)
        assert got.ws_bytes > ref.ws_bytes                      # room for the parked state of the larger half
        got.idx.fill_(-7)
        gx, gi = got.run(cat, offd)
        torch.cuda.synchronize()
        assert int(got.overflow.item()) == 0
        assert torch.equal(gi, ri) and torch.equal(gx.view(torch.int32), rx.view(torch.int32))
        for b in (0, B // 2 - 1, B // 2, B // 2 + 1, B - 1):
            if sizes[b] == 0:
                assert (gi[b] == -1).all()
                continue
            ox, oi = cbind.fps(pts[off[b]:off[b + 1]], K, 1)
            np.testing.assert_array_equal(gi[b].cpu().numpy(), oi)
            np.testing.assert_array_equal(gx[b].cpu().numpy(), ox)


@pytest.mark.parametrize("seed", [11, 12])
def test_fps_soak_slice(seed, capsys):
    import stress_fps
    _no_mismatch(stress_fps, 4, seed, capsys)


@pytest.mark.parametrize("seed", [41])
def test_fps_sliced_launch_soak_slice(seed, capsys):
    import stress_fps_sliced
    _no_mismatch(stress_fps_sliced, 6, seed, capsys)


@pytest.mark.parametrize("seed", [21])
def test_stage1_soak_slice(seed, capsys):
    import stress_stage1
    _no_mismatch(stress_stage1, 5, seed, capsys)


@pytest.mark.parametrize("seed", [31])
def test_filters_soak_slice(seed, capsys):
    import stress_filters
    _no_mismatch(stress_filters, 5, seed, capsys)
