"""GPU (-m gpu): pcp_fps_batched_static — the one entry point that allocates nothing (its workspace is a static device
array inside the library) — called through the C ABI and compared with pcp_fps_batched bit for bit, on two calls in a
row, the second with a smaller input (the shared array keeps the first call's contents)."""
import ctypes as C

import numpy as np
import pytest
import torch

from pcp import _ffi, ops

pytestmark = pytest.mark.gpu


def _call_both(clouds, K, dist_mode, algo):
    L = _ffi.load_library()
    sizes = [c.shape[0] for c in clouds]
    off_h = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    total, B, max_pts = int(off_h[-1]), len(clouds), max(sizes + [1])
    xyz = torch.from_numpy(np.concatenate(clouds).astype(np.float32)).cuda()
    off = torch.from_numpy(off_h).cuda()
    ref = ops.Fps(B, total, max_pts, K, "cuda", dist_mode, algo)            # pcp_fps_batched, caller workspace
    ref.run(xyz, off)
    idx = torch.full((B, K), -7, dtype=torch.int64, device="cuda")
    out = torch.full((B, K, 3), 7.0, dtype=torch.float32, device="cuda")
    flag = torch.full((1,), 5, dtype=torch.int32, device="cuda")
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    p = lambda t: C.c_void_p(t.data_ptr())
    rc = L.pcp_fps_batched_static(p(xyz), p(off), B, total, max_pts, K, dist_mode, algo, p(idx), p(out), p(flag), stream)
    assert rc == 0, L.pcp_strerror(rc)
    torch.cuda.synchronize()
    assert int(flag.item()) == 0 and int(ref.overflow.item()) == 0
    np.testing.assert_array_equal(idx.cpu().numpy(), ref.idx.cpu().numpy())
    np.testing.assert_array_equal(out.cpu().numpy(), ref.xyz.cpu().numpy())
    return idx.cpu().numpy()


def test_static_entry_point_equals_the_workspace_one_twice_in_a_row():
    L = _ffi.load_library()
    rng = np.random.default_rng(71)
    big = [rng.uniform(-0.5, 0.5, (m, 3)).astype(np.float32) for m in (60000, 70001, 1, 0, 5000, 64, 200000)]
    small = [rng.normal(0, 0.2, (m, 3)).astype(np.float32) for m in (3000, 777, 40000)]
    first = _call_both(big, 300, 1, 0)                                       # wide, split-bucket and brute... shapes
    assert first[0, 0] == 0 and (first[3] == -1).all()
    _call_both(small, 300, 1, 0)                                             # smaller input over the first call's leftovers
    # ... and it really allocates nothing: device memory in use is the same before and after a call
    xyz = torch.from_numpy(np.concatenate(small)).cuda()
    off = torch.from_numpy(np.concatenate([[0], np.cumsum([c.shape[0] for c in small])]).astype(np.int64)).cuda()
    idx = torch.empty((3, 64), dtype=torch.int64, device="cuda"); flag = torch.zeros(1, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    rc = L.pcp_fps_batched_static(C.c_void_p(xyz.data_ptr()), C.c_void_p(off.data_ptr()), 3, xyz.shape[0], 40000, 64, 1, 0,
                                  C.c_void_p(idx.data_ptr()), None, C.c_void_p(flag.data_ptr()),
                                  C.c_void_p(torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    assert rc == 0 and torch.cuda.mem_get_info()[0] == free0
    _call_both(small, 128, 0, 1)                                             # brute-force path uses the array too
    _call_both(big, 64, 0, 3)
    # beyond the static array: a clear error, nothing launched, outputs untouched
    n = 1_600_001
    idx = torch.full((1, 4), -7, dtype=torch.int64, device="cuda")
    off = torch.tensor([0, n], dtype=torch.int64, device="cuda")
    xyz = torch.zeros((n, 3), dtype=torch.float32, device="cuda")
    rc = L.pcp_fps_batched_static(C.c_void_p(xyz.data_ptr()), C.c_void_p(off.data_ptr()), 1, n, n, 4, 0, 0,
                                  C.c_void_p(idx.data_ptr()), None, None, C.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == -3 and b"capacity" in L.pcp_strerror(rc)
    torch.cuda.synchronize()
    assert (idx == -7).all()
    # exactly at the limit it runs
    n = 1_600_000
    pts = rng.uniform(-1, 1, (n, 3)).astype(np.float32)
    _call_both([pts[:800_000], pts[800_000:]], 32, 1, 0)
