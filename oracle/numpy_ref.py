"""
TEST INFRASTRUCTURE — NOT PRODUCT CODE.

CPU restatement (numpy) of the reference's DP3 preprocessing hot path, used only
as the parity checker by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  Nothing under point-cloud-processing_amd/ may import this.

Reference files (under /root/reference/data_processing_pipeline/):
  XPC = extract_point_clouds_one_cam.py     DP3 = dp3_point_cloud_processing.py
  S1  = pipeline_stage1_extract.py          CHK = chunking_data.py

Pinning status
  * backproject / colour_align / valid_mask / to_base: PINNED — checked bit-exact
    against golden vectors produced by importing the reference's own deproject and
    align_colors (tests/golden/make_golden.py, tests/test_oracle_golden.py).
  * workspace_crop: restatement of six comparisons (DP3:35-39); DP3 cannot be
    imported (it calls pytorch3d on CUDA immediately) — checked by hand-made cases.
  * fps: PARITY UNPINNED by the reference.  The arithmetic lives in pytorch3d
    (requirements.txt:10, unpinned, not vendored, not installed, not fetchable).
    This follows pytorch3d's published algorithm (sample_farthest_points, CPU and
    CUDA variants) and the reference's call site DP3:7-21; it is
    anchored by hand-derived known-answer cases only.
"""
import numpy as np

# Constants as the reference spells them ------------------------------------------------
# XPC:17-22 (float32 on purpose: -0.059 becomes -0.05900000035762787)
T_DEPTH_TO_COLOR = np.eye(4, dtype=np.float32)
T_DEPTH_TO_COLOR[0, 3] = np.float32(-0.059)
# S1:13-18 — the extrinsics the stage-1 driver really passes (parsed to float64, XPC:138)
BASE_TO_EXT_CAM_S1 = np.array(
    [-0.5777, 0.4385, -0.6884, 0.7814,
     0.7992, 0.1327, -0.5862, 0.4271,
     -0.1657, -0.8889, -0.4271, 0.5267,
     0.0, 0.0, 0.0, 1.0], dtype=np.float64).reshape(4, 4)
# XPC:24-29 — fallback when --extrinsics is absent (float32, XPC:141)
BASE_TO_EXT_CAM_FALLBACK = np.array(
    [-0.0056, 0.9985, -0.055, 0.0744,
     -0.9998, -0.0047, 0.0176, 0.0993,
     0.0173, 0.0551, 0.9983, 0.012,
     0.0, 0.0, 0.0, 1.0], dtype=np.float32).reshape(4, 4)
# DP3:29-33
WORK_SPACE = ((-0.05, 0.4), (-0.12, 0.35), (-0.008, 0.6))
# DP3:47
REFERENCE_K = 2500


def backproject(depth, intr):
    """XPC:58-64.  depth (H,W) any float dtype -> (H*W,3) float64, row-major pixels.

    x = ((u - cx) * z) / fx with u the COLUMN index, z float32; multiply first,
    then divide; x,y float64, z float32 widened by the stack."""
    rows, cols = depth.shape
    z = np.asarray(depth, dtype=np.float32)
    col_idx = np.arange(cols, dtype=np.int64)[None, :].repeat(rows, axis=0)
    row_idx = np.arange(rows, dtype=np.int64)[:, None].repeat(cols, axis=1)
    x = ((col_idx - np.float64(intr['cx'])) * z) / np.float64(intr['fx'])
    y = ((row_idx - np.float64(intr['cy'])) * z) / np.float64(intr['fy'])
    out = np.empty((rows * cols, 3), dtype=np.float64)
    out[:, 0] = x.reshape(-1)
    out[:, 1] = y.reshape(-1)
    out[:, 2] = z.reshape(-1)
    return out


def colour_align(pts_cam, rgb, cintr, T):
    """XPC:66-78 -> (colors (N,3) float32, proj_mask (N,) bool).

    T is applied as a 4x4 on homogeneous points through BLAS exactly as the
    reference does; for the reference's T (identity + tx) every row has one
    rounding, so the BLAS order cannot matter."""
    n = pts_cam.shape[0]
    homo = np.empty((n, 4), dtype=np.float64)
    homo[:, :3] = pts_cam
    homo[:, 3] = 1.0
    in_colour = (T @ homo.T).T
    zc = in_colour[:, 2]
    zdiv = np.maximum(zc, 0.001)                                   # XPC:70-71
    uc = ((in_colour[:, 0] * np.float64(cintr['fx'])) / zdiv) + np.float64(cintr['cx'])
    vc = ((in_colour[:, 1] * np.float64(cintr['fy'])) / zdiv) + np.float64(cintr['cy'])
    hc, wc = rgb.shape[0], rgb.shape[1]
    ui = np.rint(uc).astype(np.int64)                              # np.round == half-to-even, XPC:73
    vi = np.rint(vc).astype(np.int64)
    inside = (ui >= 0) & (ui < wc) & (vi >= 0) & (vi < hc)         # XPC:74
    proj = (zc > 0.01) & inside                                    # XPC:75
    colors = np.zeros((n, 3), dtype=np.float32)                    # XPC:76
    colors[proj] = rgb[vi[proj], ui[proj]] / 255.0                 # f64 divide, cast to f32 on store
    return colors, proj


def valid_mask(proj_mask, depth, z_min=0.01, z_max=5.0):
    """XPC:109 — compares run in float32 because depth is a float32 array."""
    d = np.asarray(depth, dtype=np.float32).reshape(-1)
    return proj_mask & (d > np.float32(z_min)) & (d < np.float32(z_max))


def to_base(pts, ext):
    """XPC:119 — ext @ [p,1] through BLAS (order/FMA use is BLAS's business;
    the C oracle and the HIP kernel use the k-ordered FMA chain, which equals
    OpenBLAS' result on x86-64 FMA hosts — asserted in tests)."""
    n = pts.shape[0]
    homo = np.empty((n, 4), dtype=np.float64)
    homo[:, :3] = pts
    homo[:, 3] = 1.0
    return (np.asarray(ext) @ homo.T).T[:, :3]


def to_base_nofma(pts, ext):
    """The other canonical order: ((r0*x + r1*y) + r2*z) + t, each op rounded."""
    e = np.asarray(ext, dtype=np.float64)
    x, y, z = pts[:, 0], pts[:, 1], pts[:, 2]
    cols = [((e[c, 0] * x + e[c, 1] * y) + e[c, 2] * z) + e[c, 3] for c in range(3)]
    return np.stack(cols, axis=1)


def crop_mask(xyz, ws=WORK_SPACE):
    """DP3:35-39 — six strict comparisons in the dtype of xyz (float64 from PLY)."""
    m = np.ones(xyz.shape[0], dtype=bool)
    for axis in range(3):
        lo, hi = ws[axis]
        m &= (xyz[:, axis] > lo) & (xyz[:, axis] < hi)
    return m


def workspace_crop(xyz, rgb, ws=WORK_SPACE):
    """DP3:35-42 — boolean gather, order preserved."""
    m = crop_mask(xyz, ws)
    return xyz[m], (rgb[m] if rgb is not None else None), m


def fps(points, K, start=0):
    """DP3:7-21 → pytorch3d.ops.sample_farthest_points(points[None], K=[K]).

    points (M,3) any float; cast to float32 first (DP3:10).  Distances use
    ((dx*dx)+(dy*dy))+(dz*dz) with every op rounded to float32 (dist_mode 0 /
    'nofma'; the FMA variant exists only in the C oracle where fmaf is exact).
    Returns (sampled (K,3) float32, idx (K,) int64); rows beyond M are 0 / -1.
    PARITY UNPINNED (see module docstring)."""
    pts = np.ascontiguousarray(points, dtype=np.float32)
    m = pts.shape[0]
    idx = np.full((K,), -1, dtype=np.int64)
    sampled = np.zeros((K, 3), dtype=np.float32)
    if m == 0 or K == 0:
        return sampled, idx
    px, py, pz = pts[:, 0].copy(), pts[:, 1].copy(), pts[:, 2].copy()
    running = np.full((m,), np.finfo(np.float32).max, dtype=np.float32)
    last = start
    idx[0] = last
    for k in range(1, min(K, m)):
        dx = px[last] - px
        dy = py[last] - py
        dz = pz[last] - pz
        d = ((dx * dx) + (dy * dy)) + (dz * dz)
        np.minimum(running, d, out=running)
        last = int(np.argmax(running))          # first occurrence == lowest index
        idx[k] = last
    sel = idx[idx >= 0]
    sampled[:len(sel)] = pts[sel]
    return sampled, idx


def gather_rgb(rgb, idx):
    """DP3:48 — pc_rgb[idx]; idx = -1 (pytorch3d padding) picks the LAST row."""
    return rgb[idx]


def process_point_cloud(pc_xyz, pc_rgb, K=REFERENCE_K, ws=WORK_SPACE):
    """DP3:23-52 end to end (crop → FPS → colour gather)."""
    xyz, rgb, _ = workspace_crop(pc_xyz, pc_rgb, ws)
    sampled, idx = fps(xyz, K)
    return sampled, gather_rgb(rgb, idx), idx


def stage1_frame(depth, rgb, dintr, cintr, T=T_DEPTH_TO_COLOR, ext=BASE_TO_EXT_CAM_S1):
    """One iteration of the XPC:96-125 loop minus file I/O."""
    pts_cam = backproject(depth, dintr)
    colors, proj = colour_align(pts_cam, rgb, cintr, T)
    mask = valid_mask(proj, depth)
    pts_v, clrs_v = pts_cam[mask], colors[mask]
    pts_g = to_base(pts_v, ext)
    return dict(pts_cam=pts_cam, colors=colors, proj_mask=proj, mask=mask,
                pts_v=pts_v, clrs_v=clrs_v, pts_g=pts_g)


def ply_colour_roundtrip(c):
    """What a colour survives as after Open3D's write (uchar) + read."""
    q = np.rint(np.clip(np.asarray(c, dtype=np.float64), 0.0, 1.0) * 255.0)
    return q / 255.0


def dp3_rows(xyz, rgb01):
    """CHK:134-144 → one (K,6) zarr row block: [xyz, colours*255]."""
    return np.concatenate([np.asarray(xyz, dtype=np.float64),
                           ply_colour_roundtrip(rgb01) * 255], axis=1).astype(np.float32)


# ---- config-3 filters (Open3D semantics; PARITY UNPINNED — Open3D is absent and the reference tree holds
# ---- no voxel / outlier filter) ----------------------------------------------------
def voxel_downsample(xyz, rgb, voxel):
    """Open3D PointCloud::VoxelDownSample restated: origin = min bound - voxel/2, index = floor((p-origin)/voxel),
    per-voxel mean with sums in ascending point order.  Voxel order (unspecified in Open3D): ascending (ix,iy,iz).
    Returns xyz64 (V,3), rgb32 (V,3) f32, counts (V,)."""
    xyz = np.asarray(xyz, dtype=np.float64)
    if xyz.shape[0] == 0:
        return np.zeros((0, 3)), np.zeros((0, 3), np.float32), np.zeros((0,), np.int32)
    origin = xyz.min(axis=0) - voxel * 0.5
    q = np.clip(np.floor((xyz - origin) / voxel), 0, 65535).astype(np.int64)
    key = (q[:, 0] << 32) | (q[:, 1] << 16) | q[:, 2]
    order = np.argsort(key, kind="stable")
    ks = key[order]
    gid = np.cumsum(np.r_[True, ks[1:] != ks[:-1]]) - 1
    nv = int(gid[-1]) + 1
    s = np.zeros((nv, 3)); c = np.zeros((nv, 3))
    np.add.at(s, gid, xyz[order])                                  # ufunc.at: unbuffered, strictly in order
    if rgb is not None:
        np.add.at(c, gid, np.asarray(rgb, dtype=np.float32).astype(np.float64)[order])
    cnt = np.bincount(gid, minlength=nv)
    return s / cnt[:, None], (c / cnt[:, None]).astype(np.float32), cnt.astype(np.int32)


def statistical_outlier(xyz, nb_neighbors=20, std_ratio=2.0):
    """Open3D PointCloud::RemoveStatisticalOutliers restated (brute-force KNN, self included).
    Returns keep mask, avg distances, (mean, std, threshold).  Small clouds only (O(M^2) memory)."""
    xyz = np.asarray(xyz, dtype=np.float64)
    M = xyz.shape[0]
    d = xyz[:, None, :] - xyz[None, :, :]
    d2 = ((d[..., 0] * d[..., 0]) + (d[..., 1] * d[..., 1])) + (d[..., 2] * d[..., 2])
    k = min(nb_neighbors, M)
    near = np.sqrt(np.sort(d2, axis=1)[:, :k])
    avg = np.zeros(M)
    for t in range(k):                                             # sequential, ascending distance
        avg = avg + near[:, t]
    avg = avg / float(k)
    total = 0.0
    for v in avg:
        if v > 0:
            total = total + v
    mean = total / M
    sq = 0.0
    for v in avg:
        if v > 0:
            sq = sq + (v - mean) * (v - mean)
    with np.errstate(invalid="ignore", divide="ignore"):
        std = float(np.sqrt(np.float64(sq) / np.float64(M - 1)))
    thr = mean + std_ratio * std
    return (avg > 0) & (avg < thr), avg, (mean, std, thr)
