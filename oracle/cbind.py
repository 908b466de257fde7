"""TEST INFRASTRUCTURE — ctypes binding of oracle/libpcp_oracle.so (pcp_oracle.c).

Checker / cpu_baseline only; never imported by the product path."""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.environ.get("PCP_ORACLE_SO") or os.path.join(_HERE, "libpcp_oracle.so")   # override: the ASan build (Makefile)


class Intr(C.Structure):
    _fields_ = [("fx", C.c_double), ("fy", C.c_double), ("cx", C.c_double), ("cy", C.c_double)]


class Params(C.Structure):
    """Mirror of pcp_params / orc_params."""
    _fields_ = [("depth", Intr), ("color", Intr), ("T_d2c", C.c_double * 16),
                ("ext", C.c_double * 16), ("ws_lo", C.c_double * 3), ("ws_hi", C.c_double * 3),
                ("z_min", C.c_float), ("z_max", C.c_float), ("zc_min", C.c_double),
                ("zc_clamp", C.c_double), ("apply_crop", C.c_int32), ("transform_fma", C.c_int32)]


def build(force=False):
    src = os.path.join(_HERE, "pcp_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "libpcp_oracle.so"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = C.CDLL(_SO)
        _lib.orc_backproject_align_crop.restype = C.c_int64
        _lib.orc_crop_mask.restype = C.c_int64
        _lib.orc_frame_pipeline.restype = C.c_int64
    return _lib


def make_params(dintr, cintr, T=None, ext=None, ws=None, apply_crop=0, transform_fma=1,
                z_min=0.01, z_max=5.0, zc_min=0.01, zc_clamp=0.001):
    from . import numpy_ref as R
    p = Params()
    for dst, src in ((p.depth, dintr), (p.color, cintr)):
        dst.fx, dst.fy, dst.cx, dst.cy = (float(src[k]) for k in ("fx", "fy", "cx", "cy"))
    T = R.T_DEPTH_TO_COLOR if T is None else T
    ext = R.BASE_TO_EXT_CAM_S1 if ext is None else ext
    ws = R.WORK_SPACE if ws is None else ws
    p.T_d2c[:] = [float(v) for v in np.asarray(T, dtype=np.float64).reshape(-1)]
    p.ext[:] = [float(v) for v in np.asarray(ext, dtype=np.float64).reshape(-1)]
    p.ws_lo[:] = [float(ws[a][0]) for a in range(3)]
    p.ws_hi[:] = [float(ws[a][1]) for a in range(3)]
    p.z_min, p.z_max = z_min, z_max
    p.zc_min, p.zc_clamp = zc_min, zc_clamp
    p.apply_crop, p.transform_fma = int(apply_crop), int(transform_fma)
    return p


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def backproject_dense(depth, rgb, params):
    H, W = depth.shape
    Hc, Wc = rgb.shape[:2]
    n = H * W
    depth = np.ascontiguousarray(depth, dtype=np.float32)
    rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
    out = dict(pts_cam=np.empty((n, 3), np.float64), colors=np.empty((n, 3), np.float32),
               proj_mask=np.empty(n, np.uint8), valid=np.empty(n, np.uint8),
               pts_base=np.empty((n, 3), np.float64), crop_mask=np.empty(n, np.uint8))
    lib().orc_backproject_dense(_ptr(depth), _ptr(rgb), H, W, Hc, Wc, C.byref(params),
                                _ptr(out["pts_cam"]), _ptr(out["colors"]), _ptr(out["proj_mask"]),
                                _ptr(out["valid"]), _ptr(out["pts_base"]), _ptr(out["crop_mask"]))
    return out


def backproject_align_crop(depth, rgb, params):
    H, W = depth.shape
    Hc, Wc = rgb.shape[:2]
    n = H * W
    depth = np.ascontiguousarray(depth, dtype=np.float32)
    rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
    cam = np.empty((n, 3), np.float64)
    g64 = np.empty((n, 3), np.float64)
    g32 = np.empty((n, 3), np.float32)
    c32 = np.empty((n, 3), np.float32)
    pix = np.empty(n, np.int32)
    m = lib().orc_backproject_align_crop(_ptr(depth), _ptr(rgb), H, W, Hc, Wc, C.byref(params),
                                         _ptr(cam), _ptr(g64), _ptr(g32), _ptr(c32), _ptr(pix))
    return dict(xyz_cam64=cam[:m].copy(), xyz64=g64[:m].copy(), xyz32=g32[:m].copy(),
                rgb32=c32[:m].copy(), pix=pix[:m].copy())


def crop_mask(xyz, lo, hi):
    xyz = np.ascontiguousarray(xyz, dtype=np.float64)
    mask = np.empty(xyz.shape[0], np.uint8)
    lo = np.asarray(lo, np.float64)
    hi = np.asarray(hi, np.float64)
    lib().orc_crop_mask(_ptr(xyz), C.c_int64(xyz.shape[0]), _ptr(lo), _ptr(hi), _ptr(mask))
    return mask.astype(bool)


def fps(points, K, dist_mode=0):
    pts = np.ascontiguousarray(points, dtype=np.float32)
    m = pts.shape[0]
    idx = np.empty(K, np.int64)
    out = np.empty((K, 3), np.float32)
    scratch = np.empty(max(m, 1), np.float32)
    lib().orc_fps(_ptr(pts), C.c_int64(m), int(K), int(dist_mode), _ptr(idx), _ptr(out), _ptr(scratch))
    return out, idx


def gather_rgb(rgb, idx, compat_neg=1):
    rgb = np.ascontiguousarray(rgb, dtype=np.float32)
    idx = np.ascontiguousarray(idx, dtype=np.int64)
    out = np.empty((idx.shape[0], 3), np.float32)
    lib().orc_gather_rgb(_ptr(rgb), C.c_int64(rgb.shape[0]), _ptr(idx), int(idx.shape[0]),
                         int(compat_neg), _ptr(out))
    return out


def pack_dp3(xyz, rgb):
    xyz = np.ascontiguousarray(xyz, dtype=np.float32)
    rgb = np.ascontiguousarray(rgb, dtype=np.float32)
    out = np.empty((xyz.shape[0], 6), np.float32)
    lib().orc_pack_dp3(_ptr(xyz), _ptr(rgb), C.c_int64(xyz.shape[0]), _ptr(out))
    return out


def frame_pipeline(depth, rgb, params, K, dist_mode=0):
    """Whole per-frame path (stage 1 + crop → FPS → gather): the cpu_baseline unit."""
    H, W = depth.shape
    Hc, Wc = rgb.shape[:2]
    n = H * W
    depth = np.ascontiguousarray(depth, dtype=np.float32)
    rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
    xyz32 = np.empty((n, 3), np.float32)
    rgb32 = np.empty((n, 3), np.float32)
    scratch = np.empty(n, np.float32)
    idx = np.empty(K, np.int64)
    oxyz = np.empty((K, 3), np.float32)
    orgb = np.empty((K, 3), np.float32)
    m = lib().orc_frame_pipeline(_ptr(depth), _ptr(rgb), H, W, Hc, Wc, C.byref(params), int(K),
                                 int(dist_mode), _ptr(xyz32), _ptr(rgb32), _ptr(scratch),
                                 _ptr(idx), _ptr(oxyz), _ptr(orgb))
    return dict(M=int(m), idx=idx, xyz=oxyz, rgb=orgb)


def voxel_downsample(xyz, rgb, voxel):
    xyz = np.ascontiguousarray(xyz, dtype=np.float64)
    M = xyz.shape[0]
    rgb = None if rgb is None else np.ascontiguousarray(rgb, dtype=np.float32)
    o64 = np.empty((M, 3)); o32 = np.empty((M, 3), np.float32); orgb = np.empty((M, 3), np.float32)
    cnt = np.empty(M, np.int32); clamped = C.c_int32(0)
    f = lib().orc_voxel_downsample
    f.restype = C.c_int64
    nv = f(_ptr(xyz), _ptr(rgb), C.c_int64(M), C.c_double(voxel), _ptr(o64), _ptr(o32), _ptr(orgb), _ptr(cnt),
           C.byref(clamped))
    return dict(xyz64=o64[:nv].copy(), xyz32=o32[:nv].copy(), rgb32=orgb[:nv].copy() if rgb is not None else None,
                count=cnt[:nv].copy(), clamped=int(clamped.value))


def statistical_outlier(xyz, nb_neighbors=20, std_ratio=2.0):
    xyz = np.ascontiguousarray(xyz, dtype=np.float64)
    M = xyz.shape[0]
    avg = np.zeros(M); keep = np.zeros(M, np.uint8); stats = np.zeros(3)
    f = lib().orc_statistical_outlier
    f.restype = C.c_int64
    kept = f(_ptr(xyz), C.c_int64(M), int(nb_neighbors), C.c_double(std_ratio), _ptr(avg), _ptr(keep), _ptr(stats))
    return dict(keep=keep.astype(bool), avg=avg, mean=stats[0], std=stats[1], threshold=stats[2], kept=int(kept))
