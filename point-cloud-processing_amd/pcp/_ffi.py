"""ctypes binding of libpcp_hip.so (include/pcp.h).  Fails loudly when the library is
missing — there is deliberately no CPU fallback on the product path."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.normpath(os.path.join(_HERE, "..", "lib", "libpcp_hip.so"))

EXPORTS = (
    "pcp_default_params", "pcp_strerror", "pcp_version", "pcp_device_count",
    "pcp_backproject_dense", "pcp_align_colors", "pcp_backproject_workspace_bytes", "pcp_backproject_align_crop",
    "pcp_backproject_align_crop_multicam", "pcp_depth_mm_to_m",
    "pcp_crop_mask", "pcp_crop_workspace_bytes", "pcp_crop_compact", "pcp_crop_compact_batched",
    "pcp_fps_workspace_bytes", "pcp_fps_batched", "pcp_fps_batched_static", "pcp_fps_static_prepare", "pcp_gather_rgb", "pcp_gather_rgb_pixels", "pcp_sampled_color_pixels", "pcp_pack_dp3",
    "pcp_voxel_workspace_bytes", "pcp_voxel_downsample", "pcp_sor_workspace_bytes", "pcp_statistical_outlier_removal",
)


class PcpError(RuntimeError):
    def __init__(self, status, msg):
        super().__init__(f"pcp status {status}: {msg}")
        self.status = status


class Intr(C.Structure):
    _fields_ = [("fx", C.c_double), ("fy", C.c_double), ("cx", C.c_double), ("cy", C.c_double)]


class Params(C.Structure):
    """pcp_params (include/pcp.h)."""
    _fields_ = [("depth", Intr), ("color", Intr), ("T_d2c", C.c_double * 16),
                ("ext", C.c_double * 16), ("ws_lo", C.c_double * 3), ("ws_hi", C.c_double * 3),
                ("z_min", C.c_float), ("z_max", C.c_float), ("zc_min", C.c_double),
                ("zc_clamp", C.c_double), ("apply_crop", C.c_int32), ("transform_fma", C.c_int32)]


def lib_path():
    return os.environ.get("PCP_HIP_LIB", _LIB)


_lib = None


def load_library():
    """CDLL handle with prototypes set; raises FileNotFoundError if the .so was not built."""
    global _lib
    if _lib is not None:
        return _lib
    path = lib_path()
    if not os.path.exists(path):
        raise FileNotFoundError(
            f"{path} not found: build it with `make -C point-cloud-processing_amd/csrc` "
            "(or __graft_entry__.build()). There is no CPU fallback.")
    L = C.CDLL(path)
    vp, i32, i64, sz = C.c_void_p, C.c_int, C.c_int64, C.c_size_t
    pp = C.POINTER(Params)
    d3 = C.POINTER(C.c_double)
    L.pcp_default_params.argtypes = [pp]; L.pcp_default_params.restype = None
    L.pcp_strerror.argtypes = [i32]; L.pcp_strerror.restype = C.c_char_p
    L.pcp_version.argtypes = []; L.pcp_version.restype = C.c_char_p
    L.pcp_device_count.argtypes = []; L.pcp_device_count.restype = i32
    L.pcp_backproject_dense.argtypes = [vp, vp, i32, i32, i32, i32, i32, pp, vp, vp, vp, vp, vp, vp, vp]
    L.pcp_backproject_dense.restype = i32
    L.pcp_align_colors.argtypes = [vp, i64, vp, i32, i32, pp, vp, vp, vp]
    L.pcp_align_colors.restype = i32
    L.pcp_backproject_workspace_bytes.argtypes = [i32, i32, i32]
    L.pcp_backproject_workspace_bytes.restype = sz
    L.pcp_backproject_align_crop.argtypes = [vp, vp, i32, i32, i32, i32, i32, pp, vp, vp, vp, vp, vp,
                                             vp, i64, vp, vp, sz, vp]
    L.pcp_backproject_align_crop.restype = i32
    L.pcp_backproject_align_crop_multicam.argtypes = [vp, vp, i32, i32, i32, i32, i32, i32, pp, vp, vp, vp, vp,
                                                      vp, vp, vp, i64, vp, vp, sz, vp]
    L.pcp_backproject_align_crop_multicam.restype = i32
    L.pcp_depth_mm_to_m.argtypes = [vp, i64, vp, vp]; L.pcp_depth_mm_to_m.restype = i32
    L.pcp_crop_mask.argtypes = [vp, i64, d3, d3, vp, vp]; L.pcp_crop_mask.restype = i32
    L.pcp_crop_workspace_bytes.argtypes = [i64]; L.pcp_crop_workspace_bytes.restype = sz
    L.pcp_crop_compact.argtypes = [vp, vp, i64, d3, d3, vp, vp, vp, i64, vp, vp, sz, vp]
    L.pcp_crop_compact.restype = i32
    L.pcp_crop_compact_batched.argtypes = [vp, vp, vp, i32, i64, d3, d3, vp, vp, vp, vp, i64, vp, sz, vp]
    L.pcp_crop_compact_batched.restype = i32
    L.pcp_fps_workspace_bytes.argtypes = [i32, i64, i64, i32, i32]; L.pcp_fps_workspace_bytes.restype = sz
    L.pcp_fps_batched.argtypes = [vp, vp, i32, i64, i64, i32, i32, i32, vp, vp, vp, vp, sz, vp]
    L.pcp_fps_batched.restype = i32
    L.pcp_fps_batched_static.argtypes = [vp, vp, i32, i64, i64, i32, i32, i32, vp, vp, vp, vp]
    L.pcp_fps_batched_static.restype = i32
    L.pcp_fps_static_prepare.argtypes = []; L.pcp_fps_static_prepare.restype = i32
    L.pcp_gather_rgb.argtypes = [vp, vp, i64, vp, i32, i32, i32, vp, vp]; L.pcp_gather_rgb.restype = i32
    L.pcp_gather_rgb_pixels.argtypes = [vp, vp, i32, i32, i32, i32, i32, pp, vp, vp, i64, vp, i32, i32, vp, vp]
    L.pcp_gather_rgb_pixels.restype = i32
    L.pcp_sampled_color_pixels.argtypes = [vp, i32, i32, i32, i32, i32, pp, vp, vp, i64, vp, i32, i32, vp, vp]
    L.pcp_sampled_color_pixels.restype = i32
    L.pcp_pack_dp3.argtypes = [vp, vp, i32, i32, vp, vp]; L.pcp_pack_dp3.restype = i32
    f64 = C.c_double
    L.pcp_voxel_workspace_bytes.argtypes = [i32, i64]; L.pcp_voxel_workspace_bytes.restype = sz
    L.pcp_voxel_downsample.argtypes = [vp, vp, vp, i32, i64, f64, f64, vp, vp, vp, vp, vp, i64, vp, vp, sz, vp]
    L.pcp_voxel_downsample.restype = i32
    L.pcp_sor_workspace_bytes.argtypes = [i32, i64]; L.pcp_sor_workspace_bytes.restype = sz
    L.pcp_statistical_outlier_removal.argtypes = [vp, vp, vp, i32, i64, i32, f64, f64, f64, vp, vp, vp, vp, vp, vp,
                                                  i64, vp, vp, sz, vp]
    L.pcp_statistical_outlier_removal.restype = i32
    _lib = L
    return L


def check(status):
    if status != 0:
        raise PcpError(status, load_library().pcp_strerror(status).decode())


def default_params():
    p = Params()
    load_library().pcp_default_params(C.byref(p))
    return p


def make_params(depth_intr, color_intr=None, T=None, ext=None, workspace=None, apply_crop=True,
                transform_fma=True):
    """Params from Python values.  Intrinsics are dicts with fx, fy, cx, cy (XPC:53-54);
    T / ext are 4x4 array-likes; workspace is ((xlo,xhi),(ylo,yhi),(zlo,zhi)) (DP3:29-33)."""
    import numpy as np
    p = default_params()
    color_intr = depth_intr if color_intr is None else color_intr
    for dst, src in ((p.depth, depth_intr), (p.color, color_intr)):
        dst.fx, dst.fy, dst.cx, dst.cy = (float(src[k]) for k in ("fx", "fy", "cx", "cy"))
    if T is not None:
        p.T_d2c[:] = [float(v) for v in np.asarray(T, dtype=np.float64).reshape(16)]
    if ext is not None:
        p.ext[:] = [float(v) for v in np.asarray(ext, dtype=np.float64).reshape(16)]
    if workspace is not None:
        p.ws_lo[:] = [float(workspace[a][0]) for a in range(3)]
        p.ws_hi[:] = [float(workspace[a][1]) for a in range(3)]
    p.apply_crop = 1 if apply_crop else 0
    p.transform_fma = 1 if transform_fma else 0
    return p
