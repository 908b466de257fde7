"""Tensor-level operators over the C ABI.  Inputs/outputs are torch ROCm tensors; every call
enqueues on torch's current HIP stream and returns without synchronising unless noted."""
import ctypes as C

import torch

from . import _ffi
from ._ffi import Params, check, load_library


def dist_mode_for(use_cuda):
    """FPS distance precision that follows the reference's device choice (DP3:11-16): `use_cuda=True` runs
    pytorch3d's CUDA kernel, whose `dist2 += diff*diff` nvcc contracts into FMAs (dist_mode 1); `use_cuda=False`
    runs the x86 CPU build, every operation rounded (dist_mode 0).  PCP_DIST_MODE=0|1 overrides both."""
    import os
    env = os.environ.get("PCP_DIST_MODE")
    if env is not None and env != "":
        if env not in ("0", "1"):
            raise ValueError("PCP_DIST_MODE must be 0 or 1")
        return int(env)
    return 1 if use_cuda else 0


def _require_cuda(*tensors):
    if not torch.cuda.is_available():
        raise RuntimeError("pcp needs a ROCm device (MI355X); there is no CPU fallback")
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise ValueError("pcp operators take device tensors")


def _chk(t, dtype, name):
    if t.dtype != dtype:
        raise ValueError(f"{name}: expected {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError(f"{name}: must be contiguous")
    return t


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _stream(device=None):
    """torch's current HIP stream ON `device` (default: the current device) as a void*."""
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def _call(device, fn, *args):
    """Run one C-ABI entry point with `device` current and torch's current stream of THAT device appended as the
    trailing `stream` argument.  libpcp launches on the current HIP device (include/pcp.h), so on a multi-GPU host a
    tensor on cuda:1 must not be processed while cuda:0 is current."""
    with torch.cuda.device(device):
        check(fn(*args, _stream(device)))


def _workspace(nbytes, device):
    ws = torch.empty(int(nbytes) + 256, dtype=torch.uint8, device=device)
    off = (-ws.data_ptr()) % 256
    return ws[off:off + int(nbytes)]


def backproject_dense(depth, rgb, params, want=("pts_cam", "colors", "proj_mask", "valid",
                                                 "pts_base", "crop_mask")):
    """deproject + align_colors + XPC:109/119 for every pixel of (B,H,W) depth."""
    _require_cuda(depth, rgb)
    _chk(depth, torch.float32, "depth"); _chk(rgb, torch.uint8, "rgb")
    B, H, W = depth.shape
    Hc, Wc = rgb.shape[1], rgb.shape[2]
    dev, n = depth.device, H * W
    spec = dict(pts_cam=((B, n, 3), torch.float64), colors=((B, n, 3), torch.float32),
                proj_mask=((B, n), torch.uint8), valid=((B, n), torch.uint8),
                pts_base=((B, n, 3), torch.float64), crop_mask=((B, n), torch.uint8))
    out = {k: torch.empty(spec[k][0], dtype=spec[k][1], device=dev) for k in want}
    g = lambda k: _ptr(out.get(k))
    _call(dev, load_library().pcp_backproject_dense,
          _ptr(depth), _ptr(rgb), B, H, W, Hc, Wc, C.byref(params), g("pts_cam"), g("colors"),
          g("proj_mask"), g("valid"), g("pts_base"), g("crop_mask"))
    return out


def align_colors(pts_cam, rgb, color_intr, T_depth_to_color):
    """XPC:66-78 on free-standing points: (colors (N,3) f32, mask (N,) u8) device tensors."""
    _require_cuda(pts_cam, rgb)
    _chk(pts_cam, torch.float64, "pts_cam"); _chk(rgb, torch.uint8, "rgb")
    n = pts_cam.shape[0]
    p = _ffi.make_params(color_intr, color_intr, T=T_depth_to_color, apply_crop=False)
    colors = torch.empty((n, 3), dtype=torch.float32, device=pts_cam.device)
    mask = torch.empty(n, dtype=torch.uint8, device=pts_cam.device)
    _call(pts_cam.device, load_library().pcp_align_colors, _ptr(pts_cam), n, _ptr(rgb), rgb.shape[0], rgb.shape[1],
          C.byref(p), _ptr(colors), _ptr(mask))
    return colors, mask


def depth_mm_to_m(depth_mm, out=None):
    """uint16 millimetres → float32 metres (XTP:68-72) on the device."""
    _require_cuda(depth_mm)
    _chk(depth_mm, torch.uint16, "depth_mm")
    if out is None:
        out = torch.empty(depth_mm.shape, dtype=torch.float32, device=depth_mm.device)
    _call(depth_mm.device, load_library().pcp_depth_mm_to_m, _ptr(depth_mm), depth_mm.numel(), _ptr(out))
    return out


class Stage1:
    """Reusable buffers for pcp_backproject_align_crop on (B,H,W) batches."""

    def __init__(self, B, H, W, device, cap=None, want=("xyz32", "rgb32")):
        L = load_library()
        self.B, self.H, self.W = B, H, W
        self.device = device = torch.device(device)
        if device.index is None:
            self.device = device = torch.device("cuda", torch.cuda.current_device())
        self.cap = int(cap if cap is not None else B * H * W)
        self.ws_bytes = L.pcp_backproject_workspace_bytes(B, H, W)
        self.ws = _workspace(self.ws_bytes, device)
        spec = dict(xyz_cam64=(3, torch.float64), xyz64=(3, torch.float64), xyz32=(3, torch.float32),
                    rgb32=(3, torch.float32), pix=(0, torch.int32))
        self.out = {}
        for k in want:
            c, dt = spec[k]
            shape = (self.cap, c) if c else (self.cap,)
            self.out[k] = torch.empty(shape, dtype=dt, device=device)
        self.offsets = torch.zeros(B + 1, dtype=torch.int64, device=device)
        self.overflow = torch.zeros(1, dtype=torch.int32, device=device)

    def run(self, depth, rgb, params, Hc=None, Wc=None):
        """rgb may be None (pass the colour image size as Hc, Wc) when no colour output was asked for: the projection
        mask needs rgb.shape only (XPC:72-75)."""
        _require_cuda(depth, rgb)
        _chk(depth, torch.float32, "depth")
        B, H, W = depth.shape
        if (B, H, W) != (self.B, self.H, self.W):
            raise ValueError("depth shape does not match the Stage1 plan")
        if rgb is None:
            if "rgb32" in self.out or Hc is None or Wc is None:
                raise ValueError("rgb = None needs Hc, Wc and a plan without the rgb32 output")
        else:
            _chk(rgb, torch.uint8, "rgb")
            Hc, Wc = rgb.shape[1], rgb.shape[2]
        g = lambda k: _ptr(self.out.get(k))
        _call(self.device, load_library().pcp_backproject_align_crop,
              _ptr(depth), _ptr(rgb), B, H, W, Hc, Wc, C.byref(params), g("xyz_cam64"), g("xyz64"),
              g("xyz32"), g("rgb32"), g("pix"), _ptr(self.offsets), self.cap, _ptr(self.overflow),
              _ptr(self.ws), self.ws_bytes)
        return self.out, self.offsets


class Stage1MultiCam(Stage1):
    """pcp_backproject_align_crop_multicam on (F,C,H,W) batches: per-camera params, fused CSR offsets."""

    def __init__(self, F, Cams, H, W, device, cap=None, want=("xyz32", "rgb32")):
        super().__init__(F * Cams, H, W, device, cap=cap, want=want)
        self.F, self.C = F, Cams
        self.fused_offsets = torch.zeros(F + 1, dtype=torch.int64, device=device)

    def run(self, depth, rgb, cams):
        _require_cuda(depth, rgb)
        _chk(depth, torch.float32, "depth"); _chk(rgb, torch.uint8, "rgb")
        F, Cn, H, W = depth.shape
        if (F, Cn, H, W) != (self.F, self.C, self.H, self.W) or len(cams) != Cn:
            raise ValueError("depth shape / camera count do not match the Stage1MultiCam plan")
        Hc, Wc = rgb.shape[2], rgb.shape[3]
        arr = (Params * Cn)(*cams)
        g = lambda k: _ptr(self.out.get(k))
        _call(self.device, load_library().pcp_backproject_align_crop_multicam,
              _ptr(depth), _ptr(rgb), F, Cn, H, W, Hc, Wc, arr, g("xyz_cam64"), g("xyz64"), g("xyz32"),
              g("rgb32"), g("pix"), _ptr(self.offsets), _ptr(self.fused_offsets), self.cap, _ptr(self.overflow),
              _ptr(self.ws), self.ws_bytes)
        return self.out, self.fused_offsets, self.offsets


def crop_mask(xyz, lo, hi):
    _require_cuda(xyz)
    _chk(xyz, torch.float64, "xyz")
    m = torch.empty(xyz.shape[0], dtype=torch.uint8, device=xyz.device)
    lo3, hi3 = (C.c_double * 3)(*lo), (C.c_double * 3)(*hi)
    _call(xyz.device, load_library().pcp_crop_mask, _ptr(xyz), xyz.shape[0], lo3, hi3, _ptr(m))
    return m


def crop_compact(xyz, rgb, lo, hi, want_xyz64=True, want_xyz32=True):
    """DP3:35-42 on one cloud.  Synchronises once to learn the surviving row count."""
    _require_cuda(xyz, rgb)
    _chk(xyz, torch.float64, "xyz")
    if rgb is not None:
        _chk(rgb, torch.float64, "rgb")
    L = load_library()
    M1, dev = xyz.shape[0], xyz.device
    nb = L.pcp_crop_workspace_bytes(M1)
    ws = _workspace(nb, dev)
    o64 = torch.empty((M1, 3), dtype=torch.float64, device=dev) if want_xyz64 else None
    o32 = torch.empty((M1, 3), dtype=torch.float32, device=dev) if want_xyz32 else None
    orgb = torch.empty((M1, 3), dtype=torch.float64, device=dev) if rgb is not None else None
    count = torch.zeros(1, dtype=torch.int64, device=dev)
    lo3, hi3 = (C.c_double * 3)(*lo), (C.c_double * 3)(*hi)
    _call(dev, L.pcp_crop_compact, _ptr(xyz), _ptr(rgb), M1, lo3, hi3, _ptr(o64), _ptr(o32), _ptr(orgb),
          M1, _ptr(count), _ptr(ws), nb)
    m = int(count.item())
    cut = lambda t: None if t is None else t[:m]
    return cut(o64), cut(o32), cut(orgb), m


def crop_compact_batched(xyz, rgb, offsets, lo, hi, want_xyz64=False):
    """DP3:35-42 on B CSR clouds in one launch sequence → (xyz64 or None, xyz32, rgb64 or None, out_offsets).
    Outputs keep the input's row capacity; cloud b's rows are [out_offsets[b], out_offsets[b+1])."""
    _require_cuda(xyz, offsets)
    _chk(xyz, torch.float64, "xyz"); _chk(offsets, torch.int64, "offsets")
    if rgb is not None:
        _chk(rgb, torch.float64, "rgb")
    L = load_library()
    n, dev, B = xyz.shape[0], xyz.device, offsets.numel() - 1
    nb = L.pcp_crop_workspace_bytes(n)
    ws = _workspace(nb, dev)
    o64 = torch.empty((n, 3), dtype=torch.float64, device=dev) if want_xyz64 else None
    o32 = torch.empty((n, 3), dtype=torch.float32, device=dev)
    orgb = torch.empty((n, 3), dtype=torch.float64, device=dev) if rgb is not None else None
    out_off = torch.empty(B + 1, dtype=torch.int64, device=dev)
    lo3, hi3 = (C.c_double * 3)(*lo), (C.c_double * 3)(*hi)
    _call(dev, L.pcp_crop_compact_batched, _ptr(xyz), _ptr(rgb), _ptr(offsets), B, n, lo3, hi3, _ptr(o64), _ptr(o32),
          _ptr(orgb), _ptr(out_off), n, _ptr(ws), nb)
    return o64, o32, orgb, out_off


class Fps:
    """Reusable workspace for pcp_fps_batched."""

    def __init__(self, B, total_cap, max_pts, K, device, dist_mode=0, algo=0):
        L = load_library()
        self.B, self.K, self.total_cap, self.max_pts = B, K, int(total_cap), int(max_pts)
        self.dist_mode, self.algo = dist_mode, algo
        self.device = device = torch.device(device)
        if device.index is None:
            self.device = device = torch.device("cuda", torch.cuda.current_device())
        self.ws_bytes = L.pcp_fps_workspace_bytes(B, self.total_cap, self.max_pts, K, algo)
        self.ws = _workspace(self.ws_bytes, device)
        self.idx = torch.empty((B, K), dtype=torch.int64, device=device)
        self.xyz = torch.empty((B, K, 3), dtype=torch.float32, device=device)
        self.overflow = torch.zeros(1, dtype=torch.int32, device=device)   # set by the kernels, see pcp.h

    def run(self, xyz, offsets):
        _require_cuda(xyz, offsets)
        _chk(xyz, torch.float32, "xyz"); _chk(offsets, torch.int64, "offsets")
        if offsets.numel() != self.B + 1:
            raise ValueError("offsets must have B+1 entries")
        if xyz.shape[0] > self.total_cap:
            raise ValueError("more points than the Fps plan was sized for")
        self.enqueue(xyz, offsets, xyz.shape[0])
        return self.xyz, self.idx

    def enqueue(self, xyz, offsets, total_pts):
        """The bare C-ABI call (no checks): rows [0, total_pts) of xyz exist."""
        _call(self.device, load_library().pcp_fps_batched,
              _ptr(xyz), _ptr(offsets), self.B, int(total_pts), self.max_pts, self.K, self.dist_mode,
              self.algo, _ptr(self.idx), _ptr(self.xyz), _ptr(self.overflow), _ptr(self.ws), self.ws_bytes)


def fps_batched(xyz, offsets, K, dist_mode=0, algo=0, max_pts=None):
    """One-shot pcp_fps_batched (allocates its plan, synchronises once to read the overflow flag): raises PcpError
    when a cloud could not be sampled — more points than `max_pts` promised, or offsets reaching past `xyz`."""
    B = offsets.numel() - 1
    if max_pts is None:
        max_pts = xyz.shape[0]
    plan = Fps(B, xyz.shape[0], max_pts, K, xyz.device, dist_mode, algo)
    out = plan.run(xyz, offsets)
    if int(plan.overflow.item()):
        raise _ffi.PcpError(-3, "a cloud exceeded max_pts / total_pts in pcp_fps_batched")
    return out


def gather_rgb(rgb, offsets, idx, compat_neg_index=True):
    _require_cuda(rgb, offsets, idx)
    _chk(rgb, torch.float32, "rgb"); _chk(offsets, torch.int64, "offsets"); _chk(idx, torch.int64, "idx")
    B, K = idx.shape
    out = torch.empty((B, K, 3), dtype=torch.float32, device=rgb.device)
    _call(rgb.device, load_library().pcp_gather_rgb, _ptr(rgb), _ptr(offsets), rgb.shape[0], _ptr(idx), B, K,
          1 if compat_neg_index else 0, _ptr(out))
    return out


def gather_rgb_pixels(depth, rgb, params, pix, offsets, idx, compat_neg_index=True, out=None):
    """pc_rgb[idx] (DP3:48) evaluated from the sampled points' source pixels (pcp_gather_rgb_pixels): same values as
    gather_rgb over stage 1's rgb32, without stage 1 ever producing rgb32."""
    _require_cuda(depth, rgb, pix, offsets, idx)
    _chk(depth, torch.float32, "depth"); _chk(rgb, torch.uint8, "rgb"); _chk(pix, torch.int32, "pix")
    _chk(offsets, torch.int64, "offsets"); _chk(idx, torch.int64, "idx")
    B, H, W = depth.shape
    K = idx.shape[1]
    if idx.shape[0] != B or offsets.numel() != B + 1:
        raise ValueError("idx / offsets do not match the batch")
    if out is None:
        out = torch.empty((B, K, 3), dtype=torch.float32, device=depth.device)
    _call(depth.device, load_library().pcp_gather_rgb_pixels, _ptr(depth), _ptr(rgb), B, H, W, rgb.shape[1], rgb.shape[2],
          C.byref(params), _ptr(pix), _ptr(offsets), pix.shape[0], _ptr(idx), K, 1 if compat_neg_index else 0, _ptr(out))
    return out


def sampled_color_pixels(depth, Hc, Wc, params, pix, offsets, idx, compat_neg_index=True, out=None):
    """Colour-image pixel (v_idx*Wc + u_idx, -1 = masked) of every sampled point (pcp_sampled_color_pixels): what a
    caller with the rgb images in HOST memory needs to finish pc_rgb[idx] (DP3:48) without ever uploading rgb."""
    _require_cuda(depth, pix, offsets, idx)
    _chk(depth, torch.float32, "depth"); _chk(pix, torch.int32, "pix")
    _chk(offsets, torch.int64, "offsets"); _chk(idx, torch.int64, "idx")
    B, H, W = depth.shape
    K = idx.shape[1]
    if idx.shape[0] != B or offsets.numel() != B + 1:
        raise ValueError("idx / offsets do not match the batch")
    if out is None:
        out = torch.empty((B, K), dtype=torch.int32, device=depth.device)
    _call(depth.device, load_library().pcp_sampled_color_pixels, _ptr(depth), B, H, W, int(Hc), int(Wc), C.byref(params),
          _ptr(pix), _ptr(offsets), pix.shape[0], _ptr(idx), K, 1 if compat_neg_index else 0, _ptr(out))
    return out


def pack_dp3(xyz, rgb):
    _require_cuda(xyz, rgb)
    _chk(xyz, torch.float32, "xyz"); _chk(rgb, torch.float32, "rgb")
    B, K = xyz.shape[0], xyz.shape[1]
    out = torch.empty((B, K, 6), dtype=torch.float32, device=xyz.device)
    _call(xyz.device, load_library().pcp_pack_dp3, _ptr(xyz), _ptr(rgb), B, K, _ptr(out))
    return out


def _csr_outputs(n, device, rgb, want_index=False):
    out = dict(xyz64=torch.empty((n, 3), dtype=torch.float64, device=device),
               xyz32=torch.empty((n, 3), dtype=torch.float32, device=device),
               rgb32=torch.empty((n, 3), dtype=torch.float32, device=device) if rgb is not None else None)
    if want_index:
        out["index"] = torch.empty((n,), dtype=torch.int64, device=device)
    return out


def voxel_downsample(xyz, rgb, offsets, voxel_size, cap_pts=None, max_extent=None):
    """Open3D-semantics voxel-grid mean filter on B CSR clouds (pcp_voxel_downsample).
    xyz (n,3) f64, rgb (n,3) f32 or None, offsets (B+1) i64 → dict(xyz64, xyz32, rgb32, count), out_offsets.
    max_extent: optional promise (metres) that no cloud's bounding box is longer along any axis — e.g. the longest
    workspace edge for cropped clouds; lets the grid sort run on 32-bit keys.  A broken promise raises the overflow flag."""
    _require_cuda(xyz, offsets)
    _chk(xyz, torch.float64, "xyz"); _chk(offsets, torch.int64, "offsets")
    if rgb is not None:
        _chk(rgb, torch.float32, "rgb")
    L = load_library()
    B, dev = offsets.numel() - 1, xyz.device
    n = int(xyz.shape[0] if cap_pts is None else cap_pts)
    nb = L.pcp_voxel_workspace_bytes(B, n)
    if nb == 0:
        raise RuntimeError("pcp_voxel_workspace_bytes failed (no device?)")
    ws = _workspace(nb, dev)
    out = _csr_outputs(n, dev, rgb)
    out["count"] = torch.empty((n,), dtype=torch.int32, device=dev)
    out_off = torch.empty(B + 1, dtype=torch.int64, device=dev)
    ovf = torch.zeros(1, dtype=torch.int32, device=dev)
    _call(dev, L.pcp_voxel_downsample, _ptr(xyz), _ptr(rgb), _ptr(offsets), B, n, float(voxel_size),
          float(max_extent or 0.0), _ptr(out["xyz64"]),
          _ptr(out["xyz32"]), _ptr(out["rgb32"]), _ptr(out["count"]), _ptr(out_off), n,
          _ptr(ovf), _ptr(ws), nb)
    out["overflow"] = ovf
    return out, out_off


def workspace_extent(params):
    """Longest edge of the workspace box of `params`: the max_extent promise for clouds cropped to it."""
    return max(float(params.ws_hi[c]) - float(params.ws_lo[c]) for c in range(3))


def sor_cell_size(xyz, n_clouds, nb_neighbors=20):
    """Search-grid edge for statistical_outlier_removal: the radius that holds ~nb_neighbors points at the
    batch's mean density (bounding-box volume / points per cloud), with 30 % head-room so that the first
    ring of cells almost always proves the k-th neighbour.  Only speed depends on it.  Synchronises."""
    n = xyz.shape[0]
    if n == 0 or n_clouds == 0:
        return 1.0
    ext = (xyz.amax(0) - xyz.amin(0)).cpu().numpy().astype(float)
    ext = [max(e, 1e-6) for e in ext]
    per_cloud = max(n / n_clouds, 1.0)
    vol = ext[0] * ext[1] * ext[2]
    cell = 1.3 * (3.0 * nb_neighbors * vol / (4.0 * 3.141592653589793 * per_cloud)) ** (1.0 / 3.0)
    return max(cell, max(ext) / 60000.0)          # keep every grid index inside 16 bits


def statistical_outlier_removal(xyz, rgb, offsets, nb_neighbors=20, std_ratio=2.0, cell_size=None, cap_pts=None,
                                max_extent=None):
    """Open3D-semantics statistical outlier removal on B CSR clouds (pcp_statistical_outlier_removal).
    cell_size None → sor_cell_size().  max_extent: see voxel_downsample.
    → dict(xyz64, xyz32, rgb32, index, avg), out_offsets."""
    _require_cuda(xyz, offsets)
    _chk(xyz, torch.float64, "xyz"); _chk(offsets, torch.int64, "offsets")
    if rgb is not None:
        _chk(rgb, torch.float32, "rgb")
    L = load_library()
    B, dev = offsets.numel() - 1, xyz.device
    n = int(xyz.shape[0] if cap_pts is None else cap_pts)
    if cell_size is None:
        cell_size = sor_cell_size(xyz[:n], B, nb_neighbors)
    nb = L.pcp_sor_workspace_bytes(B, n)
    if nb == 0:
        raise RuntimeError("pcp_sor_workspace_bytes failed (no device?)")
    ws = _workspace(nb, dev)
    out = _csr_outputs(n, dev, rgb, want_index=True)
    out["avg"] = torch.zeros((n,), dtype=torch.float64, device=dev)
    out_off = torch.empty(B + 1, dtype=torch.int64, device=dev)
    ovf = torch.zeros(1, dtype=torch.int32, device=dev)
    _call(dev, L.pcp_statistical_outlier_removal, _ptr(xyz), _ptr(rgb), _ptr(offsets), B, n, int(nb_neighbors),
          float(std_ratio), float(cell_size), float(max_extent or 0.0), _ptr(out["xyz64"]),
          _ptr(out["xyz32"]), _ptr(out["rgb32"]), _ptr(out["index"]),
          _ptr(out_off), _ptr(out["avg"]), n, _ptr(ovf), _ptr(ws), nb)
    out["overflow"] = ovf
    return out, out_off


class FramePipeline:
    """The whole hot path for batches of B frames: depth+rgb → (B,K,3) xyz, (B,K,3) rgb,
    (B,K) idx.  All buffers are allocated once; run() only enqueues kernels."""

    def __init__(self, B, H, W, K, params, device="cuda", dist_mode=None, algo=0, cap_per_frame=None):
        if dist_mode is None:                   # the loop this replaces samples with use_cuda=True (DP3:47)
            dist_mode = dist_mode_for(True)
        self.B, self.H, self.W, self.K, self.params = B, H, W, K, params
        cap = B * (cap_per_frame if cap_per_frame is not None else H * W)
        # colours are fetched for the K sampled points only (gather_rgb_pixels): stage 1 writes xyz32 + the source pixel
        self.stage1 = Stage1(B, H, W, device, cap=cap, want=("xyz32", "pix"))
        self.fps = Fps(B, cap, H * W, K, device, dist_mode, algo)
        self.rgb_out = torch.empty((B, K, 3), dtype=torch.float32, device=device)

    def run(self, depth, rgb):
        out, offsets = self.stage1.run(depth, rgb, self.params)
        f = self.fps
        # total_pts = the rows stage 1 could store: if its capacity overflowed, offsets hold the TRUE counts and the FPS
        # and gather kernels drop every cloud that reaches past `cap` (all -1 / zero rows, f.overflow set)
        f.enqueue(out["xyz32"], offsets, self.stage1.cap)
        self.gather(depth, rgb)
        return f.xyz, self.rgb_out, f.idx, offsets

    def run_host_rgb(self, depth, Hc, Wc):
        """The same path for a caller whose rgb images stay in HOST memory: nothing but depth is read on the device.
        Returns (xyz (B,K,3), cpix (B,K) i32, idx, offsets); colour = rgb_b.reshape(-1,3)[cpix] / 255, zero where
        cpix < 0 (ops.sampled_color_pixels)."""
        out, offsets = self.stage1.run(depth, None, self.params, Hc=Hc, Wc=Wc)
        f, s1 = self.fps, self.stage1
        f.enqueue(out["xyz32"], offsets, s1.cap)
        if getattr(self, "cpix_out", None) is None:
            self.cpix_out = torch.empty((self.B, self.K), dtype=torch.int32, device=f.device)
        _call(f.device, load_library().pcp_sampled_color_pixels, _ptr(depth), f.B, self.H, self.W, int(Hc), int(Wc),
              C.byref(self.params), _ptr(s1.out["pix"]), _ptr(s1.offsets), s1.cap, _ptr(f.idx), f.K, 1, _ptr(self.cpix_out))
        return f.xyz, self.cpix_out, f.idx, offsets

    def gather(self, depth, rgb):
        """pc_rgb[idx] (DP3:48) for the last run's picks → self.rgb_out."""
        f, s1 = self.fps, self.stage1
        _call(f.device, load_library().pcp_gather_rgb_pixels, _ptr(depth), _ptr(rgb), f.B, self.H, self.W,
              rgb.shape[1], rgb.shape[2], C.byref(self.params), _ptr(s1.out["pix"]), _ptr(s1.offsets), s1.cap,
              _ptr(f.idx), f.K, 1, _ptr(self.rgb_out))

    def check(self):
        """Synchronising: raise if the last run() lost rows (stage-1 capacity) or skipped a cloud (FPS bounds)."""
        flags = torch.stack([self.stage1.overflow[0], self.fps.overflow[0]]).cpu().tolist()
        if flags[0] or flags[1]:
            raise _ffi.PcpError(-3, "stage-1 row capacity exceeded" if flags[0] else
                                "a cloud exceeded max_pts / total_pts in pcp_fps_batched")
