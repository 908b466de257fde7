"""GPU (-m gpu): multi-camera fusion (BASELINE config 5) — per-camera params, cameras concatenated in
camera order then pixel order, then FPS on the fused cloud.  The oracle is the single-camera oracle
run once per camera and concatenated (the reference is single-camera; parity for the fusion order
is this build's definition)."""
import numpy as np
import pytest
import torch

from oracle import cbind, numpy_ref as R
from helpers import synth_frame

pytestmark = pytest.mark.gpu


def _cams(H, W, n):
    intr = [dict(fx=75.6 * W / 80 + 3 * c, fy=75.6 * W / 80 - 2 * c, cx=W / 2 + c, cy=H / 2 - c) for c in range(n)]
    base = R.BASE_TO_EXT_CAM_S1.reshape(4, 4)
    exts = []
    for c in range(n):
        th = np.deg2rad(7.0 * c)
        Rz = np.array([[np.cos(th), -np.sin(th), 0, 0.02 * c], [np.sin(th), np.cos(th), 0, -0.03 * c],
                       [0, 0, 1, 0.01 * c], [0, 0, 0, 1]])
        exts.append(Rz @ base)
    return intr, exts


def _run(F, C, H, W, K, seed0):
    from pcp import _ffi, ops
    intr, exts = _cams(H, W, C)
    frames = [[synth_frame(seed0 + f * C + c, H, W) for c in range(C)] for f in range(F)]
    depth = torch.from_numpy(np.stack([np.stack([fc[0] for fc in fr]) for fr in frames])).cuda()
    rgb = torch.from_numpy(np.stack([np.stack([fc[1] for fc in fr]) for fr in frames])).cuda()
    cams = [_ffi.make_params(intr[c], intr[c], ext=exts[c], apply_crop=True) for c in range(C)]
    s1 = ops.Stage1MultiCam(F, C, H, W, depth.device, want=("xyz64", "xyz32", "rgb32", "pix"))
    out, fused, per_image = s1.run(depth, rgb, cams)
    assert int(s1.overflow.item()) == 0
    fused_h, per_h = fused.cpu().numpy(), per_image.cpu().numpy()
    np.testing.assert_array_equal(fused_h, per_h[::C])
    refs = []
    for f in range(F):
        parts = [cbind.backproject_align_crop(*frames[f][c], cbind.make_params(intr[c], intr[c], ext=exts[c], apply_crop=1))
                 for c in range(C)]
        ref = {k: np.concatenate([p[k] for p in parts]) for k in ("xyz64", "xyz32", "rgb32", "pix")}
        lo, hi = fused_h[f], fused_h[f + 1]
        assert hi - lo == len(ref["pix"])
        for k in ref:
            np.testing.assert_array_equal(out[k][lo:hi].cpu().numpy(), ref[k], err_msg=f"frame {f} {k}")
        refs.append(ref)
    total = int(fused_h[-1])
    xyz_s, idx = ops.fps_batched(out["xyz32"][:total], fused, K, max_pts=int(np.diff(fused_h).max()))
    col = ops.gather_rgb(out["rgb32"][:total], fused, idx)
    for f in range(F):
        r_xyz, r_idx = cbind.fps(refs[f]["xyz32"], K, 0)
        np.testing.assert_array_equal(idx[f].cpu().numpy(), r_idx)
        np.testing.assert_array_equal(xyz_s[f].cpu().numpy(), r_xyz)
        np.testing.assert_array_equal(col[f].cpu().numpy(), cbind.gather_rgb(refs[f]["rgb32"], r_idx, 1))


def test_multicam_small_batch_matches_per_camera_oracle():
    _run(F=3, C=4, H=48, W=64, K=256, seed0=700)


def test_multicam_two_cameras_differing_from_plan_is_rejected():
    from pcp import _ffi, ops
    s1 = ops.Stage1MultiCam(1, 2, 8, 8, torch.device("cuda"))
    d = torch.zeros((1, 3, 8, 8), device="cuda"); c = torch.zeros((1, 3, 8, 8, 3), dtype=torch.uint8, device="cuda")
    with pytest.raises(ValueError):
        s1.run(d, c, [_ffi.make_params(dict(fx=1, fy=1, cx=0, cy=0))] * 3)


def test_multicam_config5_shape_4x640x480_to_8192():
    """One fused frame of BASELINE configs[4]: 4 x 640x480 (~1.2 M raw points) -> 8192-pt FPS (deep FPS shape)."""
    _run(F=1, C=4, H=480, W=640, K=8192, seed0=900)
