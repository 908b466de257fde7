#!/usr/bin/env python3
"""
Generates the golden vectors under tests/golden/ by IMPORTING the reference's own
stage-1 functions (deproject, align_colors) from /root/reference and running them
on seeded synthetic inputs.  Run in the build container only (the reference does
not exist on the GPU box); the .npz/.json outputs are committed, this script is
the record of how they were made.

    python tests/golden/make_golden.py

The reference module's top-level imports of cv2 / open3d / rosbag2_py / rclpy are
absent here; empty stand-in modules get it past those import lines (none of them
is touched by deproject / align_colors).  The two inline expressions of the
reference's frame loop that are not functions — the validity mask (XPC:109) and
the base-frame transform (XPC:119) — are evaluated here on the reference's outputs.

FPS and the workspace crop have no runnable reference here (pytorch3d / open3d are
absent), so no golden vectors exist for them: see oracle/numpy_ref.py.
"""
import hashlib
import json
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/data_processing_pipeline"


def load_reference():
    sys.dont_write_bytecode = True
    for name in ("cv2", "open3d", "rosbag2_py", "rclpy", "rclpy.serialization",
                 "rosidl_runtime_py", "rosidl_runtime_py.utilities"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["rclpy.serialization"].deserialize_message = None
    sys.modules["rosidl_runtime_py.utilities"].get_message = None
    sys.path.insert(0, REF)
    import extract_point_clouds_one_cam as xpc
    return xpc


EXT_S1 = np.fromstring(
    "-0.5777,0.4385,-0.6884,0.7814,0.7992,0.1327,-0.5862,0.4271,"
    "-0.1657,-0.8889,-0.4271,0.5267,0.0,0.0,0.0,1.0", sep=",").reshape(4, 4)


def intr(fx, fy, cx, cy):
    return {k: np.float64(v) for k, v in zip(("fx", "fy", "cx", "cy"), (fx, fy, cx, cy))}


def run_frame(xpc, depth, rgb, dintr, cintr, T, ext):
    pts_cam = xpc.deproject(depth, dintr)
    colors, proj = xpc.align_colors(pts_cam, rgb, cintr, T)
    flat = depth.reshape(-1)
    mask = proj & (flat > 0.01) & (flat < 5.0)                                     # XPC:109
    pts_v = pts_cam[mask]
    pts_g = (ext @ np.column_stack([pts_v, np.ones(len(pts_v))]).T).T[:, :3]        # XPC:119
    return dict(pts_cam=pts_cam, colors=colors, proj_mask=proj, mask=mask, pts_g=pts_g)


def pack_inputs(depth, rgb, dintr, cintr, T, ext):
    return dict(depth=depth, rgb=rgb,
                dintr=np.array([dintr[k] for k in ("fx", "fy", "cx", "cy")], np.float64),
                cintr=np.array([cintr[k] for k in ("fx", "fy", "cx", "cy")], np.float64),
                T=np.asarray(T), ext=np.asarray(ext))


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def synth(seed, H, W, Hc=None, Wc=None):
    """SURVEY.md §8d synthetic frame: u16 millimetres / 1000 → f32 metres (XTP:68-72)."""
    rng = np.random.default_rng(seed)
    depth_mm = rng.integers(300, 1500, (H, W)).astype(np.uint16)
    depth = (depth_mm.astype(np.float32) / 1000.0).astype(np.float32)
    rgb = rng.integers(0, 256, (Hc or H, Wc or W, 3), dtype=np.uint8)
    return depth, rgb


def main():
    xpc = load_reference()
    T = xpc.T_DEPTH_TO_COLOR
    assert T.dtype == np.float32 and float(T[0, 3]) == -0.05900000035762787

    # 1. small frame, same-size colour image, S1 extrinsics
    depth, rgb = synth(11, 48, 64)
    d, c = intr(60.5, 60.5, 32.0, 24.0), intr(61.25, 60.75, 31.5, 23.75)
    out = run_frame(xpc, depth, rgb, d, c, T, EXT_S1)
    np.savez_compressed(os.path.join(HERE, "stage1_48x64.npz"),
                        **pack_inputs(depth, rgb, d, c, T, EXT_S1), **out)

    # 2. colour image of a different resolution, float32 fallback extrinsics (XPC:141)
    depth, rgb = synth(12, 64, 96, 40, 60)
    d, c = intr(90.0, 91.0, 47.5, 31.5), intr(56.0, 57.0, 30.0, 20.0)
    ext32 = np.array(xpc.BASE_TO_EXT_CAM, dtype=np.float32)
    out = run_frame(xpc, depth, rgb, d, c, T, ext32)
    np.savez_compressed(os.path.join(HERE, "stage1_64x96_c40x60.npz"),
                        **pack_inputs(depth, rgb, d, c, T, ext32), **out)

    # 3. edge cases: range limits of XPC:109, zero / huge depth, NaN / inf, and
    #    identity T + power-of-two intrinsics so u_c lands exactly on .5 (half-to-even)
    H, W = 32, 32
    rng = np.random.default_rng(13)
    depth_mm = rng.integers(300, 1500, (H, W)).astype(np.uint16)
    specials = [0, 9, 10, 11, 4999, 5000, 5001, 65535, 1, 1000]
    for i, s in enumerate(specials * 6):
        depth_mm.reshape(-1)[7 * i + 3] = s
    depth = (depth_mm.astype(np.float32) / 1000.0).astype(np.float32)
    depth[5, 5] = np.nan
    depth[6, 6] = np.inf
    depth[7, 7] = -1.0
    depth[8, :] = 1.0          # z = 1: x' = u - cx exactly
    rgb = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    d = intr(1.0, 1.0, 0.0, 0.0)
    c = intr(1.0, 1.0, 0.5, 2.5)           # u_c = x + 0.5 → ties on every pixel of row 8
    T_id = np.eye(4, dtype=np.float32)
    with np.errstate(all="ignore"):
        out = run_frame(xpc, depth, rgb, d, c, T_id, EXT_S1)
    np.savez_compressed(os.path.join(HERE, "stage1_edge_32x32.npz"),
                        **pack_inputs(depth, rgb, d, c, T_id, EXT_S1), **out)

    # 4. the benchmark frame (BASELINE.json configs[0]): seed 0, 480x640, f = 605.
    #    Inputs are regenerated from the seed by the tests; only digests are stored.
    digests = {}
    for seed, (H, W, f) in ((0, (480, 640, 605.0)), (1, (480, 640, 605.0)), (0, (720, 1280, 910.0))):
        depth, rgb = synth(seed, H, W)
        d = c = intr(f, f, W / 2, H / 2)
        out = run_frame(xpc, depth, rgb, d, c, T, EXT_S1)
        key = f"seed{seed}_{H}x{W}"
        digests[key] = dict(seed=seed, H=H, W=W, f=f,
                            depth_sha=sha(depth), rgb_sha=sha(rgb),
                            M1=int(out["mask"].sum()),
                            pts_cam_sha=sha(out["pts_cam"]), colors_sha=sha(out["colors"]),
                            proj_mask_sha=sha(out["proj_mask"].astype(np.uint8)),
                            mask_sha=sha(out["mask"].astype(np.uint8)),
                            pts_g_sha=sha(out["pts_g"]),
                            pts_g_head=out["pts_g"][:4].tolist(),
                            pts_g_colsum=out["pts_g"].sum(axis=0).tolist())
    with open(os.path.join(HERE, "stage1_digests.json"), "w") as fh:
        json.dump(digests, fh, indent=1)
    print("golden vectors written to", HERE)


if __name__ == "__main__":
    main()
