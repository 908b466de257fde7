#!/usr/bin/env python3
"""With-file throughput of the GPU route (SURVEY.md §8d asks for compute-only and with-PLY-I/O figures separately):
the stage-1 mirror (extract_point_clouds_one_cam.process: .npy + .png in, two PLYs per frame out, XPC:96-125) and the
stage-2 mirror (dp3_point_cloud_processing.process: global_frame PLYs in, *_filtered.ply out, DP3:68-90) timed on a
synthetic trajectory tree written to a scratch directory.  The CPU side's with-PLY figure is tests/cpu_baselines.py's.
    python tools/bench_files.py [--frames 100] [--root /tmp/pcp_files] [--json out.json]"""
import argparse, json, os, shutil, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "point-cloud-processing_amd")]
import numpy as np


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=100)
    ap.add_argument("--root", default=None)
    ap.add_argument("--json", default=None)
    ap.add_argument("--K", type=int, default=2500)          # the reference's own K (DP3:47)
    a = ap.parse_args()
    from PIL import Image
    import torch
    from pcp import synth
    import pcp.extract_point_clouds_one_cam as xpc
    import pcp.dp3_point_cloud_processing as dp3
    root = a.root or tempfile.mkdtemp(prefix="pcp_files_")
    H, W = 480, 640
    ext = os.path.join(root, "bag0", "extracted_data", "bag0")
    os.makedirs(os.path.join(ext, "camera", "rgb"), exist_ok=True)
    os.makedirs(os.path.join(ext, "camera", "depth"), exist_ok=True)
    intr = synth.intrinsics(H, W)
    json.dump({"color": intr, "depth": intr}, open(os.path.join(ext, "intrinsics.json"), "w"))
    t0 = time.perf_counter()
    for i in range(a.frames):
        d, c = synth.frame(i, H, W)
        np.save(os.path.join(ext, "camera", "depth", f"depth_{i:05d}.npy"), d)
        Image.fromarray(c).save(os.path.join(ext, "camera", "rgb", f"rgb_{i:05d}.png"), compress_level=1)
    t_make = time.perf_counter() - t0
    pcd = os.path.join(root, "bag0", "point_clouds")
    out2 = os.path.join(root, "bag0", "filtered_point_clouds")
    res = {"frames": a.frames, "tree_s": round(t_make, 2)}
    for name, fn in (("stage1", lambda: xpc.process(ext, ext, pcd, synth.S1_EXTRINSICS)),
                     ("stage2", lambda: dp3.process(os.path.join(pcd, "global_frame"), out2, verbose=False, num_points=a.K))):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        res[name + "_s"] = round(dt, 3)
        res[name + "_frames_per_s"] = round(a.frames / dt, 1)
    res["both_stages_frames_per_s"] = round(a.frames / (res["stage1_s"] + res["stage2_s"]), 1)
    n_in = len(os.listdir(os.path.join(pcd, "global_frame"))); n_out = len(os.listdir(out2))
    res["ply_bytes_stage1"] = sum(os.path.getsize(os.path.join(pcd, s, f)) for s in ("camera_frame", "global_frame")
                                  for f in os.listdir(os.path.join(pcd, s)))
    res["files"] = {"global_frame": n_in, "filtered": n_out}
    res["K"] = a.K
    print(json.dumps(res), flush=True)
    if a.json:
        json.dump(res, open(a.json, "w"), indent=1)
    if a.root is None:
        shutil.rmtree(root, ignore_errors=True)


if __name__ == "__main__":
    main()
