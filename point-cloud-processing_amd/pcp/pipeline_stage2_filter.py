#!/usr/bin/env python3
"""Drop-in for the reference's pipeline_stage2_filter.py (S2): same CLI (`--output_root`), same
directory contract (<root>/<bag>/point_clouds/global_frame → <root>/<bag>/filtered_point_clouds),
one child process per bag running the GPU dp3_point_cloud_processing.py next to this file."""
import argparse
import os
import subprocess
import sys

OUTPUT_ROOT = os.environ.get("PCP_OUTPUT_ROOT", "./processed_data")


def run_command(cmd):
    """S2:9-18: child python with PYTHONPATH removed; False on a non-zero exit."""
    print(f"\nExecuting: {sys.executable} {' '.join(cmd)}")
    env = {k: v for k, v in os.environ.items() if k != "PYTHONPATH"}
    done = subprocess.run([sys.executable] + cmd, capture_output=False, text=True, env=env)
    if done.returncode != 0:
        print(f"Error executing command: {' '.join(cmd)}")
        return False
    return True


def main(argv=None):
    ap = argparse.ArgumentParser(description="Stage 2: Point Cloud Filtering and FPS")
    ap.add_argument("--output_root", type=str, default=OUTPUT_ROOT, help="The same output_root used in Stage 1")
    args = ap.parse_args(argv)
    if not os.path.exists(args.output_root):
        print(f"Directory not found: {args.output_root}")
        return
    script = os.path.join(os.path.dirname(os.path.abspath(__file__)), "dp3_point_cloud_processing.py")
    for bag in [d for d in os.listdir(args.output_root) if os.path.isdir(os.path.join(args.output_root, d))]:
        print(f"\n=== FILTERING PCD: {bag} ===")
        src = os.path.join(args.output_root, bag, "point_clouds", "global_frame")
        dst = os.path.join(args.output_root, bag, "filtered_point_clouds")
        if not os.path.exists(src):
            print(f"global_frame not found for {bag}, skipping...")          # S2:40-42
            continue
        if not run_command([script, "--input_root", src, "--output_root", dst]):
            continue
        print(f"Stage 2 complete for: {bag}")


if __name__ == "__main__":
    main()
