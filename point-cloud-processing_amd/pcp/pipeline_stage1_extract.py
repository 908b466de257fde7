#!/usr/bin/env python3
"""Drop-in for the reference's pipeline_stage1_extract.py (S1): same CLI (`--bag_dir --output_root
--extrinsics`) and directory layout.  Topic extraction (extract_topics_one_camera.py) needs ROS 2
and stays the reference's CPU script: it is run if present next to this file or named by
$PCP_EXTRACT_TOPICS_SCRIPT, and skipped when its outputs already exist."""
import argparse
import os
import subprocess
import sys

BAG_DIR = os.environ.get("PCP_BAG_DIR", "./bags")
OUTPUT_ROOT = os.environ.get("PCP_OUTPUT_ROOT", "./processed_data")
BASE_TO_EXT_CAM = [[-0.5777, 0.4385, -0.6884, 0.7814],            # S1:13-18
                   [0.7992, 0.1327, -0.5862, 0.4271],
                   [-0.1657, -0.8889, -0.4271, 0.5267],
                   [0.0000, 0.0000, 0.0000, 1.0000]]


def run_command(cmd):
    """S1:20-30."""
    print(f"\nExecuting: {sys.executable} {' '.join(cmd)}")
    env = {k: v for k, v in os.environ.items() if k != "PYTHONPATH"}
    done = subprocess.run([sys.executable] + cmd, capture_output=False, text=True, env=env)
    if done.returncode != 0:
        print(f"Error executing command: {' '.join(cmd)}")
        return False
    return True


def main(argv=None):
    ap = argparse.ArgumentParser(description="Stage 1: Topic and Point Cloud Extraction")
    ap.add_argument("--bag_dir", type=str, default=BAG_DIR)
    ap.add_argument("--output_root", type=str, default=OUTPUT_ROOT)
    ap.add_argument("--extrinsics", type=str, default=",".join(str(x) for row in BASE_TO_EXT_CAM for x in row))
    args = ap.parse_args(argv)
    os.makedirs(args.output_root, exist_ok=True)
    bags = [os.path.join(args.bag_dir, d) for d in os.listdir(args.bag_dir)
            if os.path.isdir(os.path.join(args.bag_dir, d))]
    bags = [p for p in bags if any(f.endswith((".db3", ".mcap")) for f in os.listdir(p))]     # S1:44-45
    if not bags:
        print(f"No valid bags found in {args.bag_dir}")
        return
    here = os.path.dirname(os.path.abspath(__file__))
    topics_script = os.environ.get("PCP_EXTRACT_TOPICS_SCRIPT", os.path.join(here, "extract_topics_one_camera.py"))
    pcd_script = os.path.join(here, "extract_point_clouds_one_cam.py")
    for bag_path in bags:
        bag = os.path.basename(bag_path.rstrip("/"))
        print(f"\n=== PROCESSING BAG: {bag} ===")
        extracted_root = os.path.join(args.output_root, bag, "extracted_data")
        os.makedirs(extracted_root, exist_ok=True)
        actual = os.path.join(extracted_root, bag)                                    # S1:67 (doubled <bag>)
        if os.path.exists(topics_script):
            if not run_command([topics_script, "--bag_path", bag_path, "--output_root", extracted_root]):
                continue
        elif not os.path.isdir(os.path.join(actual, "camera", "rgb")):
            print(f"extract_topics_one_camera.py not available and {actual} has no extracted data, skipping...")
            continue
        pcd_root = os.path.join(args.output_root, bag, "point_clouds")
        os.makedirs(pcd_root, exist_ok=True)
        if not run_command([pcd_script, "--bag_path", bag_path, "--extracted_data_root", actual,
                            "--output_root", pcd_root, f"--extrinsics={args.extrinsics}"]):
            continue
        print(f"Stage 1 complete for: {bag}")


if __name__ == "__main__":
    main()
