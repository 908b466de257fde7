#!/usr/bin/env python3
"""CLI of stage 1 (same flags as the reference's S1:33-41); the work is pcp/stage_driver.py:stage1."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pcp import stage_driver  # noqa: E402

BASE_TO_EXT_CAM = [list(stage_driver.S1_EXTRINSICS[4 * r:4 * r + 4]) for r in range(4)]


def main(argv=None):
    cli = argparse.ArgumentParser(description="Stage 1: Topic and Point Cloud Extraction")
    cli.add_argument("--bag_dir", default=os.environ.get("PCP_BAG_DIR", "./bags"))
    cli.add_argument("--output_root", default=os.environ.get("PCP_OUTPUT_ROOT", "./processed_data"))
    cli.add_argument("--extrinsics", default=",".join(str(v) for v in stage_driver.S1_EXTRINSICS))
    a = cli.parse_args(argv)
    stage_driver.stage1(a.bag_dir, a.output_root, a.extrinsics)


if __name__ == "__main__":
    main()
