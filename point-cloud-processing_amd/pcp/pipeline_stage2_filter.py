#!/usr/bin/env python3
"""CLI of stage 2 (same flag as the reference's S2:21-23); the work is pcp/stage_driver.py:stage2."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pcp import stage_driver  # noqa: E402


def main(argv=None):
    cli = argparse.ArgumentParser(description="Stage 2: Point Cloud Filtering and FPS")
    cli.add_argument("--output_root", default=os.environ.get("PCP_OUTPUT_ROOT", "./processed_data"),
                     help="The same output_root used in Stage 1")
    stage_driver.stage2(cli.parse_args(argv).output_root)


if __name__ == "__main__":
    main()
