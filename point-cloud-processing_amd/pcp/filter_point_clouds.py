#!/usr/bin/env python3
"""`filter_point_clouds.py` — the name the reference's README and the north-star use for its filter step.  In the
reference that step IS dp3_point_cloud_processing.py (workspace crop + FPS); this CLI runs the
same GPU path and adds the BASELINE config-3 filters in front of the FPS: `--voxel_size` (voxel-grid mean filter)
and `--sor NB,STD` (statistical outlier removal), both with Open3D semantics and no counterpart in the reference.
Without the extra flags the output equals dp3_point_cloud_processing.py's."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pcp import dp3_point_cloud_processing as dp3  # noqa: E402


def main(argv=None):
    cli = argparse.ArgumentParser(description="Crop, optional voxel / outlier filters, farthest-point sampling")
    cli.add_argument("--input_root", required=True)
    cli.add_argument("--output_root", required=True)
    cli.add_argument("--num_points", type=int, default=dp3.NUM_POINTS)
    cli.add_argument("--voxel_size", type=float, default=None, help="metres, e.g. 0.005")
    cli.add_argument("--sor", default=None, help="NB,STD, e.g. 20,2.0")
    a = cli.parse_args(argv)
    sor = None
    if a.sor:
        nb, std = a.sor.split(",")
        sor = (int(nb), float(std))
    dp3.process(a.input_root, a.output_root, voxel_size=a.voxel_size, sor=sor, num_points=a.num_points)


if __name__ == "__main__":
    main()
