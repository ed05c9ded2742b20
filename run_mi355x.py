#!/usr/bin/env python
"""`run.py` of pyLiDAR-SLAM with the MI355X plugins registered first.

    PYLIDAR_SLAM_ROOT=/path/to/pyLiDAR-SLAM python run_mi355x.py slam/odometry=icp_odometry_mi355x \
        dataset=kitti_mi355x device=cuda:0 num_workers=0 slam/preprocessing=grid_sample_mi355x \
        slam.odometry.data_key=input_data

The reference checkout is used as it is: its `run.py` does nothing but build `SLAMRunner(SLAMRunnerConfig(**cfg))` under
`hydra.main(config_path="config", config_name="slam")` (run.py:10-14).  This wrapper
  1. puts the checkout and this repository's package on sys.path,
  2. calls `pylidar_slam_amd.register_with_reference()` — the ODOMETRY / DATASET / FILTER / LOCAL_MAP / RIGID_ALIGNMENT
     enums and hydra's ConfigStore gain the `*_mi355x` members (a maintainer would add the two-line enum entries of
     INTEGRATION.md instead),
  3. adds this repository's `config/` directory (the `*_mi355x.yaml` group files) to hydra's search path, and
  4. hands over to the reference's own `run_slam`.
Nothing of the reference is copied or modified.
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REFERENCE = os.environ.get("PYLIDAR_SLAM_ROOT", "/root/reference")


def main():
    if not os.path.isdir(os.path.join(REFERENCE, "slam")):
        raise SystemExit(f"pyLiDAR-SLAM checkout not found at {REFERENCE} (set PYLIDAR_SLAM_ROOT)")
    sys.path[:0] = [REFERENCE, os.path.join(HERE, "pylidar-slam_amd")]
    from pylidar_slam_amd.register import register_with_reference
    register_with_reference()
    # hydra resolves `slam/odometry=icp_odometry_mi355x` & co. from this repository's config tree as well
    extra = f"hydra.searchpath=[file://{os.path.join(HERE, 'config')}]"
    if not any(a.startswith("hydra.searchpath") for a in sys.argv[1:]):
        sys.argv.append(extra)
    os.chdir(REFERENCE)  # run.py's relative config_path="config"
    import run as reference_run
    reference_run.run_slam()


if __name__ == "__main__":
    main()
