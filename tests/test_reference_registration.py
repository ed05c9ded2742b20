"""CPU, build container only: the MI355X odometry plugs into the reference's own ODOMETRY registry (reference imported
from /root/reference through oracle/shims).  Skipped where the reference checkout is absent (the GPU box)."""
import os
import sys

import pytest

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "slam")), reason="reference checkout not present")


@pytest.fixture()
def reference_on_path():
    added = [os.path.join(ROOT, "oracle", "shims"), REF]
    sys.path[:0] = added
    yield
    for p in added:
        sys.path.remove(p)


def test_registers_in_reference_enum_and_dispatches(reference_on_path):
    import logging
    logging.disable(logging.WARNING)
    import torch
    import slam.odometry as ref_odometry
    import slam.slam as ref_slam
    from slam.odometry.odometry import OdometryAlgorithm as RefABC
    from pylidar_slam_amd.register import ALGORITHM_NAME, register_with_reference
    from pylidar_slam_amd.odometry import MI355XICPFrameToModel, SphericalProjector
    from pylidar_slam_amd._lib import IcpLibraryError

    patched = register_with_reference()
    assert ALGORITHM_NAME in patched.__members__ and "icp_F2M" in patched.__members__
    assert ref_odometry.ODOMETRY is patched and ref_slam.ODOMETRY is patched
    assert register_with_reference() is patched  # idempotent
    # same abstract surface as the reference ABC
    for name in ("init", "process_next_frame", "do_process_next_frame", "get_relative_poses", "get_elapsed",
                 "pointcloud_key", "relative_pose_key"):
        assert hasattr(MI355XICPFrameToModel, name) and hasattr(RefABC, name)
    assert MI355XICPFrameToModel.relative_pose_key() == RefABC.relative_pose_key()
    assert MI355XICPFrameToModel.pointcloud_key() == RefABC.pointcloud_key()
    # the reference's loader reaches our constructor with its usual kwargs (slam/slam.py:103)
    from omegaconf import DictConfig
    cfg = DictConfig({"algorithm": ALGORITHM_NAME, "max_num_alignments": 20, "data_key": "numpy_pc"})
    kwargs = dict(projector=SphericalProjector(64, 1024), pose=None, device=torch.device("cpu"), viz_num_pointclouds=1)
    if torch.cuda.is_available():
        odo = patched.load(cfg, **kwargs)
        assert isinstance(odo, MI355XICPFrameToModel) and odo.config.max_num_alignments == 20
    else:
        with pytest.raises(IcpLibraryError):  # no GPU here: loud failure, not a CPU fallback
            patched.load(cfg, **kwargs)
