"""CPU, build container only: the reference's OWN frame loop drives the MI355X plugins.

`SLAMRunner` (slam/odometry/odometry_runner.py:74-259) builds a torch DataLoader (its `collate_fun`, pin_memory) over a
`DatasetLoader`, then `SLAM.init()` / `SLAM.process_next_frame` (slam/slam.py:81-170) run initialization ->
preprocessing -> odometry for every frame.  Here that code — imported unmodified from /root/reference through
oracle/shims — is given the `*_mi355x` registry members of `register_with_reference()`; the GPU context behind them is
replaced by the numpy `OracleContext` (no MI355X in this container; on the GPU box the reference is absent).  The
resulting trajectory is compared with the reference's own `icp_F2M` run through the very same loop.

Skipped where the reference checkout is absent.
"""
import logging
import os
import sys

import numpy as np
import pytest
import torch

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "slam")), reason="reference checkout not present")

H, W, FRAMES = 16, 256, 4


@pytest.fixture()
def reference(monkeypatch):
    added = [os.path.join(ROOT, "oracle", "shims"), REF]
    sys.path[:0] = added
    logging.disable(logging.WARNING)
    # the reference (written for Python 3.6-3.8) still says `collections.Mapping` in its collate function
    # (slam/common/torch_utils.py:147); aliases removed in Python 3.10 are restored for the duration of the test
    import collections
    import collections.abc
    for name in ("Mapping", "Sequence", "Iterable"):
        monkeypatch.setattr(collections, name, getattr(collections.abc, name), raising=False)
    torch.set_num_threads(1)  # the reference's z-buffer races under intra-op parallelism (oracle/make_golden.py)
    from oracle_context import OracleContext
    from pylidar_slam_amd import dataset as our_dataset, odometry as our_odometry
    from pylidar_slam_amd.register import register_with_reference
    monkeypatch.setattr(our_dataset, "IcpContext", OracleContext)
    monkeypatch.setattr(our_odometry, "IcpContext", OracleContext)
    monkeypatch.setattr(our_odometry, "_SHARED", {})
    register_with_reference()
    # GridSample of the reference: numba's f64 division semantics under the pure-Python numba stub (oracle/make_golden.py)
    import slam.preprocessing as pp
    from slam.common.pointcloud import voxelise
    monkeypatch.setattr(pp, "voxelise", lambda pc, a, b, c: voxelise(pc.astype(np.float64), a, b, c))
    yield
    for p in added:
        sys.path.remove(p)


def _runner(odometry: dict, preprocessing_filter: str):
    from omegaconf import OmegaConf
    from slam.odometry.odometry_runner import SLAMRunner, SLAMRunnerConfig
    cfg = OmegaConf.create({
        "dataset": {"dataset": "synthetic_mi355x", "lidar_height": H, "lidar_width": W, "num_frames": FRAMES},
        "slam": {"initialization": {"type": "cv"},
                 "preprocessing": {"filters": {"1": {"filter_name": preprocessing_filter, "voxel_size": 0.4,
                                                     "pointcloud_key": "numpy_pc"}}},
                 "odometry": odometry, "loop_closure": None, "backend": None},  # config/slam.yaml defaults: none
        "device": "cpu", "num_workers": 0, "pin_memory": True, "save_results": False})
    return SLAMRunner(SLAMRunnerConfig(**cfg))


def _drive(runner):
    """The loop of `SLAMRunner.run_odometry` (odometry_runner.py:137-168) without its result files."""
    from torch.utils.data import DataLoader
    from slam.common.torch_utils import collate_fun
    (name, dataset), = runner.load_datasets()
    loader = DataLoader(dataset, collate_fn=collate_fun, pin_memory=True, batch_size=1, num_workers=runner.num_workers)
    slam = runner.load_slam_algorithm()
    slam.init()
    for data_dict in loader:
        slam.process_next_frame(runner._send_to_device(data_dict))
    poses = slam.get_relative_poses()
    gt = runner.ground_truth(name)
    return np.asarray(poses), np.asarray(gt), slam


def _pose_error(a, b):
    import icp_oracle as O
    return O.pose_error(np.asarray(a, np.float32), np.asarray(b, np.float32))


GN = {"scheme": "geman_mcclure", "sigma": 0.3, "max_iters": 1}


def _reference_odometry():
    return {"algorithm": "icp_F2M", "data_key": "sample_points", "max_num_alignments": 10,
            "local_map": {"type": "kdtree_local_map", "local_map_size": 20},
            "alignment": {"mode": "point_to_plane_gauss_newton", "gauss_newton_config": GN}}


def test_reference_slam_loop_drives_the_mi355x_odometry(reference):
    """ODOMETRY member `icp_F2M_mi355x` + DATASET member `synthetic_mi355x` + FILTER member `grid_sample_mi355x` under
    the reference's SLAM loop, DataLoader, collate function and pinned memory; same trajectory as the reference's own
    odometry on the same frames."""
    ours = _runner({"algorithm": "icp_F2M_mi355x", "data_key": "sample_points", "max_num_alignments": 10,
                    "local_map": {"type": "kdtree_local_map", "local_map_size": 20},
                    "alignment": {"mode": "point_to_plane_gauss_newton", "gauss_newton_config": GN}},
                   "grid_sample_mi355x")
    poses, gt, slam = _drive(ours)
    from pylidar_slam_amd.odometry import MI355XICPFrameToModel
    assert isinstance(slam.odometry, MI355XICPFrameToModel)
    assert poses.shape == (FRAMES, 4, 4) and gt.shape == (FRAMES, 4, 4)
    assert np.array_equal(poses[0], np.eye(4, dtype=poses.dtype))
    ref_poses, _, ref_slam = _drive(_runner(_reference_odometry(), "grid_sample"))
    assert type(ref_slam.odometry).__name__ == "ICPFrameToModel"
    for f in range(1, FRAMES):
        dt, dr = _pose_error(poses[f], ref_poses[f])
        assert dt < 1e-4 and dr < 1e-4, (f, dt, dr)
        dt, _ = _pose_error(poses[f], gt[f])
        assert dt < 0.05, (f, dt)  # tracks the ground truth of the synthetic drive (0.4 m per frame)
    # the constant-velocity initialisation of the reference was fed by our relative poses (slam.py:139-140)
    np.testing.assert_allclose(slam.initialization.initial_estimate, poses[-1], atol=1e-6)


def test_reference_icp_runs_on_the_mi355x_inner_seams(reference):
    """The reference's OWN `ICPFrameToModel` with LOCAL_MAP member `hashgrid_local_map_mi355x` and RIGID_ALIGNMENT
    member `point_to_plane_gauss_newton_mi355x`: its loop multiplies the returned pose with its own tensor and sums the
    returned residuals (icp_odometry.py:284-297), so types, shapes and devices of the seams are what it expects."""
    cfg = _reference_odometry()
    cfg["local_map"] = {"type": "hashgrid_local_map_mi355x", "local_map_size": 20}
    cfg["alignment"] = {"mode": "point_to_plane_gauss_newton_mi355x", "gauss_newton_config": GN}
    poses, _, slam = _drive(_runner(cfg, "grid_sample"))
    from pylidar_slam_amd.odometry import HashGridLocalMap, PointToPlaneAlignment
    assert type(slam.odometry).__name__ == "ICPFrameToModel"
    assert isinstance(slam.odometry.local_map, HashGridLocalMap)
    assert isinstance(slam.odometry.rigid_alignment, PointToPlaneAlignment)
    ref_poses, _, _ = _drive(_runner(_reference_odometry(), "grid_sample"))
    for f in range(1, FRAMES):
        dt, dr = _pose_error(poses[f], ref_poses[f])
        assert dt < 1e-4 and dr < 1e-4, (f, dt, dr)
    last = slam.odometry.local_map.get_last_frame()
    assert isinstance(last, torch.Tensor) and last.ndim == 2 and last.shape[1] == 3


def test_seams_refuse_what_they_do_not_implement(reference):
    from oracle_context import OracleContext
    from pylidar_slam_amd.odometry import (MI355XICPConfig, MI355XICPFrameToModel, PointToPlaneAlignment,
                                           SphericalProjector)
    with pytest.raises(AssertionError):  # more than one Gauss-Newton step per alignment is not silently ignored
        MI355XICPFrameToModel(MI355XICPConfig(alignment=dict(mode="point_to_plane_gauss_newton",
                                                             gauss_newton_config=dict(max_iters=3))),
                              projector=SphericalProjector(H, W))
    with pytest.raises(AssertionError):
        MI355XICPFrameToModel(MI355XICPConfig(alignment=dict(mode="no_such_alignment")),
                              projector=SphericalProjector(H, W))
    al = PointToPlaneAlignment(OracleContext())
    pts = np.zeros((4, 3), np.float32)
    with pytest.raises(AssertionError):
        al.align(pts, pts, pts, mask=np.ones(4))
    with pytest.raises(AssertionError):
        al.align(pts, pts)  # normals are required
