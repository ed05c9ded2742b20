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


def test_dataset_and_filter_registries(reference_on_path, monkeypatch):
    """DATASET gains the synthetic and the MI355X KITTI loader, FILTER the GPU filters; the reference's own loaders
    (`DATASET.load`, `FILTER.load`, `Preprocessing`) reach our constructors with their usual arguments."""
    import logging
    logging.disable(logging.WARNING)
    import numpy as np
    import slam.dataset as ref_dataset
    import slam.preprocessing as ref_pre
    import slam.odometry.odometry_runner as ref_runner
    from omegaconf import DictConfig, OmegaConf
    from pylidar_slam_amd import dataset as our_dataset, odometry as our_odometry
    from pylidar_slam_amd.register import DATASET_NAMES, FILTER_NAMES, register_with_reference

    register_with_reference()
    ds, flt = ref_dataset.DATASET, ref_pre.FILTER
    assert set(DATASET_NAMES) <= set(ds.__members__) and "kitti" in ds.__members__ and ref_runner.DATASET is ds
    assert set(FILTER_NAMES) <= set(flt.__members__) and "grid_sample" in flt.__members__

    # a stand-in context: the loaders / filters are constructed by the reference's code, the GPU is not needed for that
    class _Ctx:
        def __init__(self, **kw):
            self.kw = kw

        def grid_sample(self, pc, voxel):
            return pc[:3], np.arange(3)

    monkeypatch.setattr(our_dataset, "IcpContext", _Ctx)
    loader = ds.load(DictConfig({"dataset": DATASET_NAMES[0], "lidar_height": 16, "lidar_width": 128, "num_frames": 3}))
    assert isinstance(loader, our_dataset.SyntheticDatasetLoader) and loader.config.lidar_width == 128
    proj = loader.projector()
    assert (proj.height, proj.width) == (16, 128)
    (train, names), _, _, _ = loader.sequences()
    assert len(train[0]) == 3 and loader.get_ground_truth(names[0]).shape == (3, 4, 4)

    f = flt.load(OmegaConf.create({"filter_name": FILTER_NAMES[0], "voxel_size": 0.5}), ctx=_Ctx())
    assert isinstance(f, our_odometry.GridSample) and f.config.voxel_size == 0.5
    # through the reference's own Preprocessing pipeline (filters applied in key order, slam/preprocessing.py:269-291)
    pre = ref_pre.Preprocessing(ref_pre.PreprocessingConfig(filters=OmegaConf.create(
        {"1": {"filter_name": FILTER_NAMES[0], "voxel_size": 0.5, "pointcloud_key": "numpy_pc"}})), ctx=_Ctx())
    d = {"numpy_pc": np.zeros((10, 3), np.float32)}
    pre.forward(d)
    assert d["sample_points"].shape == (3, 3) and d["sample_indices"].shape == (3,)
    # the reference's own members still load
    assert isinstance(flt.load(OmegaConf.create({"filter_name": "grid_sample"})), ref_pre.GridSample)


def test_reference_point_to_plane_align_raises_on_a_mask(reference_on_path):
    """`GaussNewtonPointToPlaneAlignment.align(.., mask=)` (slam/odometry/alignment.py:91-99) does not work in the reference
    itself: `get_residual_jac_fun` multiplies the [B, N, 6] Jacobian in place by `mask.unsqueeze(1)` = [B, 1, N, 1]
    (slam/common/optimization.py:393-394), which broadcasts to four dimensions and raises.  The MI355X alignment refuses the
    argument (AssertionError) instead of inventing a semantics the reference never executed; this pins the reference's side."""
    import logging
    logging.disable(logging.WARNING)
    import numpy as np
    import torch
    from slam.common.pose import Pose
    from slam.odometry.alignment import GaussNewtonPointToPlaneAlignment, GaussNewtonPointToPlaneConfig
    from pylidar_slam_amd.odometry import PointToPlaneAlignment
    cfg = GaussNewtonPointToPlaneConfig(gauss_newton_config=dict(max_iters=1, scheme="least_square", sigma=0.5))
    al = GaussNewtonPointToPlaneAlignment(cfg, pose=Pose("euler"))
    rng = np.random.default_rng(0)
    n = 64
    ref = torch.from_numpy(rng.normal(size=(1, n, 3)).astype(np.float32))
    tgt = ref + 0.01 * torch.from_numpy(rng.normal(size=(1, n, 3)).astype(np.float32))
    nrm = torch.nn.functional.normalize(torch.from_numpy(rng.normal(size=(1, n, 3)).astype(np.float32)), dim=-1)
    mask = torch.ones(1, n, 1)
    mask[0, ::3] = 0
    al.align(ref.clone(), tgt.clone(), nrm.clone())  # (without a mask it runs)
    with pytest.raises(RuntimeError, match="broadcast shape"):
        al.align(ref.clone(), tgt.clone(), nrm.clone(), mask=mask)
    # ... and ours says so before touching a device (no context needed to refuse)
    ours = PointToPlaneAlignment.__new__(PointToPlaneAlignment)
    with pytest.raises(AssertionError, match="mask"):
        ours.align(ref, tgt, nrm, mask=mask)
