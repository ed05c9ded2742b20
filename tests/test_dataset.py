"""Dataset side of the hot path: host-side file parsing on CPU, GPU correction / projection under
`-m gpu`."""
import os

import numpy as np
import pytest

from conftest import GOLDEN


def _write_kitti_tree(root, scans, poses_cam, tr):
    seq = os.path.join(root, "sequences", "00")
    os.makedirs(os.path.join(seq, "velodyne"))
    os.makedirs(os.path.join(root, "poses"))
    for i, s in enumerate(scans):
        s.astype(np.float32).tofile(os.path.join(seq, "velodyne", f"{i:06}.bin"))
    with open(os.path.join(seq, "calib.txt"), "w") as f:
        for k in ("P0", "P1"):
            f.write(f"{k}: " + " ".join(f"{v:.12e}" for v in np.eye(3, 4).ravel()) + "\n")
        f.write("Tr: " + " ".join(f"{v:.12e}" for v in tr[:3].ravel()) + "\n")
    np.savetxt(os.path.join(root, "poses", "00.txt"), poses_cam[:, :3].reshape(-1, 12), fmt="%.12e")


def _kitti_fixture(tmp_path, frames=3):
    from pylidar_slam_amd.synthetic import SceneConfig, make_sequence
    scans, gt = make_sequence(SceneConfig(height=16, width=256), frames)
    rng = np.random.default_rng(3)
    scans4 = [np.concatenate([s, rng.uniform(0, 1, (s.shape[0], 1)).astype(np.float32)], axis=1) for s in scans]
    tr = np.eye(4)
    tr[:3, :3] = np.array([[0.0, -1.0, 0.0], [0.0, 0.0, -1.0], [1.0, 0.0, 0.0]])  # velodyne -> camera axes
    tr[:3, 3] = [0.01, -0.07, -0.27]
    poses_cam = np.einsum("ij,njk,kl->nil", tr, gt, np.linalg.inv(tr))
    _write_kitti_tree(str(tmp_path), scans4, poses_cam, tr)
    return scans4, gt, tr


def test_kitti_file_parsers(tmp_path):
    from pylidar_slam_amd.dataset import kitti_read_scan, read_calib_file, read_ground_truth_file
    scans4, gt, tr = _kitti_fixture(tmp_path)
    root = str(tmp_path)
    s = kitti_read_scan(os.path.join(root, "sequences", "00", "velodyne", "000001.bin"))
    assert s.dtype == np.float32 and np.array_equal(s, scans4[1])
    calib = read_calib_file(os.path.join(root, "sequences", "00", "calib.txt"))
    assert set(calib) == {"P0", "P1", "Tr"} and calib["Tr"].dtype == np.float32
    np.testing.assert_allclose(calib["Tr"].reshape(3, 4), tr[:3], atol=1e-6)
    poses = read_ground_truth_file(os.path.join(root, "poses", "00.txt"))
    assert poses.shape == (3, 4, 4) and np.array_equal(poses[:, 3], np.tile([0, 0, 0, 1.0], (3, 1)))


def test_relative_poses_and_loader_surface():
    from pylidar_slam_amd.dataset import DatasetLoader, compute_relative_poses
    from pylidar_slam_amd.synthetic import SceneConfig, trajectory
    poses = trajectory(SceneConfig(), 5)
    rel = compute_relative_poses(poses)
    assert np.array_equal(rel[0], np.eye(4))
    acc = np.eye(4)
    for r in rel[1:]:
        acc = acc @ r
    np.testing.assert_allclose(poses[0] @ acc, poses[-1], atol=1e-12)
    # the reference's plugin surface (slam/dataset/configuration.py:31-119)
    for name in ("projector", "sequences", "get_ground_truth", "absolute_gt_key", "numpy_pc_key", "max_num_workers"):
        assert hasattr(DatasetLoader, name)
    with pytest.raises(TypeError):
        DatasetLoader(None)  # abstract


# ----------------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
def test_kitti_correct_scan_gpu_matches_reference():
    import torch
    assert torch.cuda.is_available()
    from pylidar_slam_amd.engine import IcpContext
    g = np.load(os.path.join(GOLDEN, "kitti_correct.npz"))
    ctx = IcpContext(height=16, width=256)
    r = ctx.kitti_correct_scan(g["scan"])
    assert r.dtype == np.float64 and r.shape == g["corrected"].shape
    assert np.array_equal(np.isnan(r), np.isnan(g["corrected"]))
    np.testing.assert_allclose(r, g["corrected"], atol=1e-12, equal_nan=True)
    # [N,3] input (stride 3) gives the same bits; empty input is a no-op
    ok = ~np.isnan(r).any(axis=1)
    assert np.array_equal(ctx.kitti_correct_scan(g["scan"][:, :3])[ok], r[ok])
    assert ctx.kitti_correct_scan(np.zeros((0, 4), np.float32)).shape == (0, 3)
    with pytest.raises(AssertionError):
        ctx.kitti_correct_scan(np.zeros((5, 2), np.float32))


@pytest.mark.gpu
def test_kitti_sequence_round_trip(tmp_path):
    import torch
    import icp_oracle as O
    from pylidar_slam_amd.dataset import KITTIConfig, KITTIDatasetLoader
    from pylidar_slam_amd.eval import compute_relative_poses
    scans4, gt, tr = _kitti_fixture(tmp_path)
    loader = KITTIDatasetLoader(KITTIConfig(kitti_sequence_dir=str(tmp_path), lidar_height=16, lidar_width=256,
                                            train_sequences=["00"], eval_sequences=[], test_sequences=["00", "17"]))
    (train, names), (ev, _), (test, tnames), _ = loader.sequences()
    assert names == ["00"] and ev is None and tnames == ["00"]  # absent sequences are dropped
    seq = train[0]
    assert len(seq) == 3
    item = seq[1]
    want = O.kitti_correct_scan(scans4[1]).astype(np.float32)
    np.testing.assert_allclose(item["numpy_pc"], want, atol=1e-6)
    assert item["numpy_pc"].dtype == np.float32
    vm = item["vertex_map"]
    assert not vm.is_cuda and tuple(vm.shape) == (3, 16, 256)  # CPU items: the reference DataLoader pins them
    ovm = O.build_projection_map(item["numpy_pc"], 16, 256, 3.0, -24.0)
    assert (np.abs(vm.cpu().numpy() - ovm).max(axis=0) > 0).sum() <= 2
    # ground truth comes back in the lidar frame
    np.testing.assert_allclose(item["absolute_pose_gt"].numpy(), gt[1], atol=1e-5)
    np.testing.assert_allclose(loader.get_ground_truth("00"), compute_relative_poses(gt), atol=1e-5)
    assert loader.get_ground_truth("05") is None
    assert loader.projector().height == 16


@pytest.mark.gpu
def test_synthetic_loader_feeds_the_odometry():
    import torch
    from pylidar_slam_amd.dataset import SyntheticDatasetConfig, SyntheticDatasetLoader
    from pylidar_slam_amd.odometry import MI355XICPConfig, MI355XICPFrameToModel
    loader = SyntheticDatasetLoader(SyntheticDatasetConfig(lidar_height=16, lidar_width=256, num_frames=4))
    (train, names), _, _, _ = loader.sequences()
    gt = loader.get_ground_truth(names[0])
    for key in ("numpy_pc", "vertex_map"):
        odo = MI355XICPFrameToModel(MI355XICPConfig(max_num_alignments=15, data_key=key), projector=loader.projector(),
                                    device=torch.device("cuda:0"))
        odo.init()
        for i in range(len(train[0])):
            odo.process_next_frame(train[0][i])
        rel = odo.get_relative_poses()
        assert rel.shape == (4, 4, 4)
        assert np.abs(rel[1:, :3, 3] - gt[1:, :3, 3]).max() < 0.05, key
