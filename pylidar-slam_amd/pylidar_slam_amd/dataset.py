"""Dataset side of the hot path: the `DatasetLoader` plugin surface, a synthetic-scan loader for
BASELINE.json's synthetic configurations, and the KITTI odometry reader with the scan correction and the spherical
projection done on the MI355X.

  reference                                                             here
  DatasetLoader            slam/dataset/configuration.py:31-119         DatasetLoader
  KITTIOdometrySequence    slam/dataset/kitti_dataset.py:93-279         KITTIOdometrySequence (correct_scan + projection on GPU)
  KITTIDatasetLoader       slam/dataset/kitti_dataset.py:311-400        KITTIDatasetLoader
  read_ground_truth_file / read_calib_file   kitti_dataset.py:40-91     same names
  (none: the reference ships no synthetic data)                         SyntheticSequence / SyntheticDatasetLoader

Items are the reference's data_dict: `numpy_pc` [N,3] float32 (host), `vertex_map` [3,H,W] float32 (device tensor),
`absolute_pose_gt` [4,4].  The GPU context lives in the main process: use these datasets with num_workers = 0.
"""
from abc import ABC, abstractmethod
from dataclasses import dataclass, field
from pathlib import Path
from typing import List, Optional

import numpy as np
import torch
from torch.utils.data import Dataset

from . import eval as _eval
from .engine import IcpContext
from .odometry import SphericalProjector, assert_debug
from .synthetic import SceneConfig, ray_directions, render_scan, trajectory

__all__ = ["DatasetLoader", "SyntheticDatasetConfig", "SyntheticSequence", "SyntheticDatasetLoader", "KITTIConfig",
           "KITTIOdometrySequence", "KITTIDatasetLoader", "read_ground_truth_file", "read_calib_file",
           "kitti_read_scan", "compute_relative_poses"]


def compute_relative_poses(absolute: np.ndarray) -> np.ndarray:
    """relative[0] = I, relative[i] = inv(abs[i-1]) @ abs[i] (trajectory expressed from its first frame; the
    reference's `eval_odometry.compute_relative_poses` keeps abs[0] as the first entry — see `eval.py`)."""
    rel = np.zeros_like(absolute)
    rel[0] = np.eye(4)
    for i in range(1, absolute.shape[0]):
        rel[i] = np.linalg.inv(absolute[i - 1]) @ absolute[i]
    return rel


class DatasetLoader(ABC):
    """slam/dataset/configuration.py:31-119 (the part the odometry runner uses)."""

    @classmethod
    def max_num_workers(cls):
        return 0  # the GPU context is not shared with DataLoader worker processes

    @staticmethod
    def absolute_gt_key():
        return "absolute_pose_gt"

    @staticmethod
    def numpy_pc_key():
        return "numpy_pc"

    def __init__(self, config):
        self.config = config

    @abstractmethod
    def projector(self) -> SphericalProjector:
        raise NotImplementedError("")

    @abstractmethod
    def sequences(self):
        """((train datasets, names), (eval ...), (test ...), transform)"""
        raise NotImplementedError("")

    @abstractmethod
    def get_ground_truth(self, sequence_name):
        return None


# ----------------------------------------------------------------------------------------------------------------------
@dataclass
class SyntheticDatasetConfig:
    dataset: str = "synthetic"
    lidar_height: int = 64
    lidar_width: int = 1024
    up_fov: float = 3.0
    down_fov: float = -24.0
    num_frames: int = 10
    seed: int = 1234
    train_sequences: List[str] = field(default_factory=lambda: ["room_00"])
    lidar_key: str = "vertex_map"
    with_numpy_pc: bool = True
    device: str = "cuda:0"  # the GPU the scans are projected on
    # False (default): items are CPU tensors, so the reference runner's DataLoader(pin_memory=True) works unchanged
    # (slam/odometry/odometry_runner.py:51,149) and `_send_to_device` moves them; True: the vertex map stays on the
    # device (no D2H + H2D round trip) — needs `pin_memory=false`, a CUDA tensor cannot be pinned
    device_items: bool = False


def _device_index(device: str) -> int:
    d = torch.device(device)
    return 0 if d.index is None else int(d.index)


class SyntheticSequence(Dataset):
    def __init__(self, config: SyntheticDatasetConfig, ctx: IcpContext, name: str):
        self.config = config
        self.ctx = ctx
        offset = sum(ord(c) for c in name) % 97
        self.scene = SceneConfig(height=config.lidar_height, width=config.lidar_width, up_fov=config.up_fov,
                                 down_fov=config.down_fov, seed=config.seed + offset)
        self.dirs = ray_directions(self.scene)
        self.poses = trajectory(self.scene, config.num_frames)

    def __len__(self):
        return self.config.num_frames

    def __getitem__(self, idx) -> dict:
        assert_debug(0 <= idx < len(self))
        scan = render_scan(self.scene, self.poses[idx], idx, self.dirs)
        d = {}
        if self.config.with_numpy_pc:
            d["numpy_pc"] = scan
        vmap = self.ctx.project(torch.from_numpy(scan).to(self.ctx.device))
        d[self.config.lidar_key] = vmap if self.config.device_items else vmap.cpu()
        d[DatasetLoader.absolute_gt_key()] = torch.from_numpy(self.poses[idx])
        return d


class SyntheticDatasetLoader(DatasetLoader):
    def __init__(self, config: SyntheticDatasetConfig):
        super().__init__(config)
        self._ctx = IcpContext(height=config.lidar_height, width=config.lidar_width, up_fov=config.up_fov,
                               down_fov=config.down_fov, device=_device_index(config.device))
        self._sequences = {n: SyntheticSequence(config, self._ctx, n) for n in config.train_sequences}

    def projector(self) -> SphericalProjector:
        c = self.config
        return SphericalProjector(c.lidar_height, c.lidar_width, 3, c.up_fov, c.down_fov)

    def sequences(self):
        names = list(self._sequences)
        return ([self._sequences[n] for n in names], names), (None, None), (None, None), lambda x: x

    def get_ground_truth(self, sequence_name):
        if sequence_name not in self._sequences:
            return None
        return compute_relative_poses(self._sequences[sequence_name].poses)


# ----------------------------------------------------------------------------------------------------------------------
def kitti_read_scan(file_path: str) -> np.ndarray:
    """kitti_dataset.py:20-37: float32 records x, y, z, reflectance."""
    return np.fromfile(file_path, dtype=np.float32).reshape((-1, 4))


def read_calib_file(file_path: str) -> dict:
    """kitti_dataset.py:40-67."""
    calib = {}
    with open(file_path, "r") as f:
        for line in f.readlines():
            tokens = line.split(" ")
            if tokens[0] == "calib_time:" or len(tokens) < 2:
                continue
            calib[tokens[0][:-1]] = np.array([float(t) for t in tokens[1:]], dtype=np.float32)
    return calib


def read_ground_truth_file(file_path: str) -> np.ndarray:
    """kitti_dataset.py:70-91: [N,12] rows -> [N,4,4] (left-camera frame)."""
    poses = np.loadtxt(file_path, dtype=np.float64).reshape(-1, 12)
    n = poses.shape[0]
    poses = np.concatenate((poses, np.zeros((n, 3)), np.ones((n, 1))), axis=1)
    return poses.reshape((n, 4, 4))


@dataclass
class KITTIConfig:
    """kitti_dataset.py:283-302."""
    kitti_sequence_dir: str = ""
    dataset: str = "kitti"
    lidar_key: str = "vertex_map"
    absolute_gt_key: str = "absolute_pose_gt"
    lidar_height: int = 64
    lidar_width: int = 1024
    up_fov: float = 3.0
    down_fov: float = -24.0
    train_sequences: List[str] = field(default_factory=lambda: [f"{i:02}" for i in range(11)])
    eval_sequences: List[str] = field(default_factory=lambda: ["09", "10"])
    test_sequences: List[str] = field(default_factory=lambda: [f"{i:02}" for i in range(22)])
    with_numpy_pc: bool = True
    device: str = "cuda:0"       # the GPU scans are corrected and projected on
    device_items: bool = False   # see SyntheticDatasetConfig.device_items


class KITTIOdometrySequence(Dataset):
    """kitti_dataset.py:93-279 without the unrectified-raw branch: .bin read on the host, `correct_scan` and the
    spherical projection on the GPU (the reference burns DataLoader workers on both)."""

    def __init__(self, sequences_root_dir: str, sequence_id: str, ctx: IcpContext,
                 corrected_lidar_channel: str = "vertex_map", ground_truth_channel: Optional[str] = None,
                 with_numpy_pc: bool = False, device_items: bool = False):
        self._device_items = device_items
        self.sequence_dir = Path(sequences_root_dir)
        self.sequence_id = sequence_id
        self.ctx = ctx
        self.corrected_lidar_channel = corrected_lidar_channel
        self.ground_truth_channel = ground_truth_channel
        self._with_numpy_pc = with_numpy_pc
        self.velodyne_path = self.sequence_dir / "sequences" / sequence_id / "velodyne"
        assert_debug(self.velodyne_path.exists(), f"{self.velodyne_path} does not exist")
        self.size = len(sorted(self.velodyne_path.glob("*.bin")))
        calib_path = self.sequence_dir / "sequences" / sequence_id / "calib.txt"
        assert_debug(calib_path.exists() and calib_path.is_file())
        self.calibration_matrices = {}
        calib = read_calib_file(str(calib_path))
        if "Tr" in calib:
            self.calibration_matrices["Tr"] = np.concatenate(
                (calib["Tr"].reshape(3, 4), np.array([[0, 0, 0, 1]], dtype=np.float32)), axis=0)
        self.poses_gt = None
        if ground_truth_channel:
            gt_file = self.sequence_dir / "poses" / f"{sequence_id}.txt"
            if gt_file.exists() and gt_file.is_file():
                self.poses_gt = self._lidar_pose_gt(read_ground_truth_file(str(gt_file)))

    def _lidar_pose_gt(self, poses_gt: np.ndarray):  # :186-194
        if "Tr" in self.calibration_matrices:
            tr = self.calibration_matrices["Tr"].astype(np.float64)
            return np.einsum("...ij,...jk->...ik", np.einsum("...ij,...jk->...ik", np.linalg.inv(tr), poses_gt), tr)
        return poses_gt

    def __len__(self):
        return self.size

    def correct_scan(self, scan: np.ndarray) -> np.ndarray:
        """:202-231 on the GPU.  float32 out: the reference's numpy expression yields float32 under the numpy 1.x it
        was written for and float64 under numpy 2; the GPU kernel computes in float64 and this rounds once."""
        return self.ctx.kitti_correct_scan(scan).astype(np.float32)

    def __getitem__(self, idx) -> dict:
        assert_debug(idx < self.size)
        d = {}
        scan_path = self.velodyne_path / f"{idx:06}.bin"
        assert_debug(scan_path.exists() and scan_path.is_file(), f"The file {scan_path} does not exist")
        scan = self.correct_scan(kitti_read_scan(str(scan_path)))
        if self._with_numpy_pc:
            d["numpy_pc"] = scan
        vmap = self.ctx.project(torch.from_numpy(scan).to(self.ctx.device))
        d[self.corrected_lidar_channel] = vmap if self._device_items else vmap.cpu()
        if self.ground_truth_channel and self.poses_gt is not None:
            d[self.ground_truth_channel] = torch.from_numpy(self.poses_gt[idx])
        return d


class KITTIDatasetLoader(DatasetLoader):
    """kitti_dataset.py:311-400."""

    def __init__(self, config: KITTIConfig):
        super().__init__(config)
        self.odometry_sequence_dir = Path(config.kitti_sequence_dir)
        assert_debug(self.odometry_sequence_dir.exists())
        self._ctx = IcpContext(height=config.lidar_height, width=config.lidar_width, up_fov=config.up_fov,
                               down_fov=config.down_fov, device=_device_index(config.device))

    def projector(self) -> SphericalProjector:
        c = self.config
        return SphericalProjector(c.lidar_height, c.lidar_width, 3, c.up_fov, c.down_fov)

    def get_ground_truth(self, sequence_name):  # :333-347
        gt = self.odometry_sequence_dir / "poses" / f"{sequence_name}.txt"
        if not gt.exists():
            return None
        poses = read_ground_truth_file(str(gt)).astype(np.float64)
        calib = read_calib_file(str(self.odometry_sequence_dir / "sequences" / sequence_name / "calib.txt"))
        tr = np.eye(4, dtype=np.float64)
        tr[:3, :4] = calib["Tr"].reshape(3, 4)
        right = np.einsum("...ij,...jk->...ik", np.einsum("...ij,...jk->...ik", np.linalg.inv(tr), poses), tr)
        return _eval.compute_relative_poses(right)  # the reference's own convention: rel[0] = abs[0]

    def sequences(self):  # :349-400
        c = self.config

        def get(seqs):
            if not seqs:
                return None
            present = [s for s in seqs if (self.odometry_sequence_dir / "sequences" / s / "velodyne").exists()]
            return [KITTIOdometrySequence(str(self.odometry_sequence_dir), s, self._ctx, c.lidar_key, c.absolute_gt_key,
                                          with_numpy_pc=c.with_numpy_pc, device_items=c.device_items)
                    for s in present], present

        tr, ev, te = get(c.train_sequences), get(c.eval_sequences), get(c.test_sequences)
        return tr or (None, None), ev or (None, None), te or (None, None), lambda x: x
