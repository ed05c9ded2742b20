"""Golden vectors for the `Distortion` filter, produced by the reference's own
slam/preprocessing.py:144-191 (imported from /root/reference through oracle/shims).  TEST INFRASTRUCTURE.

    python oracle/make_golden_distortion.py      # writes tests/golden/distortion.npz
"""
import logging
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "oracle", "shims"), "/root/reference", os.path.join(ROOT, "pylidar-slam_amd")]
logging.disable(logging.WARNING)

import numpy as np  # noqa: E402
import torch  # noqa: E402

torch.set_num_threads(1)
from slam.preprocessing import Distortion, DistortionConfig, GridSample, GridSampleConfig  # noqa: E402
from slam.common.pointcloud import voxelise  # noqa: E402
import slam.preprocessing as pp  # noqa: E402

from pylidar_slam_amd.synthetic import SceneConfig, make_sequence, pose_matrix  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def main():
    h, w = 16, 256
    scans, _ = make_sequence(SceneConfig(height=h, width=w), 1)
    pc = scans[0]
    rng = np.random.default_rng(42)
    # a spinning sensor: the timestamp follows the azimuth column, plus jitter; arbitrary offset and scale
    cols = np.tile(np.arange(w), h).astype(np.float64)
    ts = 1.6e9 + 0.1 * (cols + rng.uniform(0, 0.5, cols.shape)) / w
    out = dict(pc=pc, timestamps=ts)
    poses = {
        "small": pose_matrix(np.array([0.4, 0.02, -0.01, 0.002, -0.003, 0.012])),
        "large": pose_matrix(np.array([1.5, -0.3, 0.2, 0.05, -0.08, 0.4])),
        "identity": np.eye(4),
        "pure_translation": pose_matrix(np.array([0.7, 0.1, 0.0, 0.0, 0.0, 0.0])),
    }
    pp.voxelise = lambda p, a, b, c: voxelise(p.astype(np.float64), a, b, c)
    for name, rpose in poses.items():
        d = {"numpy_pc": pc, "numpy_pc_timestamps": ts, "init_rpose": rpose}
        Distortion(DistortionConfig(output_key="distorted")).filter(d)
        out[f"{name}_rpose"] = rpose
        out[f"{name}_distorted"] = d["distorted"]
        GridSample(GridSampleConfig(voxel_size=0.3, pointcloud_key="distorted")).filter(d)  # f64 cloud -> GridSample
        out[f"{name}_sample_indices"] = d["sample_indices"]
        print(name, d["distorted"].dtype, d["distorted"].shape, d["sample_indices"].shape)
    # pass-through cases (slam/preprocessing.py:158-163)
    d = {"numpy_pc": pc, "init_rpose": poses["small"]}
    Distortion(DistortionConfig(output_key="distorted")).filter(d)
    assert d["distorted"] is pc
    # constant timestamps -> alpha = 0 everywhere
    d = {"numpy_pc": pc, "numpy_pc_timestamps": np.full(pc.shape[0], 3.0), "init_rpose": poses["small"]}
    Distortion(DistortionConfig(output_key="distorted")).filter(d)
    out["constant_ts_distorted"] = d["distorted"]
    np.savez_compressed(os.path.join(OUT, "distortion.npz"), **out)
    print("distortion.npz", os.path.getsize(os.path.join(OUT, "distortion.npz")) // 1024, "KiB")


if __name__ == "__main__":
    main()
