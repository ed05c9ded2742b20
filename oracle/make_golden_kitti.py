"""Golden vectors for the KITTI HDL-64 scan correction, produced by the reference's own
`KITTIOdometrySequence.correct_scan` (slam/dataset/kitti_dataset.py:202-231, imported from /root/reference through
oracle/shims).  TEST INFRASTRUCTURE.

    python oracle/make_golden_kitti.py      # writes tests/golden/kitti_correct.npz
"""
import logging
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "oracle", "shims"), "/root/reference", os.path.join(ROOT, "pylidar-slam_amd")]
logging.disable(logging.WARNING)

import numpy as np  # noqa: E402

from slam.dataset.kitti_dataset import KITTIOdometrySequence  # noqa: E402

from pylidar_slam_amd.synthetic import SceneConfig, make_sequence  # noqa: E402


def main():
    scans, _ = make_sequence(SceneConfig(height=16, width=256), 1)
    rng = np.random.default_rng(7)
    xyz = scans[0][rng.choice(scans[0].shape[0], 998, replace=False)]
    # edge cases: a point on the z axis (zero rotation axis -> NaN in the reference) and one in the z = 0 plane
    xyz = np.concatenate([xyz, np.array([[0.0, 0.0, 2.5], [3.0, -4.0, 0.0]], np.float32)], axis=0)
    scan = np.concatenate([xyz, rng.uniform(0, 1, (xyz.shape[0], 1)).astype(np.float32)], axis=1).astype(np.float32)
    with np.errstate(all="ignore"):
        corrected = KITTIOdometrySequence.correct_scan(scan)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "kitti_correct.npz"), scan=scan, corrected=corrected,
                        numpy_version=np.array(np.__version__))
    print(corrected.dtype, corrected.shape, np.isnan(corrected).sum())


if __name__ == "__main__":
    main()
