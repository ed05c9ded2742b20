"""Trajectory metrics for the odometry output ("eval_odometry ATE / KITTI metrics wiring").

Host-side O(F) pose algebra on [F,4,4] arrays — the quantities the reference reports for a run
(slam/eval/eval_odometry.py:74-201): relative <-> absolute poses, ATE / ARE on the relative poses, and the KITTI
odometry benchmark's segment errors (translation in m/m, rotation in rad/m over 100..800 m segments).
"""
from typing import List, Optional, Sequence, Tuple

import numpy as np

__all__ = ["compute_relative_poses", "compute_absolute_poses", "compute_cumulative_trajectory_length",
           "compute_ate", "compute_are", "compute_kitti_metrics", "KITTI_SEGMENTS"]

KITTI_SEGMENTS = (100, 200, 300, 400, 500, 600, 700, 800)


def compute_relative_poses(poses: np.ndarray) -> np.ndarray:
    """eval_odometry.py:74-83: rel[0] = poses[0], rel[i] = inv(poses[i-1]) @ poses[i]."""
    prev = np.concatenate([np.eye(4)[None], poses[:-1, :4, :4]], axis=0)
    return np.linalg.inv(prev) @ poses


def compute_absolute_poses(relative_poses: np.ndarray) -> np.ndarray:
    """eval_odometry.py:86-96: abs[0] = rel[0], abs[i+1] = abs[i] @ rel[i+1]."""
    out = relative_poses.copy()
    for i in range(out.shape[0] - 1):
        out[i + 1] = out[i] @ relative_poses[i + 1]
    return out


def compute_cumulative_trajectory_length(trajectory: np.ndarray) -> np.ndarray:
    """eval_odometry.py:99-103 (the first step is measured from the origin, as there)."""
    xyz = trajectory[:, :3, 3]
    prev = np.concatenate([np.zeros((1, 3), xyz.dtype), xyz[:-1]], axis=0)
    return np.cumsum(np.linalg.norm(prev - xyz, axis=1))


def compute_ate(relative_predicted: np.ndarray, relative_ground_truth: np.ndarray) -> Tuple[float, float]:
    """eval_odometry.py:185-193: mean and std of the per-frame translation error of the relative poses."""
    err = np.linalg.norm(relative_predicted[:, :3, 3] - relative_ground_truth[:, :3, 3], axis=1)
    ate = err.mean()
    return ate, np.sqrt(((err - ate) ** 2).mean())


def compute_are(relative_trajectory: np.ndarray, relative_ground_truth: np.ndarray) -> Tuple[float, float]:
    """eval_odometry.py:196-201: Frobenius norm of R_gt^-1 R - I per frame, mean and std."""
    diff = np.linalg.inv(relative_ground_truth[:, :3, :3]) @ relative_trajectory[:, :3, :3] - np.eye(3)
    err = np.linalg.norm(diff, axis=(1, 2))
    are = err.mean()
    return are, np.sqrt(((err - are) ** 2).mean())


def _segment_errors(trajectory: np.ndarray, ground_truth: np.ndarray, segments: Sequence[float],
                    step_size: int) -> List[dict]:
    """eval_odometry.py:130-169 (calcSequenceErrors): for every `step_size`-th start frame and every segment length,
    the pose error between the estimated and the true motion over the first span exceeding that length."""
    dist = compute_cumulative_trajectory_length(ground_truth)
    errors = []
    for first in range(0, ground_truth.shape[0], step_size):
        for seg in segments:
            beyond = np.nonzero(dist[first:] > dist[first] + seg)[0]
            if beyond.size == 0:
                continue
            last = first + int(beyond[0])
            d_gt = np.linalg.inv(ground_truth[first]) @ ground_truth[last]
            d_tr = np.linalg.inv(trajectory[first]) @ trajectory[last]
            err = np.linalg.inv(d_tr) @ d_gt
            cos = 0.5 * (err[0, 0] + err[1, 1] + err[2, 2] - 1.0)
            r_err = np.arccos(np.clip(cos, -1.0, 1.0))
            t_err = np.linalg.norm(err[:3, 3])
            errors.append({"tr_err": t_err / seg, "r_err": r_err / seg, "segment": seg,
                           "speed": seg / (0.1 * (last - first + 1)), "first_frame": first, "last_frame": last})
    return errors


def compute_kitti_metrics(trajectory: np.ndarray, ground_truth: np.ndarray,
                          segments_sizes: Sequence[float] = KITTI_SEGMENTS,
                          step_size: int = 10) -> Tuple[Optional[float], Optional[float], List[dict]]:
    """eval_odometry.py:172-182: (mean translation error [m/m], mean rotation error [rad/m], per-segment records);
    (None, None, []) when the trajectory is shorter than the shortest segment."""
    errors = _segment_errors(trajectory, ground_truth, segments_sizes, step_size)
    if not errors:
        return None, None, []
    return (float(np.mean([e["tr_err"] for e in errors])), float(np.mean([e["r_err"] for e in errors])), errors)
