"""Pins the HEADLINE configuration (BASELINE.json configs[1], "C2") on the reference itself.  TEST INFRASTRUCTURE.

Runs the reference's own `ICPFrameToModel.register_new_frame` (slam/odometry/icp_odometry.py:248-299; unmodified,
imported from /root/reference through oracle/shims) at the full C2 size — a 64x2048 synthetic scan (131 072 points)
against a 100 000-point map loaded with `KdTreeLocalMap.set_map_pointcloud` (slam/odometry/local_map.py:289-299), 20
forced iterations (threshold_delta_pose = 0) — for the three schemes of SURVEY.md §8(d), and stores per scheme the
final pose / parameters and the per-iteration loss and delta pose (< 4 KB in all) with the sha1 of both inputs.  The
inputs are exactly those of tests/test_gpu_parity.py::test_c2_full_size_registration_vs_reference (seeded generator).

    python oracle/make_golden_c2.py        # build container only; ~1-2 min per scheme on 8 cores

pykdtree -> scipy cKDTree (oracle/shims/pykdtree); torch pinned to 1 thread as in make_golden.py, so the f32 `J^T J`
BLAS accumulation (slam/common/optimization.py:332-333) is the single-threaded one.
"""
import hashlib
import logging
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "oracle", "shims"), "/root/reference", os.path.join(ROOT, "pylidar-slam_amd")]
logging.disable(logging.WARNING)

import numpy as np  # noqa: E402
import torch  # noqa: E402

torch.set_num_threads(int(os.environ.get("GOLDEN_TORCH_THREADS", "1")))

from slam.common.pose import Pose  # noqa: E402
from slam.common.projection import SphericalProjector  # noqa: E402
from slam.odometry.alignment import GaussNewtonPointToPlaneConfig  # noqa: E402
from slam.odometry.icp_odometry import ICPFrameToModel, ICPFrameToModelConfig  # noqa: E402
from slam.odometry.local_map import KdTreeLocalMapConfig  # noqa: E402

from pylidar_slam_amd.synthetic import SceneConfig, make_fixed_map, make_sequence  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
SCHEMES = (("least_square", 0.5), ("geman_mcclure", 0.3), ("neighborhood", 0.2))
ITERS = 20


def sha(a: np.ndarray) -> str:
    return hashlib.sha1(np.ascontiguousarray(a).tobytes()).hexdigest()


def c2_inputs():
    """The scan / map pair of the C2 parity tests: map = 100k points of scans 0..7 in the frame of pose 7, target =
    scan 8 (never part of the map), identity initial guess (0.4 m / 0.01 rad from the truth)."""
    cfg = SceneConfig(height=64, width=2048)
    scans, poses = make_sequence(cfg, 9)
    model = make_fixed_map(cfg, scans[:8], poses[:8], ref_frame=7, num_points=100_000)
    return scans[8], model, np.linalg.inv(poses[7]) @ poses[8]


def run_reference(scan, model, scheme, sigma, iters=ITERS):
    pose = Pose("euler")
    cfg = ICPFrameToModelConfig(
        max_num_alignments=iters, threshold_delta_pose=0.0, data_key="numpy_pc",
        local_map=KdTreeLocalMapConfig(local_map_size=20),
        alignment=GaussNewtonPointToPlaneConfig(gauss_newton_config=dict(max_iters=1, scheme=scheme, sigma=sigma)))
    odo = ICPFrameToModel(cfg, projector=SphericalProjector(64, 2048, 3, 3.0, -24.0), pose=pose,
                          device=torch.device("cpu"))
    odo.init()
    odo.local_map.set_map_pointcloud(model)
    dxs = []
    orig_align = odo.rigid_alignment.align

    def align(nb, tg, nm, **kw):
        o = orig_align(nb, tg, nm, **kw)
        dxs.append(o[1][0].numpy().copy())
        return o

    odo.rigid_alignment.align = align
    params, mat, losses = odo.register_new_frame(torch.from_numpy(scan))
    return dict(pose=mat[0].numpy().astype(np.float32), params=params.numpy().reshape(6).astype(np.float32),
                loss=np.array([float(l) for l in losses], np.float64), dx=np.stack(dxs).astype(np.float32))


def main():
    scan, model, gt_rel = c2_inputs()
    out = dict(scan_sha=np.array(sha(scan)), model_sha=np.array(sha(model)), gt_rel=gt_rel,
               schemes=np.array([s for s, _ in SCHEMES]), sigmas=np.array([g for _, g in SCHEMES]),
               iters=np.array(ITERS), torch_threads=np.array(torch.get_num_threads()))
    for scheme, sigma in SCHEMES:
        t0 = time.perf_counter()
        r = run_reference(scan, model, scheme, sigma)
        for k, v in r.items():
            out[f"{scheme}_{k}"] = v
        err = np.linalg.norm(gt_rel[:3, 3] - r["pose"][:3, 3])
        print(f"{scheme}: {time.perf_counter() - t0:.1f} s, {len(r['loss'])} iterations, loss {r['loss'][-1]:.4f}, "
              f"|t - t_gt| = {err:.2e} m", flush=True)
    np.savez_compressed(os.path.join(OUT, "c2_reference.npz"), **out)
    print("c2_reference.npz", os.path.getsize(os.path.join(OUT, "c2_reference.npz")), "bytes")


if __name__ == "__main__":
    main()
