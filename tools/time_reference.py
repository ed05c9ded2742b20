#!/usr/bin/env python
"""Times the REFERENCE's own `ICPFrameToModel` (unmodified, imported from /root/reference through oracle/shims) on the
bench workload, in the build container — context for `bench.py`'s `cpu_baseline` (which times the numpy/cKDTree port on
the GPU box, where the reference does not exist).  VERDICT r2 item 8.

    python tools/time_reference.py [--frames 6] [--threads 8]  ->  profiles/r03_reference_cpu_timing.json

One frame = what a bench step does: spherical projection of the 131 072-point scan, `register_new_frame` (20 forced
iterations against the 100 000-point map: kd-tree queries, lazy kNN normals, Gauss-Newton step), `local_map.update(pose)`
(re-expression + kd-tree rebuild).  pykdtree is replaced by scipy's cKDTree (oracle/shims/pykdtree, workers = -1) and
numba by plain Python (not on this path): the figure is the reference's PyTorch/numpy code on this container's vCPUs."""
import argparse
import json
import logging
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "oracle", "shims"), "/root/reference", os.path.join(ROOT, "pylidar-slam_amd"), ROOT]
logging.disable(logging.WARNING)

import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=6)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--threads", type=int, default=os.cpu_count())
    ap.add_argument("--no-save", action="store_true", help="print the JSON line only (bench.py's cpu_baseline calls it so)")
    args = ap.parse_args()
    torch.set_num_threads(args.threads)
    from slam.common.pose import Pose
    from slam.common.projection import SphericalProjector
    from slam.odometry.alignment import GaussNewtonPointToPlaneConfig
    from slam.odometry.icp_odometry import ICPFrameToModel, ICPFrameToModelConfig
    from slam.odometry.local_map import KdTreeLocalMapConfig
    import bench
    scans, poses, model, order, start = bench.make_workload(0, "pingpong", args.frames + args.warmup)
    cfg = ICPFrameToModelConfig(
        max_num_alignments=20, threshold_delta_pose=0.0, data_key="numpy_pc",
        local_map=KdTreeLocalMapConfig(local_map_size=20),
        alignment=GaussNewtonPointToPlaneConfig(gauss_newton_config=dict(max_iters=1, scheme="geman_mcclure", sigma=0.3)))
    projector = SphericalProjector(64, 2048, 3, 3.0, -24.0)
    odo = ICPFrameToModel(cfg, projector=projector, pose=Pose("euler"), device=torch.device("cpu"))
    odo.init()
    t0 = time.perf_counter()
    odo.local_map.set_map_pointcloud(model)
    build_s = time.perf_counter() - t0
    times, last, prev, errs = [], torch.eye(4).unsqueeze(0), start, []
    for i in range(args.frames + args.warmup):
        f = order[i % len(order)]
        scan = torch.from_numpy(scans[f])
        t0 = time.perf_counter()
        projector.build_projection_map(scan.unsqueeze(0))
        params, mat, losses = odo.register_new_frame(scan, last)
        odo.local_map.update(mat)
        times.append(time.perf_counter() - t0)
        gt = np.linalg.inv(poses[prev]) @ poses[f]
        errs.append(float(np.linalg.norm(gt[:3, 3] - mat[0].numpy()[:3, 3])))
        last, prev = mat, f
        print(f"frame {i}: {times[-1]:.2f} s, |t - t_gt| = {errs[-1]:.2e} m", file=sys.stderr, flush=True)
    timed = sorted(times[args.warmup:])
    med = timed[len(timed) // 2]
    out = {"kind": "reference", "what": "slam.odometry.icp_odometry.ICPFrameToModel (unmodified) through oracle/shims: "
                                        "projection + register_new_frame (20 iterations) + local_map.update, C2 bench "
                                        "workload (131072-pt scan vs 100000-pt map, untracked-map ping-pong)",
           "value": 1.0 / med, "unit": "scans/s", "frame_s": {"min": timed[0], "median": med, "max": timed[-1]},
           "frames": args.frames, "warmup": args.warmup, "torch_threads": torch.get_num_threads(),
           "cores": os.cpu_count(), "kd_tree": "scipy cKDTree standing in for pykdtree", "tree_build_s": build_s,
           "max_pose_error_vs_ground_truth_m": max(errs[args.warmup:]),
           "where": "the host cores of the box bench.py runs on" if args.no_save else "build container (no GPU)"}
    if not args.no_save:
        json.dump(out, open(os.path.join(ROOT, "profiles", "r03_reference_cpu_timing.json"), "w"), indent=1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
