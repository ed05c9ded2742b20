#!/usr/bin/env python
"""Workload for the rocprofv3 run of the kernels OUTSIDE the C2 kd-tree frame: every secondary op at
the headline size, a few dozen times each, so that `--kernel-trace --stats` yields stable per-kernel averages.

    cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d OUT -o t -- python tools/secondary.py

Ops: projective frame (64x2048 vertex maps, window of 20 maps, 20 iterations: normal map + model re-projection +
`k_pm_iterate`), grid sample at 131 072 and 200 064 points (f32 and f64), de-skew, voxel statistics, the point-to-point
registration loop, Procrustes, icp_compact_targets.  Prints the sizes it used (tools/secondary_summary.py needs them)."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pylidar-slam_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from pylidar_slam_amd.engine import IcpContext  # noqa: E402
from pylidar_slam_amd.odometry import MI355XICPConfig, MI355XICPFrameToModel, SphericalProjector  # noqa: E402
from pylidar_slam_amd.synthetic import SceneConfig, make_fixed_map, make_sequence  # noqa: E402

REPS = 20


def timed(name, fn, reps=REPS):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3 / reps
    print(f"{name}: {ms:.3f} ms per call (wall, {reps} calls)", flush=True)
    return ms


def main():
    dev = torch.device("cuda:0")
    out = {}
    cfg = SceneConfig(height=64, width=2048)
    scans, poses = make_sequence(cfg, 24)
    d131 = [torch.from_numpy(s).to(dev) for s in scans]
    big = SceneConfig(height=128, width=1563, up_fov=22.5, down_fov=-22.5)
    s200 = make_sequence(big, 2)[0]
    d200 = torch.from_numpy(s200[1]).to(dev)
    ctx = IcpContext(height=64, width=2048, max_num_alignments=20, threshold_delta_pose=0.0, scheme="geman_mcclure",
                     sigma=0.3)
    ctx.use_torch_stream()
    # ---- grid sample (a4-a6)
    for name, pts in (("grid_sample_131k", d131[3]), ("grid_sample_200k", d200)):
        v = ctx.grid_sample(pts, 0.4)[0].shape[0]
        out[name] = {"ms": timed(name, lambda p=pts: ctx.grid_sample(p, 0.4)), "n": int(pts.shape[0]), "samples": int(v)}
    p64 = d131[3].to(torch.float64)
    out["grid_sample_f64_131k"] = {"ms": timed("grid_sample_f64_131k", lambda: ctx.grid_sample_f64(p64, 0.4)),
                                   "n": int(p64.shape[0])}
    # ---- de-skew (f1)
    ts = torch.linspace(0.0, 0.1, d131[3].shape[0], dtype=torch.float64, device=dev)
    rel = np.linalg.inv(poses[2]) @ poses[3]
    out["distort_131k"] = {"ms": timed("distort_131k", lambda: ctx.distort(d131[3], ts, rel)), "n": int(d131[3].shape[0])}
    # ---- voxel statistics (f4; host arrays in and out, like the reference filter)
    out["voxel_statistics_131k"] = {"ms": timed("voxel_statistics_131k", lambda: ctx.voxel_statistics(scans[3], 0.4), 8),
                                    "n": int(scans[3].shape[0])}
    # ---- compaction of vertex-map targets
    vm = ctx.project(ctx.grid_sample(d131[3], 0.4)[0])
    pix = vm.permute(1, 2, 0).reshape(-1, 3).contiguous()
    out["compact_targets"] = {"ms": timed("compact_targets", lambda: ctx.compact_targets(pix, 8192)), "n": int(pix.shape[0])}
    # ---- point-to-point registration loop + Procrustes (f4)
    model = make_fixed_map(cfg, scans[:8], poses[:8], ref_frame=7, num_points=100_000)
    p2p = IcpContext(height=64, width=2048, max_num_alignments=20, threshold_delta_pose=0.0)
    p2p.use_torch_stream()
    p2p.set_cost("point_to_point_gauss_newton")
    p2p.map_set(torch.from_numpy(model).to(dev))
    out["p2p_register_c2"] = {"ms": timed("p2p_register_c2", lambda: p2p.register(d131[8]), 10), "n": 131072, "m": 100000}
    out["procrustes_131k"] = {"ms": timed("procrustes_131k", lambda: ctx.weighted_procrustes(d131[8], d131[7]), 10),
                              "n": 131072}
    # ---- projective frame (a19): vertex-map input, window of 20 maps, 20 forced iterations
    pcfg = MI355XICPConfig(max_num_alignments=20, threshold_delta_pose=0.0, data_key="vertex_map",
                           local_map=dict(type="projective_local_map", local_map_size=20),
                           alignment=dict(mode="point_to_plane_gauss_newton",
                                          gauss_newton_config=dict(max_iters=1, scheme="geman_mcclure", sigma=0.3)))
    odo = MI355XICPFrameToModel(pcfg, projector=SphericalProjector(64, 2048), device=dev)
    odo.init()
    vmaps = [odo.ctx.project(p) for p in d131]
    last = None

    def frames(seq):
        nonlocal last
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for vmap in seq:
            d = {"vertex_map": vmap, "init_rpose": last}
            odo.process_next_frame(d)
            if "odometry_pose" in d:
                last = d["odometry_pose"].astype(np.float64)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) * 1e3 / len(seq)

    grow = frames(vmaps)                    # the window grows to 20 maps (buffers reallocated as it does)
    full = frames(vmaps[::-1] + vmaps[1:])  # the drive back and forth again with the window full: the steady state
    print(f"projective_frame: {full:.3f} ms per frame with the window full (wall; {grow:.3f} while it grows)", flush=True)
    out["projective_frame"] = {"ms": full, "pixels": 64 * 2048, "maps": int(odo.ctx.pmap_num_maps()), "iterations": 20}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
