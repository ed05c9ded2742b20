"""Dev tool: wall-clock of the full odometry loop on two other BASELINE.json shapes (not the bench metric).
  S1  C1-like plumbing: 64x1024 scans -> GridSample 0.3 m -> 20-iter ICP, sliding map of 20 clouds, CV init
  S2  C4-size on ONE GPU: 128x1563 scan (200k pts) vs a 1M-point map, 20 iterations"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pylidar-slam_amd"))
from pylidar_slam_amd.synthetic import SceneConfig, make_sequence
from pylidar_slam_amd.odometry import (MI355XICPConfig, MI355XICPFrameToModel, SphericalProjector, GridSample,
                                       GridSampleConfig, ConstantVelocityInitialization)
from pylidar_slam_amd.engine import IcpContext

def s1(frames=40):
    cfg = SceneConfig(height=64, width=1024)
    scans, gt = make_sequence(cfg, frames)
    oc = MI355XICPConfig(max_num_alignments=20, threshold_delta_pose=1e-4, data_key="sample_points",
                         alignment=dict(mode="point_to_plane_gauss_newton",
                                        gauss_newton_config=dict(max_iters=1, scheme="neighborhood", sigma=0.2)))
    odo = MI355XICPFrameToModel(oc, projector=SphericalProjector(64, 1024), device=torch.device("cuda:0"))
    odo.init(); init = ConstantVelocityInitialization(); init.init()
    gs = GridSample(GridSampleConfig(voxel_size=0.3), ctx=odo.ctx)
    times = []; errs = []
    for f, s in enumerate(scans):
        d = {"numpy_pc": s}
        t0 = time.perf_counter()
        init.next_frame(d); gs.filter(d); odo.process_next_frame(d)
        times.append(time.perf_counter() - t0)
        if f:
            init.save_real_motion(d["odometry_pose"], d)
            rel = np.linalg.inv(gt[f - 1]) @ gt[f]
            errs.append(np.linalg.norm(rel[:3, 3] - d["odometry_pose"][:3, 3]))
    t = np.array(times[5:])
    print(f"S1 C1-like loop (host numpy in, grid_sample + ICP, lazy normals): median {np.median(t)*1e3:.2f} ms/frame "
          f"({1/np.median(t):.0f} frames/s), map {odo.ctx.map_size()} pts, targets {d['sample_points'].shape[0]}, "
          f"iters last {odo.last_result.iterations}, max |t-t_gt| {max(errs)*1e3:.2f} mm")

def s2():
    cfg = SceneConfig(height=128, width=1563, up_fov=22.5, down_fov=-22.5)
    scans, gt = make_sequence(cfg, 7)
    rng = np.random.default_rng(0)
    clouds = []
    for k in range(6):
        rel = np.linalg.inv(gt[5]) @ gt[k]
        clouds.append(scans[k].astype(np.float64) @ rel[:3, :3].T + rel[:3, 3])
    cloud = np.concatenate(clouds)
    model = cloud[np.sort(rng.choice(cloud.shape[0], 1_000_000, replace=False))].astype(np.float32)
    ctx = IcpContext(height=128, width=1563, up_fov=22.5, down_fov=-22.5, max_num_alignments=20, threshold_delta_pose=0.0,
                     scheme="geman_mcclure", sigma=0.3)
    dm = torch.from_numpy(model).cuda(); ds = torch.from_numpy(scans[6]).cuda()
    ts = []
    for r in range(6):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        ctx.map_set(dm); res = ctx.register(ds)
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    rel = np.linalg.inv(gt[5]) @ gt[6]
    print(f"S2 C4-size, 1 GPU: {scans[6].shape[0]} pts vs 1M map, 20 iters: {min(ts)*1e3:.2f} ms (map build + registration), "
          f"normals computed {res.normals_computed}, |t-t_gt| {np.linalg.norm(rel[:3,3]-res.pose[:3,3])*1e3:.2f} mm")

def s3(frames=30):
    """The reference's published projective configuration (docs/results/KITTI/kitti_benchmark.md:12,21: 116.6 ms/frame on
    an unnamed CUDA GPU): 64x720 range image, 15 iterations, local map of 20 maps, neighborhood sigma 0.2, CV init."""
    cfg = SceneConfig(height=64, width=720)
    scans, gt = make_sequence(cfg, frames)
    oc = MI355XICPConfig(max_num_alignments=15, threshold_delta_pose=1e-4, data_key="vertex_map",
                         local_map=dict(type="projective_local_map", local_map_size=20),
                         alignment=dict(mode="point_to_plane_gauss_newton",
                                        gauss_newton_config=dict(max_iters=1, scheme="neighborhood", sigma=0.2)))
    proj = SphericalProjector(64, 720)
    odo = MI355XICPFrameToModel(oc, projector=proj, device=torch.device("cuda:0"))
    odo.init(); init = ConstantVelocityInitialization(); init.init()
    times, errs = [], []
    for f, s in enumerate(scans):
        vm = odo.ctx.project(torch.from_numpy(s).cuda())   # the dataset's projection (kitti_dataset.py:249)
        d = {"vertex_map": vm}
        torch.cuda.synchronize(); t0 = time.perf_counter()
        init.next_frame(d); odo.process_next_frame(d)
        times.append(time.perf_counter() - t0)
        if f:
            init.save_real_motion(d["odometry_pose"], d)
            rel = np.linalg.inv(gt[f - 1]) @ gt[f]
            errs.append(np.linalg.norm(rel[:3, 3] - d["odometry_pose"][:3, 3]))
    t = np.array(times[5:])
    print(f"S3 projective F2M (64x720, 15 iters max, 20 maps): median {np.median(t)*1e3:.2f} ms/frame "
          f"({1/np.median(t):.0f} frames/s), maps {odo.ctx.pmap_num_maps()}, iters last {odo.last_result.iterations}, "
          f"max |t-t_gt| {max(errs)*1e3:.2f} mm")


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    if which in ("all", "s1"):
        s1()
    if which in ("all", "s2"):
        s2()
    if which in ("all", "s3"):
        s3()
