"""dev: where the HOST spends a frame of the published-configuration loop (bench.py::odometry_loop_leg with timers around every call)"""
import os, sys, time, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "pylidar-slam_amd")]
import numpy as np, torch
from pylidar_slam_amd.odometry import (ConstantVelocityInitialization, Distortion, DistortionConfig, GridSample, GridSampleConfig,
                                       MI355XICPConfig, MI355XICPFrameToModel, SphericalProjector, ToDevice, ToDeviceConfig, ToTensor, ToTensorConfig)
from pylidar_slam_amd.synthetic import SceneConfig, make_sequence
from pylidar_slam_amd import engine as E
dev = torch.device("cuda", 0)
frames = 36
scans, _ = make_sequence(SceneConfig(height=64, width=2048), frames)
cfg = MI355XICPConfig(max_num_alignments=20, threshold_delta_pose=1.0e-4, data_key="input_data",
                      local_map=dict(type="kdtree_local_map", local_map_size=30, num_neighbors_normals=10),
                      alignment=dict(mode="point_to_plane_gauss_newton", gauss_newton_config=dict(max_iters=1, scheme="neighborhood", sigma=0.2)))
odo = MI355XICPFrameToModel(cfg, projector=SphericalProjector(64, 2048), device=dev)
filters = [ToDevice(ToDeviceConfig(device=str(dev)), device=dev),
           Distortion(DistortionConfig(pointcloud_key="pc_device", timestamps_key="timestamps_device", output_key="distorted")),
           GridSample(GridSampleConfig(voxel_size=0.4, pointcloud_key="distorted", padded=True)),
           ToTensor(ToTensorConfig(device=str(dev), keys={"sample_points": "input_data"}, dtype="float32"), device=dev)]
init = ConstantVelocityInitialization()
acc = collections.defaultdict(float); cnt = collections.Counter()
def timed(obj, name, label):
    f = getattr(obj, name)
    def w(*a, **k):
        t = time.perf_counter(); r = f(*a, **k); acc[label] += time.perf_counter() - t; cnt[label] += 1; return r
    setattr(obj, name, w)
ctx = odo.ctx
for n in ("register_launch", "register_end", "map_update_staged", "map_update", "stage_insert", "project", "project_rows", "grid_sample_padded", "use_torch_stream", "distort"):
    if hasattr(ctx, n): timed(ctx, n, "ctx." + n)
for i, f in enumerate(filters): timed(f, "filter", f"filter{i}:{type(f).__name__}")
timed(odo, "_read_input", "odo._read_input"); timed(odo, "_rows_to_host", "odo._rows_to_host")
timed(odo.local_map, "update", "local_map.update"); timed(odo.local_map, "stage", "local_map.stage")
def one_pass(record):
    odo.init(); init.init()
    if record: acc.clear(); cnt.clear()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for f in range(frames):
        d = {"numpy_pc": scans[f]}
        init.next_frame(d)
        for flt in filters: flt.filter(d)
        t = time.perf_counter(); odo.process_next_frame(d); acc["process_next_frame (total)"] += time.perf_counter() - t
        if odo.relative_pose_key() in d: init.save_real_motion(d[odo.relative_pose_key()], d)
    torch.cuda.synchronize()
    return time.perf_counter() - t0
one_pass(False)
el = one_pass(True)
print(f"frame {el/ (frames-1)*1e6:.1f} us (with the timers)")
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]): print(f"  {k:40s} {v/(frames-1)*1e6:8.1f} us/frame  ({cnt[k]} calls)")
