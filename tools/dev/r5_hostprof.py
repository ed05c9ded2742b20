"""dev: where the HOST spends a frame of the published-configuration loop (perf_counter around every stage)."""
import sys, os, time, collections
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "pylidar-slam_amd")]
import numpy as np, torch
from pylidar_slam_amd import odometry as our
from pylidar_slam_amd.synthetic import SceneConfig, make_sequence
dev = torch.device("cuda:0")
scans, gt = make_sequence(SceneConfig(height=64, width=2048), 36)
cfg = our.MI355XICPConfig(max_num_alignments=20, threshold_delta_pose=1e-4, data_key="input_data",
                          local_map=dict(type="kdtree_local_map", local_map_size=30, num_neighbors_normals=10),
                          alignment=dict(mode="point_to_plane_gauss_newton", gauss_newton_config=dict(max_iters=1, scheme="neighborhood", sigma=0.2)))
odo = our.MI355XICPFrameToModel(cfg, projector=our.SphericalProjector(64, 2048), device=dev)
filters = [our.ToDevice(our.ToDeviceConfig(device=str(dev), pinned_staging=os.environ.get("PIN", "1") == "1"), device=dev),
           our.Distortion(our.DistortionConfig(pointcloud_key="pc_device", timestamps_key="timestamps_device", output_key="distorted")),
           our.GridSample(our.GridSampleConfig(voxel_size=0.4, pointcloud_key="distorted", padded=os.environ.get("PAD", "1") == "1")),
           our.ToTensor(our.ToTensorConfig(device=str(dev), keys={"sample_points": "input_data"}, dtype="float32"), device=dev)]
init = our.ConstantVelocityInitialization()
T = collections.defaultdict(float)
def timed(name, fn, *a, **k):
    t0 = time.perf_counter(); r = fn(*a, **k); T[name] += time.perf_counter() - t0; return r
# wrap the inner calls of the plugin
ctx = odo.ctx
for name in ("project", "compact_targets", "register_launch", "register_end", "map_update", "map_stage_cloud", "map_update_staged", "use_torch_stream"):
    if hasattr(ctx, name):
        orig = getattr(ctx, name)
        setattr(ctx, name, (lambda o, n: (lambda *a, **k: timed("ctx." + n, o, *a, **k)))(orig, name))
for p in range(2):
    odo.init(); init.init(); T.clear()
    torch.cuda.synchronize(); t_all = time.perf_counter()
    for f in range(36):
        d = {"numpy_pc": scans[f]}
        timed("init.next_frame", init.next_frame, d)
        for flt in filters:
            timed(type(flt).__name__, flt.filter, d)
        timed("process_next_frame", odo.process_next_frame, d)
        if odo.relative_pose_key() in d:
            timed("save_real_motion", init.save_real_motion, d[odo.relative_pose_key()], d)
    torch.cuda.synchronize(); total = time.perf_counter() - t_all
print(f"PIN={os.environ.get('PIN','1')} PAD={os.environ.get('PAD','1')} pass total {total * 1e3 / 35:.3f} ms per frame")
for k, v in sorted(T.items(), key=lambda kv: -kv[1]):
    print(f"  {k:28s} {v * 1e6 / 36:8.1f} us per frame")
