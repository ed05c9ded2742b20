"""dev: where the HOST spends a frame of the PLUGIN leg (numpy frames in, pose + host cloud out, fixed 100k map)."""
import sys, os, time, collections
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "pylidar-slam_amd"), ROOT]
import numpy as np, torch
import bench
from pylidar_slam_amd import odometry as our
args = bench.parse.__wrapped__() if hasattr(bench.parse, "__wrapped__") else None
sys.argv = ["bench.py"]
args = bench.parse()
scans, poses, model, order, start = bench.make_workload(0, "pingpong", 80)
cfg = our.MI355XICPConfig(max_num_alignments=20, threshold_delta_pose=0.0, data_key="numpy_pc", threshold_trans=float("inf"), threshold_rot=float("inf"),
                          local_map=dict(type="kdtree_local_map", local_map_size=20, num_neighbors_normals=10),
                          alignment=dict(mode="point_to_plane_gauss_newton", gauss_newton_config=dict(max_iters=1, scheme="geman_mcclure", sigma=0.3)))
odo = our.MI355XICPFrameToModel(cfg, projector=our.SphericalProjector(64, 2048), device=torch.device("cuda:0"))
init = our.ConstantVelocityInitialization()
odo.init(); init.init()
odo.process_next_frame({"numpy_pc": scans[start]})
odo.local_map.set_map_pointcloud(model)
T = collections.defaultdict(float)
def timed(name, fn, *a, **k):
    t0 = time.perf_counter(); r = fn(*a, **k); T[name] += time.perf_counter() - t0; return r
ctx = odo.ctx
for name in ("project", "register_launch", "register_end", "map_update", "use_torch_stream"):
    orig = getattr(ctx, name)
    setattr(ctx, name, (lambda o, n: (lambda *a, **k: timed("ctx." + n, o, *a, **k)))(orig, name))
for name in ("_upload", "_rows_to_host", "_read_input"):
    orig = getattr(odo, name)
    setattr(odo, name, (lambda o, n: (lambda *a, **k: timed("odo." + n, o, *a, **k)))(orig, name))
cur = 0
def run(k):
    global cur
    for _ in range(k):
        f = order[cur % len(order)]; d = {"numpy_pc": scans[f]}
        init.next_frame(d)
        timed("process_next_frame", odo.process_next_frame, d)
        init.save_real_motion(d[odo.relative_pose_key()], d); cur += 1
run(6); T.clear(); torch.cuda.synchronize(); t0 = time.perf_counter(); run(60); torch.cuda.synchronize()
print(f"plugin leg {1e3 * (time.perf_counter() - t0) / 60:.3f} ms per frame")
for k, v in sorted(T.items(), key=lambda kv: -kv[1]):
    print(f"  {k:28s} {v * 1e6 / 60:8.1f} us per frame")
