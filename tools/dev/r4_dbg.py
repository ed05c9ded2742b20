import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pylidar-slam_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
from pylidar_slam_amd.engine import IcpContext
from pylidar_slam_amd.synthetic import SceneConfig, make_fixed_map, make_sequence
cfg = SceneConfig(height=32, width=1024)
scans, poses = make_sequence(cfg, 6)
model = make_fixed_map(cfg, scans[:4], poses[:4], ref_frame=3, num_points=30_000)
for hoods in (1, 2):
    ctx = IcpContext(height=32, width=1024, max_num_alignments=4, threshold_delta_pose=0.0)
    ctx.set_option("hoods", hoods)
    ctx.map_set(model)
    t0 = time.time(); r = ctx.register(scans[4]); torch.cuda.synchronize(); print("hoods", hoods, "register ok", time.time() - t0, r.normals_computed, flush=True)
    _, n, _ = ctx.nearest_neighbor_search(scans[4][::5]); print(" normals sum", float(np.abs(n).sum()), flush=True)
    ctx.close()
