import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "pylidar-slam_amd")]
import numpy as np, torch
from pylidar_slam_amd.engine import IcpContext
from pylidar_slam_amd.synthetic import SceneConfig, make_fixed_map, make_sequence
cfg = SceneConfig(height=32, width=1024)
scans, poses = make_sequence(cfg, 9)
model = make_fixed_map(cfg, scans[:4], poses[:4], ref_frame=3, num_points=30_000)
for carry in (0, 1):
    for tail in (0, 3):
        ctx = IcpContext(height=32, width=1024, max_num_alignments=10, threshold_delta_pose=0.0, scheme="geman_mcclure", sigma=0.3)
        ctx.set_option("carry_normals", carry); ctx.set_option("resident_tail", tail)
        ctx.map_set(model)
        init = None; out = []
        for f in (4, 5, 6, 7):
            ctx.register_launch(scans[f], init)
            ctx.map_update(None, None)
            r = ctx.register_end()
            out.append((r.normals_computed, r.iterations))
            init = r.pose
        print("carry", carry, "tail", tail, out, "fallbacks", ctx.handoff_fallbacks())
        ctx.close()
