import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "pylidar-slam_amd")]
import numpy as np, torch
from pylidar_slam_amd.engine import IcpContext
from pylidar_slam_amd.synthetic import SceneConfig, make_fixed_map, make_sequence
h, w, iters = 32, 1024, 10
cfg = SceneConfig(height=h, width=w)
scans, poses = make_sequence(cfg, 7)
model = make_fixed_map(cfg, scans[:4], poses[:4], ref_frame=3, num_points=30_000)
for first_tail in (1, 0):
    ctx = IcpContext(height=h, width=w, max_num_alignments=iters, threshold_delta_pose=0.0, scheme="geman_mcclure", sigma=0.3)
    ctx.set_option("resident_tail_max_blocks", 4096)
    ctx.set_option("search_stats", 2)
    ctx.map_set(model)
    if not first_tail:
        ctx.set_option("resident_tail", 0)
    init = None
    for f in (4, 5):
        if f == 5:
            ctx.set_option("resident_tail", 3)
        t0 = time.perf_counter()
        ctx.register_launch(scans[f], init)
        ctx.map_update(None, None)
        r = ctx.register_end()
        print("first_tail", first_tail, "frame", f, r.iterations, round((time.perf_counter() - t0) * 1e3, 2), ctx.handoff_fallbacks(), flush=True)
        sys.stderr.flush()
        init = r.pose
    ctx.close()
