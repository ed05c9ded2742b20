"""Dev measurement: the point-to-point registration loop (unfused: k_search_rows + k_reduce_p2p per iteration) from a
constant-velocity-quality guess, with and without the seeds of the previous iteration ("nn_cache" 2 / 1).  Wall clock."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pylidar-slam_amd"))
from pylidar_slam_amd.engine import IcpContext  # noqa: E402
from pylidar_slam_amd.synthetic import SceneConfig, make_fixed_map, make_sequence  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    cfg = SceneConfig(height=64, width=2048)
    scans, poses = make_sequence(cfg, 10)
    model = make_fixed_map(cfg, scans[:8], poses[:8], ref_frame=7, num_points=100_000)
    scan = torch.from_numpy(scans[8]).to(dev)
    gt = (np.linalg.inv(poses[7]) @ poses[8]).astype(np.float32)
    guess = gt.copy()
    guess[:3, 3] += np.array([0.02, -0.01, 0.005], np.float32)  # a constant-velocity guess is centimetres off
    for name, init in (("cv_guess", guess), ("identity_0.4m_off", None)):
        res = {}
        for seeds in (2, 1, 2, 1):
            ctx = IcpContext(height=64, width=2048, max_num_alignments=20, threshold_delta_pose=0.0)
            ctx.use_torch_stream()
            ctx.set_cost("point_to_point_gauss_newton")
            ctx.set_option("nn_cache", seeds)
            ctx.map_set(torch.from_numpy(model).to(dev))
            r = ctx.register(scan, init)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(10):
                r = ctx.register(scan, init)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) * 100.0
            err = float(np.linalg.norm(r.pose[:3, 3] - gt[:3, 3]))
            res.setdefault(seeds, []).append(ms)
            print(f"{name}: nn_cache={seeds}: {ms:.3f} ms per 20-iteration registration, |t - t_gt| = {err:.4f} m", flush=True)
            ctx.close()
        print(f"{name}: seeded {min(res[2]):.3f} ms vs unseeded {min(res[1]):.3f} ms", flush=True)


if __name__ == "__main__":
    main()
