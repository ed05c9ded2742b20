"""dev: the bench's throughput leg on its own, the per-sequence maximum pose errors printed (run-to-run comparison)"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "pylidar-slam_amd"), ROOT]
import numpy as np, torch
import bench
sys.argv = ["bench.py"] + sys.argv[1:]
args = bench.parse()
torch.cuda.set_device(0)
main_tr = bench.Tracker(args, 0, args.trajectory, args.warmup + args.steps, 0)
main_tr.run(20)
orig = bench.SequenceThread.run
errs = {}
res = bench.throughput_leg(args, 4, 0, main_tr)
print("throughput", round(res["value"]), "max_err_mm", round(res["max_pose_error_vs_ground_truth_m"] * 1e3, 4), "main max step ms", round(res["ms_per_step_spread_main_sequence"]["max"], 3), flush=True)
