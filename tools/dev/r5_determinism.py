"""dev: are the poses of S sequences driven by S host threads on S streams of one GPU the same from run to run?
(the bench's throughput arrangement; every sequence has its own context, scans and map)"""
import sys, os, threading, hashlib
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "pylidar-slam_amd"), ROOT]
import numpy as np, torch
import bench
sys.argv = ["bench.py"] + sys.argv[1:]
args = bench.parse()
S, FR = 4, int(os.environ.get('DET_FRAMES', '70'))
WITH_MAIN = os.environ.get('DET_MAIN', '1') == '1'
work = {j: bench.make_workload(100 + j, "pingpong", FR) for j in range(S)}

def one_run(opts):
    out = {}
    def body(j):
        torch.cuda.set_device(0)
        stream = torch.cuda.Stream()
        with torch.cuda.stream(stream):
            tr = bench.Tracker(args, 100 + j, "pingpong", FR, 0, workload=work[j])
            for k, v in opts.items():
                tr.ctx.set_option(k, v)
            poses = []
            for _ in range(FR):
                f = tr.order[tr.cursor % len(tr.order)]
                res = tr.step(f, tr.last)
                tr._account(res, f, tr.prev)
                poses.append(res.pose.copy()); tr.prev = f; tr.cursor += 1
            stream.synchronize()
            out[j] = (np.stack(poses), tr.max_err, tr.ctx.handoff_fallbacks())
            tr.close()
    ts = [threading.Thread(target=body, args=(j,)) for j in range(1 if WITH_MAIN else 0, S)]
    [t.start() for t in ts]
    if WITH_MAIN:  # sequence 0 on the main thread and the DEFAULT stream, like the bench's headline tracker
        tr = bench.Tracker(args, 100, "pingpong", FR, 0, workload=work[0])
        for k, v in opts.items():
            tr.ctx.set_option(k, v)
        poses = []
        for _ in range(FR):
            f = tr.order[tr.cursor % len(tr.order)]
            res = tr.step(f, tr.last)
            tr._account(res, f, tr.prev)
            poses.append(res.pose.copy()); tr.prev = f; tr.cursor += 1
        torch.cuda.synchronize()
        out[0] = (np.stack(poses), tr.max_err, tr.ctx.handoff_fallbacks())
        tr.close()
    [t.join() for t in ts]
    return out

for name, opts in (("throughput-leg options", {"lead_solve": 0, "wide_until": 0}),
                   ("+ frame_seed 0", {"lead_solve": 0, "wide_until": 0, "frame_seed": 0})):
    runs = [one_run(opts) for _ in range(int(os.environ.get('DET_RUNS', '6')))]
    ref = runs[0]
    diffs = []
    for r in runs[1:]:
        for j in range(S):
            if not np.array_equal(r[j][0], ref[j][0]):
                bad = np.flatnonzero(np.any(r[j][0].reshape(FR, -1) != ref[j][0].reshape(FR, -1), axis=1))
                diffs.append((j, int(bad[0]), float(np.abs(r[j][0] - ref[j][0]).max())))
    print(f"{name}: max_err {[round(ref[j][1] * 1e3, 4) for j in range(S)]} mm, fallbacks {[ref[j][2] for j in range(S)]}; "
          f"runs that differ from the first (sequence, first frame, max |dpose|): {diffs}", flush=True)
