import sys, threading, hashlib
sys.path[:0] = [__import__("os").path.join(__import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__)))), "pylidar-slam_amd"), __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))]
import numpy as np
import bench
def digest(w):
    scans, poses, model, order, start = w
    h = hashlib.sha1()
    for f in sorted(scans): h.update(scans[f].tobytes())
    h.update(model.tobytes())
    return h.hexdigest()[:12]
ref = {j: digest(bench.make_workload(100 + j, "pingpong", 70)) for j in (1, 2, 3)}
print("sequential:", ref)
for rep in range(6):
    out = {}
    def body(j): out[j] = digest(bench.make_workload(100 + j, "pingpong", 70))
    ts = [threading.Thread(target=body, args=(j,)) for j in (1, 2, 3)]
    [t.start() for t in ts]; [t.join() for t in ts]
    print("concurrent", rep, {j: (out[j], "SAME" if out[j] == ref[j] else "DIFFERENT") for j in out})
