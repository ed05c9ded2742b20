#!/usr/bin/env python
"""dev: per-iteration step lengths (|dx| translation / rotation) of a few bench frames."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench
sys.argv = [sys.argv[0], "--no-cpu-baseline"] + sys.argv[1:]
args = bench.parse()
tr = bench.Tracker(args, 0, "pingpong", 30, 0)
for k in range(12):
    f = tr.order[tr.cursor % len(tr.order)]
    res = tr.step(f, tr.last)
    tr._account(res, f, tr.prev); tr.prev = f; tr.cursor += 1
    dx = res.dx
    print(f"frame {f:2d}: |dt| mm", " ".join(f"{np.linalg.norm(d[:3])*1e3:7.3f}" for d in dx[:12]))
    print(f"          |dr| mrad", " ".join(f"{np.linalg.norm(d[3:])*1e3:7.3f}" for d in dx[:12]))
