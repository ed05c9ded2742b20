#!/usr/bin/env python
"""rocprofv3 --kernel-trace CSV of a bench.py run -> profiles/rocprof_iterate_kernel.json: the average launch duration
of the dominant kernel (k_iterate_compact, launch-weighted over its shapes), by shape and by iteration index within a
frame, the duration rocprofv3 sees for the 20 us spin kernel of the event calibration (k_event_floor) — the figures bench.py prints next to its live
HIP-event timing as `roofline.rocprof_*`.

usage: tools/rocprof_iterate_summary.py <dir with *kernel_trace.csv> <out.json> [commit] [command]"""
import collections
import csv
import glob
import json
import statistics
import sys


def main():
    src, out = sys.argv[1], sys.argv[2]
    head = sys.argv[3] if len(sys.argv) > 3 else "unknown"
    command = sys.argv[4] if len(sys.argv) > 4 else ""
    files = glob.glob(src + "/**/*kernel_trace.csv", recursive=True)
    rows = list(csv.DictReader(open(files[0])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    dur = lambda r: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    it = [r for r in rows if "k_iterate_compact" in r["Kernel_Name"]]
    by_shape = collections.defaultdict(list)
    for r in it:
        name = r["Kernel_Name"]
        shape = name[name.index("<"):name.index(">") + 1] if "<" in name else name
        by_shape[shape].append(dur(r))
    d = [dur(r) for r in it]
    iters = 20
    frames = [d[i:i + iters] for i in range(0, len(d) - len(d) % iters, iters)]
    per_iter = [statistics.mean(fr[i] for fr in frames) for i in range(iters)] if frames else []
    # the calibration kernel of icp_profile_event_floor: the bracketed launches spin for 20 us of the device clock
    null = [dur(r) for r in rows if "k_event_floor" in r["Kernel_Name"] and dur(r) >= 15.0]
    rec = {"kernel": "k_iterate_compact", "head": head, "command": command, "launches": len(d),
           "avg_launch_us": statistics.mean(d) if d else None,
           "by_shape_us": {k: {"launches": len(v), "avg": statistics.mean(v), "min": min(v), "max": max(v)}
                           for k, v in by_shape.items()},
           "by_iteration_us": per_iter,
           "spin_kernel_us": statistics.median(null) if null else None,
           "source": "rocprofv3 --kernel-trace (End_Timestamp - Start_Timestamp per dispatch)"}
    if rec["spin_kernel_us"] is None:
        rec.pop("spin_kernel_us")
    json.dump(rec, open(out, "w"), indent=1)
    print(json.dumps(rec))


if __name__ == "__main__":
    main()
