#!/bin/bash
set -u
OUT=gpurun_out/r6_s10; mkdir -p $OUT
if [ "${SKIP_TESTS:-0}" != "1" ]; then
timeout 2400 python -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
fi
T0=$(date +%s); timeout 900 python bench.py > $OUT/bench_line.json 2> $OUT/bench_line.err; echo "default bench: $(( $(date +%s) - T0 )) s wall"
T0=$(date +%s); timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/bench_line_20.json 2> $OUT/bench_line_20.err; echo "20-step bench: $(( $(date +%s) - T0 )) s wall"
python - <<'PY'
import json
for f in ("bench_line","bench_line_20"):
    try:
        d=json.loads(open(f"gpurun_out/r6_s10/{f}.json").read().strip().splitlines()[-1])
    except Exception as e:
        print(f,"FAILED",e); continue
    r=d.get("roofline",{})
    print(f, round(d["value"],1),"scans/s", round(d["ms_per_step"],4),"ms; spread",{k:round(v,3) for k,v in d["ms_per_step_spread"].items()})
    print("  roofline frac",round(r.get("frac",0),4),"avg_launch_us",round(r.get("avg_launch_us",0),2),"rocprof",r.get("rocprof_avg_launch_us"))
    for k in ("headline_60","reference_schedule","plugin","loop","throughput"):
        if k in d: print("  ",k, round(d[k].get("value",0),1), d[k].get("error",""))
    o=d.get("odometry_loop",{}); print("   odometry_loop ms/frame", o.get("ms_per_frame"), o.get("error",""), "other-count frames", o.get("frames_with_other_iteration_count"))
    tb=d.get("throughput_batched",{}); print("   throughput_batched", tb.get("value"), tb.get("error",""), {k:round(v["value"]) for k,v in tb.get("by_B",{}).items()}, "frac", tb.get("whole_path_frac_of_hbm_peak"))
    c=d.get("cpu_baseline",{}); print("   cpu_baseline", c.get("kind"), c.get("value"), "dev from oracle", c.get("max_pose_deviation_from_oracle_m"), c.get("max_pose_deviation_from_oracle_rad"), c.get("reference_timing"))
PY
