#!/bin/bash
# dev: the two bench lines at HEAD (driver style and default)
set -u
TAG=${1:-r5ao}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/bench_line_20.json 2> $OUT/bench_line_20.err; echo "bench20 rc=$?"
timeout 600 python bench.py > $OUT/bench_line.json 2> $OUT/bench_line.err; echo "bench60 rc=$?"
for f in $OUT/bench_line_20.json $OUT/bench_line.json; do python - "$f" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d.get("roofline",{})
print(f"{sys.argv[1]}: {d['value']:.1f} scans/s {d['ms_per_step']:.3f} ms (h60 {d.get('headline_60',{}).get('value',0):.0f}) ref-sched {d.get('reference_schedule',{}).get('value',0):.0f} iter-kernel {r.get('avg_launch_us',0):.2f} us (raw {r.get('avg_launch_us_raw_events',0):.2f}, overhead {r.get('event_overhead_us',0):.2f}, rocprof {r.get('rocprof_avg_launch_us',0) or 0:.2f}) frac {r.get('frac',0):.4f} plugin {d.get('plugin',{}).get('value',0):.0f} ({d.get('plugin',{}).get('frac_of_engine_headline',0):.2f}) odometry_loop {d.get('odometry_loop',{}).get('ms_per_frame',0):.3f} ms loop {d.get('loop',{}).get('value',0):.0f} throughput {d.get('throughput',{}).get('value',0):.0f} cpu {d.get('cpu_baseline',{}).get('value',0):.3f}")
PY
done
