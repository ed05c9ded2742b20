#!/bin/bash
set -u
TAG=${1:-r5am}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
for i in 1 2 3; do
for v in "" "--no-profile"; do
  timeout 200 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --loop-steps 0 --plugin-steps 0 --odometry-loop 0 --throughput-leg 0 $v > $OUT/h.json 2> $OUT/h.err
  python - "$v" $OUT/h.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[2]).read().strip().splitlines()[-1]); r=d.get("roofline",{})
print(f"[{sys.argv[1]:14s}] {d['value']:.0f} scans/s {d['ms_per_step']:.4f} ms median {d['ms_per_step_spread']['median']:.4f} | avg_launch {r.get('avg_launch_us')} raw {r.get('avg_launch_us_raw_events')} overhead {r.get('event_overhead_us')} pair-on-spin {r.get('event_pair_on_20us_spin_kernel_us')} frac {r.get('frac')}")
PY
done; done
timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --loop-steps 0 --plugin-steps 0 --odometry-loop 0 --throughput-leg 0 > $OUT/h20.json 2> $OUT/h20.err
python - $OUT/h20.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d.get("roofline",{})
print(f"[20 steps] {d['value']:.0f} scans/s h60 {d.get('headline_60',{}).get('value')} avg_launch {r.get('avg_launch_us')} raw {r.get('avg_launch_us_raw_events')} long {r.get('avg_launch_us_long')} frac {r.get('frac')}")
PY
