#!/bin/bash
# round 4 dev: the live roofline figure of a short and a default run (event overhead calibration)
set -u
TAG=${1:-r4roof}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
for st in 20 60; do
timeout 200 python bench.py --steps $st --warmup 5 --no-cpu-baseline --loop-steps 0 --plugin-steps 0 --odometry-loop 0 --throughput-leg 0 > $OUT/b_$st.json 2> $OUT/b_$st.err
python - $OUT/b_$st.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d['roofline']
print(d['value'], {k:(round(v,3) if isinstance(v,float) else v) for k,v in r.items() if k in ('avg_launch_us','avg_launch_us_raw_events','avg_launch_us_long','event_overhead_us','frac','frac_long','rocprof_avg_launch_us','rocprof_frac')})
PY
done
