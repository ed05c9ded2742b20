#!/bin/bash
set -u
TAG=${1:-r5at}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp BENCH_DEV_SKIP_CPU_TIMING=1
for v in "" "insert_by_cell=0"; do
for i in 1 2 3 4 5 6 7 8; do
  extra=""; for o in ${v//,/ }; do extra="$extra --option $o"; done
  timeout 200 python bench.py --steps 20 --warmup 5 --odometry-loop 0 --plugin-steps 0 --loop-steps 0 $extra > $OUT/b.json 2> $OUT/b.err
  python - "$v" $OUT/b.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[2]).read().strip().splitlines()[-1]); t=d.get("throughput",{})
e=[round(x*1e3,4) for x in t.get("max_pose_error_by_sequence_m",[])]
print(f"[{sys.argv[1]:18s}] headline {round(d['value'])} throughput {round(t.get('value',0))} err {e} {'<-- DIFFERENT' if e[1:]!=[0.9549,1.3476,1.1083] else ''}")
PY
done; done
