#!/bin/bash
set -u
TAG=${1:-r5au}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp BENCH_DEV_SKIP_CPU_TIMING=1
run() { # label, args...
  local label="$1"; shift
  timeout 200 python bench.py --steps 20 --warmup 5 --odometry-loop 0 --plugin-steps 0 --loop-steps 0 "$@" > $OUT/b.json 2> $OUT/b.err
  python - "$label" $OUT/b.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[2]).read().strip().splitlines()[-1]); t=d.get("throughput",{})
e=[round(x*1e3,4) for x in t.get("max_pose_error_by_sequence_m",[])]
print(f"[{sys.argv[1]:22s}] headline {round(d['value'])} throughput {round(t.get('value',0))} err {e}")
PY
}
for i in 1 2 3 4 5 6 7 8 9 10; do run "cell-size 0.5" --cell-size 0.5; done
for i in 1 2 3 4 5 6; do run "nn_cache=0" --option nn_cache=0; done
