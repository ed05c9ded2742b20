#!/bin/bash
set -u
TAG=${1:-r5av}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp BENCH_DEV_SKIP_CPU_TIMING=1
run() { # label, args...
  local label="$1"; shift
  timeout 200 python bench.py --steps 20 --warmup 5 --odometry-loop 0 --plugin-steps 0 --loop-steps 0 --cell-size 0.5 "$@" > $OUT/b.json 2> $OUT/b.err
  python - "$label" $OUT/b.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[2]).read().strip().splitlines()[-1]); t=d.get("throughput",{})
e=[round(x*1e3,4) for x in t.get("max_pose_error_by_sequence_m",[])]
print(f"[{sys.argv[1]:22s}] thr {round(t.get('value',0))} err {e} {'<--' if e[1:]!=[0.9549,1.3476,1.1083] else ''}")
PY
}
for i in 1 2 3 4 5 6 7; do run "nn_cache=1" --option nn_cache=1; done
for i in 1 2 3 4 5 6 7; do run "frame_seed=0" --option frame_seed=0; done
for i in 1 2 3 4 5 6 7; do run "carry_normals=0" --option carry_normals=0; done
