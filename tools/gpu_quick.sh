#!/bin/bash
# dev: a few bench variants -> gpurun_out/$1 ; args after the first: "name|bench args"
set -u
OUT=gpurun_out/$1; shift; mkdir -p $OUT
export TMPDIR=/tmp
for spec in "$@"; do
  name=${spec%%|*}; args=${spec#*|}
  timeout 300 python bench.py --steps 84 --warmup 6 --no-cpu-baseline --loop-steps 0 --no-profile $args > $OUT/$name.json 2> $OUT/$name.err
  python - $OUT/$name.json $name <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); s=d['ms_per_step_spread']
    print(f"{sys.argv[2]:12s} {d['value']:8.1f} scans/s mean {d['ms_per_step']:.3f} median {s['median']:.3f} p90 {s['p90']:.3f} err {d['max_pose_error_vs_ground_truth_m']:.4f}")
except Exception as e: print(sys.argv[2],'FAILED',e)
PY
done
