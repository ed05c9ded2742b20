#!/bin/bash
# round 3 dev: option sweep on the headline loop.  usage: tools/r3_sweep.sh TAG "opt=val opt=val" "opt=val" ...
set -u
TAG=$1; shift; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
i=0
for combo in "$@"; do
  i=$((i+1)); extra=""
  for o in $combo; do [ "$o" != "-" ] && extra="$extra --option $o"; done
  timeout 300 python bench.py --steps 84 --warmup 6 --no-cpu-baseline --loop-steps 0 --no-profile $extra > $OUT/s_$i.json 2> $OUT/s_$i.err
  python - $OUT/s_$i.json "$combo" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); s=d['ms_per_step_spread']
    print(f"{sys.argv[2]:40s} {d['value']:8.1f} scans/s mean {d['ms_per_step']:.3f} median {s['median']:.3f} p90 {s['p90']:.3f} max {s['max']:.3f} err {d['max_pose_error_vs_ground_truth_m']:.4f}")
except Exception as e: print(sys.argv[2], "FAILED", e)
PY
done
