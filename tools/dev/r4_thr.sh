#!/bin/bash
# round 4 dev: throughput mode (S sequences on S streams of one process) for option sets
set -u
TAG=${1:-r4t}; shift; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
i=0
for v in "$@"; do i=$((i+1))
  S=${v%%:*}; opts=${v#*:}
  extra=""; for o in ${opts//,/ }; do [ -n "$o" ] && extra="$extra --option $o"; done
  timeout 300 python bench.py --steps 60 --warmup 6 --no-cpu-baseline --loop-steps 0 --no-profile --plugin-steps 0 --odometry-loop 0 --sequences-per-gpu $S $extra > $OUT/t_$i.json 2> $OUT/t_$i.err
  python - $OUT/t_$i.json "$v" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(f"{sys.argv[2]:40s} {d['value']:8.1f} scans/s  ms/step {d['ms_per_step']:.3f} err {d['max_pose_error_vs_ground_truth_m']:.4f}")
except Exception as e: print(sys.argv[2], "FAILED", e)
PY
done
