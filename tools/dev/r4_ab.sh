#!/bin/bash
# round 3 dev: headline with the defaults and each option set, interleaved twice, no tests / traces (cheap A/B)
set -u
TAG=${1:-r4x}; shift; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
run() {
  local extra=""; for o in ${2//,/ }; do extra="$extra --option $o"; done
  timeout 120 python bench.py --steps 84 --warmup 6 --no-cpu-baseline --loop-steps 0 --no-profile --plugin-steps 0 --odometry-loop 0 $extra > $OUT/b_$1.json 2> $OUT/b_$1.err
  python - $OUT/b_$1.json "$1 $2" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); s=d['ms_per_step_spread']
    print(f"{sys.argv[2]:48s} {d['value']:8.1f} scans/s mean {d['ms_per_step']:.3f} median {s['median']:.3f} p90 {s['p90']:.3f} err {d['max_pose_error_vs_ground_truth_m']:.4f}")
except Exception as e: print(sys.argv[2], "FAILED", e)
PY
}
for rep in 1 2; do
  run base$rep ""
  i=0; for v in "$@"; do i=$((i+1)); run v${i}_$rep "$v"; done
done
