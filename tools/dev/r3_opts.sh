#!/bin/bash
# round 3 dev: GPU tests, then the headline with the defaults and with each option set of the arguments, interleaved twice
# usage: tools/r3_opts.sh TAG "opt=val[,opt=val]" ...      -> gpurun_out/TAG
set -u
TAG=${1:-r3x}; shift; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest.log
run() {  # name, option string
  local extra=""; for o in ${2//,/ }; do extra="$extra --option $o"; done
  timeout 300 python bench.py --steps 84 --warmup 6 --no-cpu-baseline --loop-steps 0 --no-profile --plugin-steps 0 --odometry-loop 0 $extra > $OUT/b_$1.json 2> $OUT/b_$1.err
  python - $OUT/b_$1.json "$1 $2" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); s=d['ms_per_step_spread']
    print(f"{sys.argv[2]:40s} {d['value']:8.1f} scans/s mean {d['ms_per_step']:.3f} median {s['median']:.3f} p90 {s['p90']:.3f} err {d['max_pose_error_vs_ground_truth_m']:.4f}")
except Exception as e: print(sys.argv[2], "FAILED", e)
PY
}
for rep in 1 2; do
  run base$rep ""
  i=0; for v in "$@"; do i=$((i+1)); run v${i}_$rep "$v"; done
done
bash tools/gpu_trace.sh $TAG/trace | tail -22
timeout 120 python bench.py --steps 3 --warmup 3 --no-cpu-baseline --loop-steps 0 --no-profile --plugin-steps 0 --odometry-loop 0 --option search_stats=2 > /dev/null 2> $OUT/phases.err
echo "== phases (last frame)"; grep "icp phases\|icp lead" $OUT/phases.err | tail -34 | cut -c1-260
timeout 120 python bench.py --steps 3 --warmup 3 --no-cpu-baseline --loop-steps 0 --no-profile --plugin-steps 0 --odometry-loop 0 --option search_stats=1 > /dev/null 2> $OUT/phases_stats.err
grep "icp deciles" $OUT/phases_stats.err | tail -16 | head -8 | cut -c1-400
grep "icp stats" $OUT/phases_stats.err | tail -2 | cut -c1-300
