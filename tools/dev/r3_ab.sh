#!/bin/bash
# round 3 dev: GPU tests, then the headline with an option off / on (A/B/A) -> gpurun_out/$1 ; usage: tools/r3_ab.sh TAG OPTION
set -u
TAG=${1:-r3c}; OPT=${2:-lead_solve}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest.log
for v in off on off2 on2; do
  extra="--option $OPT=0"; [ ${v:0:2} = on ] && extra="--option $OPT=1"
  timeout 300 python bench.py --steps 84 --warmup 6 --no-cpu-baseline --loop-steps 0 --no-profile $extra > $OUT/b_$v.json 2> $OUT/b_$v.err
  python - $OUT/b_$v.json $v <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); s=d['ms_per_step_spread']
    print(f"{sys.argv[2]:8s} {d['value']:8.1f} scans/s mean {d['ms_per_step']:.3f} median {s['median']:.3f} p90 {s['p90']:.3f} err {d['max_pose_error_vs_ground_truth_m']:.4f}")
except Exception as e: print(sys.argv[2], "FAILED", e)
PY
done
bash tools/gpu_trace.sh $TAG/trace | tail -22
# in-kernel phase stamps of three frames (dev option search_stats = 2), lead on / off
for v in 1 0; do
  timeout 120 python bench.py --steps 3 --warmup 3 --no-cpu-baseline --loop-steps 0 --no-profile --option $OPT=$v --option search_stats=2 > /dev/null 2> $OUT/phases_$v.err
  echo "== phases with $OPT=$v (last frame)"; grep "icp phases" $OUT/phases_$v.err | tail -20 | cut -c1-260
done
# path counters per workgroup (search_stats = 1): what the slow workgroups have that the others do not
timeout 120 python bench.py --steps 3 --warmup 3 --no-cpu-baseline --loop-steps 0 --no-profile --option $OPT=1 --option search_stats=1 > /dev/null 2> $OUT/phases_stats.err
grep "icp deciles" $OUT/phases_stats.err | tail -16 | head -8 | cut -c1-400
grep "icp stats" $OUT/phases_stats.err | tail -2 | cut -c1-300
