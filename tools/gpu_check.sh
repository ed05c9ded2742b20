#!/bin/bash
# dev: full GPU test suite + headline bench (with and without the event timing) + kernel trace -> gpurun_out/$1
set -u
OUT=gpurun_out/${1:-check}; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
for v in noprof prof noprof2; do
  extra="--no-profile"; [ $v = prof ] && extra=""
  timeout 300 python bench.py --steps 84 --warmup 6 --no-cpu-baseline --loop-steps 0 $extra > $OUT/b_$v.json 2> $OUT/b_$v.err
  python - $OUT/b_$v.json $v <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); s=d['ms_per_step_spread']
print(f"{sys.argv[2]:8s} {d['value']:8.1f} scans/s mean {d['ms_per_step']:.3f} median {s['median']:.3f} p90 {s['p90']:.3f} err {d['max_pose_error_vs_ground_truth_m']:.4f}")
PY
done
bash tools/gpu_trace.sh ${1:-check}/trace | tail -18
