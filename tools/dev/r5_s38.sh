#!/bin/bash
set -u
TAG=${1:-r5an}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; tail -2 $OUT/pytest.log
for v in "--pipeline 1" "--pipeline 2" "--pipeline 1" "--pipeline 2"; do
  timeout 200 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --loop-steps 0 --no-profile --plugin-steps 0 --odometry-loop 0 --throughput-leg 0 $v > $OUT/h.json 2> $OUT/h.err
  python - "$v" $OUT/h.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
print(f"[{sys.argv[1]:14s}] {d['value']:.0f} scans/s {d['ms_per_step']:.4f} ms median {d['ms_per_step_spread']['median']:.4f}")
PY
done
