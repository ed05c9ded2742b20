#!/bin/bash
# Generic A/B runner for a gpurun session: tools/gpu_ab.sh OUTDIR [--tests] "name|bench args" ...
set -u
OUT=gpurun_out/$1; shift
mkdir -p $OUT
export TMPDIR=/tmp
if [ "${1:-}" = "--tests" ]; then
  shift
  timeout 1200 python -m pytest tests -m gpu -q -s > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest.log
  tail -12 $OUT/pytest.log
fi
for spec in "$@"; do
  name=${spec%%|*}; args=${spec#*|}
  timeout 300 python bench.py --steps 60 --warmup 5 --no-cpu-baseline --loop-steps 0 $args > $OUT/ab_$name.json 2> $OUT/ab_$name.err
  python - "$OUT/ab_$name.json" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r=d.get("roofline",{})
    print(f"{sys.argv[1]:45s} {d['value']:8.1f} scans/s  {d['ms_per_step']:.3f} ms  iter-kernel {r.get('avg_launch_us',0):.1f} us  normals {d.get('normals_ms_per_step',0):.3f} ms  err {d['max_pose_error_vs_ground_truth_m']:.4f}")
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
  grep "icp stats\|icp phases" $OUT/ab_$name.err | tail -12
done
