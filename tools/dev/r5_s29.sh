#!/bin/bash
set -u
TAG=${1:-r5ad}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
i=0
for v in "" "lead_after_dense=0" "wide_until=2" "wide_until=4" "narrow_from=2" ""; do i=$((i+1))
  extra=""; for o in ${v//,/ }; do extra="$extra --option $o"; done
  timeout 200 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --loop-steps 0 --no-profile --plugin-steps 0 --odometry-loop 0 --throughput-leg 0 $extra > $OUT/h_$i.json 2> $OUT/h_$i.err
  python - "$v" $OUT/h_$i.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
print(f"[{sys.argv[1]:24s}] {d['value']:.0f} scans/s {d['ms_per_step']:.4f} ms spread {d['ms_per_step_spread']}")
PY
done
