#!/bin/bash
set -u
TAG=${1:-r5ah}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; tail -5 $OUT/pytest.log
i=0
for v in "" "chunk_rotation=0" "" "chunk_rotation=0" "xcd_sectors=0" "chunk_rotation=0,xcd_sectors=0"; do i=$((i+1))
  extra=""; for o in ${v//,/ }; do extra="$extra --option $o"; done
  timeout 200 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --loop-steps 20 --no-profile --plugin-steps 0 --odometry-loop 0 --throughput-leg 0 $extra > $OUT/h_$i.json 2> $OUT/h_$i.err
  python - "$v" $OUT/h_$i.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
print(f"[{sys.argv[1]:32s}] {d['value']:.0f} scans/s {d['ms_per_step']:.4f} ms spread {d['ms_per_step_spread']} loop {d.get('loop',{}).get('value')}")
PY
done
