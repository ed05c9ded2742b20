#!/bin/bash
set -u
TAG=${1:-r5ae}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
i=0
for v in "" "refresh_margin=0.004" "refresh_margin=0.008" "refresh_at=3" "refresh_at=3,refresh_margin=0.004" "refresh_margin=0.001" ""; do i=$((i+1))
  extra=""; for o in ${v//,/ }; do extra="$extra --option $o"; done
  timeout 200 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --loop-steps 0 --no-profile --plugin-steps 0 --odometry-loop 0 --throughput-leg 0 $extra > $OUT/h_$i.json 2> $OUT/h_$i.err
  python - "$v" $OUT/h_$i.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
print(f"[{sys.argv[1]:36s}] {d['value']:.0f} scans/s {d['ms_per_step']:.4f} ms median {d['ms_per_step_spread']['median']:.4f} p90 {d['ms_per_step_spread']['p90']:.4f}")
PY
done
