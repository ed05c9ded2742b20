#!/bin/bash
# round 4 dev: the odometry_loop leg under option sets
set -u
TAG=${1:-r4odo}; shift; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
i=0
for v in "$@"; do i=$((i+1))
  extra=""; for o in ${v//,/ }; do [ "$o" != "none" ] && extra="$extra --option $o"; done
  timeout 150 python bench.py --leg odometry_loop --no-cpu-baseline $extra > $OUT/o_$i.json 2> $OUT/o_$i.err
  python - $OUT/o_$i.json "$v" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])['odometry_loop']
    print(f"odometry_loop [{sys.argv[2]:40s}] {d['ms_per_frame']:.3f} ms/frame median {d['ms_per_frame_spread']['median']:.3f} iters {d['iterations_per_frame']['mean']:.2f} dev {d.get('max_translation_deviation_from_reference_run_m',0):.2e}")
except Exception as e: print(sys.argv[2], "FAILED", e)
PY
done
