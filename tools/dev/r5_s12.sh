#!/bin/bash
set -u
OUT=gpurun_out/r5l; mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_loop.py -m gpu -q -x -k "schedule_options or loop or lazy" > $OUT/pytest1.log 2>&1; echo "pytest1 rc=$?"; tail -12 $OUT/pytest1.log | cut -c1-250
for v in "lazy1|--option lazy_fused=1" "lazy0|--option lazy_fused=0" "lazy1b|" "lazy0b|--option lazy_fused=0"; do
  name=${v%%|*}; args=${v#*|}
  timeout 200 python bench.py --leg odometry_loop $args > $OUT/odo_$name.json 2> $OUT/odo_$name.err
  python - $OUT/odo_$name.json $name <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])['odometry_loop']
    print(sys.argv[2], 'ms/frame %.3f'%d['ms_per_frame'], 'full window %.3f'%d['ms_per_frame_full_window'], d['ms_per_frame_spread'], 'dev', d.get('max_translation_deviation_from_reference_run_m'), d.get('frames_with_other_iteration_count'))
except Exception as e: print(sys.argv[2],'FAILED',e)
PY
done
tail -3 $OUT/odo_lazy1.err
