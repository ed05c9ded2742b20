#!/bin/bash
# round 5 session 5: tail with the full search: headline A/B, odometry_loop A/B
set -u
OUT=gpurun_out/r5e; mkdir -p $OUT
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "schedule_options or timed_out" > $OUT/pytest1.log 2>&1; echo "pytest1 rc=$?"; tail -3 $OUT/pytest1.log
bash tools/gpu_quick.sh r5e "tail|" "notail|--option resident_tail=0" "tail7|--option resident_tail=7"
for v in "t0|--option resident_tail=0" "t3|--option resident_tail=3" "t1|--option resident_tail=1 --option wide_until=0" "t1w1|--option resident_tail=1 --option wide_until=1" "t0b|--option resident_tail=0"; do
  name=${v%%|*}; args=${v#*|}
  timeout 200 python bench.py --leg odometry_loop $args > $OUT/odo_$name.json 2> $OUT/odo_$name.err
  python - $OUT/odo_$name.json $name <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])['odometry_loop']
    print(sys.argv[2], 'ms/frame %.3f'%d['ms_per_frame'], 'full window %.3f'%d['ms_per_frame_full_window'], d['ms_per_frame_spread'], d['iterations_per_frame'], 'dev', d.get('max_translation_deviation_from_reference_run_m'))
except Exception as e: print(sys.argv[2],'FAILED',e)
PY
done
