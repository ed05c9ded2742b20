#!/bin/bash
set -u
OUT=gpurun_out/r5i; mkdir -p $OUT
export TMPDIR=/tmp
for v in "c1|1|" "c0|0|" "c1b|1|" "c0b|0|" "c0_narrow|0|--option narrow_from=-1" "c1_nowide|1|--option wide_until=0"; do
  name=${v%%|*}; rest=${v#*|}; comp=${rest%%|*}; args=${rest#*|}
  BENCH_ODO_COMPACT=$comp timeout 200 python bench.py --leg odometry_loop $args > $OUT/odo_$name.json 2> $OUT/odo_$name.err
  python - $OUT/odo_$name.json $name <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])['odometry_loop']
    print(sys.argv[2], 'ms/frame %.3f'%d['ms_per_frame'], 'full window %.3f'%d['ms_per_frame_full_window'], d['ms_per_frame_spread'], 'dev', d.get('max_translation_deviation_from_reference_run_m'), d.get('frames_with_other_iteration_count'))
except Exception as e: print(sys.argv[2],'FAILED',e)
PY
done
