#!/bin/bash
set -u
OUT=gpurun_out/r5m; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest.log | cut -c1-250
for v in "ov1|" "ov0|--option overlap_map_update=0" "ov1b|" "ov0b|--option overlap_map_update=0"; do
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
bash tools/gpu_quick.sh r5m "head_ov1|" "head_ov0|--option overlap_map_update=0" "head_ov1b|"
