#!/bin/bash
set -u
OUT=gpurun_out/r5k; mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_loop.py tests/test_gpu_pipeline.py -m gpu -q -x -k "grid_sample or loop or pipeline" > $OUT/pytest1.log 2>&1; echo "pytest1 rc=$?"; tail -3 $OUT/pytest1.log | cut -c1-250
python tools/dev/r5_hostprof.py 2>&1 | grep -E "pass total|ToDevice|GridSample|register_end|stage_cloud|update_staged|ctx.map_update |process_next"
for v in "pad1|1|" "pad0|0|" "pad1b|1|"; do
  name=${v%%|*}; rest=${v#*|}; pad=${rest%%|*}; args=${rest#*|}
  BENCH_ODO_PADDED=$pad timeout 200 python bench.py --leg odometry_loop $args > $OUT/odo_$name.json 2> $OUT/odo_$name.err
  python - $OUT/odo_$name.json $name <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])['odometry_loop']
    print(sys.argv[2], 'ms/frame %.3f'%d['ms_per_frame'], 'full window %.3f'%d['ms_per_frame_full_window'], d['ms_per_frame_spread'], 'dev', d.get('max_translation_deviation_from_reference_run_m'), d.get('frames_with_other_iteration_count'))
except Exception as e: print(sys.argv[2],'FAILED',e)
PY
done
bash tools/dev/r5_prof1.sh r5k_prof 2>&1 | sed -n '/== odo/,/== head/p' | head -30
