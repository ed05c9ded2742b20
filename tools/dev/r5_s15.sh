#!/bin/bash
set -u
OUT=gpurun_out/r5o; mkdir -p $OUT
export TMPDIR=/tmp
timeout 60 tools/dev/t/solve_bench.bin
timeout 900 python -m pytest tests -m gpu -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log | cut -c1-250
bash tools/gpu_quick.sh r5o "a|" "b|" 
timeout 600 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r5o/bench_default.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step','ms_per_step_spread')})
for k in ('headline_60','reference_schedule','plugin','odometry_loop','throughput','loop'):
    v=d.get(k)
    if v: print(k, {kk:v[kk] for kk in v if kk in ('value','ms_per_step','ms_per_frame','ms_per_step_spread','error','max_translation_deviation_from_reference_run_m','steps_per_sequence','frames_with_other_iteration_count')})
print(d.get('roofline',{}).get('avg_launch_us'), d.get('roofline',{}).get('frac'), d.get('cpu_baseline'))
PY
