#!/bin/bash
# round 5 session 1: GPU tests + carry_normals A/B + default bench line
set -u
OUT=gpurun_out/r5a; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -15 $OUT/pytest.log
bash tools/gpu_quick.sh r5a "carry1|" "carry0|--option carry_normals=0" "carry1b|" "carry0b|--option carry_normals=0"
timeout 600 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc=$?"; tail -c 600 $OUT/bench_default.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r5a/bench_default.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step','ms_per_step_spread')})
for k in ('headline_60','reference_schedule','plugin','odometry_loop','throughput','loop'):
    v=d.get(k)
    if v: print(k, {kk:v[kk] for kk in v if kk in ('value','ms_per_step','ms_per_frame','ms_per_step_spread','error','max_translation_deviation_from_reference_run_m','steps_per_sequence')})
print(d.get('roofline',{}).get('avg_launch_us'), d.get('roofline',{}).get('frac'))
PY
