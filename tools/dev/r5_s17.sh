#!/bin/bash
set -u
OUT=gpurun_out/r5q; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log | cut -c1-250
timeout 300 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --loop-steps 0 --no-profile --sequences-per-gpu 4 > $OUT/s4.json 2> $OUT/s4.err
timeout 300 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --loop-steps 0 --no-profile --sequences-per-gpu 4 --option lead_solve=1 > $OUT/s4lead.json 2> $OUT/s4lead.err
python - <<'PY'
import json
for n in ('s4','s4lead'):
    d=json.loads(open(f'gpurun_out/r5q/{n}.json').read().strip().splitlines()[-1]); print(n, d['value'], d['ms_per_step'])
PY
