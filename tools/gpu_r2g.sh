#!/bin/bash
# quick check of a kNN-normal build: its tests, phase stamps, A/B -> gpurun_out/$1
set -u
OUT=gpurun_out/${1:-r2g}; mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "knn or schedule or normals or tiny" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
for v in 0 1; do
  timeout 200 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --loop-steps 0 --no-profile --option knn_select=$v --option search_stats=1 > $OUT/stats_$v.json 2> $OUT/stats_$v.err
  echo "== knn_select=$v"; grep "icp normals" $OUT/stats_$v.err | tail -2
done
for v in 0 1 0 1; do
  BENCH_PROF_MASK=5 timeout 300 python bench.py --steps 60 --warmup 5 --no-cpu-baseline --loop-steps 0 --option knn_select=$v > $OUT/ab_$v.json 2> $OUT/ab_$v.err
  python - $OUT/ab_$v.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d.get("roofline",{})
print(f"{sys.argv[1]:34s} {d['value']:8.1f} scans/s {d['ms_per_step']:.3f} ms (median {d['ms_per_step_spread']['median']:.3f}) normals {d.get('normals_ms_per_step',0)*1e3:.1f} us")
PY
done
for v in 0 1; do
  timeout 300 python bench.py --steps 60 --warmup 5 --no-cpu-baseline --loop-steps 0 --no-profile --sequences-per-gpu 4 --option knn_select=$v > $OUT/s4_$v.json 2> $OUT/s4_$v.err
  python -c "import json,sys; d=json.loads(open('$OUT/s4_$v.json').read().strip().splitlines()[-1]); print('S=4 knn_select=$v', round(d['value'],1))"
done
