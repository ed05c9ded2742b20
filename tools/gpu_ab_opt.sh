#!/bin/bash
# dev: tests touching the kNN normals, then A/B of one library option on the headline workload: tools/gpu_ab_opt.sh OUT name=value
set -u
OUT=gpurun_out/${1:-ab}; OPT=${2:-knn_cells=1}; mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_multi.py -m gpu -q -x -k "knn or schedule or normals or tiny or c4 or c2" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
for v in base opt base opt; do
  extra=""; [ $v = opt ] && extra="--option $OPT"
  BENCH_PROF_MASK=5 timeout 300 python bench.py --steps 60 --warmup 5 --no-cpu-baseline --loop-steps 0 $extra > $OUT/ab_$v.json 2> $OUT/ab_$v.err
  python - $OUT/ab_$v.json $v <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(f"{sys.argv[2]:6s} {d['value']:8.1f} scans/s {d['ms_per_step']:.3f} ms (median {d['ms_per_step_spread']['median']:.3f}) normals {d.get('normals_ms_per_step',0)*1e3:.1f} us  err {d['max_pose_error_vs_ground_truth_m']:.4f}")
PY
done
for v in base opt base opt; do
  extra=""; [ $v = opt ] && extra="--option $OPT"
  timeout 300 python bench.py --steps 60 --warmup 5 --no-cpu-baseline --loop-steps 0 --no-profile --sequences-per-gpu 4 $extra > $OUT/s4_$v.json 2> $OUT/s4_$v.err
  python -c "import json; d=json.loads(open('$OUT/s4_$v.json').read().strip().splitlines()[-1]); print('S=4 $v', round(d['value'],1))"
done
