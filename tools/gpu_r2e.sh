#!/bin/bash
# round-2 session e: full GPU test suite, then A/B of the kNN-by-selection normals ("knn_select") and of more hardware
# queues for the throughput mode, then a kernel trace of the selected build -> gpurun_out/r2e
set -u
OUT=gpurun_out/r2e; mkdir -p $OUT
export TMPDIR=/tmp
git rev-parse --short HEAD > gpurun_out/.head_sha 2>/dev/null || true
timeout 900 python -m pytest tests -m gpu -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest.log
tail -15 $OUT/pytest.log
run() {  # name, env, args
  name=$1; shift; envs=$1; shift
  env $envs BENCH_PROF_MASK=5 timeout 300 python bench.py --steps 60 --warmup 5 --no-cpu-baseline --loop-steps 0 "$@" > $OUT/ab_$name.json 2> $OUT/ab_$name.err
  python - "$OUT/ab_$name.json" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d.get("roofline",{})
    print(f"{sys.argv[1]:40s} {d['value']:8.1f} scans/s  {d['ms_per_step']:.3f} ms (median {d['ms_per_step_spread']['median']:.3f})  iter-kernel {r.get('avg_launch_us',0):.1f} us  normals {d.get('normals_ms_per_step',0)*1e3:.1f} us  err {d['max_pose_error_vs_ground_truth_m']:.4f}")
except Exception as e: print(sys.argv[1],"FAILED",e)
PY
}
run base A=1
run sel A=1 --option knn_select=1
run base2 A=1
run sel2 A=1 --option knn_select=1
run s4_base A=1 --sequences-per-gpu 4
run s4_sel A=1 --sequences-per-gpu 4 --option knn_select=1
run s6_q8 GPU_MAX_HW_QUEUES=8 --sequences-per-gpu 6
run s6_q8_sel GPU_MAX_HW_QUEUES=8 --sequences-per-gpu 6 --option knn_select=1
run s8_q8_sel GPU_MAX_HW_QUEUES=8 --sequences-per-gpu 8 --option knn_select=1
bash tools/gpu_trace.sh r2e/trace_sel --option knn_select=1
bash tools/gpu_trace.sh r2e/trace_base
