#!/bin/bash
set -u
OUT=gpurun_out/r6_s2; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_batch.py -x -q > $OUT/pytest_batch.log 2>&1; echo "batch rc=$?"; tail -3 $OUT/pytest_batch.log
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "carried_normals" > $OUT/pytest_carry.log 2>&1; echo "carry rc=$?"; tail -3 $OUT/pytest_carry.log
bash tools/batch_trace.sh r6_s2/wide8 8
BENCH_BATCH_OPTIONS=wide_until=0 bash tools/batch_trace.sh r6_s2/narrow8 8
