#!/bin/bash
# round 6, session 1: batch tests + the small-item tests, then the batched bench leg
set -u
OUT=gpurun_out/r6_s1; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_batch.py -x -q > $OUT/pytest_batch.log 2>&1; echo "batch rc=$?"; tail -15 $OUT/pytest_batch.log
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "carried_normals" > $OUT/pytest_carry.log 2>&1; echo "carry rc=$?"; tail -5 $OUT/pytest_carry.log
timeout 600 python bench.py --leg throughput_batched --steps 100 --batched-leg 1,2,4,8 > $OUT/batched.json 2> $OUT/batched.err; tail -c 3000 $OUT/batched.json; tail -5 $OUT/batched.err
BENCH_BATCH_OPTIONS=wide_until=0 timeout 600 python bench.py --leg throughput_batched --steps 100 --batched-leg 4,8 > $OUT/batched_narrow.json 2> $OUT/batched_narrow.err; tail -c 2000 $OUT/batched_narrow.json
