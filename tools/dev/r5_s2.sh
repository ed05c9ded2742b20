#!/bin/bash
# round 5 session 2: the resident tail: parity tests first (bounded), then A/B
set -u
OUT=gpurun_out/r5b; mkdir -p $OUT
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "schedule_options_are_bit_identical or timed_out or carried" > $OUT/pytest1.log 2>&1; echo "pytest1 rc=$?"; tail -25 $OUT/pytest1.log
bash tools/gpu_quick.sh r5b "tail|" "notail|--option resident_tail=0" "tail7|--option resident_tail=7" "tailb|" "notailb|--option resident_tail=0"
timeout 900 python -m pytest tests -m gpu -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -15 $OUT/pytest.log
