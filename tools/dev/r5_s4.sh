#!/bin/bash
# round 5 session 4: tail v2 (register-resident queries, 16-byte row loads): bit-identity, stamps, A/B
set -u
OUT=gpurun_out/r5d; mkdir -p $OUT
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "schedule_options or timed_out" > $OUT/pytest1.log 2>&1; echo "pytest1 rc=$?"; tail -5 $OUT/pytest1.log
timeout 200 python bench.py --steps 6 --warmup 4 --no-cpu-baseline --loop-steps 0 --no-profile --option search_stats=2 > $OUT/st_tail.json 2> $OUT/st_tail.err
grep -E "icp lead|icp phases" $OUT/st_tail.err | tail -40 | cut -c1-200
bash tools/gpu_quick.sh r5d "tail|" "notail|--option resident_tail=0" "tailb|" "notailb|--option resident_tail=0"
