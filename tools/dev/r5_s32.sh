#!/bin/bash
set -u
TAG=${1:-r5ag}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 300 python bench.py --steps 8 --warmup 10 --no-cpu-baseline --loop-steps 0 --plugin-steps 0 --odometry-loop 0 --throughput-leg 0 --no-profile --option search_stats=4 > $OUT/st4.json 2> $OUT/st4.err
grep -c "icp blocks" $OUT/st4.err
