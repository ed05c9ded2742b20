#!/bin/bash
set -u
TAG=${1:-r5y}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 300 python bench.py --steps 16 --warmup 10 --no-cpu-baseline --loop-steps 0 --plugin-steps 0 --odometry-loop 0 --throughput-leg 0 --no-profile --option search_stats=2 > $OUT/st2.json 2> $OUT/st2.err
grep -E "icp phases\] it +[0-3]:" $OUT/st2.err | tail -64 | cut -c1-420
timeout 300 python bench.py --steps 16 --warmup 10 --no-cpu-baseline --loop-steps 0 --plugin-steps 0 --odometry-loop 0 --throughput-leg 0 --no-profile --option search_stats=1 > $OUT/st1.json 2> $OUT/st1.err
grep -E "icp stats" $OUT/st1.err | tail -16 | cut -c1-420
