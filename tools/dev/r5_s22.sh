#!/bin/bash
set -u
TAG=${1:-r5v}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 200 python bench.py --leg odometry_loop --no-cpu-baseline --option search_stats=1 > $OUT/st1.json 2> $OUT/st1.err
grep -n "icp stats" $OUT/st1.err | sed -n 36,70p
timeout 200 python bench.py --leg odometry_loop --no-cpu-baseline --option search_stats=2 > $OUT/st2.json 2> $OUT/st2.err
grep -c "icp phases" $OUT/st2.err
