#!/bin/bash
# round 4 dev: per-workgroup phase durations of the early launches (ICP_STATS_BLOCKS; search_stats 2: + what the ball search
# left and when; search_stats 1: + the path counters of each workgroup)
set -u
TAG=${1:-r4blk}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
for s in 2 1; do
ICP_STATS_BLOCKS=1 timeout 200 python bench.py --steps 12 --warmup 6 --no-cpu-baseline --loop-steps 0 --no-profile --plugin-steps 0 --odometry-loop 0 --throughput-leg 0 --option search_stats=$s > $OUT/blocks$s.json 2> $OUT/blocks$s.err
grep -c "icp blocks" $OUT/blocks$s.err
done
