#!/bin/bash
set -u
OUT=gpurun_out/${1:-blocks}; mkdir -p $OUT
export TMPDIR=/tmp
timeout 200 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --loop-steps 0 --no-profile --option knn_select=1 --option search_stats=3 > $OUT/stats.json 2> $OUT/stats.err
grep "icp normals" $OUT/stats.err | tail -1 | cut -c1-200
cp /tmp/icp_normal_blocks.csv $OUT/ 2>/dev/null; wc -l $OUT/icp_normal_blocks.csv
