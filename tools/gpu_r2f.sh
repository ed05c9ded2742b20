#!/bin/bash
# diagnostics of the eager kNN-normal kernel, both builds: in-kernel phase stamps and SQ counters -> gpurun_out/r2f
set -u
OUT=gpurun_out/r2f; mkdir -p $OUT
export TMPDIR=/tmp
for v in 0 1; do
  timeout 200 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --loop-steps 0 --no-profile --option knn_select=$v --option search_stats=1 > $OUT/stats_$v.json 2> $OUT/stats_$v.err
  echo "== knn_select=$v"; grep "icp normals" $OUT/stats_$v.err | tail -3; grep "icp stats" $OUT/stats_$v.err | tail -1 | sed 's/.*knn:/knn:/'
  bash tools/pmc_kernel.sh k_normals_all --option knn_select=$v > $OUT/pmc_$v.txt 2>&1; cat $OUT/pmc_$v.txt
done
