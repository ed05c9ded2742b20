#!/bin/bash
# round 5 session 3: in-kernel stamps of the late iterations, tail vs per-iteration launches
set -u
OUT=gpurun_out/r5c; mkdir -p $OUT
export TMPDIR=/tmp
for v in tail notail; do
  extra=""; [ $v = notail ] && extra="--option resident_tail=0"
  timeout 200 python bench.py --steps 6 --warmup 4 --no-cpu-baseline --loop-steps 0 --no-profile --option search_stats=2 $extra > $OUT/st_$v.json 2> $OUT/st_$v.err
  echo "== $v"; grep -E "icp lead|icp phases" $OUT/st_$v.err | tail -44
done
