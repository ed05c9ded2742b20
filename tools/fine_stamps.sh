#!/bin/bash
# dev tool: library variants with fine stamps inside the 4-lane search (variants/libicp_f<it>.so) -> "[icp fine]" lines
R=$PWD; TAG=${1:-r3f}; shift; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
L=$R/pylidar-slam_amd/pylidar_slam_amd/_lib
cp $L/libicp_mi355x.so /tmp/libicp_base.so
for v in "$@"; do
  cp $L/variants/libicp_$v.so $L/libicp_mi355x.so
  timeout 120 python bench.py --steps 4 --warmup 3 --no-cpu-baseline --loop-steps 0 --no-profile --plugin-steps 0 --odometry-loop 0 --option search_stats=2 > /dev/null 2> $OUT/fine_$v.err
  echo "== $v"; grep "icp fine" $OUT/fine_$v.err | tail -5; grep "icp phases\] it  [0-3]" $OUT/fine_$v.err | tail -4 | cut -c1-200
done
cp /tmp/libicp_base.so $L/libicp_mi355x.so
