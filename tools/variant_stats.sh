#!/bin/bash
# dev tool: run the bench with a library variant (variants/libicp_<name>.so) and search_stats = 1, print the stats lines
R=$PWD; TAG=${1:-r4s}; V=${2:-d}; shift; shift; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
L=$R/pylidar-slam_amd/pylidar_slam_amd/_lib
cp $L/libicp_mi355x.so /tmp/libicp_base.so; cp $L/variants/libicp_$V.so $L/libicp_mi355x.so
timeout 60 python bench.py --steps 6 --warmup 3 --no-cpu-baseline --loop-steps 0 --no-profile --plugin-steps 0 --odometry-loop 0 --option search_stats=1 "$@" > /dev/null 2> $OUT/stats_$V.err
grep "icp stats" $OUT/stats_$V.err | tail -6 | cut -c1-320; grep "late miss" $OUT/stats_$V.err | tail -60
cp /tmp/libicp_base.so $L/libicp_mi355x.so
