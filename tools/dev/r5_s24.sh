#!/bin/bash
# dev: kernel timeline of the headline frames (anchor: k_pack_targets), outliers listed
set -u
TAG=${1:-r5x}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
R=$PWD
cd /tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/$OUT/h -o t -- python $R/bench.py --steps 60 --warmup 10 --no-cpu-baseline --loop-steps 0 --plugin-steps 0 --odometry-loop 0 --throughput-leg 0 --no-profile > $R/$OUT/head.json 2> $R/$OUT/head.err
cd $R
f=$(ls $OUT/h/*kernel_trace.csv 2>/dev/null | head -1)
python tools/dev/r5_timeline.py $f k_pack_targets | tee $OUT/timeline.txt | head -120
rm -rf $OUT/h
