#!/bin/bash
set -u
TAG=${1:-r5aa}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
R=$PWD
cd /tmp
timeout 200 rocprofv3 --kernel-trace --output-format csv -d $R/$OUT/odo -o t -- python $R/bench.py --leg odometry_loop --no-cpu-baseline --option lazy_fused=2 --option carry_normals=0 > $R/$OUT/odo.json 2> $R/$OUT/odo.err
cd $R
f=$(ls $OUT/odo/*kernel_trace.csv 2>/dev/null | head -1)
python tools/dev/r5_timeline.py $f k_dedupe_clear | tee $OUT/timeline.txt | head -40
rm -rf $OUT/odo
