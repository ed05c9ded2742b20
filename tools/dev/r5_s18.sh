#!/bin/bash
# dev: kernel timeline of the published-configuration frame + in-kernel phase stamps of its iteration launches
set -u
TAG=${1:-r5r}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
R=$PWD
cd /tmp
timeout 200 rocprofv3 --kernel-trace --output-format csv -d $R/$OUT/odo -o t -- python $R/bench.py --leg odometry_loop --no-cpu-baseline > $R/$OUT/odo.json 2> $R/$OUT/odo.err
cd $R
f=$(ls $OUT/odo/*kernel_trace.csv 2>/dev/null | head -1)
python tools/dev/r5_timeline.py $f k_dedupe_clear | tee $OUT/timeline.txt
rm -rf $OUT/odo
timeout 200 python bench.py --leg odometry_loop --no-cpu-baseline --option search_stats=2 > $OUT/st.json 2> $OUT/st.err
grep -E "icp phases\]|icp stats|icp lead\]" $OUT/st.err | tail -40
cat $OUT/odo.json | head -c 600
