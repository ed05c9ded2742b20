#!/bin/bash
set -u
TAG=${1:-r3sec}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
R=$PWD
timeout 600 python tools/secondary.py > $OUT/secondary_wall.log 2>&1; tail -14 $OUT/secondary_wall.log | cut -c1-400
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof -o t -- python $R/tools/secondary.py > $R/$OUT/prof.log 2>&1)
python tools/secondary_summary.py $OUT/prof $OUT/r03_secondary
