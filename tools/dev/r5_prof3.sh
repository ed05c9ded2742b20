#!/bin/bash
set -u
TAG=${1:-r5prof3}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
R=$PWD
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/odo -o t -- python $R/bench.py --leg odometry_loop --no-cpu-baseline ${2:-} ${3:-} > $R/$OUT/odo.json 2> $R/$OUT/odo.err
cd $R
f=$(ls $OUT/odo/*kernel_stats.csv 2>/dev/null | head -1)
cp $f $OUT/odo_kernel_stats.csv
python - $f <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time {tot/1e6:.2f} ms = {tot/72e3:.1f} us per frame")
for r in rows[:12]:
    print(f'{r["Name"][:86]:86s} calls {int(r["Calls"]):5d} total {float(r["TotalDurationNs"])/1e3:9.1f} us avg {float(r["AverageNs"])/1e3:7.1f} max {float(r["MaxNs"])/1e3:7.1f} {float(r["Percentage"]):5.1f}%')
PY
rm -rf $OUT/odo
