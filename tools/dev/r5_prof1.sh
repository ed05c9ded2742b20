#!/bin/bash
# round 5: kernel tables of the published-configuration loop and of the headline loop (rocprofv3 --kernel-trace --stats)
set -u
TAG=${1:-r5prof1}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
R=$PWD
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/odo -o t -- python $R/bench.py --leg odometry_loop --no-cpu-baseline > $R/$OUT/odo.json 2> $R/$OUT/odo.err
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/head -o t -- python $R/bench.py --steps 28 --warmup 6 --no-cpu-baseline --loop-steps 0 --no-profile > $R/$OUT/head.json 2> $R/$OUT/head.err
cd $R
for w in odo head; do
  f=$(ls $OUT/$w/*kernel_stats.csv 2>/dev/null | head -1)
  [ -n "$f" ] && cp $f $OUT/${w}_kernel_stats.csv && echo "== $w" && python - $f <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time {tot/1e6:.2f} ms")
for r in rows[:26]:
    print(f'{r["Name"][:78]:78s} calls {int(r["Calls"]):6d} total {float(r["TotalDurationNs"])/1e3:10.1f} us avg {float(r["AverageNs"])/1e3:8.1f} us {float(r["Percentage"]):5.1f}%')
PY
done
rm -rf $OUT/odo $OUT/head
