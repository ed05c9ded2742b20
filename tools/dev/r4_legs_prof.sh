#!/bin/bash
# round 4: kernel tables (rocprofv3 --kernel-trace --stats) of the two workloads outside the headline loop:
# the published configuration as a loop (bench.py --leg odometry_loop) and C4 on one GPU (bench.py --workload c4)
set -u
TAG=${1:-r4legs}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
R=$PWD
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/odo -o t -- python $R/bench.py --leg odometry_loop --no-cpu-baseline > $R/$OUT/odo.json 2> $R/$OUT/odo.err
timeout 280 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/c4 -o t -- python $R/bench.py --workload c4 --steps 6 --warmup 2 --no-cpu-baseline > $R/$OUT/c4.json 2> $R/$OUT/c4.err
cd $R
for w in odo c4; do
  f=$(ls $OUT/$w/*kernel_stats.csv 2>/dev/null | head -1)
  [ -n "$f" ] && cp $f $OUT/${w}_kernel_stats.csv && echo "== $w" && python - $f <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time {tot/1e6:.2f} ms")
for r in rows[:22]:
    print(f'{r["Name"][:70]:70s} calls {int(r["Calls"]):6d} total {float(r["TotalDurationNs"])/1e3:10.1f} us avg {float(r["AverageNs"])/1e3:8.1f} us {float(r["Percentage"]):5.1f}%')
PY
done
tail -c 600 $OUT/odo.json; echo; tail -c 400 $OUT/c4.json; echo
