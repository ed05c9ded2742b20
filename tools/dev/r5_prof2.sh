#!/bin/bash
set -u
OUT=gpurun_out/r5prof2; mkdir -p $OUT
export TMPDIR=/tmp
R=$PWD
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/hp -o t -- python $R/tools/dev/r5_hostprof.py > $R/$OUT/hp.log 2>&1
cd $R
f=$(ls $OUT/hp/*kernel_stats.csv 2>/dev/null | head -1)
python - $f <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time {tot/1e6:.2f} ms")
for r in rows[:14]:
    print(f'{r["Name"][:78]:78s} calls {int(r["Calls"]):6d} total {float(r["TotalDurationNs"])/1e3:10.1f} us avg {float(r["AverageNs"])/1e3:8.1f} us max {float(r["MaxNs"])/1e3:8.1f} {float(r["Percentage"]):5.1f}%')
PY
tail -16 $OUT/hp.log
rm -rf $OUT/hp
