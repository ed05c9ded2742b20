#!/bin/bash
# round 3 dev: the plugin / odometry_loop legs alone + a kernel trace of each -> gpurun_out/$1
set -u
TAG=${1:-r3b}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
R=$PWD
for leg in plugin odometry_loop; do
  timeout 300 python bench.py --leg $leg > $OUT/leg_$leg.json 2> $OUT/leg_$leg.err; tail -c 1500 $OUT/leg_$leg.json; echo
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof_$leg -o t -- python $R/bench.py --leg $leg > $R/$OUT/prof_$leg.log 2>&1)
  python - $OUT/prof_$leg <<'PY'
import csv,glob,sys,collections
f=glob.glob(sys.argv[1]+"/*kernel_trace.csv")
rows=list(csv.DictReader(open(f[0]))); rows.sort(key=lambda r:int(r["Start_Timestamp"]))
half=rows[len(rows)//2:]
tot=collections.Counter(); cnt=collections.Counter()
for r in half:
    k=r["Kernel_Name"].split("(")[0][:70]; tot[k]+=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3; cnt[k]+=1
span=(int(half[-1]["End_Timestamp"])-int(half[0]["Start_Timestamp"]))/1e3
print(f"second half of the trace: span {span/1e3:.2f} ms, kernel time {sum(tot.values())/1e3:.2f} ms, {len(half)} launches")
for k,v in tot.most_common(22): print(f"{k:72s} {cnt[k]:6d} x {v/cnt[k]:8.1f} us = {v/1e3:8.2f} ms")
PY
done
