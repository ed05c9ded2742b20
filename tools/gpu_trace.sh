#!/bin/bash
# per-iteration kernel durations + per-frame kernel totals from a rocprofv3 kernel trace: tools/gpu_trace.sh OUTDIR "bench args"
set -u
OUT=gpurun_out/$1; shift
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 30 --warmup 5 --no-cpu-baseline --loop-steps 0 --no-profile $@ > $GRAFT_REPO_ROOT/$OUT/prof_bench.json 2> $GRAFT_REPO_ROOT/$OUT/prof_bench.err
cd $GRAFT_REPO_ROOT
python - "$OUT" <<'PY'
import csv,glob,collections,statistics,sys
f=glob.glob(sys.argv[1]+"/prof/*kernel_trace.csv")
rows=list(csv.DictReader(open(f[0])))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
it=[r for r in rows if "k_iterate" in r["Kernel_Name"]]
d=[(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3 for r in it]
frames=[d[i:i+20] for i in range(0,len(d),20)]
last=[fr for fr in frames[-20:] if len(fr)==20]
print("per-iteration kernel us (median over last 20 frames):", [round(statistics.median(fr[i] for fr in last),1) for i in range(20)])
half=rows[len(rows)//2:]
tot=collections.Counter()
for r in half: tot[r["Kernel_Name"].split("(")[0][:60]]+=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3
n=max(1,len([r for r in half if "k_project(" in r["Kernel_Name"]]))  # frames: one projection each
span=(int(half[-1]["End_Timestamp"])-int(half[0]["Start_Timestamp"]))/1e3/n
busy=sum(tot.values())/n
print(f"frames {n}: wall {span:.1f} us/frame, kernel time {busy:.1f} us/frame, gaps {span-busy:.1f}")
for k,v in tot.most_common(14): print(f"{k:62s} {v/n:8.1f} us/frame")
PY
