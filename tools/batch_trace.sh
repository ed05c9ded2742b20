#!/bin/bash
# kernel trace of the batched throughput leg: tools/batch_trace.sh TAG B [bench args]  -> gpurun_out/TAG/{kernel_stats.txt,timeline.txt}
set -u
TAG=$1; B=$2; shift 2
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/prof -o t -- python $GRAFT_REPO_ROOT/bench.py --leg throughput_batched --batched-leg $B --steps 30 --warmup 5 "$@" > $OUT/bench.json 2> $OUT/bench.err
cd $GRAFT_REPO_ROOT
python - "$OUT" "$B" <<'PY'
import csv,glob,collections,sys
out,B=sys.argv[1],int(sys.argv[2])
f=glob.glob(out+"/prof/**/*kernel_trace.csv",recursive=True)
rows=list(csv.DictReader(open(f[0])))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
# the last timed window: the last 30 steps = the launches behind the 30th-from-last k_sum_solve_batch... simply the last third
half=rows[len(rows)*2//3:]
tot=collections.Counter(); cnt=collections.Counter()
for r in half:
    k=r["Kernel_Name"].split("(")[0][:70]; tot[k]+=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3; cnt[k]+=1
steps=max(1,len([r for r in half if "k_sum_solve_batch" in r["Kernel_Name"]]))
span=(int(half[-1]["End_Timestamp"])-int(half[0]["Start_Timestamp"]))/1e3/steps
busy=sum(tot.values())/steps
with open(out+"/kernel_stats.txt","w") as o:
    print(f"# B={B}: {steps} steps in the sampled part; wall {span:.1f} us/step, kernel time {busy:.1f} us/step, gaps {span-busy:.1f} us/step; per frame {span/B:.1f} us", file=o)
    for k,v in tot.most_common(24): print(f"{k:72s} {cnt[k]/steps:6.1f} launches/step {v/steps:9.1f} us/step  avg {v/cnt[k]:8.1f} us", file=o)
print(open(out+"/kernel_stats.txt").read())
import json
it=[r for r in half if "k_iterate" in r["Kernel_Name"]]
d=[(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3 for r in it]
frames=[d[i:i+20] for i in range(0,len(d)-len(d)%20,20)]
json.dump({"sequences_per_launch":B,"launches":len(d),"avg_launch_us":sum(d)/max(1,len(d)),
           "by_iteration_us":[sum(f[i] for f in frames)/len(frames) for i in range(20)] if frames else [],
           "wall_us_per_step":span,"kernel_us_per_step":busy}, open(out+"/iterate_batch.json","w"), indent=1)
# timeline of one step in the middle of the sampled part
idx=[i for i,r in enumerate(half) if "k_sum_solve_batch" in r["Kernel_Name"]]
if len(idx)>3:
    a,b=idx[len(idx)//2],idx[len(idx)//2+1]
    t0=int(half[a]["End_Timestamp"])
    with open(out+"/timeline.txt","w") as o:
        for r in half[a+1:b+1]:
            print(f"{(int(r['Start_Timestamp'])-t0)/1e3:9.1f} +{(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3:7.1f} us  grid {r.get('Grid_Size_X','?'):>8s} wg {r.get('Workgroup_Size_X','?'):>5s}  {r['Kernel_Name'].split('(')[0][:80]}", file=o)
PY
rm -rf $OUT/prof
