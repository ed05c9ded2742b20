#!/bin/bash
# round-2 GPU session B: full parity suite, per-iteration kernel trace, search statistics, A/B on both trajectories.
set -u
OUT=gpurun_out/r2b
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q -s > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest.log
tail -15 $OUT/pytest.log
B="timeout 300 python bench.py --steps 60 --warmup 5 --no-cpu-baseline --loop-steps 0"
run() { name=$1; shift; $B "$@" > $OUT/ab_$name.json 2> $OUT/ab_$name.err; }
run default
run rows_kernel --option compact_misses=0
run nocache --option nn_cache=0
run occ3 --option target_occupancy=3
run occ8 --option target_occupancy=8
run occ12 --option target_occupancy=12
run rings1 --max-rings 1
run s4_new --sequences-per-gpu 4
run s4_old --sequences-per-gpu 4 --option compact_misses=0
run s8_new --sequences-per-gpu 8
run loop --trajectory loop
run stats --option search_stats=1 --steps 4 --warmup 1
BENCH_PROF_MASK=5 run prof_normals_one --option normals_two_pass=0
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof -o r2b -- python $GRAFT_REPO_ROOT/bench.py --steps 30 --warmup 5 --no-cpu-baseline --loop-steps 0 --no-profile > $GRAFT_REPO_ROOT/$OUT/prof_bench.json 2> $GRAFT_REPO_ROOT/$OUT/prof_bench.err
cd $GRAFT_REPO_ROOT
ls $OUT/prof
python - <<'PY'
import csv,glob,collections
f=glob.glob("gpurun_out/r2b/prof/*kernel_trace.csv")
if f:
    rows=list(csv.DictReader(open(f[0])))
    rows.sort(key=lambda r:int(r["Start_Timestamp"]))
    it=[r for r in rows if "k_iterate" in r["Kernel_Name"]]
    # last 10 frames: 20 launches each
    d=[(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3 for r in it]
    frames=[d[i:i+20] for i in range(0,len(d),20)]
    import statistics
    print("per-iteration kernel us (median over last 20 frames):")
    last=frames[-20:]
    print([round(statistics.median(fr[i] for fr in last if len(fr)==20),1) for i in range(20)])
    # gaps: one steady frame
    st=[r for r in rows][-400:]
    tot=collections.Counter()
    for r in rows[len(rows)//2:]:
        tot[r["Kernel_Name"].split("(")[0][:60]]+= (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3
    n=len([r for r in rows[len(rows)//2:] if "k_sum_solve" in r["Kernel_Name"]])/20
    for k,v in tot.most_common(25): print(f"{k:62s} {v/n:8.1f} us/frame")
PY
for f in $OUT/ab_*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r=d.get("roofline",{})
    print(f"{sys.argv[1]:45s} {d['value']:8.1f} scans/s  {d['ms_per_step']:.3f} ms  iter-kernel {r.get('avg_launch_us',0):.1f} us  normals {d.get('normals_ms_per_step',0):.3f} ms  err {d['max_pose_error_vs_ground_truth_m']:.4f}")
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done
grep "icp stats" $OUT/ab_stats.err | tail -3
