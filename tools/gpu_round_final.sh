#!/bin/bash
# End-of-round measurements on HEAD -> gpurun_out/$1 (copy what is to be judged into profiles/): GPU tests, smoke, the bench
# lines (default and driver-style), kernel traces + rocprof summaries (headline, reference schedule, published configuration,
# batched throughput), PMC traffic, phase stamps, the two-rank dry run on the one GPU (gloo), C4.
set -u
TAG=${1:-final}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
HEAD_SHA=$(cat tools/.head_sha 2>/dev/null || echo unknown)
R=$PWD
timeout 1800 python -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest.log; tail -3 $OUT/pytest.log | cut -c1-200
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
LEGS_OFF="--no-cpu-baseline --loop-steps 0 --plugin-steps 0 --odometry-loop 0 --throughput-leg 0 --batched-leg="
CMD="python bench.py --steps 30 --warmup 5 $LEGS_OFF"
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof -o t -- python $R/bench.py --steps 30 --warmup 5 $LEGS_OFF > $R/$OUT/prof_bench.json 2> $R/$OUT/prof_bench.err )
python tools/rocprof_iterate_summary.py $OUT/prof $OUT/rocprof_iterate_kernel.json $HEAD_SHA "rocprofv3 --kernel-trace --stats -- $CMD" > /dev/null
cp $(ls $OUT/prof/*kernel_stats.csv | head -1) $OUT/kernel_stats.csv 2>/dev/null
python tools/frame_timeline.py $(ls $OUT/prof/*kernel_trace.csv | head -1) k_pack_targets > $OUT/headline_timeline.txt 2>&1
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof_ref -o t -- python $R/bench.py --steps 30 --warmup 5 $LEGS_OFF --no-profile --option carry_normals=0 > $R/$OUT/prof_ref_bench.json 2> $R/$OUT/prof_ref_bench.err )
cp $(ls $OUT/prof_ref/*kernel_stats.csv | head -1) $OUT/reference_schedule_kernel_stats.csv 2>/dev/null
rm -rf $OUT/prof $OUT/prof_ref
bash tools/pmc.sh k_iterate_compact > $OUT/pmc.log 2>&1; cp gpurun_out/pmc_k_iterate_compact.json $OUT/pmc_search_kernel.json 2>/dev/null; tail -c 300 $OUT/pmc.log; echo
# the batched throughput leg: kernel table + timeline + the dominant kernel's average per B
for B in 8 16; do bash tools/batch_trace.sh $TAG/batch$B $B > /dev/null 2>&1; head -14 $OUT/batch$B/kernel_stats.txt; done
python - $OUT <<'PY'
import json,sys,os
out=sys.argv[1]; rec={"kernel":"k_iterate_batch","head":open("tools/.head_sha").read().strip() if os.path.exists("tools/.head_sha") else "unknown",
  "command":"rocprofv3 --kernel-trace -- python bench.py --leg throughput_batched --batched-leg B --steps 30 --warmup 5 (tools/batch_trace.sh)","by_B":{}}
for B in (8,16):
    try: rec["by_B"][str(B)]=json.load(open(f"{out}/batch{B}/iterate_batch.json"))
    except Exception as e: pass
json.dump(rec,open(out+"/rocprof_iterate_batch.json","w"),indent=1)
PY
bash tools/batch_pmc.sh $TAG/batch8_pmc 8 > /dev/null 2>&1; head -24 $OUT/batch8_pmc/pmc.txt | cut -c1-200
# the bench lines (the committed rocprof / PMC summaries they quote are those of THIS session: copied first)
cp $OUT/rocprof_iterate_kernel.json profiles/rocprof_iterate_kernel.json 2>/dev/null
[ -s $OUT/pmc_search_kernel.json ] && cp $OUT/pmc_search_kernel.json profiles/pmc_search_kernel.json
cp $OUT/rocprof_iterate_batch.json profiles/rocprof_iterate_batch.json 2>/dev/null
T0=$(date +%s); timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/bench_line_20.json 2> $OUT/bench_line_20.err; echo "bench20 rc=$? $(( $(date +%s) - T0 )) s"
T0=$(date +%s); timeout 900 python bench.py > $OUT/bench_line.json 2> $OUT/bench_line.err; echo "bench60 rc=$? $(( $(date +%s) - T0 )) s"
for f in $OUT/bench_line_20.json $OUT/bench_line.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d.get("roofline",{}); tb=d.get("throughput_batched",{}); c=d.get("cpu_baseline",{})
    print(f"{sys.argv[1]}: {d['value']:.1f} scans/s {d['ms_per_step']:.3f} ms (h60 {d.get('headline_60',{}).get('value',0):.0f}) ref-sched {d.get('reference_schedule',{}).get('value',0):.0f} iter-kernel {r.get('avg_launch_us',0):.2f} us (raw {r.get('avg_launch_us_raw_events',0):.2f}, rocprof {r.get('rocprof_avg_launch_us',0) or 0:.2f}) frac {r.get('frac',0):.4f} plugin {d.get('plugin',{}).get('value',0):.0f} odometry_loop {d.get('odometry_loop',{}).get('ms_per_frame',0):.3f} ms loop {d.get('loop',{}).get('value',0):.0f} throughput {d.get('throughput',{}).get('value',0):.0f} batched {tb.get('value',0):.0f} {dict((k,round(v['value'])) for k,v in tb.get('by_B',{}).items())} frac {tb.get('whole_path_frac_of_hbm_peak',0):.4f} cpu {c.get('value',0):.3f} dev-from-oracle {c.get('max_pose_deviation_from_oracle_m')}")
except Exception as e: print(sys.argv[1],"FAILED",e)
PY
done
BENCH_DIST_BACKEND=gloo timeout 900 python bench.py --gpus 2 --steps 20 --warmup 5 > $OUT/bench_2ranks_gloo.json 2> $OUT/bench_2ranks_gloo.err; echo "2-rank rc=$?"
python - $OUT/bench_2ranks_gloo.json <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]).read().strip().splitlines() if l.startswith("{")][-1])
    print("replicas x2", round(d["value"],1)); print("sharded", json.dumps(d.get("sharded"))[:500]); print("c4", json.dumps(d.get("c4"))[:400])
except Exception as e: print("FAILED", e)
PY
timeout 600 python bench.py --workload c4 --steps 6 --warmup 2 > $OUT/bench_c4.json 2> $OUT/bench_c4.err; echo "c4 rc=$?"; tail -c 400 $OUT/bench_c4.json; echo
( cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/odo -o t -- python $R/bench.py --leg odometry_loop --no-cpu-baseline > $R/$OUT/odo.json 2> $R/$OUT/odo.err )
cp $(ls $OUT/odo/*kernel_stats.csv | head -1) $OUT/odometry_loop_kernel_stats.csv 2>/dev/null
python tools/frame_timeline.py $(ls $OUT/odo/*kernel_trace.csv | head -1) k_dedupe_clear > $OUT/odometry_loop_timeline.txt 2>&1
rm -rf $OUT/odo
timeout 100 python bench.py --steps 14 --warmup 6 $LEGS_OFF --no-profile --option search_stats=2 > $OUT/stamps.json 2> $OUT/stamps.err; grep "icp phases\|icp lead" $OUT/stamps.err | tail -60 > $OUT/phase_stamps.txt; wc -l $OUT/phase_stamps.txt
