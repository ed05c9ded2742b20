#!/bin/bash
# round 5, end-of-session measurements on HEAD: GPU tests, smoke, kernel trace + rocprof summary of the dominant kernel,
# PMC traffic, the driver-style bench line (20 steps) and the 60-step one, the two-rank self-launched run on the one GPU
# (gloo), C4, the kernel tables of the odometry_loop workload, the reference schedule's trace -> gpurun_out/$1
set -u
TAG=${1:-r5final}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
HEAD_SHA=$(cat tools/.head_sha 2>/dev/null || echo unknown)
timeout 900 python -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest.log; tail -3 $OUT/pytest.log | cut -c1-200
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
R=$PWD
CMD="python bench.py --steps 30 --warmup 5 --no-cpu-baseline --loop-steps 0 --plugin-steps 0 --odometry-loop 0 --throughput-leg 0"
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof -o t -- python $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline --loop-steps 0 --plugin-steps 0 --odometry-loop 0 --throughput-leg 0 > $R/$OUT/prof_bench.json 2> $R/$OUT/prof_bench.err )
python tools/rocprof_iterate_summary.py $OUT/prof $OUT/rocprof_iterate_kernel.json $HEAD_SHA "rocprofv3 --kernel-trace --stats -- $CMD" > /dev/null && cp $OUT/rocprof_iterate_kernel.json profiles/rocprof_iterate_kernel.json
cp $(ls $OUT/prof/*kernel_stats.csv | head -1) $OUT/kernel_stats.csv 2>/dev/null
python tools/dev/r5_timeline.py $(ls $OUT/prof/*kernel_trace.csv | head -1) k_pack_targets > $OUT/headline_timeline.txt 2>&1
# the reference's schedule (normals cleared and re-estimated behind every map update): same trace
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof_ref -o t -- python $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline --loop-steps 0 --plugin-steps 0 --odometry-loop 0 --throughput-leg 0 --no-profile --option carry_normals=0 > $R/$OUT/prof_ref_bench.json 2> $R/$OUT/prof_ref_bench.err )
cp $(ls $OUT/prof_ref/*kernel_stats.csv | head -1) $OUT/reference_schedule_kernel_stats.csv 2>/dev/null
rm -rf $OUT/prof $OUT/prof_ref
bash tools/pmc.sh k_iterate_compact > $OUT/pmc.log 2>&1; cp gpurun_out/pmc_k_iterate_compact.json $OUT/ 2>/dev/null; tail -c 300 $OUT/pmc.log; echo
[ -s gpurun_out/pmc_k_iterate_compact.json ] && cp gpurun_out/pmc_k_iterate_compact.json profiles/pmc_search_kernel.json
timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/bench_line_20.json 2> $OUT/bench_line_20.err; echo "bench20 rc=$?"
timeout 600 python bench.py > $OUT/bench_line.json 2> $OUT/bench_line.err; echo "bench60 rc=$?"
for f in $OUT/bench_line_20.json $OUT/bench_line.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d.get("roofline",{})
    print(f"{sys.argv[1]}: {d['value']:.1f} scans/s {d['ms_per_step']:.3f} ms (h60 {d.get('headline_60',{}).get('value',0):.0f}) ref-sched {d.get('reference_schedule',{}).get('value',0):.0f} iter-kernel {r.get('avg_launch_us',0):.2f} us (raw {r.get('avg_launch_us_raw_events',0):.2f}, overhead {r.get('event_overhead_us',0):.2f}, rocprof {r.get('rocprof_avg_launch_us',0) or 0:.2f}) frac {r.get('frac',0):.4f} plugin {d.get('plugin',{}).get('value',0):.0f} ({d.get('plugin',{}).get('frac_of_engine_headline',0):.2f}) odometry_loop {d.get('odometry_loop',{}).get('ms_per_frame',0):.3f} ms loop {d.get('loop',{}).get('value',0):.0f} throughput {d.get('throughput',{}).get('value',0):.0f} cpu {d.get('cpu_baseline',{}).get('value',0):.3f}")
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
python tools/dev/r5_timeline.py $(ls $OUT/odo/*kernel_trace.csv | head -1) k_dedupe_clear > $OUT/odometry_loop_timeline.txt 2>&1
rm -rf $OUT/odo
python - $OUT/odometry_loop_kernel_stats.csv <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print(f"odometry_loop: total kernel time {tot/72e3:.1f} us per frame; rocprim/at rows:", [r["Name"][:40] for r in rows if "rocprim" in r["Name"] or "at::" in r["Name"]])
PY
timeout 100 python bench.py --steps 14 --warmup 6 --no-cpu-baseline --loop-steps 0 --no-profile --plugin-steps 0 --odometry-loop 0 --throughput-leg 0 --option search_stats=2 > $OUT/stamps.json 2> $OUT/stamps.err; grep -c "icp phases" $OUT/stamps.err
timeout 100 python bench.py --steps 14 --warmup 6 --no-cpu-baseline --loop-steps 0 --no-profile --plugin-steps 0 --odometry-loop 0 --throughput-leg 0 --option search_stats=2 --option resident_tail=3 > $OUT/stamps_tail.json 2> $OUT/stamps_tail.err; grep -c "icp phases" $OUT/stamps_tail.err
[ -x tools/dev/t/stale_poll.bin ] && timeout 60 tools/dev/t/stale_poll.bin > $OUT/stale_poll.txt 2>&1; tail -3 $OUT/stale_poll.txt
