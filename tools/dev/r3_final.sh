#!/bin/bash
# round 3, end-of-session measurements on HEAD: GPU tests, smoke, PMC traffic, the driver-style bench line (20 steps) and
# the 60-step one, the two-rank run on the one GPU (gloo rendezvous), C4, kernel trace, PMC counters -> gpurun_out/$1
set -u
TAG=${1:-r3final}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest.log; tail -3 $OUT/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
# the PMC traffic of this build first: the bench lines below read it from profiles/pmc_search_kernel.json
bash tools/pmc.sh k_iterate_compact > $OUT/pmc.log 2>&1; cp gpurun_out/pmc_k_iterate_compact.json $OUT/ 2>/dev/null; tail -c 300 $OUT/pmc.log; echo
[ -s gpurun_out/pmc_k_iterate_compact.json ] && cp gpurun_out/pmc_k_iterate_compact.json profiles/pmc_search_kernel.json
timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/bench_line_20.json 2> $OUT/bench_line_20.err; echo "bench20 rc=$?"
timeout 600 python bench.py > $OUT/bench_line.json 2> $OUT/bench_line.err; echo "bench60 rc=$?"
for f in $OUT/bench_line_20.json $OUT/bench_line.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d.get("roofline",{})
    print(f"{sys.argv[1]}: {d['value']:.1f} scans/s {d['ms_per_step']:.3f} ms (h60 {d.get('headline_60',{}).get('value',0):.0f}) iter-kernel {r.get('avg_launch_us',0):.1f} us frac {r.get('frac',0):.4f} plugin {d.get('plugin',{}).get('value',0):.0f} ({d.get('plugin',{}).get('frac_of_engine_headline',0):.2f}) odometry_loop {d.get('odometry_loop',{}).get('ms_per_frame',0):.3f} ms loop {d.get('loop',{}).get('value',0):.0f} throughput {d.get('throughput',{}).get('value',0):.0f} cpu {d.get('cpu_baseline',{}).get('value',0):.3f}")
except Exception as e: print(sys.argv[1],"FAILED",e)
PY
done
export BENCH_DIST_BACKEND=gloo
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29531 bench.py --gpus 2 --steps 20 --warmup 5 > $OUT/bench_2ranks_gloo.json 2> $OUT/bench_2ranks_gloo.err; echo "2-rank rc=$?"
unset BENCH_DIST_BACKEND
python - $OUT/bench_2ranks_gloo.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("replicas x2", round(d["value"],1)); print("sharded", json.dumps(d.get("sharded"))[:500]); print("c4", json.dumps(d.get("c4"))[:400])
except Exception as e: print("FAILED", e)
PY
timeout 600 python bench.py --workload c4 --steps 6 --warmup 2 > $OUT/bench_c4.json 2> $OUT/bench_c4.err; echo "c4 rc=$?"; tail -c 700 $OUT/bench_c4.json; echo
bash tools/gpu_trace.sh $TAG/trace | tail -20
bash tools/pmc_kernel.sh k_iterate_compact > $OUT/pmc_sq_iterate.txt 2>&1; cat $OUT/pmc_sq_iterate.txt
bash tools/pmc_by_iter.sh $TAG > /dev/null 2>&1; head -30 $OUT/pmc_by_iter.txt | cut -c1-240
