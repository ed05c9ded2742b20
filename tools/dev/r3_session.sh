#!/bin/bash
# round 3: GPU tests, the bench line (with the plugin / odometry_loop legs), the multi-rank legs on the one GPU (gloo
# rendezvous, two ranks sharing it: exercises the code path of `sharded` / `c4`), C4 on one GPU, kernel trace
set -u
TAG=${1:-r3a}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest.log; tail -5 $OUT/pytest.log
timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/bench_line_20.json 2> $OUT/bench_line_20.err; echo "bench rc=$?"; tail -c 300 $OUT/bench_line_20.err
python - $OUT/bench_line_20.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("headline", round(d["value"],1), "ms", round(d["ms_per_step"],3), "h60", d.get("headline_60",{}).get("value"))
print("plugin", json.dumps(d.get("plugin"))[:600])
print("odometry_loop", json.dumps(d.get("odometry_loop"))[:900])
print("throughput", d.get("throughput",{}).get("value"), "cpu", d.get("cpu_baseline",{}).get("value"))
PY
export BENCH_DIST_BACKEND=gloo
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29521 bench.py --gpus 2 --steps 20 --warmup 5 > $OUT/bench_2ranks_gloo.json 2> $OUT/bench_2ranks_gloo.err; echo "2-rank rc=$?"; tail -c 300 $OUT/bench_2ranks_gloo.err
python - $OUT/bench_2ranks_gloo.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("replicas x2", round(d["value"],1)); print("sharded", json.dumps(d.get("sharded"))[:800]); print("c4", json.dumps(d.get("c4"))[:800])
except Exception as e: print("FAILED", e)
PY
unset BENCH_DIST_BACKEND
timeout 600 python bench.py --workload c4 --steps 6 --warmup 2 > $OUT/bench_c4.json 2> $OUT/bench_c4.err; echo "c4 rc=$?"; tail -c 600 $OUT/bench_c4.json; echo
bash tools/gpu_trace.sh $TAG/trace | tail -25
