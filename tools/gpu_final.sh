#!/bin/bash
# end-of-session measurements on HEAD: GPU test suite, bench line, throughput mode, kernel trace, PMC -> gpurun_out/$1
set -u
TAG=${1:-final}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest.log; tail -3 $OUT/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
timeout 600 python bench.py > $OUT/bench_line.json 2> $OUT/bench_line.err; tail -c 400 $OUT/bench_line.json; echo
timeout 300 python bench.py --steps 20 --warmup 5 > $OUT/bench_line_20.json 2> $OUT/bench_line_20.err
for s in 2 4; do timeout 300 python bench.py --sequences-per-gpu $s --no-cpu-baseline --loop-steps 0 > $OUT/bench_s$s.json 2> $OUT/bench_s$s.err; done
for f in $OUT/bench_*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d.get("roofline",{})
    print(f"{sys.argv[1]:42s} {d['value']:8.1f} scans/s {d['ms_per_step']:.3f} ms iter-kernel {r.get('avg_launch_us',0):.1f} us frac {r.get('frac',0):.4f} loop {d.get('loop',{}).get('value',0):.0f} cpu {d.get('cpu_baseline',{}).get('value',0):.3f}")
except Exception as e: print(sys.argv[1],"FAILED",e)
PY
done
bash tools/gpu_trace.sh $TAG/trace
bash tools/pmc.sh k_iterate_compact > $OUT/pmc.log 2>&1; cp gpurun_out/pmc_k_iterate_compact.json $OUT/ 2>/dev/null; tail -c 300 $OUT/pmc.log; echo
bash tools/pmc_kernel.sh k_iterate_compact > $OUT/pmc_sq_iterate.txt 2>&1; cat $OUT/pmc_sq_iterate.txt
bash tools/pmc_kernel.sh k_normals_all > $OUT/pmc_sq_normals.txt 2>&1
