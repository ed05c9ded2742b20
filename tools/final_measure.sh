#!/bin/bash
# round-end measurement batch on one box: GPU tests, bench line (with cpu_baseline), kernel stats, HBM counters, scenarios
mkdir -p gpurun_out
timeout 200 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
timeout 300 python bench.py --steps 112 --warmup 5 > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err; tail -c 600 gpurun_out/bench_final.json; echo
timeout 200 ./tools/prof.sh ${TAG:-r01_u} > /dev/null; head -12 gpurun_out/${TAG:-r01_u}_stats.txt
timeout 260 ./tools/pmc.sh k_iterate_rows | tail -c 400; echo
timeout 200 python tools/scenarios.py 2>&1 | grep "^S[123]"
