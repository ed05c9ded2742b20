#!/bin/bash
# end-of-round measurements: full bench line, throughput mode, round-1 trajectory, kernel trace -> gpurun_out/<dir>
set -u
OUT=gpurun_out/${1:-final}; mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python bench.py > $OUT/bench_line.json 2> $OUT/bench_line.err; tail -c 600 $OUT/bench_line.json; echo
for s in 2 3 4; do timeout 300 python bench.py --sequences-per-gpu $s --no-cpu-baseline --loop-steps 0 > $OUT/bench_s$s.json 2> $OUT/bench_s$s.err; done
timeout 300 python bench.py --trajectory pingpong_r01 --no-cpu-baseline --loop-steps 0 > $OUT/bench_r01traj.json 2> $OUT/bench_r01traj.err
timeout 300 python bench.py --pipeline 2 --no-cpu-baseline --loop-steps 0 > $OUT/bench_pipe2.json 2> $OUT/bench_pipe2.err
for f in $OUT/bench_*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d.get("roofline",{})
    print(f"{sys.argv[1]:42s} {d['value']:8.1f} scans/s {d['ms_per_step']:.3f} ms iter-kernel {r.get('avg_launch_us',0):.1f} us frac {r.get('frac',0):.4f} loop {d.get('loop',{}).get('value',0):.0f} cpu {d.get('cpu_baseline',{}).get('value',0):.3f}")
except Exception as e: print(sys.argv[1],"FAILED",e)
PY
done
bash tools/gpu_trace.sh ${1:-final}
