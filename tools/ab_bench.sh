#!/bin/bash
# A/B of schedule knobs on one box: one bench line per variant to gpurun_out/ab_*.json.  usage: ab_bench.sh name ENV=.. [name ENV=..]...
mkdir -p gpurun_out
run() { name=$1; shift; env "$@" timeout 300 python bench.py --steps ${STEPS:-112} --warmup 5 --no-cpu-baseline ${EXTRA} > gpurun_out/ab_$name.json 2> gpurun_out/ab_$name.err; echo "$name: $(python -c "import json,sys; d=json.load(open('gpurun_out/ab_$name.json')); print(round(d['value'],1),'scans/s', round(d['ms_per_step'],4),'ms', 'iter_us', round(d['roofline']['avg_launch_us'],2), 'err', round(d['max_pose_error_vs_ground_truth_m'],5), 'normals_ms', round(d.get('normals_ms_per_step',0),4))" 2>&1 | tail -1)"; }
while [ $# -ge 2 ]; do run "$1" "$2"; shift 2; done
