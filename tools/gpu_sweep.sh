#!/bin/bash
# dev: sweep of the schedule knobs on the headline workload -> gpurun_out/$1 (each line: value, ms/step, median)
set -u
OUT=gpurun_out/${1:-sweep}; mkdir -p $OUT
export TMPDIR=/tmp
run() { name=$1; shift
  timeout 200 python bench.py --steps 84 --warmup 6 --no-cpu-baseline --loop-steps 0 --no-profile "$@" > $OUT/$name.json 2> $OUT/$name.err
  python - "$OUT/$name.json" "$name" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); s=d['ms_per_step_spread']
    print(f"{sys.argv[2]:22s} {d['value']:8.1f} scans/s  mean {d['ms_per_step']:.3f}  median {s['median']:.3f}  p90 {s['p90']:.3f}  max {s['max']:.3f}  err {d['max_pose_error_vs_ground_truth_m']:.4f}")
except Exception as e: print(sys.argv[2],"FAILED",e)
PY
}
run base
run rings1 --max-rings 1
run rings3 --max-rings 3
run wm8 --option wave_misses=8
run wm48 --option wave_misses=48
run wm128 --option wave_misses=128
run nf4 --option narrow_from=4
run nf5 --option narrow_from=5
run nf8 --option narrow_from=8
run occ6 --option target_occupancy=6
run occ16 --option target_occupancy=16
run occ24 --option target_occupancy=24
run base_again
run pipe2 --pipeline 2
