#!/bin/bash
# round 3 dev: GPU tests, headline twice, the C4 workload and the 4-sequence throughput mode -> gpurun_out/$1
set -u
TAG=${1:-r4k}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
show() { python - "$1" "$2" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); s=d.get('ms_per_step_spread',{})
    print(f"{sys.argv[2]:28s} {d['value']:8.1f} {d['unit']} mean {d['ms_per_step']:.3f} median {s.get('median',0):.3f} p90 {s.get('p90',0):.3f} err {d.get('max_pose_error_vs_ground_truth_m',0):.4f}")
    for k in ('plugin','odometry_loop','throughput','loop'):
        if k in d: print('   ',k, {kk:(round(v,3) if isinstance(v,float) else v) for kk,v in d[k].items() if kk in ('value','ms_per_step','ms_per_frame','sequences_per_gpu')})
except Exception as e: print(sys.argv[2], "FAILED", e)
PY
}
for rep in 1 2; do
  timeout 300 python bench.py --steps 84 --warmup 6 --no-cpu-baseline --loop-steps 0 --no-profile --plugin-steps 0 --odometry-loop 0 > $OUT/b_base$rep.json 2> $OUT/b_base$rep.err; show $OUT/b_base$rep.json base$rep
done
timeout 300 python bench.py --workload c4 --no-cpu-baseline > $OUT/b_c4.json 2> $OUT/b_c4.err; show $OUT/b_c4.json c4
timeout 300 python bench.py --steps 60 --warmup 6 --no-cpu-baseline --loop-steps 0 --no-profile --plugin-steps 0 --odometry-loop 0 --sequences-per-gpu 4 --option lead_solve=0 > $OUT/b_s4.json 2> $OUT/b_s4.err; show $OUT/b_s4.json "4 sequences"
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/b_legs.json 2> $OUT/b_legs.err; show $OUT/b_legs.json "all legs, 20 steps"
