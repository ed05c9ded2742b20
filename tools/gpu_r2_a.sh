#!/bin/bash
# round-2 GPU session A: parity suite, A/B of the new schedules, kernel trace.  Run from the repo root under gpurun.
set -u
mkdir -p gpurun_out/r2a
export TMPDIR=/tmp
OUT=gpurun_out/r2a
timeout 900 python -m pytest tests -m gpu -x -q -s > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest.log
tail -5 $OUT/pytest.log
B="timeout 300 python bench.py --steps 60 --warmup 5 --no-cpu-baseline --loop-steps 0"
$B > $OUT/ab_default.json 2> $OUT/ab_default.err
$B --option compact_misses=0 > $OUT/ab_rows_kernel.json 2> $OUT/ab_rows_kernel.err
$B --option iterate_dense=0 > $OUT/ab_sparse_build.json 2> $OUT/ab_sparse_build.err
$B --option frame_seed=0 > $OUT/ab_no_frame_seed.json 2> $OUT/ab_no_frame_seed.err
$B --option normals_two_pass=0 > $OUT/ab_one_pass_normals.json 2> $OUT/ab_one_pass_normals.err
$B --option compact_misses=0 --option normals_two_pass=0 --option frame_seed=0 --trajectory pingpong_r01 > $OUT/ab_r01_config.json 2> $OUT/ab_r01_config.err
$B --trajectory pingpong_r01 > $OUT/ab_r01_traj.json 2> $OUT/ab_r01_traj.err
BENCH_PROF_MASK=5 $B > $OUT/ab_prof_normals.json 2> $OUT/ab_prof_normals.err
$B --sequences-per-gpu 4 > $OUT/ab_s4.json 2> $OUT/ab_s4.err
timeout 400 python bench.py --steps 60 --warmup 5 > $OUT/bench_full.json 2> $OUT/bench_full.err
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/prof -o r2a -- python $GRAFT_REPO_ROOT/bench.py --steps 40 --warmup 5 --no-cpu-baseline --loop-steps 0 --no-profile > $GRAFT_REPO_ROOT/$OUT/prof_bench.json 2> $GRAFT_REPO_ROOT/$OUT/prof_bench.err
cd $GRAFT_REPO_ROOT
find $OUT/prof -name "*kernel_stats*" | head
for f in $OUT/ab_*.json $OUT/bench_full.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r=d.get("roofline",{})
    print(f"{sys.argv[1]:45s} {d['value']:8.1f} scans/s  {d['ms_per_step']:.3f} ms  iter-kernel {r.get('avg_launch_us',0):.1f} us  normals {d.get('normals_ms_per_step',0):.3f} ms  err {d['max_pose_error_vs_ground_truth_m']:.4f}")
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done
