#!/bin/bash
# A/B of two BUILDS of the library inside one gpurun call (boxes differ by up to 7 % from call to call):
#   tools/gpu_ab_lib.sh TAG [--tests "pytest -k expression"] [--batched "8,16,48x4"] [--rounds N] [--steps K] [--extra "bench args"]
# expects tools/ab/base_libicp_mi355x.so (the build to compare with: `git worktree add /tmp/base <commit>; make` and copy it
# there; *.so is git-ignored and still travels to the box) next to the in-tree build.  Alternates base / new, N rounds.
set -u
TAG=$1; shift
TESTS=""; BATCHED=""; ROUNDS=2; STEPS=70; EXTRA=""
while [ $# -gt 0 ]; do
  case $1 in
    --tests) TESTS=$2; shift 2;;
    --batched) BATCHED=$2; shift 2;;
    --rounds) ROUNDS=$2; shift 2;;
    --steps) STEPS=$2; shift 2;;
    --extra) EXTRA=$2; shift 2;;   # further bench.py arguments of the headline run, e.g. "--option carry_normals=0"
    *) echo "unknown $1"; exit 2;;
  esac
done
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
LIB=$R/pylidar-slam_amd/pylidar_slam_amd/_lib/libicp_mi355x.so
cp $LIB /tmp/new_lib.so
BASE=$R/tools/ab/base_libicp_mi355x.so
if [ -n "$TESTS" ]; then
  timeout 1500 python -m pytest tests -m gpu -x -q -k "$TESTS" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest.log
fi
for r in $(seq 1 $ROUNDS); do
  for which in base new; do
    if [ $which = base ]; then cp $BASE $LIB; else cp /tmp/new_lib.so $LIB; fi
    timeout 300 python bench.py --steps $STEPS --no-cpu-baseline --loop-steps 0 $EXTRA > $OUT/head_$which.json 2> $OUT/head_$which.err
    python - $OUT/head_$which.json "$which" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d.get("roofline",{})
print(f"{sys.argv[2]:5s} headline {d['value']:7.1f} scans/s {d['ms_per_step']:.4f} ms", {k:round(v,3) for k,v in d["ms_per_step_spread"].items()},
      "iter kernel", round(r.get("avg_launch_us",0),2), "us; by iter", [round(v,1) for v in r.get("avg_launch_us_by_iteration_raw",[])[:20]])
for k in ("reference_schedule","plugin","odometry_loop","throughput"):
    v=d.get(k)
    if isinstance(v,dict): print(f"      {k}:", v.get("value", v.get("ms_per_frame")), v.get("unit",""))
PY
    if [ -n "$BATCHED" ]; then
      timeout 900 python bench.py --leg throughput_batched --steps 100 --warmup 20 --batched-leg $BATCHED > $OUT/batched_$which.json 2> $OUT/batched_$which.err
      python - $OUT/batched_$which.json "$which" <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))["throughput_batched"]
print(f"{sys.argv[2]:5s} batched", "  ".join(f"B={B}: {round(r['value'])} {[round(v) for v in r['windows_scans_per_s']]}" for B,r in d["by_B"].items()))
PY
    fi
  done
done
cp /tmp/new_lib.so $LIB
