#!/bin/bash
# A/B/C.. of several BUILDS inside one gpurun call: tools/gpu_ab_multi.sh TAG "base O2 Os" [rounds] [batched list]
# expects tools/ab/<name>_libicp_mi355x.so for every name (see tools/gpu_ab_lib.sh)
set -u
TAG=$1; NAMES=$2; ROUNDS=${3:-2}; BATCHED=${4:-16,48x4}
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; exec > >(tee $OUT/stdout.txt) 2>&1
LIB=$R/pylidar-slam_amd/pylidar_slam_amd/_lib/libicp_mi355x.so
cp $LIB /tmp/intree_lib.so
cd $R
for r in $(seq 1 $ROUNDS); do for n in $NAMES; do
  cp $R/tools/ab/${n}_libicp_mi355x.so $LIB
  timeout 300 python bench.py --steps 70 --no-cpu-baseline --loop-steps 0 > $OUT/head_$n.json 2> $OUT/head_$n.err
  timeout 900 python bench.py --leg throughput_batched --steps 100 --warmup 20 --batched-leg $BATCHED > $OUT/batched_$n.json 2> $OUT/batched_$n.err
  python - $OUT/head_$n.json $OUT/batched_$n.json "$n" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d.get("roofline",{})
    b=json.load(open(sys.argv[2]))["throughput_batched"]
    print(f"{sys.argv[3]:8s} headline {d['value']:7.1f} median {d['ms_per_step_spread']['median']:.4f} late {[round(v,1) for v in r.get('avg_launch_us_by_iteration_raw',[])[12:16]]}  batched", "  ".join(f"{B}: {round(v['value'])}" for B,v in b["by_B"].items()))
except Exception as e: print(sys.argv[3], "FAILED", e)
PY
done; done
cp /tmp/intree_lib.so $LIB
