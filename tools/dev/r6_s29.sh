#!/bin/bash
# two ranks on the one GPU (gloo dry run), base build against the in-tree one, alternating
set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r6_s29; mkdir -p $OUT; exec > >(tee $OUT/stdout.txt) 2>&1
LIB=$R/pylidar-slam_amd/pylidar_slam_amd/_lib/libicp_mi355x.so
cp $LIB /tmp/new_lib.so
cd $R
for r in 1 2 3 4 5; do for which in base new; do
[ $which = base ] && cp $R/tools/ab/base_libicp_mi355x.so $LIB || cp /tmp/new_lib.so $LIB
BENCH_DIST_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 20 --warmup 5 --multi-gpu-legs 1 > $OUT/two_$which.json 2> $OUT/two_$which.err
python - $OUT/two_$which.json $which <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]).read().strip().splitlines() if l.startswith("{")][-1])
    s=d.get("sharded",{})
    print(sys.argv[2], "replicas x2", round(d["value"]), "library", round(s["library"]["value"]), "collective", round(s["collective"]["value"]), "c4", round(d.get("c4",{}).get("value",0)))
except Exception as e: print(sys.argv[2], "FAILED", e)
PY
done; done
cp /tmp/new_lib.so $LIB
