#!/bin/bash
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6_s26; mkdir -p $OUT; exec > >(tee $OUT/stdout.txt) 2>&1
cd $GRAFT_REPO_ROOT
for v in 0 0.8 1.0 1.2 1.5 2.0 0; do
timeout 300 python bench.py --steps 70 --no-cpu-baseline --loop-steps 0 --option two_stage=$v > $OUT/head.json 2> $OUT/head.err
python - $OUT/head.json "two_stage=$v" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d.get("roofline",{})
print(f"{sys.argv[2]:14s} {d['value']:7.1f} scans/s {d['ms_per_step']:.4f} ms", {k:round(v,3) for k,v in d["ms_per_step_spread"].items()}, "by iter", [round(v,1) for v in r.get("avg_launch_us_by_iteration_raw",[])[:6]])
PY
done
