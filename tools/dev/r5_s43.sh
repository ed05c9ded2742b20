#!/bin/bash
set -u
TAG=${1:-r5as}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
for i in 1 2 3 4; do
  timeout 400 python bench.py --odometry-loop 0 --plugin-steps 0 --loop-steps 0 > $OUT/b$i.json 2> $OUT/b$i.err
  python - $OUT/b$i.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); t=d.get("throughput",{})
print("headline", round(d["value"]), "throughput", round(t.get("value",0)), "err by sequence (mm)", [round(x*1e3,4) for x in t.get("max_pose_error_by_sequence_m",[])], "main max step", round(t.get("ms_per_step_spread_main_sequence",{}).get("max",0),3))
PY
done
