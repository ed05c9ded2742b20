#!/bin/bash
set -u
TAG=${1:-r5aq}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
for i in 1 2 3 4; do
  timeout 400 python bench.py --cpu-baseline-frames 2 > $OUT/b$i.json 2> $OUT/b$i.err || timeout 400 python bench.py > $OUT/b$i.json 2> $OUT/b$i.err
  python - $OUT/b$i.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("headline", round(d["value"]), "throughput", round(d.get("throughput",{}).get("value",0)), "plugin", round(d.get("plugin",{}).get("value",0)), "odo", round(d.get("odometry_loop",{}).get("ms_per_frame",0),4), "loop", round(d.get("loop",{}).get("value",0)))
PY
done
