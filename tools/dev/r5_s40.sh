#!/bin/bash
set -u
TAG=${1:-r5ap}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
for i in 1 2 3 4 5; do
  timeout 200 python bench.py --leg plugin --no-cpu-baseline > $OUT/p.json 2> $OUT/p.err
  python - $OUT/p.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
p=d.get("plugin",d)
print("plugin alone", round(p["value"]), round(p["ms_per_step"],4))
PY
done
for i in 1 2 3; do
  timeout 300 python bench.py --no-cpu-baseline --odometry-loop 0 --loop-steps 0 > $OUT/b.json 2> $OUT/b.err
  python - $OUT/b.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("in sequence: headline", round(d["value"]), "throughput", round(d.get("throughput",{}).get("value",0)), "plugin", round(d.get("plugin",{}).get("value",0)))
PY
done
