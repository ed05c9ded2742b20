#!/bin/bash
set -u
TAG=${1:-r5w}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
for r in 1 2; do
timeout 200 python bench.py --leg odometry_loop --no-cpu-baseline > $OUT/odo_$r.json 2> $OUT/odo_$r.err
python - $OUT/odo_$r.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))["odometry_loop"]
print(f"odo ms/frame {d['ms_per_frame']:.4f} full-window {d['ms_per_frame_full_window']:.4f} dev {d.get('max_translation_deviation_from_reference_run_m')}")
print(d["ms_by_frame"])
PY
done
timeout 300 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --loop-steps 0 --plugin-steps 20 --odometry-loop 0 --throughput-leg 0 > $OUT/head.json 2> $OUT/head.err
python - $OUT/head.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("headline", d["value"], d["ms_per_step"], d.get("frame_ms_spread") or d.get("ms_per_step_spread"), "plugin", d.get("plugin",{}).get("value"))
PY
