#!/bin/bash
set -u
TAG=${1:-r5t}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_pipeline.py -m gpu -x -q > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
for r in 1 2; do
timeout 200 python bench.py --leg odometry_loop --no-cpu-baseline > $OUT/odo_$r.json 2> $OUT/odo_$r.err
python - $OUT/odo_$r.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))["odometry_loop"]
print(f"odo ms/frame {d['ms_per_frame']:.4f} full-window {d['ms_per_frame_full_window']:.4f}")
print(d["ms_by_frame"]); print(d["iterations_by_frame"])
PY
done
