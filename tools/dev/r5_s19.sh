#!/bin/bash
# dev: after the frame-path trims (wave-parallel state init, padding fills in the table clear, two-launch compaction,
# projection rows): tests, the odometry_loop leg under a few schedules, the headline
set -u
TAG=${1:-r5s}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
for v in "" "wide_until=1" "wide_until=0" "wide_until=2"; do
  extra=""; for o in ${v//,/ }; do extra="$extra --option $o"; done
  for r in 1 2; do
  timeout 200 python bench.py --leg odometry_loop --no-cpu-baseline $extra > $OUT/odo_${v}_$r.json 2> $OUT/odo_${v}_$r.err
  python - "$v" $OUT/odo_${v}_$r.json <<'PY'
import json,sys
d=json.load(open(sys.argv[2]))["odometry_loop"]
print(f"odo [{sys.argv[1]:14s}] ms/frame {d['ms_per_frame']:.4f} full-window {d['ms_per_frame_full_window']:.4f} median {d['ms_per_frame_spread']['median']:.4f} dev {d.get('max_translation_deviation_from_reference_run_m')} other {d.get('frames_with_other_iteration_count')}")
PY
  done
done
timeout 300 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --loop-steps 0 --plugin-steps 20 --odometry-loop 0 --throughput-leg 0 > $OUT/head.json 2> $OUT/head.err
python - $OUT/head.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("headline", d["value"], d["ms_per_step"], "plugin", d.get("plugin",{}).get("value"), d.get("plugin",{}).get("ms_per_step"))
PY
