#!/bin/bash
set -u
TAG=${1:-r5z}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
for v in "" "overlap_map_update=1" "lazy_fused=2,carry_normals=0" "hoods=0"; do
  extra=""; for o in ${v//,/ }; do extra="$extra --option $o"; done
  for r in 1 2; do
  timeout 200 python bench.py --leg odometry_loop --no-cpu-baseline $extra > $OUT/odo_${v}_$r.json 2> $OUT/odo_${v}_$r.err
  python - "$v" $OUT/odo_${v}_$r.json <<'PY'
import json,sys
d=json.load(open(sys.argv[2]))["odometry_loop"]
print(f"odo [{sys.argv[1]:30s}] ms/frame {d['ms_per_frame']:.4f} full-window {d['ms_per_frame_full_window']:.4f} median {d['ms_per_frame_spread']['median']:.4f} dev {d.get('max_translation_deviation_from_reference_run_m')} other {d.get('frames_with_other_iteration_count')}")
PY
  done
done
