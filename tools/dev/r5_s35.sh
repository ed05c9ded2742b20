#!/bin/bash
set -u
TAG=${1:-r5ak}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
i=0
for v in "" "insert_by_cell=0" "" "insert_by_cell=0" "" "insert_by_cell=0"; do i=$((i+1))
  extra=""; for o in ${v//,/ }; do extra="$extra --option $o"; done
  timeout 200 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --loop-steps 0 --no-profile --plugin-steps 0 --odometry-loop 0 --throughput-leg 0 $extra > $OUT/h_$i.json 2> $OUT/h_$i.err
  python - "$v" $OUT/h_$i.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
print(f"[{sys.argv[1]:24s}] {d['value']:.0f} scans/s {d['ms_per_step']:.4f} ms median {d['ms_per_step_spread']['median']:.4f}")
PY
done
for v in "" "insert_by_cell=0" "" "insert_by_cell=0"; do
  extra=""; for o in ${v//,/ }; do extra="$extra --option $o"; done
  timeout 200 python bench.py --leg odometry_loop --no-cpu-baseline $extra > $OUT/odo.json 2> $OUT/odo.err
  python - "$v" $OUT/odo.json <<'PY'
import json,sys
d=json.load(open(sys.argv[2]))["odometry_loop"]
print(f"odo [{sys.argv[1]:24s}] ms/frame {d['ms_per_frame']:.4f} full-window {d['ms_per_frame_full_window']:.4f} median {d['ms_per_frame_spread']['median']:.4f}")
PY
done
