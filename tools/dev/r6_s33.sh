#!/bin/bash
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6_s33; mkdir -p $OUT; exec > >(tee $OUT/stdout.txt) 2>&1
cd $GRAFT_REPO_ROOT
echo skip tests
for r in 1 2 3 4; do for mode in numpy memmove; do
  if [ $mode = numpy ]; then export ICP_DEV_NUMPY_COPY=1; else unset ICP_DEV_NUMPY_COPY; fi
  timeout 300 python bench.py --leg odometry_loop > $OUT/odo_$mode.json 2> $OUT/odo_$mode.err
  timeout 300 python bench.py --leg plugin > $OUT/plugin_$mode.json 2> $OUT/plugin_$mode.err
  python - $OUT/odo_$mode.json $OUT/plugin_$mode.json $mode <<'PY'
import json,sys
o=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])["odometry_loop"]; p=json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])["plugin"]
print(f"{sys.argv[3]:8s} odometry_loop {o['ms_per_frame']:.4f} ms/frame (full window {o['ms_per_frame_full_window']:.4f}, median {o['ms_per_frame_spread']['median']:.4f})   plugin {p['value']:.1f} scans/s")
PY
done; done
