#!/bin/bash
set -u
OUT=gpurun_out/r6_s12; mkdir -p $OUT
for o in "overlap_map_update=0" "overlap_map_update=1" "normals_tail_stream=1" "overlap_map_update=1 --option normals_tail_stream=1" "overlap_map_update=0"; do
timeout 300 python bench.py --leg odometry_loop --option $o > $OUT/odo.json 2> $OUT/odo.err
python - $OUT/odo.json "$o" <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))["odometry_loop"]
print(sys.argv[2], "ms/frame", round(d["ms_per_frame"],4), "full window", round(d["ms_per_frame_full_window"],4), d["ms_per_frame_spread"], "max dev", d.get("max_translation_deviation_on_frames_of_equal_iteration_count_m"), d.get("frames_with_other_iteration_count"))
PY
done
python - <<'PY'
# host-side profile of the published-configuration loop: where the Python time of a frame goes
import cProfile, pstats, sys, os, io
sys.argv=["bench.py","--leg","odometry_loop"]
sys.path.insert(0, os.getcwd())
import bench
args=bench.parse()
pr=cProfile.Profile(); pr.enable(); out=bench.odometry_loop_leg(args,0); pr.disable()
s=io.StringIO(); pstats.Stats(pr,stream=s).sort_stats("cumulative").print_stats(35); print(s.getvalue()[:6000])
print("ms/frame under cProfile", out["ms_per_frame"])
PY
