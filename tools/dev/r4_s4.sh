#!/bin/bash
# round 4 dev: odometry_loop leg and C4 under option sets (no profiler)
set -u
TAG=${1:-r4u}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
i=0
for v in "" "eager_normals_limit=0" "narrow_from=3,ball_search=0" "narrow_from=-1" "hoods=1" "narrow_from=3,ball_search=0,hoods=1"; do i=$((i+1))
  extra=""; for o in ${v//,/ }; do extra="$extra --option $o"; done
  timeout 150 python bench.py --leg odometry_loop --no-cpu-baseline $extra > $OUT/o_$i.json 2> $OUT/o_$i.err
  python - $OUT/o_$i.json "$v" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])['odometry_loop']
    print(f"odometry_loop [{sys.argv[2]:40s}] {d['ms_per_frame']:.3f} ms/frame median {d['ms_per_frame_spread']['median']:.3f} iters {d['iterations_per_frame']['mean']:.2f} dev {d.get('max_translation_deviation_from_reference_run_m',0):.2e}")
except Exception as e: print(sys.argv[2], "FAILED", e)
PY
done
i=0
for v in "" "hoods=1" "hoods=0"; do i=$((i+1))
  extra=""; for o in ${v//,/ }; do extra="$extra --option $o"; done
  timeout 250 python bench.py --workload c4 --steps 6 --warmup 2 --no-cpu-baseline $extra > $OUT/c_$i.json 2> $OUT/c_$i.err
  python - $OUT/c_$i.json "$v" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])['c4']
    print(f"c4 [{sys.argv[2]:20s}] sharded-normals {d['map_sharded_normals'].get('ms_per_step',0):.3f} ms, lazy {d.get('lazy_normals',{}).get('ms_per_step',0):.3f} ms")
except Exception as e: print(sys.argv[2], "FAILED", e)
PY
done
