#!/bin/bash
# round 4 dev: the headline under several builds of the library (pylidar_slam_amd/_lib/variants/lib<X>.so), interleaved twice,
# plus the workgroup start skew / early-launch spans of each build
set -u
TAG=${1:-r4var}; shift; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
L=pylidar-slam_amd/pylidar_slam_amd/_lib
cp $L/libicp_mi355x.so $L/variants/libORIG.so
for rep in 1 2; do
 for v in "$@"; do
  lib=${v%%:*}; extra=""; [ "$lib" != "$v" ] && for o in ${v#*:}; do extra="$extra --option ${o//,/ --option }"; done
  cp $L/variants/lib$lib.so $L/libicp_mi355x.so
  timeout 120 python bench.py --steps 84 --warmup 6 --no-cpu-baseline --loop-steps 0 --no-profile --plugin-steps 0 --odometry-loop 0 --throughput-leg 0 $extra > $OUT/b_${v}_$rep.json 2> $OUT/b_${v}_$rep.err
  python - $OUT/b_${v}_$rep.json "$v $rep" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); s=d['ms_per_step_spread']
    print(f"{sys.argv[2]:12s} {d['value']:8.1f} scans/s mean {d['ms_per_step']:.3f} median {s['median']:.3f} p90 {s['p90']:.3f} max {s['max']:.3f} err {d['max_pose_error_vs_ground_truth_m']:.4f}")
except Exception as e: print(sys.argv[2], "FAILED", e)
PY
 done
done
for v in "$@"; do
  lib=${v%%:*}; extra=""; [ "$lib" != "$v" ] && for o in ${v#*:}; do extra="$extra --option ${o//,/ --option }"; done
  cp $L/variants/lib$lib.so $L/libicp_mi355x.so
  ICP_STATS_BLOCKS=1 timeout 100 python bench.py $extra --steps 14 --warmup 6 --no-cpu-baseline --loop-steps 0 --no-profile --plugin-steps 0 --odometry-loop 0 --throughput-leg 0 --option search_stats=2 > $OUT/s_$v.json 2> $OUT/s_$v.err
  echo "== $v"; grep -E "icp phases\] it +[0-2]:" $OUT/s_$v.err | tail -42 | awk '{print $3,$4,"skew",$7,"span",$9,"B max",$(20)}' | sort | awk '{k=$1" "$2; s[k]+=$6; n[k]++; sk[k]+=$4} END{for(k in s) printf "%s mean span %.1f skew %.2f (n=%d)\n",k,s[k]/n[k],sk[k]/n[k],n[k]}' | sort
done
cp $L/variants/libORIG.so $L/libicp_mi355x.so
