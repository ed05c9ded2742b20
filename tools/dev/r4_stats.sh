#!/bin/bash
# round 4 dev: in-kernel phase stamps of the iteration launches (search_stats 2 / 1) for a few frames, option sets as args
set -u
TAG=${1:-r4s}; shift; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
i=0
for v in "$@"; do i=$((i+1))
  extra=""; for o in ${v//,/ }; do extra="$extra --option $o"; done
  timeout 200 python bench.py --steps 4 --warmup 6 --no-cpu-baseline --loop-steps 0 --no-profile --plugin-steps 0 --odometry-loop 0 $extra > $OUT/s_$i.json 2> $OUT/s_$i.err
  echo "== $v"; grep -E "icp phases\] it +[0-6]:|icp stats|icp lead\] it +[1-4]:" $OUT/s_$i.err | tail -24
done
