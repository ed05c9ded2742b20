#!/bin/bash
# dev: phase stamps of the eager normal kernel for every library variant under tools/variants -> gpurun_out/$1
set -u
OUT=gpurun_out/${1:-var}; mkdir -p $OUT
export TMPDIR=/tmp
LIB=pylidar-slam_amd/pylidar_slam_amd/_lib/libicp_mi355x.so
cp $LIB /tmp/lib_default.so
for f in tools/variants/libicp_*.so; do
  v=$(basename $f .so); v=${v#libicp_}
  cp $f $LIB
  timeout 200 python bench.py --steps 16 --warmup 3 --no-cpu-baseline --loop-steps 0 --no-profile --option knn_select=1 --option search_stats=2 > $OUT/stats_$v.json 2> $OUT/stats_$v.err
  echo "== $v: $(python -c "import json;d=json.loads(open('$OUT/stats_$v.json').read().strip().splitlines()[-1]);print(round(d['value'],1),'scans/s median',round(d['ms_per_step_spread']['median'],3))")"; grep "icp normals" $OUT/stats_$v.err | tail -2 | cut -c1-175
done
cp /tmp/lib_default.so $LIB
