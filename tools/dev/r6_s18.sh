#!/bin/bash
# timing only: the pair pass of k_normals_hood2 without its stragglers (tools/ab/skip_libicp_mi355x.so: wrong normals for 0.3 % of the map)
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6_s18; mkdir -p $OUT; exec > >(tee $OUT/stdout.txt) 2>&1
LIB=$GRAFT_REPO_ROOT/pylidar-slam_amd/pylidar_slam_amd/_lib/libicp_mi355x.so
cp $LIB /tmp/new_lib.so
export TMPDIR=/tmp; cd /tmp
for which in new skip; do
[ $which = skip ] && cp $GRAFT_REPO_ROOT/tools/ab/skip_libicp_mi355x.so $LIB || cp /tmp/new_lib.so $LIB
rm -rf /tmp/pn_$which
timeout 300 rocprofv3 --kernel-trace --stats -f csv -d /tmp/pn_$which -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 40 --no-cpu-baseline --no-profile --loop-steps 0 --plugin-steps 0 --odometry-loop 0 --multi-gpu-legs 0 --throughput-leg 0 --batched-leg "" --option carry_normals=0 > $OUT/c2_$which.json 2> $OUT/c2_$which.err
f=$(find /tmp/pn_$which -name "*kernel_stats.csv" | head -1)
echo "== C2 carry_normals=0 $which"
python3 -c "
import csv,sys
for r in csv.DictReader(open('$f')):
    n=r['Name']
    if 'normals' in n or 'hood' in n: print(n.split('(')[0][:50], r['Calls'], 'avg', round(float(r['AverageNs'])/1e3,1), 'min', round(float(r['MinNs'])/1e3,1), 'max', round(float(r['MaxNs'])/1e3,1))
"
rm -rf /tmp/pn2
timeout 300 rocprofv3 --kernel-trace --stats -f csv -d /tmp/pn2 -o p -- python $GRAFT_REPO_ROOT/bench.py --leg odometry_loop > $OUT/odo_$which.json 2> $OUT/odo_$which.err
f=$(find /tmp/pn2 -name "*kernel_stats.csv" | head -1)
echo "== odometry_loop $which"
python3 -c "
import csv,sys
for r in csv.DictReader(open('$f')):
    n=r['Name']
    if 'normals' in n or 'hood' in n: print(n.split('(')[0][:50], r['Calls'], 'avg', round(float(r['AverageNs'])/1e3,1), 'min', round(float(r['MinNs'])/1e3,1), 'max', round(float(r['MaxNs'])/1e3,1))
"
done
cp /tmp/new_lib.so $LIB
