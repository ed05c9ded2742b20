#!/bin/bash
# normals kernel: main pass vs stragglers (normals_tail_stream splits them into two kernels), straggler counts
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6_s17; mkdir -p $OUT; exec > >(tee $OUT/stdout.txt) 2>&1
export TMPDIR=/tmp; cd /tmp
for tail in 0 1; do
rm -rf /tmp/pn$tail
timeout 300 rocprofv3 --kernel-trace --stats -f csv -d /tmp/pn$tail -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 40 --no-cpu-baseline --no-profile --loop-steps 0 --plugin-steps 0 --odometry-loop 0 --multi-gpu-legs 0 --throughput-leg 0 --batched-leg "" --option carry_normals=0 --option normals_tail_stream=$tail > $OUT/c2_tail$tail.json 2> $OUT/c2_tail$tail.err
f=$(find /tmp/pn$tail -name "*kernel_stats.csv" | head -1)
echo "== C2 carry_normals=0 normals_tail_stream=$tail"; python3 -c "
import csv,sys
for r in csv.DictReader(open('$f')):
    n=r['Name']
    if 'normals' in n or 'hood' in n: print(n.split('(')[0][:50], r['Calls'], 'avg', round(float(r['AverageNs'])/1e3,1), 'min', round(float(r['MinNs'])/1e3,1), 'max', round(float(r['MaxNs'])/1e3,1))
"
done
timeout 200 python $GRAFT_REPO_ROOT/bench.py --steps 20 --no-cpu-baseline --no-profile --loop-steps 0 --plugin-steps 0 --odometry-loop 0 --multi-gpu-legs 0 --throughput-leg 0 --batched-leg "" --option carry_normals=0 --option search_stats=1 2>&1 | grep "icp stats" | tail -3
rm -rf /tmp/pn2
timeout 300 rocprofv3 --kernel-trace --stats -f csv -d /tmp/pn2 -o p -- python $GRAFT_REPO_ROOT/bench.py --leg odometry_loop > $OUT/odo.json 2> $OUT/odo.err
f=$(find /tmp/pn2 -name "*kernel_stats.csv" | head -1)
echo "== odometry_loop"; python3 -c "
import csv,sys
for r in csv.DictReader(open('$f')):
    n=r['Name']
    if 'normals' in n or 'hood' in n: print(n.split('(')[0][:50], r['Calls'], 'avg', round(float(r['AverageNs'])/1e3,1), 'min', round(float(r['MinNs'])/1e3,1), 'max', round(float(r['MaxNs'])/1e3,1))
"
