#!/bin/bash
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6_s20; mkdir -p $OUT; exec > >(tee $OUT/stdout.txt) 2>&1
export TMPDIR=/tmp; cd /tmp
for dv in 0 1 3 7 15; do
rm -rf /tmp/pn_$dv
ICP_DEV_TAIL=$dv timeout 300 rocprofv3 --kernel-trace --stats -f csv -d /tmp/pn_$dv -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 40 --no-cpu-baseline --no-profile --loop-steps 0 --plugin-steps 0 --odometry-loop 0 --multi-gpu-legs 0 --throughput-leg 0 --batched-leg "" --option carry_normals=0 > $OUT/c2_$dv.json 2> $OUT/c2_$dv.err
f=$(find /tmp/pn_$dv -name "*kernel_stats.csv" | head -1)
echo "== C2 carry_normals=0 dev=$dv"
python3 -c "
import csv,sys
for r in csv.DictReader(open('$f')):
    n=r['Name']
    if 'normals' in n: print(n.split('(')[0][:50], r['Calls'], 'avg', round(float(r['AverageNs'])/1e3,1), 'min', round(float(r['MinNs'])/1e3,1), 'max', round(float(r['MaxNs'])/1e3,1))
"
done
