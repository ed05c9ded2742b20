#!/bin/bash
# the stragglers of the kNN normals in a launch of their own: tests, kernel times, A/B of the legs that estimate normals
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6_s19; mkdir -p $OUT; exec > >(tee $OUT/stdout.txt) 2>&1
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q -k "knn_normals or schedule_options or bit_identical or carried or map_normals or sharded or loop" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/pytest.log
export TMPDIR=/tmp; cd /tmp
for nl in 1 0; do
rm -rf /tmp/pn_$nl
timeout 300 rocprofv3 --kernel-trace --stats -f csv -d /tmp/pn_$nl -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 40 --no-cpu-baseline --no-profile --loop-steps 0 --plugin-steps 0 --odometry-loop 0 --multi-gpu-legs 0 --throughput-leg 0 --batched-leg "" --option carry_normals=0 --option normals_list=$nl > $OUT/c2_$nl.json 2> $OUT/c2_$nl.err
f=$(find /tmp/pn_$nl -name "*kernel_stats.csv" | head -1)
echo "== C2 carry_normals=0 normals_list=$nl"
python3 -c "
import csv,sys
for r in csv.DictReader(open('$f')):
    n=r['Name']
    if 'normals' in n or 'hood' in n: print(n.split('(')[0][:50], r['Calls'], 'avg', round(float(r['AverageNs'])/1e3,1), 'min', round(float(r['MinNs'])/1e3,1), 'max', round(float(r['MaxNs'])/1e3,1))
"
done
rm -rf /tmp/pn2
timeout 300 rocprofv3 --kernel-trace --stats -f csv -d /tmp/pn2 -o p -- python $GRAFT_REPO_ROOT/bench.py --leg odometry_loop > $OUT/odo_prof.json 2> $OUT/odo_prof.err
f=$(find /tmp/pn2 -name "*kernel_stats.csv" | head -1)
echo "== odometry_loop (default)"
python3 -c "
import csv,sys
for r in csv.DictReader(open('$f')):
    n=r['Name']
    if 'normals' in n or 'hood' in n: print(n.split('(')[0][:50], r['Calls'], 'avg', round(float(r['AverageNs'])/1e3,1), 'min', round(float(r['MinNs'])/1e3,1), 'max', round(float(r['MaxNs'])/1e3,1))
"
cd $GRAFT_REPO_ROOT
for r in 1 2; do for nl in 1 0; do
timeout 300 python bench.py --steps 70 --no-cpu-baseline --loop-steps 0 --multi-gpu-legs 0 --throughput-leg 0 --batched-leg "" --option normals_list=$nl > $OUT/head_$nl.json 2> $OUT/head_$nl.err
python - $OUT/head_$nl.json "normals_list=$nl" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], "headline", round(d["value"],1), "| reference_schedule", d.get("reference_schedule",{}).get("value"), "| plugin", d.get("plugin",{}).get("value"), "| odometry_loop ms/frame", d.get("odometry_loop",{}).get("ms_per_frame"))
PY
done; done
