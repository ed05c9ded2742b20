#!/bin/bash
# round 4 dev: P processes x S sequences sharing the one GPU (each process has its own four hardware queues)
set -u
TAG=${1:-r4t2}; P=${2:-2}; S=${3:-4}; OPTS=${4:-}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
extra=""; for o in ${OPTS//,/ }; do [ -n "$o" ] && extra="$extra --option $o"; done
for p in $(seq 1 $P); do
  timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --loop-steps 0 --no-profile --plugin-steps 0 --odometry-loop 0 --sequences-per-gpu $S $extra > $OUT/p${P}s${S}_$p.json 2> $OUT/p${P}s${S}_$p.err &
done
wait
python - $OUT $P $S <<'PY'
import json,sys,glob
tot=0
for f in sorted(glob.glob(f"{sys.argv[1]}/p{sys.argv[2]}s{sys.argv[3]}_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); tot+=d['value']; print(f, round(d['value'],1), d['ms_per_step'])
    except Exception as e: print(f,"FAILED",e)
print(f"P={sys.argv[2]} S={sys.argv[3]} total (sum of the processes' own windows) {tot:.0f} scans/s")
PY
