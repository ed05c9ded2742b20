#!/bin/bash
# round 3 dev: option sweep in throughput mode (4 sequences on one GPU).  usage: tools/r3_sweep4.sh TAG "opts" ...
set -u
TAG=$1; shift; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
i=0
for combo in "$@"; do
  i=$((i+1)); extra=""
  for o in $combo; do [ "$o" != "-" ] && extra="$extra --option $o"; done
  timeout 300 python bench.py --steps 60 --warmup 6 --sequences-per-gpu 4 --no-cpu-baseline --loop-steps 0 --no-profile $extra > $OUT/s4_$i.json 2> $OUT/s4_$i.err
  python - $OUT/s4_$i.json "$combo" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(f"{sys.argv[2]:40s} {d['value']:8.1f} scans/s (4 sequences) ms/frame each {d['ms_per_step']:.3f}")
except Exception as e: print(sys.argv[2], "FAILED", e)
PY
done
