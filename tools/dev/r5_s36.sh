#!/bin/bash
set -u
TAG=${1:-r5al}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
for v in "" "insert_by_cell=0" ""; do
  extra=""; for o in ${v//,/ }; do extra="$extra --option $o"; done
  BENCH_DIST_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 20 --warmup 5 --no-cpu-baseline $extra > $OUT/b2.json 2> $OUT/b2.err; echo "rc=$?"
  python - "$v" $OUT/b2.json <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[2]).read().strip().splitlines() if l.startswith("{")][-1])
    s=d.get("sharded",{})
    print(f"[{sys.argv[1]:20s}] replicas {d['value']:.0f} library {s.get('library',{}).get('value')} collective {s.get('collective',{}).get('value')} c4 {d.get('c4',{}).get('value')}")
except Exception as e: print("FAILED", e)
PY
done
