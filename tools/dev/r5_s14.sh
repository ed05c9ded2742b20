#!/bin/bash
# throughput leg knobs (4 sequences on 4 streams)
set -u
OUT=gpurun_out/r5n; mkdir -p $OUT
export TMPDIR=/tmp
for v in "base|" "nowide|--option wide_until=0" "q128|--option narrow_from=-1" "q128from3|--option narrow_from=3" "seq6|--sequences-per-gpu 6" "seq3|--sequences-per-gpu 3"; do
  name=${v%%|*}; args=${v#*|}
  case "$args" in *sequences-per-gpu*) sp="";; *) sp="--sequences-per-gpu 4";; esac
  timeout 300 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --loop-steps 0 --no-profile $sp $args > $OUT/$name.json 2> $OUT/$name.err
  python - $OUT/$name.json $name <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); s=d['ms_per_step_spread']
    print(f"{sys.argv[2]:12s} {d['value']:8.1f} scans/s  per-seq ms {d['ms_per_step']:.3f} median {s['median']:.3f} err {d['max_pose_error_vs_ground_truth_m']:.4f}")
except Exception as e: print(sys.argv[2],'FAILED',e)
PY
done
