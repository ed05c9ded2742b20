#!/bin/bash
set -u
OUT=gpurun_out/r6_s7; mkdir -p $OUT
for g in 1 2 4; do
BENCH_BATCH_GROUPS=$g BENCH_BATCH_OPTIONS=wide_until=0 timeout 600 python bench.py --leg throughput_batched --steps 100 --batched-leg 8,16,32 > $OUT/batched_g$g.json 2> $OUT/batched_g$g.err; python - $OUT/batched_g$g.json "groups=$g" <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))["throughput_batched"]
for B,r in d["by_B"].items(): print(sys.argv[2],"B",B,round(r["value"]),[round(v) for v in r["windows_scans_per_s"]], max(r["max_pose_error_by_sequence_m"]))
PY
tail -2 $OUT/batched_g$g.err
done
BENCH_BATCH_GROUPS=2 BENCH_BATCH_OPTIONS=wide_until=3 timeout 600 python bench.py --leg throughput_batched --steps 100 --batched-leg 16 > $OUT/batched_g2w.json 2> $OUT/batched_g2w.err; python - $OUT/batched_g2w.json "groups=2 wide" <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))["throughput_batched"]
for B,r in d["by_B"].items(): print(sys.argv[2],"B",B,round(r["value"]),[round(v) for v in r["windows_scans_per_s"]], max(r["max_pose_error_by_sequence_m"]))
PY
