#!/bin/bash
set -u
OUT=gpurun_out/r6_s3; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_batch.py -x -q > $OUT/pytest_batch.log 2>&1; echo "batch rc=$?"; tail -5 $OUT/pytest_batch.log
timeout 600 python bench.py --leg throughput_batched --steps 100 --batched-leg 4,8,16 > $OUT/batched.json 2> $OUT/batched.err; python - <<'PY'
import json
d=json.load(open("gpurun_out/r6_s3/batched.json"))["throughput_batched"]
for B,r in d["by_B"].items(): print("wide  B",B,round(r["value"]),r["windows_scans_per_s"])
PY
BENCH_BATCH_OPTIONS=wide_until=0 timeout 600 python bench.py --leg throughput_batched --steps 100 --batched-leg 4,8,16 > $OUT/batched_narrow.json 2> $OUT/batched_narrow.err; python - <<'PY'
import json
d=json.load(open("gpurun_out/r6_s3/batched_narrow.json"))["throughput_batched"]
for B,r in d["by_B"].items(): print("narrow B",B,round(r["value"]),r["windows_scans_per_s"])
PY
BENCH_BATCH_OPTIONS=wide_until=0 bash tools/batch_trace.sh r6_s3/narrow8 8 | head -14
BENCH_BATCH_OPTIONS=wide_until=0 bash tools/batch_trace.sh r6_s3/narrow16 16 | head -14
