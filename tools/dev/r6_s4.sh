#!/bin/bash
set -u
OUT=gpurun_out/r6_s4; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/pytest.log
for o in 1 0; do
timeout 300 python bench.py --steps 60 --no-cpu-baseline --loop-steps 0 --option hit_records=$o > $OUT/head_rec$o.json 2> $OUT/head_rec$o.err
python - $OUT/head_rec$o.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d.get("roofline",{})
print(sys.argv[1], round(d["value"],1), "scans/s", round(d["ms_per_step"],4), "ms; iter kernel", round(r.get("avg_launch_us",0),2), "us; by iter", [round(v,1) for v in r.get("avg_launch_us_by_iteration_raw",[]) if v])
PY
done
for o in 1 0; do
BENCH_BATCH_OPTIONS=wide_until=0,hit_records=$o timeout 600 python bench.py --leg throughput_batched --steps 100 --batched-leg 8,16 > $OUT/batched_rec$o.json 2> $OUT/batched_rec$o.err; python - $OUT/batched_rec$o.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))["throughput_batched"]
for B,r in d["by_B"].items(): print(sys.argv[1],"B",B,round(r["value"]),[round(v) for v in r["windows_scans_per_s"]], max(r["max_pose_error_by_sequence_m"]))
PY
done
BENCH_BATCH_OPTIONS=wide_until=0 bash tools/batch_trace.sh r6_s4/narrow8 8 | head -14
