#!/bin/bash
set -u
OUT=gpurun_out/r6_s5; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_batch.py tests/test_gpu_parity.py -x -q -k "batch or schedule_options or far_from or c2_full" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/pytest.log
for o in "late_from=3" "late_from=-1" "late_from=3 --option late_waves=6" "late_from=2" "late_from=-1 --option hit_records=0"; do
timeout 300 python bench.py --steps 60 --no-cpu-baseline --loop-steps 0 --option $o > $OUT/head.json 2> $OUT/head.err
python - $OUT/head.json "$o" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d.get("roofline",{})
print(sys.argv[2], round(d["value"],1), "scans/s", round(d["ms_per_step"],4), "ms; iter kernel", round(r.get("avg_launch_us",0),2), "us; by iter", [round(v,1) for v in r.get("avg_launch_us_by_iteration_raw",[]) if v])
PY
done
for o in "late_from=3" "late_from=-1" "late_from=3,late_waves=6" "late_from=2" "late_from=1"; do
BENCH_BATCH_OPTIONS=wide_until=0,$o timeout 600 python bench.py --leg throughput_batched --steps 100 --batched-leg 8,16 > $OUT/batched.json 2> $OUT/batched.err; python - $OUT/batched.json "$o" <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))["throughput_batched"]
for B,r in d["by_B"].items(): print(sys.argv[2],"B",B,round(r["value"]),[round(v) for v in r["windows_scans_per_s"]], max(r["max_pose_error_by_sequence_m"]))
PY
done
BENCH_BATCH_OPTIONS=wide_until=0 bash tools/batch_trace.sh r6_s5/narrow8 8 | head -16
