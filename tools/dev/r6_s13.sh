#!/bin/bash
set -u
OUT=gpurun_out/r6_s13; mkdir -p $OUT
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_batch.py -x -q -k "schedule_options or far_from or c2_full or batch or bit_identical" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest.log
for o in 1 0 1 0; do
timeout 300 python bench.py --steps 70 --no-cpu-baseline --loop-steps 0 --option ball_empty=$o > $OUT/head.json 2> $OUT/head.err
python - $OUT/head.json "ball_empty=$o" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d.get("roofline",{})
print(sys.argv[2], round(d["value"],1), "scans/s", round(d["ms_per_step"],4), "ms", {k:round(v,3) for k,v in d["ms_per_step_spread"].items()}, "by iter", [round(v,1) for v in r.get("avg_launch_us_by_iteration_raw",[])[:6]])
PY
done
for o in 1 0; do
BENCH_BATCH_OPTIONS=ball_empty=$o timeout 600 python bench.py --leg throughput_batched --steps 100 --warmup 20 --batched-leg 8,16,48x4 > $OUT/batched.json 2> $OUT/batched.err; python - $OUT/batched.json "ball_empty=$o" <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))["throughput_batched"]
for B,r in d["by_B"].items(): print(sys.argv[2],"B",B,round(r["value"]),[round(v) for v in r["windows_scans_per_s"]], max(r["max_pose_error_by_sequence_m"]))
PY
done
