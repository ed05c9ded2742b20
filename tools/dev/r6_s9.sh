#!/bin/bash
set -u
OUT=gpurun_out/r6_s9; mkdir -p $OUT
timeout 2400 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/pytest.log
for o in 1 0 1 0; do
timeout 300 python bench.py --steps 60 --no-cpu-baseline --loop-steps 0 --option cell_lists=$o > $OUT/head.json 2> $OUT/head.err
python - $OUT/head.json "cell_lists=$o" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d.get("roofline",{})
print(sys.argv[2], round(d["value"],1), "scans/s", round(d["ms_per_step"],4), "ms; median", round(d["ms_per_step_spread"]["median"],4), "refsched", round(d.get("reference_schedule",{}).get("value",0)))
PY
done
for o in 1 0; do
BENCH_BATCH_OPTIONS=wide_until=0,cell_lists=$o timeout 600 python bench.py --leg throughput_batched --steps 100 --warmup 20 --batched-leg 8,16,32x4 > $OUT/batched.json 2> $OUT/batched.err; python - $OUT/batched.json "cell_lists=$o" <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))["throughput_batched"]
for B,r in d["by_B"].items(): print(sys.argv[2],"B",B,round(r["value"]),[round(v) for v in r["windows_scans_per_s"]], max(r["max_pose_error_by_sequence_m"]))
PY
done
cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/prof -o t --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 30 --warmup 5 --no-cpu-baseline --loop-steps 0 --no-profile > /dev/null 2>&1; cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv,glob
f=glob.glob("gpurun_out/r6_s9/prof/**/*kernel_stats.csv",recursive=True)
for r in list(csv.DictReader(open(f[0])))[:16]: print(r["Name"][:60].ljust(60), r["Calls"].rjust(6), ("%.1f"%(float(r["AverageNs"])/1e3)).rjust(8),"us")
PY
rm -rf $OUT/prof
