#!/bin/bash
# sweep of schedule knobs for the batched leg at saturation (48 sequences as 4 batches) and at B = 16
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6_s34; mkdir -p $OUT; exec > >(tee $OUT/stdout.txt) 2>&1
cd $GRAFT_REPO_ROOT
for o in "" "target_occupancy=12" "target_occupancy=20" "target_occupancy=24" "wide_until=1" "ball_max=128" "wave_misses=8" "far_max=256" "refresh_margin=0.002" ""; do
BENCH_BATCH_OPTIONS=$o timeout 600 python bench.py --leg throughput_batched --steps 100 --warmup 20 --batched-leg 16,48x4 > $OUT/b.json 2> $OUT/b.err
python - $OUT/b.json "${o:-default}" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1]))["throughput_batched"]
    print(f"{sys.argv[2]:24s}", "  ".join(f"{B}: {round(r['value'])} {[round(v) for v in r['windows_scans_per_s']]}" for B,r in d["by_B"].items()))
except Exception as e: print(sys.argv[2], "FAILED", e)
PY
done
