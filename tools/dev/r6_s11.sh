#!/bin/bash
set -u
OUT=gpurun_out/r6_s11; mkdir -p $OUT
run() { tag=$1; shift; BENCH_BATCH_OPTIONS=${OPTS:-} timeout 900 python bench.py --leg throughput_batched --steps 100 --warmup 20 --batched-leg "$@" > $OUT/$tag.json 2> $OUT/$tag.err; python - $OUT/$tag.json "$tag" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1]))["throughput_batched"]
    for B,r in d["by_B"].items(): print(sys.argv[2],"B",B,round(r["value"]),[round(v) for v in r["windows_scans_per_s"]])
except Exception as e: print(sys.argv[2],"FAILED",e)
PY
}
BENCH_WORKLOAD_WORKERS=32 run sweep "32x2,32x4,32x8,48x4,64x4,64x8"
OPTS=cell_lists=0 BENCH_WORKLOAD_WORKERS=32 run nolists "32x4,64x4"
OPTS=far_min=8 BENCH_WORKLOAD_WORKERS=32 run farmin8 "32x4"
OPTS=wave_misses=24 BENCH_WORKLOAD_WORKERS=32 run wm24 "32x4"
OPTS=ball_lanes=2 BENCH_WORKLOAD_WORKERS=32 run bl2 "32x4"
OPTS=xcd_sectors=0 BENCH_WORKLOAD_WORKERS=32 run noxcd "32x4"
