#!/bin/bash
# HBM traffic of the dominant kernel from PMC counters (separate passes, --kernel-trace only; MI355X_MICROARCH.md §HBM)
# usage: tools/pmc.sh <kernel-name-substring>  -> gpurun_out/pmc_<name>.json
R=$PWD; K=${1:-k_iterate_compact}
HEAD_SHA=$(cat $R/tools/.head_sha 2>/dev/null || echo unknown)
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 100 rocprofv3 --pmc $C --kernel-trace -f csv -d $R/gpurun_out/pmc_$C -o pmc -- python $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-profile --loop-steps 0 > $R/gpurun_out/pmc_$C.log 2>&1
done
cd $R && python - <<PY
import csv, glob, json
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"gpurun_out/pmc_{c}/**/*counter_collection.csv", recursive=True)
    vals = []
    for path in f:
        for row in csv.DictReader(open(path)):
            if "$K" in row.get("Kernel_Name", "") and row.get("Counter_Name") == c:
                vals.append(float(row["Counter_Value"]))
    out[c] = {"launches": len(vals), "mean": sum(vals) / max(1, len(vals))}
fetch_kb, write_kb = out["FETCH_SIZE"]["mean"], out["WRITE_SIZE"]["mean"]
res = {"kernel": "$K", "head": "$HEAD_SHA",
       "workload": "bench.py default (C2, untracked-map ping-pong), 6 timed frames = 120 launches",
       "counters": out, "unit_note": "FETCH_SIZE / WRITE_SIZE are KiB per dispatch (rocprofv3, separate "
       "--pmc passes). gfx950 correction of MI355X_MICROARCH.md (HBM section): FETCH_SIZE counts 128-B requests at 64 B for "
       "16-B-per-lane loads, which is what every load of this kernel is (float4 targets, float4 map points / normals, int2 "
       "rows) -> doubled; WRITE_SIZE is uncalibrated and taken as is. Infinity-Cache hits are counted, so this is "
       "fabric-side traffic, an upper bound on HBM bytes.",
       "fetch_bytes_corrected": 2 * fetch_kb * 1024.0, "write_bytes": write_kb * 1024.0,
       "hbm_bytes_per_launch_raw": (fetch_kb + write_kb) * 1024.0,
       "hbm_bytes_per_launch": (2 * fetch_kb + write_kb) * 1024.0}
json.dump(res, open("gpurun_out/pmc_$K.json", "w"), indent=1)
print(json.dumps(res))
PY
rm -rf gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE
