#!/bin/bash
# dev tool: hardware counters of the search micro-probe kernels (one rocprofv3 pass per counter group)
R=$PWD; H=${1:-0.33}
python tools/search_probe.py >/dev/null
cd /tmp && export TMPDIR=/tmp
i=0
for G in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VALU" \
         "TCP_TOTAL_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum" \
         "TCP_TCP_LATENCY_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_CACHE_ACCESSES_sum" \
         "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" \
         "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TD_TD_BUSY_sum" \
         "GRBM_GUI_ACTIVE TA_FLAT_READ_WAVEFRONTS_sum" \
         "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum" \
         "TCP_TAGRAM0_REQ_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_GATE_EN1_sum"; do
  i=$((i+1))
  PROBE_REPS=2 timeout 60 rocprofv3 --pmc $G --kernel-trace -f csv -d /tmp/pp/$i -o p -- $R/tools/search_probe.bin /tmp/probe.bin $H > /tmp/pp_$i.log 2>&1 || echo "pass $i ($G) failed: $(tail -2 /tmp/pp_$i.log | cut -c1-200)"
done
python - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob("/tmp/pp/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(path)):
        k = row["Kernel_Name"].split("(")[0].replace("void ", "")
        acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
names = sorted({c for k in acc for c in acc[k]})
kernels = [k for k in acc if any(t in k for t in ("v2_own", "v4_probe", "v3_full", "v5_flat<256, 4>", "v0_noop"))]
print("counter".ljust(40), *[k[-16:].rjust(18) for k in kernels])
for c in names:
    print(c.ljust(40), *[f"{sum(acc[k][c]) / max(1, len(acc[k][c])):18.0f}" for k in kernels])
PY
