#!/bin/bash
# dev tool: "knock-out" profile — library variants with one piece of the iteration kernel removed (built by hand into
# pylidar_slam_amd/_lib/variants/libicp_<name>.so); per variant: VALU / SALU instructions and duration by iteration index
# usage (on the GPU box): tools/knock_profile.sh TAG name ...
R=$PWD; TAG=${1:-r3k}; shift; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
L=$R/pylidar-slam_amd/pylidar_slam_amd/_lib
cp $L/libicp_mi355x.so /tmp/libicp_base.so
export TMPDIR=/tmp
B="python $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-profile --loop-steps 0 --plugin-steps 0 --odometry-loop 0"
for v in base "$@"; do
  if [ $v = base ]; then cp /tmp/libicp_base.so $L/libicp_mi355x.so; else cp $L/variants/libicp_$v.so $L/libicp_mi355x.so; fi
  rm -rf /tmp/kp; cd /tmp
  timeout 150 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY --kernel-trace -f csv -d /tmp/kp/1 -o p -- $B > /tmp/kp1.log 2>&1 || tail -3 /tmp/kp1.log
  timeout 150 rocprofv3 --kernel-trace -f csv -d /tmp/kp/2 -o p -- $B > /tmp/kp2.log 2>&1 || tail -3 /tmp/kp2.log
  cd $R
  python - $v <<'PY' | tee $OUT/knock_$v.txt
import csv, glob, collections, sys, statistics
per = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob("/tmp/kp/1/**/*counter_collection.csv", recursive=True):
    by = {}
    for r in csv.DictReader(open(path)):
        if "k_iterate_compact" in r["Kernel_Name"]:
            by.setdefault(int(r["Dispatch_Id"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
    ids = sorted(by); ids = ids[len(ids) % 20:]
    for k, i in enumerate(ids):
        if k >= len(ids) - 100:
            for c, v in by[i].items(): per[c][k % 20].append(v)
dur = collections.defaultdict(list)
for path in glob.glob("/tmp/kp/2/**/*kernel_trace.csv", recursive=True):
    rows = sorted((r for r in csv.DictReader(open(path)) if "k_iterate_compact" in r["Kernel_Name"]), key=lambda r: int(r["Start_Timestamp"]))
    rows = rows[len(rows) % 20:]
    for k, r in enumerate(rows):
        if k >= len(rows) - 100: dur[k % 20].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print(f"== {sys.argv[1]}")
print("iteration".ljust(18) + "".join(f"{i:>9d}" for i in range(8)))
print("duration us".ljust(18) + "".join(f"{statistics.median(dur[i]) if dur[i] else 0:9.1f}" for i in range(8)))
for c in sorted(per):
    print(c.ljust(18) + "".join(f"{sum(per[c][i]) / max(1, len(per[c][i])) / 1e3:9.0f}" for i in range(8)) + "  (thousands)")
PY
done
cp /tmp/libicp_base.so $L/libicp_mi355x.so
