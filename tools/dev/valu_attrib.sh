#!/bin/bash
# dev: VALU instructions per wave of the late batched launches for several BUILDS (tools/ab/<name>_libicp_mi355x.so) — timing-only
# variants with one piece of the hit path switched off attribute the instruction count to the pieces
set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/valu_attrib; mkdir -p $OUT; exec > >(tee $OUT/stdout.txt) 2>&1
LIB=$R/pylidar-slam_amd/pylidar_slam_amd/_lib/libicp_mi355x.so
cp $LIB /tmp/intree_lib.so
export TMPDIR=/tmp; cd /tmp
for n in "$@"; do
  cp $R/tools/ab/${n}_libicp_mi355x.so $LIB
  rm -rf /tmp/va_$n
  timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES --kernel-trace -f csv -d /tmp/va_$n -o p -- python $R/bench.py --leg throughput_batched --batched-leg 8 --steps 20 --warmup 5 > /tmp/va_$n.log 2>&1 || tail -3 /tmp/va_$n.log
  python3 - /tmp/va_$n "$n" <<'PY'
import csv, glob, collections, sys
per = collections.defaultdict(list)
for path in sorted(glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)):
    rows = list(csv.DictReader(open(path))); rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    for r in rows:
        if "k_iterate" in r["Kernel_Name"]: per[r["Counter_Name"]].append(float(r["Counter_Value"]))
n = min(len(v) for v in per.values()); steps = n // 20
def at(c, i): return sum(per[c][s * 20 + i] for s in range(steps - 10, steps)) / 10
for i in (0, 2, 15):
    w = at("SQ_WAVES", i)
    print(f"{sys.argv[2]:6s} iteration {i:2d}: per wave VALU {at('SQ_INSTS_VALU', i)/w:7.1f}  SALU {at('SQ_INSTS_SALU', i)/w:6.1f}  LDS {at('SQ_INSTS_LDS', i)/w:5.1f}  VMEM_RD {at('SQ_INSTS_VMEM_RD', i)/w:5.1f}  wave cycles {at('SQ_WAVE_CYCLES', i)/w:7.0f}")
PY
done
cp /tmp/intree_lib.so $LIB
