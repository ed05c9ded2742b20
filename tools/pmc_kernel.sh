#!/bin/bash
# dev tool: SQ counters of one kernel of bench.py.  usage: tools/pmc_kernel.sh <kernel substring> [bench args]
R=$PWD; K=${1:-k_normals_all}; shift; EXTRA="$@"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pk
timeout 120 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VALU --kernel-trace -f csv -d /tmp/pk/1 -o p -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-profile --loop-steps 0 $EXTRA > /tmp/pk1.log 2>&1 || tail -3 /tmp/pk1.log
timeout 120 rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT --kernel-trace -f csv -d /tmp/pk/2 -o p -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-profile --loop-steps 0 $EXTRA > /tmp/pk2.log 2>&1 || tail -3 /tmp/pk2.log
python - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for path in glob.glob("/tmp/pk/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(path)):
        if "$K" in row["Kernel_Name"]:
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
for c in sorted(acc):
    v = acc[c]
    print(f"{c:28s} n={len(v):4d} mean={sum(v)/len(v):14.0f}")
PY
