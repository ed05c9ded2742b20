#!/bin/bash
# dev tool: SQ / TCC counters of the iteration kernel by iteration index within the frame (dispatch order, 20 per frame)
# usage: tools/pmc_by_iter.sh TAG [bench args]   -> gpurun_out/TAG/pmc_by_iter.txt
R=$PWD; TAG=${1:-r3y}; shift; EXTRA="$@"; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pk
B="python $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-profile --loop-steps 0 --plugin-steps 0 --odometry-loop 0 $EXTRA"
timeout 150 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VALU --kernel-trace -f csv -d /tmp/pk/1 -o p -- $B > /tmp/pk1.log 2>&1 || tail -3 /tmp/pk1.log
timeout 150 rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_WR SQ_INSTS_SMEM --kernel-trace -f csv -d /tmp/pk/2 -o p -- $B > /tmp/pk2.log 2>&1 || tail -3 /tmp/pk2.log
timeout 150 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum GRBM_GUI_ACTIVE --kernel-trace -f csv -d /tmp/pk/3 -o p -- $B > /tmp/pk3.log 2>&1 || tail -3 /tmp/pk3.log
python - > $OUT/pmc_by_iter.txt <<PY
import csv, glob, collections
per = collections.defaultdict(lambda: collections.defaultdict(list))   # counter -> iteration -> values
for d in ("1", "2", "3"):
    for path in glob.glob(f"/tmp/pk/{d}/**/*counter_collection.csv", recursive=True):
        rows = [r for r in csv.DictReader(open(path))]
        by_disp = collections.OrderedDict()
        for r in rows:
            if "k_iterate_compact" in r["Kernel_Name"]:
                by_disp.setdefault(int(r["Dispatch_Id"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
        ids = sorted(by_disp)
        ids = ids[len(ids) % 20:]           # whole frames, counted from the end
        for k, i in enumerate(ids):
            if k < len(ids) - 120: continue  # the last six frames
            for c, v in by_disp[i].items(): per[c][k % 20].append(v)
print("counter".ljust(30) + "".join(f"{i:>10d}" for i in range(20)))
for c in sorted(per):
    print(c.ljust(30) + "".join(f"{sum(per[c][i]) / max(1, len(per[c][i])):10.0f}" for i in range(20)))
PY
cat $OUT/pmc_by_iter.txt
