#!/bin/bash
# the VALU instructions of the batched iteration launches BY CLASS (and thread cycles, LDS loads / stores, branches), launch by launch: tools/batch_pmc_classes.sh TAG B -> gpurun_out/TAG/pmc.txt
set -u
TAG=$1; B=$2; shift 2
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
i=0
for SET in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32" \
           "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_INT64 SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_BRANCH SQ_INSTS_SALU" \
           "SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_WAIT_INST_LDS SQ_INSTS_VSKIPPED SQ_INST_CYCLES_SALU SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES"; do
  i=$((i+1)); rm -rf /tmp/bp$i
  timeout 300 rocprofv3 --pmc $SET --kernel-trace -f csv -d /tmp/bp$i -o p -- python $GRAFT_REPO_ROOT/bench.py --leg throughput_batched --batched-leg $B --steps 20 --warmup 5 "$@" > /tmp/bp$i.log 2>&1 || tail -3 /tmp/bp$i.log
done
cd $GRAFT_REPO_ROOT
python - "$OUT" <<'PY'
import csv, glob, collections, sys
out = sys.argv[1]
per = collections.defaultdict(lambda: collections.defaultdict(list))  # kernel -> counter -> values in dispatch order
for path in sorted(glob.glob("/tmp/bp*/**/*counter_collection.csv", recursive=True)):
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    for r in rows:
        k = r["Kernel_Name"].split("(")[0]
        if "k_iterate" in k:
            per["iterate"][r["Counter_Name"]].append(float(r["Counter_Value"]))
        elif "k_grid" in k or "k_project" in k or "k_pack" in k or "k_sum_solve" in k:
            per[k[:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(out + "/pmc.txt", "w") as o:
    it = per["iterate"]
    names = sorted(it)
    n = min(len(v) for v in it.values())
    steps = n // 20
    print(f"# k_iterate(_late)_batch: {n} launches = {steps} steps of 20; mean per iteration index over the last {min(steps, 20)} steps", file=o)
    print("iter " + " ".join(f"{c[3:] if c.startswith('SQ_') else c:>16s}" for c in names), file=o)
    use = min(steps, 20)
    for i in range(20):
        vals = []
        for c in names:
            v = it[c][:steps * 20]
            sel = [v[s * 20 + i] for s in range(steps - use, steps)]
            vals.append(sum(sel) / len(sel))
        print(f"{i:4d} " + " ".join(f"{x:16.0f}" for x in vals), file=o)
    for k, cs in per.items():
        if k == "iterate":
            continue
        print(f"# {k}: " + ", ".join(f"{c}={sum(v)/len(v):.0f}" for c, v in sorted(cs.items())), file=o)
print(open(out + "/pmc.txt").read())
PY
