#!/bin/bash
# usage: prof.sh <tag> [bench args]  -> gpurun_out/<tag>_stats.txt
R=$PWD; tag=$1; shift
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $R/gpurun_out/$tag -o bench -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline "$@" > $R/gpurun_out/$tag.log 2>&1
cd $R && python - <<PY > gpurun_out/${tag}_stats.txt
import sqlite3,glob
c=sqlite3.connect(glob.glob('gpurun_out/$tag/*.db')[0])
rows=c.execute("select name, count(*), sum(end-start)/1e3, avg(end-start)/1e3, min(end-start)/1e3, max(end-start)/1e3 from kernels group by name order by 3 desc").fetchall()
tot=sum(r[2] for r in rows)
print(f"# total kernel time {tot/1e3:.2f} ms")
print(f"{'kernel':72s} {'calls':>6s} {'total_ms':>9s} {'avg_us':>8s} {'min_us':>8s} {'max_us':>8s} {'pct':>6s}")
for r in rows:
    print(f"{r[0][:72]:72s} {r[1]:6d} {r[2]/1e3:9.2f} {r[3]:8.1f} {r[4]:8.1f} {r[5]:8.1f} {100*r[2]/tot:6.1f}")
# per-iteration durations of the fused iteration kernel for the last frames (us), one line per frame
it=[r[0] for r in c.execute("select (end-start)/1e3 from kernels where name like '%k_iterate_rows%' order by start").fetchall()]
print("# k_iterate_rows per iteration (us), last 6 frames:")
for f in range(max(0,len(it)//20-6), len(it)//20):
    print("#  "+" ".join(f"{v:5.1f}" for v in it[20*f:20*f+20]))
PY
rm -rf gpurun_out/$tag
head -14 gpurun_out/${tag}_stats.txt
