#!/bin/bash
# usage: tools/prof_any.sh <tag> <command...>  -> gpurun_out/<tag>_stats.txt (per-kernel totals from rocprofv3 --kernel-trace)
R=$PWD; tag=$1; shift
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $R/gpurun_out/$tag -o t -- "$@" > $R/gpurun_out/$tag.log 2>&1
cd $R && python - <<PY > gpurun_out/${tag}_stats.txt
import sqlite3,glob
c=sqlite3.connect(glob.glob('gpurun_out/$tag/*.db')[0])
rows=c.execute("select name, count(*), sum(end-start)/1e3, avg(end-start)/1e3, min(end-start)/1e3, max(end-start)/1e3 from kernels group by name order by 3 desc").fetchall()
tot=sum(r[2] for r in rows)
span=c.execute("select (max(end)-min(start))/1e6 from kernels").fetchone()[0]
print(f"# total kernel time {tot/1e3:.2f} ms over a {span:.1f} ms span, {sum(r[1] for r in rows)} launches")
print(f"{'kernel':72s} {'calls':>6s} {'total_ms':>9s} {'avg_us':>8s} {'min_us':>8s} {'max_us':>8s} {'pct':>6s}")
for r in rows:
    print(f"{r[0][:72]:72s} {r[1]:6d} {r[2]/1e3:9.2f} {r[3]:8.1f} {r[4]:8.1f} {r[5]:8.1f} {100*r[2]/tot:6.1f}")
PY
rm -rf gpurun_out/$tag
head -30 gpurun_out/${tag}_stats.txt
