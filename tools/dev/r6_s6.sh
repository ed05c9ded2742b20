#!/bin/bash
set -u
OUT=gpurun_out/r6_s6; mkdir -p $OUT
for o in 1 0; do
timeout 300 python bench.py --steps 14 --warmup 14 --no-cpu-baseline --loop-steps 0 --no-profile --option late_from=-1 --option hit_records=$o --option search_stats=2 > $OUT/stats_rec$o.json 2> $OUT/stats_rec$o.err
grep "icp phases" $OUT/stats_rec$o.err | tail -280 | awk '{it=$4; sub(":","",it); a[it]+=$16; b[it]+=$20; r[it]+=$(NF-2); sp[it]+=$10; n[it]++} END{for(i=0;i<20;i++) printf "it %2d: span %.2f A %.2f B %.2f reduce %.2f (n=%d)\n", i, sp[i]/n[i], a[i]/n[i], b[i]/n[i], r[i]/n[i], n[i]}' > $OUT/phases_rec$o.txt
echo "records=$o"; cat $OUT/phases_rec$o.txt
grep "icp phases" $OUT/stats_rec$o.err | tail -3
grep "icp lead" $OUT/stats_rec$o.err | tail -3
done
