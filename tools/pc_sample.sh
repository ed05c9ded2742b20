#!/bin/bash
# dev tool: PC sampling of the bench (rocprofv3 beta) -> hottest source lines / instructions of the iteration kernel
# usage: tools/pc_sample.sh TAG [method: stochastic|host_trap] [bench args]
R=$PWD; TAG=${1:-r3pc}; M=${2:-stochastic}; shift; shift; EXTRA="$@"; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pcs
B="python $R/bench.py --steps 40 --warmup 3 --no-cpu-baseline --no-profile --loop-steps 0 --plugin-steps 0 --odometry-loop 0 $EXTRA"
if [ $M = stochastic ]; then U="--pc-sampling-unit cycles --pc-sampling-interval 1048576"; else U="--pc-sampling-unit time --pc-sampling-interval 1"; fi
ROCPROFILER_PC_SAMPLING_BETA_ENABLED=1 timeout 150 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-method $M $U --kernel-trace -f csv -d /tmp/pcs -o s -- $B > $OUT/pcs_$M.log 2>&1
echo "rc=$?"; tail -5 $OUT/pcs_$M.log
ls -la /tmp/pcs/* 2>/dev/null | head
for f in $(find /tmp/pcs -name "*pc_sampling*.csv"); do
  echo "== $f"; head -3 $f | cut -c1-400; wc -l $f
  gzip -c $f > $OUT/$(basename $f).gz
done
cp $(find /tmp/pcs -name "*kernel_trace.csv" | head -1) $OUT/pcs_kernel_trace.csv 2>/dev/null
ls -la $OUT
