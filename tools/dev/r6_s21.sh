#!/bin/bash
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6_s21; mkdir -p $OUT; exec > >(tee $OUT/stdout.txt) 2>&1
cd $GRAFT_REPO_ROOT
python -c "
import cProfile, pstats, sys, io
sys.argv=['bench.py','--leg','odometry_loop']
import runpy
pr=cProfile.Profile()
pr.enable()
try:
    runpy.run_path('bench.py', run_name='__main__')
except SystemExit: pass
pr.disable()
s=io.StringIO(); pstats.Stats(pr,stream=s).sort_stats('tottime').print_stats(45); print(s.getvalue()[:9000])
" 2>&1 | tail -75
