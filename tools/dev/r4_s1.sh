#!/bin/bash
# round 4, session 1: the new one-lane ball search — schedule tests (small and benchmark size), A/B of the shapes, trace
set -u
TAG=${1:-r4a}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_pipeline.py -m gpu -q -x -k "schedule or nearest or tiny or chunked or calls_between or c2_full or library_loaded or fresh_context" > $OUT/pytest_sel.log 2>&1; echo "pytest(selected) rc=$?"; tail -15 $OUT/pytest_sel.log
bash tools/r4_ab.sh $TAG/ab "ball_search=0" "narrow_from=0" "narrow_from=1" "narrow_from=0,wave_misses=0" 2>&1 | tee $OUT/ab.txt
bash tools/gpu_trace.sh $TAG/trace_default 2>&1 | tail -22
bash tools/gpu_trace.sh $TAG/trace_narrow0 --option narrow_from=0 2>&1 | tail -22
