#!/bin/bash
# round 4 dev: kNN + schedule tests, A/B of the normals kernels, trace
set -u
TAG=${1:-r4h}; shift; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_multi.py -m gpu -q -x -k "schedule or knn or nearest or tiny or map_sharded or lazy" > $OUT/pytest_sel.log 2>&1; echo "pytest(selected) rc=$?"; tail -8 $OUT/pytest_sel.log
bash tools/r4_ab.sh $TAG/ab "$@" 2>&1 | tee $OUT/ab.txt
bash tools/gpu_trace.sh $TAG/trace 2>&1 | tail -20
