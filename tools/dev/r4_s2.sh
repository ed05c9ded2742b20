#!/bin/bash
# round 4 dev: schedule tests + A/B + phase stamps for option sets given as args
set -u
TAG=${1:-r4c}; shift; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "schedule or nearest or tiny" > $OUT/pytest_sel.log 2>&1; echo "pytest(selected) rc=$?"; tail -5 $OUT/pytest_sel.log
bash tools/r4_ab.sh $TAG/ab "$@" 2>&1 | tee $OUT/ab.txt
bash tools/r4_stats.sh $TAG/st "search_stats=2,narrow_from=0" 2>&1 | grep -E "it +[0-3]:" | head -16
