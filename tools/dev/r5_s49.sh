#!/bin/bash
set -u
TAG=${1:-r5bc}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1; tail -2 $OUT/pytest.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
bash tools/dev/r5_s39.sh $TAG
