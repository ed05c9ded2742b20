#!/bin/bash
# dev session: parity subset + A/B list (see tools/gpu_ab.sh)
OUT=gpurun_out/$1; shift
mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/pytest.log
bash tools/gpu_ab.sh ${OUT#gpurun_out/} "$@"
