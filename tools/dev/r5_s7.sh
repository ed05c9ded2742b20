#!/bin/bash
set -u
OUT=gpurun_out/r5g; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x -s > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|frames with another" $OUT/pytest.log | tail -8
