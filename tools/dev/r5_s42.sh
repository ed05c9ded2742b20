#!/bin/bash
for i in 1 2 3 4 5 6 7 8; do timeout 120 python tools/dev/r5_thr_repeat.py 2>&1 | grep throughput; done
