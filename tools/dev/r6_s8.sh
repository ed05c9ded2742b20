#!/bin/bash
BENCH_BATCH_OPTIONS=wide_until=0 bash tools/batch_pmc.sh r6_s8/narrow8 8
