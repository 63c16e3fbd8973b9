#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
OUT=$REPO/gpurun_out/r3_run3
mkdir -p $OUT
timeout 1500 python -m pytest tests/test_hip_parity.py tests/test_overflow_protocol.py tests/test_baseline_configs.py -m gpu -x -q > $OUT/parity.log 2>&1
tail -5 $OUT/parity.log
timeout 600 python tools/ab_forest.py base:0 rows1:128 rows4:256 > $OUT/forest.log 2>&1
timeout 600 python tools/ab_kernel.py 50 5 base:0 rows2:512 > $OUT/c2.log 2>&1
cat $OUT/forest.log $OUT/c2.log
