#!/bin/bash
# 40-byte entries: parity subset, then A/B against 48-byte entries (variant 128) on one box
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
OUT=$REPO/gpurun_out/r3_run21
mkdir -p $OUT
timeout 1500 python -m pytest tests/test_hip_parity.py tests/test_overflow_protocol.py tests/test_baseline_configs.py -m gpu -x -q > $OUT/tests.log 2>&1
tail -5 $OUT/tests.log
timeout 600 python tools/ab_kernel.py 50 5 short:0 full:128 > $OUT/ab.log 2>&1
cat $OUT/ab.log | tail -4
