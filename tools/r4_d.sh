#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
OUT=$REPO/gpurun_out/r4_d
mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q -x > $OUT/gpu_tests.log 2>&1
tail -3 $OUT/gpu_tests.log
timeout 1500 python tools/ab_libs.py 3 c2,c5 base old pe3 p233 pnone 2>&1 | tee $OUT/ab_prio.log | tail -12
