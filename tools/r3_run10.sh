#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
OUT=$REPO/gpurun_out/r3_run10
mkdir -p $OUT
timeout 1800 python -m pytest tests/test_envelope.py -m gpu -q -s > $OUT/envelope.log 2>&1
grep -E "implementation-defined|passed|failed" $OUT/envelope.log
timeout 2400 python -m pytest tests -m gpu -q > $OUT/gpu_tests.log 2>&1
tail -3 $OUT/gpu_tests.log
timeout 600 python tools/ab_forest.py base:0 > $OUT/forest.log 2>&1; cat $OUT/forest.log
timeout 600 python tools/ab_kernel.py 50 5 base:0 > $OUT/c2.log 2>&1; tail -1 $OUT/c2.log
