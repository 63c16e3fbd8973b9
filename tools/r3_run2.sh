#!/bin/bash
# round-3 GPU run 2: full GPU suite, forest SQ counters
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
OUT=$REPO/gpurun_out/r3_run2
mkdir -p $OUT
timeout 2400 python -m pytest tests -m gpu -x -q > $OUT/gpu_tests.log 2>&1
tail -8 $OUT/gpu_tests.log
bash tools/pmc_forest.sh r3a > $OUT/pmc.log 2>&1
tail -150 $OUT/pmc.log
