#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
OUT=$REPO/gpurun_out/r4_e
mkdir -p $OUT
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_baseline_configs.py tests/test_envelope.py tests/test_overflow_protocol.py -m gpu -q -x > $OUT/parity.log 2>&1
tail -4 $OUT/parity.log
timeout 600 python tools/ab_forest.py col:0 nocol:256 2>&1 | tee $OUT/ab_forest.log | cut -c1-400
timeout 600 python tools/ab_kernel.py 50 4 col:0 nocol:256 2>&1 | tail -2 | cut -c1-400
