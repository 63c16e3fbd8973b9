#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
OUT=$REPO/gpurun_out/r4_c
mkdir -p $OUT
# correctness of the experimental build first: the parity suite with the e3 library in place of the product's
GEOGRAYPHER_AMD_LIB=$REPO/geograypher_amd/csrc/libgeograster_e3.so timeout 900 python -m pytest tests/test_hip_parity.py tests/test_baseline_configs.py tests/test_overflow_protocol.py -m gpu -q -x > $OUT/parity_e3.log 2>&1
tail -3 $OUT/parity_e3.log
timeout 1500 python tools/ab_libs.py 3 c2,c5 base e1:1 e3:3 2>&1 | tee $OUT/ab_e3.log | tail -10
