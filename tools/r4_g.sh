#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
OUT=$REPO/gpurun_out/r4_g
mkdir -p $OUT
timeout 1500 python tools/ab_libs.py 4 c2,c5 base rot old 2>&1 | tee $OUT/ab_plain_vs_rot.log | tail -8
timeout 600 python -m pytest tests/test_hip_parity.py tests/test_photo_resize.py tests/test_api_pipelines.py -m gpu -q -x 2>&1 | tail -3
