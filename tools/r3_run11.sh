#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
OUT=$REPO/gpurun_out/r3_run11
mkdir -p $OUT
timeout 1500 python -m pytest tests/test_hip_parity.py tests/test_overflow_protocol.py tests/test_baseline_configs.py tests/test_envelope.py -m gpu -x -q > $OUT/parity.log 2>&1
tail -5 $OUT/parity.log
bash tools/ab_builds.sh 3 gpurun_tmp/lib_old.so gpurun_tmp/lib_new.so 2>&1 | tee $OUT/ab.log
for L in old new; do cp gpurun_tmp/lib_$L.so geograypher_amd/csrc/libgeograster.so; echo $L; timeout 600 python tools/ab_forest.py base:0 2>/dev/null | cut -c1-200; done | tee $OUT/forest.log
