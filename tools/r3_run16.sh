#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
bash tools/ab_builds.sh 2 gpurun_tmp/lib_old.so gpurun_tmp/lib_new.so gpurun_tmp/lib_c21.so gpurun_tmp/lib_c32.so gpurun_tmp/lib_c41.so 2>&1
for L in old new c21 c32 c41; do cp gpurun_tmp/lib_$L.so geograypher_amd/csrc/libgeograster.so; echo $L; timeout 600 python tools/ab_forest.py base:0 2>/dev/null | cut -c1-130; done
