#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
bash tools/ab_builds.sh 3 gpurun_tmp/lib_base.so gpurun_tmp/lib_sprio.so gpurun_tmp/lib_w4.so gpurun_tmp/lib_w6.so 2>&1
cp gpurun_tmp/lib_base.so geograypher_amd/csrc/libgeograster.so
python tools/ab_kernel.py 50 3 g128:0 g64:256 g256:512 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['variant'], d['plain']['setup_ms'], d['fused']['setup_ms'])"
for L in base sprio w4 w6; do cp gpurun_tmp/lib_$L.so geograypher_amd/csrc/libgeograster.so; echo $L; timeout 600 python tools/ab_forest.py base:0 2>/dev/null | cut -c1-110; done
