#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
bash tools/ab_builds.sh 3 "$@" 2>&1
L=$2
cp $L geograypher_amd/csrc/libgeograster.so
timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -x -q 2>&1 | tail -2
for L in "$@"; do cp $L geograypher_amd/csrc/libgeograster.so; echo $L; timeout 600 python tools/ab_forest.py base:0 2>/dev/null | cut -c1-150; done
