#!/bin/bash
# tools/ab_builds.sh <rounds> <lib.so> [lib.so ...] -- alternate builds of libgeograster.so on ONE box (box-to-box spread is
# larger than most kernel changes): each round copies every build into place in turn and runs tools/ab_kernel.py once.
N=$1; shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
for i in $(seq $N); do
  for L in "$@"; do
    cp $L geograypher_amd/csrc/libgeograster.so
    echo -n "$(basename $L) "
    timeout 300 python tools/ab_kernel.py 50 5 x:0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('setup', d['plain']['setup_ms'], 'plain', d['plain']['raster_ms'], 'fused', d['fused']['raster_ms'], 'vote', d['fused']['vote_ms'])"
  done
done
