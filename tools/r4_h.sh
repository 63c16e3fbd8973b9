#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
OUT=$REPO/gpurun_out/r4_h
mkdir -p $OUT
# 2 x 2: key reads (b64 / the original dword pairs) x column-list code (in / out of the kernel), 6 rounds on one box
timeout 2000 python tools/ab_libs.py 6 c2 base nocol colold nocolold 2>&1 | tee $OUT/ab_2x2.log | tail -6
for w in c2 c5; do timeout 300 python tools/tile_phases.py $w > $OUT/phases_$w.log 2>&1; tail -1 $OUT/phases_$w.log | cut -c1-1700; done
