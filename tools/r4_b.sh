#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
OUT=$REPO/gpurun_out/r4_b
mkdir -p $OUT
for w in c2 c5; do timeout 300 python tools/tile_phases.py $w > $OUT/phases_$w.log 2>&1; tail -1 $OUT/phases_$w.log | cut -c1-1500; done
timeout 300 python tools/tile_phases.py c2 50 fused > $OUT/phases_c2_fused.log 2>&1; tail -1 $OUT/phases_c2_fused.log | cut -c1-1500
timeout 1500 python tools/ab_libs.py 3 c2,c5 base e1:1 2>&1 | tee $OUT/ab_e1.log | tail -12
