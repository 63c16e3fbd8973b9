#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
OUT=$REPO/gpurun_out/r3_run4
mkdir -p $OUT
bash tools/pmc_forest.sh r3rows2 base:0 2 > $OUT/rows2.log 2>&1
bash tools/pmc_forest.sh r3rows1 rows1:128 2 > $OUT/rows1.log 2>&1
for t in rows2 rows1; do echo "== $t"; grep -A30 "k_raster_tile<6, 5, 256, false, 1, 5, .> grid=30320640" gpurun_out/pmc_r3$t/summary.txt | grep -E "k_raster|SQ_INSTS_VALU|SQ_INSTS_SALU|SQ_INSTS_LDS|SQ_BUSY_CYCLES|SQ_ACTIVE_INST_VALU|SQ_LDS_BANK|SQ_LDS_IDX|SQ_WAIT_INST_LDS|SQ_WAIT_ANY |SQ_WAVE_CYCLES"; done
