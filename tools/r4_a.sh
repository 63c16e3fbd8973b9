#!/bin/bash
# round 4, first GPU call: GPU tests, bench line, tile-kernel ablations on C2 and C5
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
OUT=$REPO/gpurun_out/r4_a
mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q -x > $OUT/gpu_tests.log 2>&1
tail -5 $OUT/gpu_tests.log
( time timeout 900 python bench.py ) > $OUT/bench.json 2> $OUT/bench.err
tail -3 $OUT/bench.err
python -c "
import json
d=json.loads(open('$OUT/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['roofline']['frac'], d['roofline']['kernel_ms_per_launch'], d['aggregate']['views_per_s'], d['c5']['raster_mpix_per_s'], d['c5']['raster_kernel_frac_of_hbm_peak'], d['workload_2']['scale_1']['mpix_per_s'], d['workload_2']['scale_0.25']['mpix_per_s'])
print(d['api'])
"
# ablations: full, no scanline loop (1), no stores/epilogue (2), no triangles (4), 1+2
timeout 600 python tools/ab_kernel.py 50 5 base:0 noitems:0:1 nostore:0:2 notri:0:4 skel:0:3 > $OUT/ab_c2.log 2>&1
cat $OUT/ab_c2.log | cut -c1-400
AB_WORKLOAD=c5 timeout 600 python tools/ab_kernel.py 20 4 base:0 noitems:0:1 nostore:0:2 notri:0:4 skel:0:3 > $OUT/ab_c5.log 2>&1
cat $OUT/ab_c5.log | cut -c1-400
for w in c2 c5; do timeout 300 python tools/tile_phases.py $w > $OUT/phases_$w.log 2>&1; tail -1 $OUT/phases_$w.log | cut -c1-1500; done
timeout 300 python tools/tile_phases.py c2 50 fused > $OUT/phases_c2_fused.log 2>&1; tail -1 $OUT/phases_c2_fused.log | cut -c1-1500
