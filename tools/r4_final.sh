#!/bin/bash
# final evidence of round 4: GPU tests, bench line, rocprofv3 trace + PMC passes (summary_r04.txt, traffic.json, valu.json), SQ
# counters of the tile kernel on config 5, phase stamps
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
OUT=$REPO/gpurun_out/r4_final
mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q > $OUT/gpu_tests.log 2>&1
tail -3 $OUT/gpu_tests.log
( time timeout 900 python bench.py ) > $OUT/bench.json 2> $OUT/bench.err
tail -4 $OUT/bench.err
python -c "
import json
d=json.loads(open('$OUT/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['roofline']['frac'], d['roofline']['kernel_ms_per_launch'], d['aggregate']['views_per_s'], d['c5']['raster_mpix_per_s'], d['c5']['raster_kernel_frac_of_hbm_peak'], d['workload_2']['scale_1']['mpix_per_s'], d['workload_2']['scale_0.25']['mpix_per_s'])
print({k: v for k, v in d['api'].items() if 'photo' in k or 'resize' in k})
"
bash tools/profile.sh r04 > $OUT/profile.log 2>&1
grep -E "k_raster_tile|k_setup_cull|k_vote" gpurun_out/prof_r04/summary_r04.txt | cut -c1-170
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $REPO/gpurun_out/prof_r04/pmc_sq_c5 -o sq -- python3 $REPO/tools/prof_c5.py 20 3 > $OUT/pmc_c5.log 2>&1
echo "c5 sq rc=$?"
cd $REPO
for w in c2 c5; do timeout 300 python tools/tile_phases.py $w > $OUT/phases_$w.log 2>&1; tail -1 $OUT/phases_$w.log | cut -c1-400; done
# the bench line again inside the profiled command's shape is in prof_r04/trace_bench.log
tail -2 gpurun_out/prof_r04/trace_bench.log | cut -c1-300
