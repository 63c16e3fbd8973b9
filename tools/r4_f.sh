#!/bin/bash
# round 4: bench line on the current tree + rocprofv3 trace / PMC passes (profiles/summary_r04.txt, traffic.json, valu.json)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
OUT=$REPO/gpurun_out/r4_f
mkdir -p $OUT
( time timeout 900 python bench.py ) > $OUT/bench.json 2> $OUT/bench.err
tail -4 $OUT/bench.err
python -c "
import json
d=json.loads(open('$OUT/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['roofline']['frac'], d['roofline']['kernel_ms_per_launch'], d['aggregate']['views_per_s'], d['c5']['raster_mpix_per_s'], d['c5']['raster_kernel_frac_of_hbm_peak'], d['workload_2']['scale_1']['mpix_per_s'], d['workload_2']['scale_0.25']['mpix_per_s'])
print({k: v for k, v in d['workload_2']['scale_1'].items() if 'retries' in k})
print(d['api'])
print(d['io'])
print(d['rooflines']['k_vote_labels'])
print({k: d['roofline'][k] for k in d['roofline'] if 'culled' in k or 'traffic' in k})
"
bash tools/profile.sh r04 > $OUT/profile.log 2>&1
tail -30 $OUT/profile.log | cut -c1-200
# SQ counters of the tile kernel on config 5 (20 views 6000x4000)
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $REPO/gpurun_out/prof_r04/pmc_sq_c5 -o sq -- python3 $REPO/tools/prof_c5.py 20 3 > $OUT/pmc_c5.log 2>&1
echo "c5 sq rc=$?"
