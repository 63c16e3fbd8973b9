#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
OUT=$REPO/gpurun_out/r3_run26
mkdir -p $OUT
timeout 2400 python -m pytest tests -m gpu -q -x > $OUT/gpu_tests.log 2>&1
tail -3 $OUT/gpu_tests.log
( time timeout 900 python bench.py ) > $OUT/bench.json 2> $OUT/bench.err
tail -3 $OUT/bench.err
python -c "
import json
d=json.loads(open('$OUT/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['roofline']['frac'], d['roofline']['kernel_ms_per_launch'], d['aggregate']['views_per_s'], d['c5']['raster_mpix_per_s'], d['workload_2']['scale_1']['mpix_per_s'], d['workload_2']['scale_0.25']['mpix_per_s'])
print(json.dumps(d['workload_2'])[:1500])
"
