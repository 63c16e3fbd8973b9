#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
OUT=$REPO/gpurun_out/r3_run8
mkdir -p $OUT
( time timeout 900 python bench.py ) > $OUT/bench.json 2> $OUT/bench.err
tail -3 $OUT/bench.err
timeout 2400 python -m pytest tests -m gpu -q > $OUT/gpu_tests.log 2>&1
tail -4 $OUT/gpu_tests.log
python -c "
import json
d=json.loads(open('$OUT/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['timed_gpu_s'])
print(json.dumps(d['roofline']['valu'])); 
for k,v in d['rooflines'].items(): print(k, v['frac'], v['traffic'], v['valu'] and v['valu']['frac'])
print(json.dumps(d['api'],indent=1)); print(json.dumps(d['workload_2'])[:900])
"
