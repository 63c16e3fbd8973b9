#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
OUT=$REPO/gpurun_out/r3_run5
mkdir -p $OUT
( time timeout 900 python bench.py ) > $OUT/bench.json 2> $OUT/bench.err
tail -3 $OUT/bench.err
python -c "
import json
d=json.loads(open('$OUT/bench.json').read().strip().splitlines()[-1])
for k in ('value','ms_per_step','timed_gpu_s','ms_per_step_windows'): print(k, d[k])
print(json.dumps(d['roofline'],indent=1)); print(json.dumps(d['rooflines'],indent=1))
for k in ('aggregate','c4','c5','workload_2','api'): print(k, json.dumps(d[k],indent=1))
print(json.dumps(d['cpu_baseline'])[:600])
"
timeout 2400 python -m pytest tests -m gpu -x -q > $OUT/gpu_tests.log 2>&1
tail -5 $OUT/gpu_tests.log
