#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
OUT=$REPO/gpurun_out/r3_run6
mkdir -p $OUT
timeout 1200 python -m pytest tests/test_api_pipelines.py tests/test_next_rows.py tests/test_hip_parity.py::test_end_to_end_api_matches_oracle_pipeline tests/test_integration_stub.py tests/test_envelope.py -m gpu -x -q > $OUT/tests.log 2>&1
tail -15 $OUT/tests.log
( time timeout 900 python bench.py --no-c4 --no-c5 --no-workload2 --no-cpu-baseline --no-aggregate ) > $OUT/bench.json 2> $OUT/bench.err
tail -3 $OUT/bench.err
python -c "
import json
d=json.loads(open('$OUT/bench.json').read().strip().splitlines()[-1])
print(json.dumps(d['api'],indent=1))
"
