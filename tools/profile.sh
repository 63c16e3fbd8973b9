#!/bin/bash
# tools/profile.sh <tag>  -- GPU box only.  rocprofv3 kernel-trace stats + two PMC passes (FETCH_SIZE, WRITE_SIZE) of
# the bench command; raw output under gpurun_out/prof_<tag>/, summaries via tools/summarize_profile.py.
TAG=${1:-r01}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $REPO/bench.py --steps 20 --warmup 2 --windows 2 --no-cpu-baseline --no-workload2 --no-c4"
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- $BENCH > $OUT/trace_bench.log 2>&1
echo "trace rc=$?" >> $OUT/trace_bench.log
SHORT="python3 $REPO/bench.py --steps 2 --warmup 1 --windows 1 --no-cpu-baseline --no-aggregate --no-workload2 --no-c4"
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o fetch -- $SHORT > $OUT/pmc_fetch.log 2>&1
echo "fetch rc=$?" >> $OUT/pmc_fetch.log
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o write -- $SHORT > $OUT/pmc_write.log 2>&1
echo "write rc=$?" >> $OUT/pmc_write.log
cd $REPO && python3 tools/summarize_profile.py $OUT $TAG
