#!/bin/bash
# tools/profile.sh <tag>  -- GPU box only.  rocprofv3 kernel-trace stats + PMC passes (FETCH_SIZE, WRITE_SIZE, SQ_INSTS_VALU) of
# the bench command; raw output under gpurun_out/prof_<tag>/, summaries via tools/summarize_profile.py (copy the ones to be
# judged into profiles/: summary_<tag>.txt/.json, traffic.json, valu.json).
TAG=${1:-r01}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
SKIP="--no-cpu-baseline --no-workload2 --no-c4 --no-c5 --no-api --no-io"
BENCH="python3 $REPO/bench.py --steps 20 --warmup 2 --windows 2 --min-timed-s 0 $SKIP"
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- $BENCH > $OUT/trace_bench.log 2>&1
echo "trace rc=$?" >> $OUT/trace_bench.log
# the counter passes run tools/prof_pipeline.py: the same kernels on the same workloads (50 C2 views per pix2face launch, 64 C3
# views per fused launch) without torch's elementwise kernels and without the library's side stream -- rocprofv3's counter
# collection segfaults on either
SHORT="python3 $REPO/tools/prof_pipeline.py 50 3 64"
timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o fetch -- $SHORT > $OUT/pmc_fetch.log 2>&1
echo "fetch rc=$?" >> $OUT/pmc_fetch.log
timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o write -- $SHORT > $OUT/pmc_write.log 2>&1
echo "write rc=$?" >> $OUT/pmc_write.log
timeout 400 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $OUT/pmc_sq -o sq -- $SHORT > $OUT/pmc_sq.log 2>&1
echo "sq rc=$?" >> $OUT/pmc_sq.log
cd $REPO && python3 tools/summarize_profile.py $OUT $TAG
