#!/bin/bash
# round-3 GPU run 1: atomic microbenchmark, parity of the new set-up kernel, forest + C2 stage times
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
OUT=$REPO/gpurun_out/r3_run1
mkdir -p $OUT
hipcc --offload-arch=gfx950 -O3 -o /tmp/atomic_rate tools/ubench/atomic_rate.hip && timeout 120 /tmp/atomic_rate > $OUT/atomic_rate.txt 2>&1
timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -x -q > $OUT/parity.log 2>&1
tail -5 $OUT/parity.log
timeout 600 python tools/ab_forest.py base:0 biglist:64 > $OUT/forest.log 2>&1
timeout 600 python tools/ab_kernel.py 50 5 base:0 biglist:64 > $OUT/c2.log 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/forest_trace -o t -- python3 $REPO/tools/ab_forest.py base:0 > $OUT/forest_trace.log 2>&1
cd $REPO
python3 - <<PY
import csv, glob
for f in glob.glob("$OUT/forest_trace/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    with open("$OUT/forest_kernel_stats.txt", "w") as fo:
        for r in rows[:14]:
            fo.write(f"{r['Name'][:90]:90s} calls {r['Calls']:>6s} total_ns {r['TotalDurationNs']:>12s} avg_ns {r['AverageNs']:>12s}\n")
PY
cat $OUT/atomic_rate.txt $OUT/forest.log $OUT/c2.log $OUT/forest_kernel_stats.txt
