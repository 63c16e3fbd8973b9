#!/bin/bash
# tools/pmc_mem.sh <tag> -- memory-side counter passes (TA / TCP / TCC) over tools/prof_raster.py (GPU box only).
# Every pass runs under `timeout`: a pass with the TCP latency counters or GRBM_GUI_ACTIVE hung the profiler for 20 minutes.
TAG=${1:-mem}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/pmcmem_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
P1="TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_ATOMIC_WITH_RET_sum TCP_TOTAL_ATOMIC_WITHOUT_RET_sum"
P2="TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TOTAL_ACCESSES_sum"
P3="TCC_BUSY_sum TCC_REQ_sum TCC_ATOMIC_sum TCC_TAG_STALL_sum TCC_EA0_WRREQ_STALL_sum TCC_WRITE_sum TCC_READ_sum TCC_CYCLE_sum"
i=0
for P in "$P1" "$P2" "$P3"; do
  i=$((i+1))
  timeout 150 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $OUT/p$i -o p$i -- python3 $REPO/tools/prof_raster.py 0 5 32 2 > $OUT/p$i.log 2>&1
  echo "pass $i rc=$?" >> $OUT/p$i.log
done
cd $REPO && python3 - <<PY
import csv, glob, collections
out = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
        out[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open("$OUT/summary.txt", "w") as fo:
    for k, cs in sorted(out.items()):
        if not (k.startswith("k_setup") or k.startswith("k_raster") or k.startswith("k_cull")):
            continue
        fo.write(f"{k}\n")
        for c, v in sorted(cs.items()):
            fo.write(f"    {c:40s} n={len(v):3d} avg={sum(v)/len(v):16.1f}\n")
print(open("$OUT/summary.txt").read())
PY
