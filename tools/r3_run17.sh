#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/pmc_c5
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
P1="SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU"
timeout 300 rocprofv3 --pmc $P1 --kernel-trace --output-format csv -d $OUT/p1 -o p1 -- python3 $REPO/tools/prof_c5.py 20 3 > $OUT/p1.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/p2 -o p2 -- python3 $REPO/tools/prof_c5.py 20 3 > $OUT/p2.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/p3 -o p3 -- python3 $REPO/tools/prof_c5.py 20 3 > $OUT/p3.log 2>&1
cd $REPO && python3 - <<PY
import csv, glob, collections
out = collections.defaultdict(lambda: collections.defaultdict(list)); dur = collections.defaultdict(list)
for f in glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
        out[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in glob.glob("$OUT/p1/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
        dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, cs in sorted(out.items()):
    if not k.startswith("k_"): continue
    print(k, "avg_us", round(sum(dur[k]) / max(len(dur[k]), 1), 1))
    for c, v in sorted(cs.items()): print(f"    {c:26s} n={len(v):3d} avg={sum(v)/len(v):16.1f}")
PY
