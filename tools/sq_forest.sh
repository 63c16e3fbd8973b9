#!/bin/bash
# SQ counters of the set-up and tile kernels on the hostile forest (20 oblique views per launch, both scales): tools/prof_forest.py
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/sq_forest
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $OUT/pmc -o sq -- python3 $REPO/tools/prof_forest.py 3 > $OUT/pmc.log 2>&1
echo "rc=$?"
cd $REPO
python3 - <<'PY'
import csv, glob, os, collections
repo = os.environ.get("GRAFT_REPO_ROOT", os.getcwd())
f = (glob.glob(f"{repo}/gpurun_out/sq_forest/pmc/**/*counter_collection.csv", recursive=True) or [None])[0]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    if n.startswith("k_setup_cull") or n.startswith("k_raster_tile"):
        acc[n[:60] + " grid " + r.get("Grid_Size", "?")][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(f"{repo}/gpurun_out/sq_forest/summary.txt", "w") as out:
    for k in sorted(acc):
        out.write(k + "\n")
        for c in sorted(acc[k]):
            v = acc[k][c]
            out.write(f"    {c:28s} n={len(v):3d} avg={sum(v)/len(v):16.1f}\n")
print(open(f"{repo}/gpurun_out/sq_forest/summary.txt").read())
PY
