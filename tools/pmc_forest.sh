#!/bin/bash
# tools/pmc_forest.sh <tag> -- SQ / LDS counter passes over the hostile workload (tools/ab_forest.py base:0: terrain + 20 000
# trees, 20 oblique views at 4000x3000 and 1000x750) for the set-up and the tile kernel (GPU box only)
TAG=${1:-forest}; SPEC=${2:-base:0}; PASSES=${3:-4}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU"
P2="SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU"
P3="SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_THREAD_CYCLES_VALU SQ_INSTS_FLAT SQ_INSTS_GDS"
P4="TCP_TOTAL_ATOMIC_WITH_RET_sum TCP_TOTAL_ATOMIC_WITHOUT_RET_sum TCC_ATOMIC_sum TCC_REQ_sum"
i=0
for P in "$P1" "$P2" "$P3" "$P4"; do
  i=$((i+1))
  [ $i -gt $PASSES ] && break
  timeout 300 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $OUT/p$i -o p$i -- python3 $REPO/tools/ab_forest.py $SPEC > $OUT/p$i.log 2>&1
  echo "pass $i rc=$?" >> $OUT/p$i.log
done
cd $REPO && python3 - <<PY
import csv, glob, collections
out = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
        # two image sizes per run: keep them apart by grid size
        k = f"{k} grid={r.get('Grid_Size', '?')}"
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
