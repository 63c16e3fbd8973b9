#!/bin/bash
# tools/pmc_entries.sh -- SQ and HBM counters of the C2 pipeline with 40-byte (variant 0) and 48-byte (variant 128) entries
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/pmc_entries
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
P1="SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS"
P2="SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_LDS_ADDR_CONFLICT"
P3="FETCH_SIZE"
P4="WRITE_SIZE"
for V in 0 128; do
  i=0
  for P in "$P1" "$P2" "$P3" "$P4"; do
    i=$((i+1))
    timeout 300 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $OUT/v$V/p$i -o p$i -- python3 $REPO/tools/prof_pipeline.py 50 2 0 $V > $OUT/v${V}_p$i.log 2>&1
    echo "pass $i rc=$?" >> $OUT/v${V}_p$i.log
  done
done
cd $REPO && python3 - <<PY
import csv, glob, collections
for V in (0, 128):
    out = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f"$OUT/v{V}/p*/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
            if "k_raster_tile" in k or "k_setup_cull" in k:
                out[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    with open(f"$OUT/summary_v{V}.txt", "w") as fo:
        for k, cs in sorted(out.items()):
            fo.write(f"{k}\n")
            for c, v in sorted(cs.items()):
                fo.write(f"    {c:28s} n={len(v):3d} avg={sum(v)/len(v):16.1f}\n")
    print("variant", V); print(open(f"$OUT/summary_v{V}.txt").read())
PY
