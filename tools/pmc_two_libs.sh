#!/bin/bash
# tools/pmc_two_libs.sh <lib.so> <lib.so> ... -- SQ counters of the C2 pipeline (tools/prof_pipeline.py) for each build, one box
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/pmc_libs
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
P1="SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS"
for L in "$@"; do
  N=$(basename $L .so)
  cp $REPO/$L $REPO/geograypher_amd/csrc/libgeograster.so
  timeout 300 rocprofv3 --pmc $P1 --kernel-trace --output-format csv -d $OUT/$N -o p1 -- python3 $REPO/tools/prof_pipeline.py 50 2 0 > $OUT/$N.log 2>&1
done
cd $REPO && python3 - "$@" <<'PY'
import csv, glob, collections, sys, os
for L in sys.argv[1:]:
    N = os.path.basename(L)[:-3]
    out = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f"gpurun_out/pmc_libs/{N}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
            if "k_raster_tile" in k or "k_setup_cull" in k:
                out[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(N)
    for k, cs in sorted(out.items()):
        print("  ", k, {c: round(sum(v) / len(v) / 1e6, 2) for c, v in sorted(cs.items())})
PY
