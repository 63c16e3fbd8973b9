#!/bin/bash
# tools/prof_libs.sh <workload> <tag> [<tag> ...] -- rocprofv3 kernel-trace statistics of tools/ab_kernel.py for library builds
# (csrc/libgeograster_<tag>.so; "base" = the product).  GPU box only.  Prints the set-up and tile kernels' average durations.
W=$1; shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for TAG in "$@"; do
  if [ "$TAG" = base ]; then LIB=$REPO/geograypher_amd/csrc/libgeograster.so; else LIB=$REPO/geograypher_amd/csrc/libgeograster_$TAG.so; fi
  export GEOGRAYPHER_AMD_LIB=$LIB AB_WORKLOAD=$W
  OUT=$REPO/gpurun_out/prof_libs_${W}_$TAG
  rm -rf $OUT; mkdir -p $OUT
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o t -- python3 $REPO/tools/ab_kernel.py 50 5 x:0 > $OUT/run.log 2>&1
  F=$(find $OUT -name "*kernel_stats.csv" | head -1)
  echo "== $W $TAG"
  python3 - "$F" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    n = r["Name"]
    if any(k in n for k in ("k_setup_cull", "k_cull_blocks", "k_clip_faces", "k_bin_stats", "k_bin_init", "k_vote", "k_raster_tile")):
        print(f"  {n[:70]:70s} calls {r['Calls']:>5s} avg_us {float(r['AverageNs'])/1e3:9.2f} min_us {float(r['MinNs'])/1e3:9.2f} max_us {float(r['MaxNs'])/1e3:9.2f}")
PY
done
