#!/bin/bash
# tools/pmc_fetch_calib.sh -- FETCH_SIZE / WRITE_SIZE of tools/ubench/fetch_calib (known byte counts per access shape).  GPU box only.
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/fetch_calib
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -o f -- $REPO/tools/ubench/fetch_calib > $OUT/fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -o w -- $REPO/tools/ubench/fetch_calib > $OUT/write.log 2>&1
cd $REPO && python3 - <<'PY'
import csv, glob, collections, os
repo = os.environ.get("GRAFT_REPO_ROOT", os.getcwd())
out = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"{repo}/gpurun_out/fetch_calib/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        out[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
GiB = 1 << 30
known = {"k_read16": ("read", GiB), "k_read4": ("read", GiB), "k_gather1": ("read 1 B gathers", GiB // 52), "k_write4": ("write", GiB // 2)}
lines = ["# rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (KiB) -- tools/ubench/fetch_calib: known bytes per access shape, 1 GiB buffers, cold",
         f"{'kernel':12s} {'known bytes':>14s} {'FETCH_SIZE B':>14s} {'ratio':>7s} {'WRITE_SIZE B':>14s} {'ratio':>7s}"]
for k, (what, nbytes) in known.items():
    c = out.get(k, {})
    fe = sum(c.get("FETCH_SIZE", [0])) / max(len(c.get("FETCH_SIZE", [1])), 1) * 1024
    wr = sum(c.get("WRITE_SIZE", [0])) / max(len(c.get("WRITE_SIZE", [1])), 1) * 1024
    lines.append(f"{k:12s} {nbytes:14d} {fe:14.0f} {fe / nbytes:7.3f} {wr:14.0f} {wr / nbytes:7.3f}   ({what})")
open(f"{repo}/gpurun_out/fetch_calib/summary.txt", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
