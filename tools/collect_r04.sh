#!/bin/bash
# copy the round-4 evidence out of gpurun_out/ (scratch) into profiles/ (tracked)
cd "$(dirname "$0")/.."
P=gpurun_out/prof_r04; F=gpurun_out/r4_final
cp $P/summary_r04.txt $P/summary_r04.json $P/traffic.json $P/valu.json profiles/
cp $P/trace/trace_kernel_stats.csv profiles/r04_kernel_stats.csv
tail -1 $F/bench.json > profiles/r04_bench.json
grep '^{' $P/trace_bench.log | tail -1 > profiles/r04_bench_under_rocprof.json
tail -4 $F/gpu_tests.log > profiles/r04_gpu_tests.log
cp $F/phases_c2.log profiles/r04_ab/tile_phases_c2_final.log; cp $F/phases_c5.log profiles/r04_ab/tile_phases_c5_final.log
python3 - <<'PY'
import csv, glob, collections
out = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/prof_r04/pmc_sq_c5/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
        out[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open("profiles/r04_sq_counters_c5.txt", "w") as fo:
    fo.write("# rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT --kernel-trace -- python3 tools/prof_c5.py 20 3\n")
    fo.write("# BASELINE config 5: 4 999 122 faces, 20 views 6000x4000 per launch; averages per launch\n")
    for k, cs in sorted(out.items()):
        if not k.startswith("k_"):
            continue
        a = {c: sum(v) / len(v) for c, v in cs.items()}
        cyc = a.get("SQ_BUSY_CYCLES", 0) / 32.0
        fo.write(f"{k}\n")
        for c, v in sorted(a.items()):
            fo.write(f"    {c:24s} n={len(cs[c]):3d} avg={v:16.1f}\n")
        if cyc and "SQ_ACTIVE_INST_VALU" in a:
            fo.write(f"    -> cycles per launch {cyc:.0f}; VALU busy {a['SQ_ACTIVE_INST_VALU'] * 4 / (1024 * cyc):.3f}; "
                     f"LDS busy {a.get('SQ_LDS_IDX_ACTIVE', 0) / (256 * cyc):.3f}; bank conflicts / LDS cycles "
                     f"{a.get('SQ_LDS_BANK_CONFLICT', 0) / max(a.get('SQ_LDS_IDX_ACTIVE', 1), 1):.3f}\n")
print(open("profiles/r04_sq_counters_c5.txt").read()[:1500])
PY
