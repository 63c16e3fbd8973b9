#!/bin/bash
# copy the round-6 evidence out of gpurun_out/ (scratch) into profiles/ (tracked)
cd "$(dirname "$0")/.."
P=gpurun_out/prof_r06; F=gpurun_out/r6_final
cat $P/summary_r06.txt $P/summary_r06_c5_forest.txt > profiles/summary_r06.txt
cp $P/summary_r06.json $P/traffic.json $P/valu.json profiles/
cp $P/trace/trace_kernel_stats.csv profiles/r06_kernel_stats.csv
grep '^{' $F/bench.json | tail -1 > profiles/r06_bench.json
grep '^{' $P/trace_bench.log | tail -1 > profiles/r06_bench_under_rocprof.json
tail -24 $F/gpu_tests.log > profiles/r06_gpu_tests.log
for w in c2 c5 c2q; do tail -1 $F/phases_$w.log > profiles/r06_ab/tile_phases_$w.log; done
for w in c2 c2q forest; do tail -1 $F/setup_phases_$w.log > profiles/r06_ab/setup_phases_$w.log; done
python3 - <<'PY'
import csv, glob, collections
out = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/prof_r06/pmc_sq_c5/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
        out[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open("profiles/r06_sq_counters_c5.txt", "w") as fo:
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
print(open("profiles/r06_sq_counters_c5.txt").read()[:1500])
PY
