#!/bin/bash
# final evidence of round 6: GPU tests, bench line, rocprofv3 trace + PMC passes (summary_r06.txt, traffic.json, valu.json),
# kernel-trace statistics of config 5 and of the hostile forest (absent in round 4), phase stamps of both kernels
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
OUT=$REPO/gpurun_out/r6_final
mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q --durations=15 > $OUT/gpu_tests.log 2>&1
tail -3 $OUT/gpu_tests.log
bash tools/profile.sh r06 > $OUT/profile.log 2>&1
grep -E "k_raster_tile|k_setup_cull|k_vote|k_cull" gpurun_out/prof_r06/summary_r06.txt | cut -c1-170
# the counter files of THIS tree in place before the bench line is taken, so that the line carries `traffic` / `valu`
cp gpurun_out/prof_r06/traffic.json gpurun_out/prof_r06/valu.json profiles/
( time timeout 900 python bench.py ) > $OUT/bench.json 2> $OUT/bench.err
tail -4 $OUT/bench.err
python -c "
import json
d=json.loads(open('$OUT/bench.json').read().strip().splitlines()[-1])
r=d['roofline']
print(d['value'], d['ms_per_step'], r['frac'], r['kernel_ms_per_launch'], r['stage_ms_per_view'])
print({k: r.get(k) for k in ('binding_resource','valu_frac','c5_kernel_frac','c3_views_per_s','hostile_gpix_scale_1','hostile_gpix_scale_0.25','c2_quarter_scale_gpix','hostile_overflow_retries_cold')})
print(d['quarter_scale'])
print({k: v for k, v in d['io']['aggregate_from_label_png_files'].items() if 'views_per_s' in k})
print({k: v for k, v in d['api'].items() if 'photo' in k})
"
cd /tmp && export TMPDIR=/tmp
# kernel-trace statistics of config 5 (ids kernel, 20 views per launch) and of the hostile forest (20 oblique views, both scales)
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/prof_r06/trace_c5 -o t -- python3 $REPO/tools/prof_c5.py 20 6 > $OUT/trace_c5.log 2>&1
echo "c5 trace rc=$?"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/prof_r06/trace_forest -o t -- python3 $REPO/tools/prof_forest.py 6 > $OUT/trace_forest.log 2>&1
echo "forest trace rc=$?"
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $REPO/gpurun_out/prof_r06/pmc_sq_c5 -o sq -- python3 $REPO/tools/prof_c5.py 20 3 > $OUT/pmc_c5.log 2>&1
echo "c5 sq rc=$?"
cd $REPO
python3 - <<'PY'
import csv, glob, os
repo = os.environ.get("GRAFT_REPO_ROOT", os.getcwd())
with open(f"{repo}/gpurun_out/prof_r06/summary_r06_c5_forest.txt", "w") as out:
    for tag, title in (("trace_c5", "config 5: 20 views 6000x4000 per launch (tools/prof_c5.py 20 6)"),
                       ("trace_forest", "hostile forest: 20 oblique views per launch at 4000x3000 and at 1000x750 (tools/prof_forest.py 6)")):
        f = (glob.glob(f"{repo}/gpurun_out/prof_r06/{tag}/**/*kernel_stats.csv", recursive=True) or [None])[0]
        out.write(f"# rocprofv3 --kernel-trace --stats -- {title}\n")
        if not f:
            out.write("  (no trace)\n"); continue
        out.write(f"{'kernel':76s} {'calls':>6s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s}\n")
        for r in csv.DictReader(open(f)):
            n = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")
            if n.startswith("k_"):
                out.write(f"{n[:76]:76s} {r['Calls']:>6s} {float(r['AverageNs'])/1e3:10.2f} {float(r['MinNs'])/1e3:10.2f} {float(r['MaxNs'])/1e3:10.2f}\n")
        out.write("\n")
print(open(f"{repo}/gpurun_out/prof_r06/summary_r06_c5_forest.txt").read())
PY
for w in c2 c5 c2q; do timeout 300 python tools/tile_phases.py $w > $OUT/phases_$w.log 2>&1; tail -1 $OUT/phases_$w.log | cut -c1-700; done
for w in c2 c2q forest; do timeout 300 python tools/setup_phases.py $w 50 > $OUT/setup_phases_$w.log 2>&1; tail -1 $OUT/setup_phases_$w.log | cut -c1-900; done
tail -2 gpurun_out/prof_r06/trace_bench.log | cut -c1-300
