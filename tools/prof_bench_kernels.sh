REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/cull_prof -o t -- python3 $REPO/bench.py --no-cpu-baseline --no-api --no-io --no-c4 --no-c5 > $REPO/gpurun_out/cull_prof.log 2>&1
cd $REPO
python3 - <<'PY'
import csv, glob, os, json
repo = os.environ.get("GRAFT_REPO_ROOT", os.getcwd())
f = glob.glob(f"{repo}/gpurun_out/cull_prof/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    n = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    if n.startswith("k_"): print(f"{n[:60]:60s} {r['Calls']:>6s} avg {float(r['AverageNs'])/1e3:9.2f} min {float(r['MinNs'])/1e3:9.2f} max {float(r['MaxNs'])/1e3:9.2f}")
j = json.loads([l for l in open(f"{repo}/gpurun_out/cull_prof.log") if l.startswith("{")][-1])
print(j["value"], j["ms_per_step"], j["aggregate"]["views_per_s"])
PY
