#!/usr/bin/env python3
"""tools/summarize_profile.py <raw_dir> <tag> -- reduce rocprofv3 CSV output to small text/JSON summaries
(gpurun_out/prof_<tag>/summary_*.{txt,json}); copy the ones to be judged into profiles/."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def kernel_sha():
    """sha256 of the library sources the counters were measured on (bench.kernel_source_sha256): bench.py drops the figures of
    traffic.json / valu.json from its line as soon as the tree holds other kernels."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench

    return bench.kernel_source_sha256()


def find(pattern):
    hits = glob.glob(pattern, recursive=True)
    return hits[0] if hits else None


def short(name):
    name = name.replace("(anonymous namespace)::", "")
    return name.split("(")[0].strip()


def kernel_stats(raw):
    path = find(os.path.join(raw, "trace", "**", "*kernel_stats.csv"))
    rows = []
    if path:
        with open(path) as f:
            for r in csv.DictReader(f):
                rows.append(r)
    return path, rows


def kernel_trace_durations(raw, sub):
    path = find(os.path.join(raw, sub, "**", "*kernel_trace.csv"))
    d = defaultdict(list)
    if path:
        with open(path) as f:
            for r in csv.DictReader(f):
                d[short(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    return d


def pmc(raw, sub, counter):
    path = find(os.path.join(raw, sub, "**", "*counter_collection.csv"))
    d = defaultdict(list)
    if path:
        with open(path) as f:
            for r in csv.DictReader(f):
                if r.get("Counter_Name") == counter:
                    d[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    return d


def views_per_launch(raw, sub):
    """Views of a launch = the grid's y extent (z for k_winner) of the per-view kernels, from the kernel trace of the pass."""
    path = find(os.path.join(raw, sub, "**", "*kernel_trace.csv"))
    d = defaultdict(list)
    if path:
        with open(path) as f:
            for r in csv.DictReader(f):
                k = short(r["Kernel_Name"])
                if key_of(k) in ("k_bin_stats", "k_scan_tiles"):   # one workgroup per view along x
                    d[k].append(int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1))
                else:
                    d[k].append(int(r["Grid_Size_Y"]) // max(int(r["Workgroup_Size_Y"]), 1))
    # the vote kernel's grid is one thread per face: its views are those of the fused tile launch it follows
    fused = [k for k in d if key_of(k) == "k_raster_tile_fused"]
    if fused:
        for k in list(d):
            if key_of(k) == "k_vote_labels":
                d[k] = list(d[fused[0]])
    return d


def key_of(kernel):
    """traffic.json / valu.json key: the kernel's name without template arguments; the fused tile kernel gets its own."""
    name = kernel.replace("void ", "").strip()
    base = name.split("<")[0].strip()
    if base in ("k_raster_tile", "k_raster_tile_roll") and "<" in name:   # the fused kernel runs rolling chains since round 5
        args = [a.strip() for a in name.split("<", 1)[1].rstrip(">").split(",")]
        if len(args) > 3 and args[3] == "true":
            return "k_raster_tile_fused"
        return "k_raster_tile"
    return base


def main():
    raw, tag = sys.argv[1], sys.argv[2]
    out_txt = os.path.join(raw, f"summary_{tag}.txt")
    lines = []
    path, rows = kernel_stats(raw)
    lines.append(f"# rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 20 --warmup 2 --windows 2 --min-timed-s 0 --no-cpu-baseline --no-workload2 --no-c4 --no-c5 --no-api --no-io   [{tag}]")
    lines.append(f"# source: {path}")
    # median from the per-launch trace of the same run (the first launches after a mesh upload run cold and pull the mean up)
    trace = kernel_trace_durations(raw, "trace")
    lines.append(f"{'kernel':58s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>10s} {'median_us':>10s} {'min_us':>9s} {'max_us':>9s} {'pct':>6s}")
    for r in rows[:25]:
        d = sorted(trace.get(short(r['Name']), []))
        med = d[len(d) // 2] if d else float("nan")
        lines.append(f"{short(r['Name'])[:58]:58s} {r['Calls']:>7s} {float(r['TotalDurationNs'])/1e6:10.3f} "
                     f"{float(r['AverageNs'])/1e3:10.2f} {med:10.2f} {float(r['MinNs'])/1e3:9.2f} {float(r['MaxNs'])/1e3:9.2f} "
                     f"{float(r['Percentage']):6.2f}")
    fetch = pmc(raw, "pmc_fetch", "FETCH_SIZE")
    write = pmc(raw, "pmc_write", "WRITE_SIZE")
    dur = kernel_trace_durations(raw, "pmc_fetch")
    lines.append("")
    lines.append("# PMC passes (separate runs): python3 tools/prof_pipeline.py 50 3 64  (50 C2 views per pix2face launch, 64 C3 views per fused launch)")
    lines.append("# FETCH_SIZE / WRITE_SIZE are reported in KiB by rocprofv3; on gfx950 FETCH_SIZE counts 64 B per 128-B request")
    lines.append("# for wide streaming reads (MI355X_MICROARCH.md, HBM): the 'fetch_x2' column doubles it as the guide prescribes.")
    lines.append(f"{'kernel':40s} {'launches':>8s} {'fetch_MB':>10s} {'fetch_x2_MB':>12s} {'write_MB':>10s} {'avg_us':>9s}")
    summary = {}
    for k in sorted(set(fetch) | set(write)):
        fm = sum(fetch[k]) / max(len(fetch[k]), 1) * 1024 / 1e6 if k in fetch else float("nan")
        wm = sum(write[k]) / max(len(write[k]), 1) * 1024 / 1e6 if k in write else float("nan")
        du = sum(dur[k]) / max(len(dur[k]), 1) if k in dur else float("nan")
        lines.append(f"{k[:40]:40s} {len(fetch.get(k, write.get(k, []))):8d} {fm:10.2f} {2*fm:12.2f} {wm:10.2f} {du:9.2f}")
        summary[k] = {"launches": len(fetch.get(k, [])), "fetch_MB_per_launch": fm, "fetch_x2_MB_per_launch": 2 * fm,
                      "write_MB_per_launch": wm, "avg_us_in_pmc_pass": du}
    with open(out_txt, "w") as f:
        f.write("\n".join(lines) + "\n")
    with open(os.path.join(raw, f"summary_{tag}.json"), "w") as f:
        json.dump(summary, f, indent=1)
    # traffic.json: HBM bytes per launch of each kernel = 2 x FETCH_SIZE (gfx950 correction) + WRITE_SIZE, and per view
    # (the per-view kernels carry the views of a launch in their grid's y extent)
    vpl = views_per_launch(raw, "pmc_fetch")
    traffic = {}
    for k, v in summary.items():
        name = key_of(k)
        fm, wm = v["fetch_x2_MB_per_launch"], v["write_MB_per_launch"]
        if fm == fm and wm == wm:
            nv = sum(vpl.get(k, [1])) / max(len(vpl.get(k, [1])), 1)
            traffic[name] = {"hbm_bytes_per_launch": (fm + wm) * 1e6, "fetch_x2_MB": fm, "write_MB": wm,
                             "launches": v["launches"], "views_per_launch": nv, "hbm_bytes_per_view": (fm + wm) * 1e6 / max(nv, 1),
                             "source": f"profiles/summary_{tag}.txt"}
    traffic["_source"] = f"tools/profile.sh {tag}: summary_{tag}.txt"
    traffic["_kernel_sha256"] = kernel_sha()
    with open(os.path.join(raw, "traffic.json"), "w") as f:
        json.dump(traffic, f, indent=1)
    # valu.json: wave-level VALU instructions per view of the kernels the bench prices against the VALU-issue roofline
    sq = {c: pmc(raw, "pmc_sq", c) for c in ("SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_BUSY_CYCLES", "SQ_INSTS_SALU", "SQ_INSTS_LDS",
                                               "SQ_LDS_IDX_ACTIVE", "SQ_LDS_BANK_CONFLICT")}
    vpl = views_per_launch(raw, "pmc_sq")
    valu = {}
    lines.append("")
    lines.append("# SQ pass (separate run, same command as the PMC passes): per launch, averages")
    lines.append(f"{'kernel':44s} {'launches':>8s} {'views':>6s} {'VALU_insts':>14s} {'VALU_busy':>10s} {'SALU_insts':>14s} {'LDS_insts':>12s} {'LDS_busy':>9s} {'bank_conf':>10s}")
    for k in sorted(sq["SQ_INSTS_VALU"]):
        n = len(sq["SQ_INSTS_VALU"][k])
        avg = {c: sum(sq[c].get(k, [0.0])) / max(len(sq[c].get(k, [0.0])), 1) for c in sq}
        nv = sum(vpl.get(k, [1])) / max(len(vpl.get(k, [1])), 1)
        cycles = avg["SQ_BUSY_CYCLES"] / 32.0  # the counter sums over 32 shader engines
        busy = avg["SQ_ACTIVE_INST_VALU"] * 4.0 / (1024.0 * cycles) if cycles else float("nan")
        lds_busy = avg["SQ_LDS_IDX_ACTIVE"] / (256.0 * cycles) if cycles else float("nan")
        conf = avg["SQ_LDS_BANK_CONFLICT"] / avg["SQ_LDS_IDX_ACTIVE"] if avg["SQ_LDS_IDX_ACTIVE"] else float("nan")
        lines.append(f"{k[:44]:44s} {n:8d} {nv:6.1f} {avg['SQ_INSTS_VALU']:14.0f} {busy:10.3f} {avg['SQ_INSTS_SALU']:14.0f} "
                     f"{avg['SQ_INSTS_LDS']:12.0f} {lds_busy:9.3f} {conf:10.3f}")
        if k.replace("void ", "").startswith("k_"):
            valu[key_of(k)] = {"valu_insts_per_launch": avg["SQ_INSTS_VALU"], "views_per_launch": nv,
                               "valu_insts_per_view": avg["SQ_INSTS_VALU"] / max(nv, 1), "valu_busy_in_pmc_pass": busy,
                               "lds_busy_in_pmc_pass": lds_busy, "lds_bank_conflict_share": conf, "launches": n,
                               "source": f"profiles/summary_{tag}.txt"}
    valu["_source"] = f"tools/profile.sh {tag}: summary_{tag}.txt"
    valu["_kernel_sha256"] = kernel_sha()
    with open(os.path.join(raw, "valu.json"), "w") as f:
        json.dump(valu, f, indent=1)
    with open(out_txt, "w") as f:
        f.write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
