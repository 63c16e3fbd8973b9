#!/usr/bin/env python3
"""tools/ab_libs.py <rounds> <workloads: c2,c5> <tag[:-D defines, comma separated[:GR_OPT_VARIANT bits]]> ... -- A/B of library BUILDS on one
box.  GPU box only.

Every tag is a build of the library: `base` = the product's libgeograster.so, anything else libgeograster_<tag>.so (built here
with the given -D defines when missing, e.g. `wide:GR_SETUP_BPW=6u`; a library saved from an earlier tree under that name is used as it is).  Rounds x builds x workloads, alternating (box-to-box spread is larger than most kernel
changes): each cell is one run of tools/ab_kernel.py in a child process with GEOGRAYPHER_AMD_LIB pointing at the build.  The
FIRST build's ids and votes are hashed; every other build must reproduce them (bit-exact A/B).  Prints the medians."""
import json
import os
import statistics
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from geograypher_amd import build as gbuild


def main():
    rounds = int(sys.argv[1])
    workloads = sys.argv[2].split(",")
    libs = []
    for spec in sys.argv[3:]:
        tag, _, rest = spec.partition(":")
        bits, _, var = rest.partition(":")
        name = f"{tag}/{var}" if var else tag
        if tag == "base":
            libs.append((name, gbuild.build(), var or "0"))
            continue
        path = gbuild.CSRC / f"libgeograster_{tag}.so"
        if not path.is_file():
            gbuild.build_variant(tag, [d for d in bits.split(",") if d])
        libs.append((name, path, var or "0"))
    acc = {(t, w): [] for t, _, _ in libs for w in workloads}
    sums = {}
    for r in range(rounds):
        for w in workloads:
            for tag, path, var in libs[r % len(libs):] + libs[:r % len(libs)]:   # rotate the order: no build is always first
                env = dict(os.environ, GEOGRAYPHER_AMD_LIB=str(path), AB_WORKLOAD=w, AB_CHECKSUM="1")
                nv, reps = ("20", "4") if w in ("c5", "forest", "forestq") else ("50", "5")
                res = subprocess.run([sys.executable, str(ROOT / "tools" / "ab_kernel.py"), nv, reps, f"x:{var}"], env=env,
                                     capture_output=True, text=True)
                lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
                if res.returncode != 0 or not lines:
                    print(f"{tag} {w}: FAILED rc={res.returncode} {res.stderr[-400:]}", flush=True)
                    continue
                d = json.loads(lines[-1])
                acc[(tag, w)].append(d)
                chk = d.get("checksum")
                if chk is not None:
                    ref = sums.setdefault(w, (tag, chk))
                    if ref[1] != chk:
                        print(f"MISMATCH {tag} {w}: checksum {chk} differs from {ref[0]}'s {ref[1]}", flush=True)
                print(f"round {r} {w} {tag:12s} setup {d['plain']['setup_ms']:6.2f} plain {d['plain']['raster_ms']:6.2f} "
                      f"fused {d['fused']['raster_ms']:6.2f} vote {d['fused']['vote_ms']:5.2f}", flush=True)
    print("--- medians (us per view)")
    for w in workloads:
        for tag, _, _ in libs:
            runs = acc[(tag, w)]
            if not runs:
                continue
            med = lambda k1, k2: statistics.median(r[k1][k2] for r in runs)
            print(json.dumps({"workload": w, "build": tag, "setup": med("plain", "setup_ms"), "plain": med("plain", "raster_ms"),
                              "fused": med("fused", "raster_ms"), "vote": med("fused", "vote_ms"), "runs": len(runs),
                              "setup_runs": [r["plain"]["setup_ms"] for r in runs], "plain_runs": [r["plain"]["raster_ms"] for r in runs]}))


if __name__ == "__main__":
    main()
