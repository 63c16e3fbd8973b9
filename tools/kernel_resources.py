#!/usr/bin/env python3
"""tools/kernel_resources.py [unit.hip ...] [-D...] -- registers, LDS, scratch and occupancy of every kernel of the library's
translation units, from hipcc's -Rpass-analysis=kernel-resource-usage (no GPU needed).  Workgroups of 256 threads per CU are
limited by min(8, LDS, VGPRs, floor(800 / (ceil(sgpr / 16) * 16 + 16))) (MI355X_MICROARCH.md, Residency)."""
import re
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from geograypher_amd import build as gbuild


def main():
    defs = [a for a in sys.argv[1:] if a.startswith("-D")]
    units = [a for a in sys.argv[1:] if not a.startswith("-D")] or [p.name for p in gbuild.SOURCES]
    flags = [f for f in gbuild.HIPCC_FLAGS if f not in ("-shared", "-fPIC")]
    print(f"{'kernel':78s} {'VGPR':>5s} {'SGPR':>5s} {'LDS':>7s} {'scratch':>7s} {'waves/SIMD':>10s} {'WG/CU(sgpr)':>11s}")
    for unit in units:
        cmd = [gbuild.hipcc_path(), *flags, *defs, "-c", "--cuda-device-only", f"-I{gbuild.INCLUDE}", f"-I{gbuild.CSRC}",
               "-Rpass-analysis=kernel-resource-usage", "-o", "/dev/null", str(gbuild.CSRC / unit)]
        err = subprocess.run(cmd, capture_output=True, text=True).stderr
        cur = {}
        for line in err.splitlines():
            m = re.search(r"remark:\s+(Function Name|VGPRs|TotalSGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\S+)", line)
            if not m:
                continue
            cur[m.group(1)] = m.group(2)
            if m.group(1).startswith("LDS Size"):
                name = subprocess.run(["c++filt", cur["Function Name"]], capture_output=True, text=True).stdout.strip()
                name = name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
                if "rocprim" in name or "hipcub" in name:
                    cur = {}
                    continue
                sg = int(cur["TotalSGPRs"])
                wg_sgpr = min(8, 800 // (-(-sg // 16) * 16 + 16))
                print(f"{name[:78]:78s} {cur['VGPRs']:>5s} {sg:5d} {cur['LDS Size [bytes/block]']:>7s} {cur['ScratchSize [bytes/lane]']:>7s} "
                      f"{cur['Occupancy [waves/SIMD]']:>10s} {wg_sgpr:11d}")
                cur = {}


if __name__ == "__main__":
    main()
