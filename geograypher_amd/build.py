"""Build libgeograster.so in-tree with hipcc for gfx950 (no torch, no cmake: one translation unit, seconds)."""
import os
import shutil
import subprocess
from pathlib import Path

ROOT = Path(__file__).resolve().parent
SRC = ROOT / "csrc" / "geograster.hip"
OUT = ROOT / "csrc" / "libgeograster.so"
INCLUDE = ROOT.parent / "include"

# -ffp-contract=off: the rule-set of DESIGN.md rounds every floating-point operation individually.
HIPCC_FLAGS = [
    "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
    "-fhip-fp32-correctly-rounded-divide-sqrt",
]


def hipcc_path() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and Path(cand).exists():
            return cand
    raise RuntimeError("hipcc not found (needed to build the gfx950 extension)")


def needs_build() -> bool:
    if not OUT.is_file():
        return True
    newest = max(SRC.stat().st_mtime, (INCLUDE / "geograster.h").stat().st_mtime)
    return OUT.stat().st_mtime < newest


def build(force: bool = False, verbose: bool = False) -> Path:
    if not force and not needs_build():
        return OUT
    cmd = [hipcc_path(), *HIPCC_FLAGS, f"-I{INCLUDE}", "-o", str(OUT), str(SRC)]
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError(f"hipcc failed:\n{' '.join(cmd)}\n{res.stdout}\n{res.stderr}")
    if verbose:
        print(" ".join(cmd))
    return OUT


if __name__ == "__main__":
    print(build(force=True, verbose=True))
