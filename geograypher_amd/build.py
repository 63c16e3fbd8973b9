"""Build libgeograster.so in-tree with hipcc for gfx950 (no torch, no cmake: a handful of translation units compiled in
parallel and linked, seconds)."""
import os
import shutil
import subprocess
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

ROOT = Path(__file__).resolve().parent
CSRC = ROOT / "csrc"
# translation units of the library (csrc/gr_internal.hpp says what lives where); raster_tile.hip holds the dominant kernel
SOURCES = [CSRC / n for n in ("geograster.hip", "mesh_upload.hip", "binning.hip", "raster_tile.hip", "project.hip", "warp.hip",
                              "resize.hip")]
HEADERS = [CSRC / "gr_internal.hpp", CSRC / "dev_common.hpp"]
SRC = CSRC / "raster_tile.hip"   # the tile kernel's source (tests/test_isa_waits.py compiles it to assembly)
OUT = CSRC / "libgeograster.so"
OBJ = CSRC / "_obj"
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


def _newest_header() -> float:
    return max(p.stat().st_mtime for p in HEADERS + [INCLUDE / "geograster.h", Path(__file__)])


def needs_build() -> bool:
    if not OUT.is_file():
        return True
    newest = max(max(p.stat().st_mtime for p in SOURCES), _newest_header())
    return OUT.stat().st_mtime < newest


def _compile(src: Path, force: bool) -> Path:
    obj = OBJ / (src.stem + ".o")
    if not force and obj.is_file() and obj.stat().st_mtime >= max(src.stat().st_mtime, _newest_header()):
        return obj
    flags = [f for f in HIPCC_FLAGS if f != "-shared"]
    cmd = [hipcc_path(), *flags, "-c", f"-I{INCLUDE}", f"-I{CSRC}", "-o", str(obj), str(src)]
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError(f"hipcc failed:\n{' '.join(cmd)}\n{res.stdout}\n{res.stderr}")
    return obj


def build_variant(tag: str, defines) -> Path:
    """A diagnostic build beside the product library: csrc/libgeograster_<tag>.so compiled with extra -D flags (e.g.
    GR_STAMPS: in-kernel phase stamps of the tile kernel, tools/tile_phases.py).  Load it with GEOGRAYPHER_AMD_LIB=<path>."""
    import tempfile

    out = CSRC / f"libgeograster_{tag}.so"
    # objects of diagnostic builds live outside the tree: everything under the repo travels to the GPU box with every run
    obj_dir = Path(tempfile.gettempdir()) / f"geograster_obj_{tag}"
    obj_dir.mkdir(parents=True, exist_ok=True)
    flags = [f for f in HIPCC_FLAGS if f != "-shared"] + [f"-D{d}" for d in defines]

    def one(src):
        obj = obj_dir / (src.stem + ".o")
        cmd = [hipcc_path(), *flags, "-c", f"-I{INCLUDE}", f"-I{CSRC}", "-o", str(obj), str(src)]
        res = subprocess.run(cmd, capture_output=True, text=True)
        if res.returncode != 0:
            raise RuntimeError(f"hipcc failed:\n{' '.join(cmd)}\n{res.stdout}\n{res.stderr}")
        return obj

    with ThreadPoolExecutor(max_workers=min(len(SOURCES), os.cpu_count() or 1)) as pool:
        objs = list(pool.map(one, SOURCES))
    cmd = [hipcc_path(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", str(out), *[str(o) for o in objs]]
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError(f"link failed:\n{' '.join(cmd)}\n{res.stdout}\n{res.stderr}")
    return out


def build(force: bool = False, verbose: bool = False) -> Path:
    if not force and not needs_build():
        return OUT
    OBJ.mkdir(exist_ok=True)
    with ThreadPoolExecutor(max_workers=min(len(SOURCES), os.cpu_count() or 1)) as pool:
        objs = list(pool.map(lambda s: _compile(s, force), SOURCES))
    cmd = [hipcc_path(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", str(OUT), *[str(o) for o in objs]]
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError(f"link failed:\n{' '.join(cmd)}\n{res.stdout}\n{res.stderr}")
    if verbose:
        print(" ".join(cmd))
    return OUT


if __name__ == "__main__":
    print(build(force=True, verbose=True))
