"""The C-ABI library loads and exports every symbol include/geograster.h declares (no compute without a GPU)."""
import ctypes
import re
from pathlib import Path

import pytest

from geograypher_amd import _hip

ROOT = Path(__file__).resolve().parents[1]


def _declared_symbols():
    text = (ROOT / "include" / "geograster.h").read_text()
    return sorted(set(re.findall(r"\b(gr_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree():
    assert _declared_symbols() == sorted(_hip.EXPORTED_SYMBOLS)


def test_library_exports_every_declared_symbol():
    lib = ctypes.CDLL(str(_hip.library_path()))
    for name in _declared_symbols():
        assert hasattr(lib, name), f"libgeograster.so does not export {name}"
    lib.gr_version.restype = ctypes.c_int
    assert lib.gr_version() == 123


def _header_struct(name):
    """[(field, C type)] of `typedef struct <name> { ... }` in include/geograster.h."""
    text = (ROOT / "include" / "geograster.h").read_text()
    body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (name, name), text, re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    return [(m.group(2), m.group(1)) for m in re.finditer(r"\b(int64_t|int32_t|float|double)\s+(\w+)\s*;", body)]


C_TYPES = {"int64_t": ctypes.c_int64, "int32_t": ctypes.c_int32, "float": ctypes.c_float, "double": ctypes.c_double}


@pytest.mark.parametrize("struct,mirror", [("gr_raster_stats", _hip.RasterStats), ("gr_stage_times", _hip.StageTimes)])
def test_binding_structures_mirror_the_header(struct, mirror):
    """The library WRITES these structures through the caller's pointer: a binding with a shorter mirror is a buffer overrun."""
    want = [(n, C_TYPES[t]) for n, t in _header_struct(struct)]
    assert len(want) >= 8
    assert [(n, t) for n, t in mirror._fields_] == want


def test_integration_stub_structure_mirrors_the_header():
    """... and so is the reference-side stub of INTEGRATION.md (executed as written by tests/test_integration_stub.py on the GPU)."""
    text = (ROOT / "INTEGRATION.md").read_text()
    block = re.search(r"class Stats\(ctypes.Structure\):.*?_fields_ = (.*?)\n\s*stats, v0", text, re.S).group(1)
    fields = eval(block.replace("\\\n", " "), {"ctypes": ctypes})
    assert fields == [(n, C_TYPES[t]) for n, t in _header_struct("gr_raster_stats")]


def test_no_gpu_fails_loudly():
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _hip.HipRaster()
    from geograypher_amd.meshes import TexturedPhotogrammetryMesh
    from geograypher_amd.utils import synthetic

    (mesh, _colors) = synthetic.make_simple_mesh([], None)
    tm = TexturedPhotogrammetryMesh(mesh, log_level="ERROR")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        tm.pix2face(synthetic.make_simple_camera_set(), apply_distortion=False)


def test_product_never_imports_the_oracle():
    for path in (ROOT / "geograypher_amd").rglob("*.py"):
        text = path.read_text()
        assert "import oracle" not in text and "from oracle" not in text, path
