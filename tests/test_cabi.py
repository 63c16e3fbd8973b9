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
