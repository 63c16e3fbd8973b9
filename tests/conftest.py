import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

GOLDEN = ROOT / "tests" / "golden"

# hermetic: the library's learned-binning table is not read from / written to the user's cache folder by the suite (a test
# that wants persistence points gr_learned_cache_file at its own tmp_path)
os.environ.setdefault("GEOGRAYPHER_AMD_CACHE", "off")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    with np.load(GOLDEN / "reference_numpy_stages.npz", allow_pickle=False) as d:
        return {k: d[k] for k in d.files}


@pytest.fixture(scope="session")
def golden_cameras():
    import numpy as np

    with np.load(GOLDEN / "reference_cameras.npz", allow_pickle=False) as d:
        return {k: d[k] for k in d.files}


@pytest.fixture(scope="session")
def golden_warp():
    import numpy as np

    with np.load(GOLDEN / "reference_warp.npz", allow_pickle=False) as d:
        return {k: d[k] for k in d.files}


@pytest.fixture(scope="session")
def oracle_backend_cls():
    from tests.oracle_backend import OracleBackend

    return OracleBackend


@pytest.fixture(scope="session")
def hip():
    """One HIP context for the GPU session (fails loudly if the extension or the GPU is missing)."""
    from geograypher_amd._hip import HipRaster

    return HipRaster(0)
