"""-m gpu: the binding INTEGRATION.md hands to a reference maintainer is executed as written -- extracted from the
markdown, dropped next to a minimal stand-in for the reference's `TexturedPhotogrammetryMesh` (pyvista-style mesh with
`.points` / VTK-format `.faces`, `get_mesh_in_cameras_coords`) -- and must reproduce the oracle."""
import re
from pathlib import Path

import numpy as np
import pytest

from geograypher_amd import _hip
from geograypher_amd.cameras import PhotogrammetryCamera
from geograypher_amd.utils import synthetic
from oracle import oracle_c

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parents[1]


def test_integration_stub_runs_as_written(hip):
    text = (ROOT / "INTEGRATION.md").read_text()
    code = re.search(r"```python\n(import ctypes.*?)```", text, re.S).group(1)

    class _PolyData:  # what get_mesh_in_cameras_coords returns in the reference: a pyvista mesh
        def __init__(self, points, faces):
            self.points = points
            self.faces = np.hstack([np.full((faces.shape[0], 1), 3), faces]).ravel()

    (points, faces), cams = synthetic.config1_scene()

    class TexturedPhotogrammetryMesh:  # the slice of the reference's base class the stub relies on
        def __init__(self):
            self._mesh = _PolyData(points.astype(np.float64), faces)

        def get_mesh_in_cameras_coords(self, cameras):
            return self._mesh

    ns = {"TexturedPhotogrammetryMesh": TexturedPhotogrammetryMesh, "PhotogrammetryCamera": PhotogrammetryCamera,
          "CACHE_FOLDER": "/tmp/unused"}
    exec(code, ns)
    mesh = ns["TexturedPhotogrammetryMeshHIP"](libgeograster=str(_hip.library_path()))
    ids = mesh.pix2face(cams, render_img_scale=0.5)
    assert ids.shape == (8, 240, 320) and ids.dtype == np.int64
    one = mesh.pix2face(cams[3], render_img_scale=0.5)
    np.testing.assert_array_equal(one, ids[3])
    # the stub's near plane is the product's (VTK's clipping-range rule): the same mesh + cameras through the two bindings
    # clip the same faces
    from geograypher_amd.cameras.cameras import vtk_like_near_planes

    lo, hi = points.min(axis=0), points.max(axis=0)
    nears = vtk_like_near_planes(np.stack([np.asarray(c.cam_to_world_transform, dtype=np.float64) for c in cams.cameras]),
                                 np.array([lo[0], hi[0], lo[1], hi[1], lo[2], hi[2]]))
    recs = cams.get_raster_records(0.5, near=list(nears))
    for v in (0, 3, 7):
        want = oracle_c.raster(points, faces, recs[v], 240, 320)
        np.testing.assert_array_equal(ids[v], want)
    from geograypher_amd.meshes import TexturedPhotogrammetryMesh as ProductMesh

    product = ProductMesh((points, faces), log_level="ERROR", backend=hip)
    np.testing.assert_array_equal(product.pix2face(cams, render_img_scale=0.5, apply_distortion=False), ids)
