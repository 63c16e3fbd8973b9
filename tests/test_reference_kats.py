"""The reference's own hot-path tests, restated against this package's classes.

    tests/test_derived_meshes.py:16-76     test_perspective_camera  (parametrized over the mesh class there;
                                           here parametrized over the backend: CPU oracle / HIP)
    tests/test_derived_cameras.py:316-415  TestPix2Face (the no-distortion properties; the warp itself is row f1)

The fixtures are the reference's closed-form ones (utils/test_utils.py), rebuilt in numpy by
geograypher_amd.utils.synthetic."""
import numpy as np
import pytest

from geograypher_amd.cameras import PhotogrammetryCamera, PhotogrammetryCameraSet
from geograypher_amd.constants import EARTH_CENTERED_EARTH_FIXED_CRS
from geograypher_amd.meshes import TexturedPhotogrammetryMesh
from geograypher_amd.utils.synthetic import downward_view, make_simple_camera_set, make_simple_mesh


def _backend(kind, request):
    if kind == "oracle":
        return request.getfixturevalue("oracle_backend_cls")()
    return request.getfixturevalue("hip")


BACKENDS = [pytest.param("oracle", id="oracle"), pytest.param("hip", id="hip", marks=pytest.mark.gpu)]


@pytest.mark.parametrize("kind", BACKENDS)
def test_perspective_camera(kind, request):
    fill_pixels = np.array([[10, 20], [15, 190], [195, 5], [50, 100], [150, 120]])
    empty_pixels = np.array([[30, 40], [160, 180], [120, 40], [100, 150], [180, 100]])
    mesh, point_colors = make_simple_mesh(pixels=fill_pixels, color=[255, 0, 0], background=80, buffer=1)
    textured_mesh = TexturedPhotogrammetryMesh(
        mesh=mesh, input_CRS=EARTH_CENTERED_EARTH_FIXED_CRS, texture=point_colors, log_level="ERROR",
        backend=_backend(kind, request),
    )
    cameras = make_simple_camera_set()
    renders = list(textured_mesh.render_flat(cameras=cameras, return_camera=False, apply_distortion=False))
    assert len(renders) == 1
    render = np.asarray(renders[0])
    assert render.ndim == 3
    assert render.shape[2] == 3
    assert render.shape[:2] == (200, 200)
    assert np.allclose(render[fill_pixels[:, 0], fill_pixels[:, 1]], [255, 0, 0])
    assert np.allclose(render[empty_pixels[:, 0], empty_pixels[:, 1]], [80, 80, 80])
    assert not np.any(np.isnan(render))  # the plane fills the whole view


@pytest.mark.parametrize("kind", BACKENDS)
@pytest.mark.parametrize("render_img_scale", [0.5, 0.7, 0.9, 1.0])
def test_pix2face_properties(kind, request, render_img_scale):
    """The `ideal` half of TestPix2Face.test_dewarp_pix2face (tests/test_derived_cameras.py:339-415)."""
    mesh, point_colors = make_simple_mesh(pixels=[], color=None)
    n_faces = mesh[1].shape[0]
    textured_mesh = TexturedPhotogrammetryMesh(
        mesh=mesh, input_CRS=EARTH_CENTERED_EARTH_FIXED_CRS, texture=point_colors, log_level="ERROR",
        backend=_backend(kind, request),
    )
    sensor = 2**8 + 1
    HT = downward_view(scene_width=4, focal=100, sensor_width=sensor)
    cam = PhotogrammetryCamera(None, HT, f=100, cx=0, cy=0, image_width=sensor, image_height=sensor,
                               local_to_epsg_4978_transform=np.eye(4))
    cameras = PhotogrammetryCameraSet([cam], local_to_epsg_4978_transform=np.eye(4))
    ideal = textured_mesh.pix2face(cameras=cameras, cache_folder=None, distortion_set=cameras,
                                   render_img_scale=render_img_scale, apply_distortion=False)
    assert len(ideal) == 1
    image = ideal[0]
    scaled_sensor = int(sensor * render_img_scale)
    assert isinstance(image, np.ndarray)
    assert image.dtype == np.int64
    assert image.shape == (scaled_sensor, scaled_sensor)
    assert image.min() >= -1
    assert image.max() < n_faces
    assert image.max() > 0.95 * n_faces
    for rows in (slice(None, 10), slice(-10, None)):
        for cols in (slice(None, 10), slice(-10, None)):
            assert len(np.unique(image[rows, cols])) > 1


@pytest.mark.parametrize("kind", BACKENDS)
def test_nadir_plane_ids_are_the_analytic_ones(kind, request):
    """Known answer beyond the reference's tests: the simple plane seen by the simple camera maps pixel (i, j) to
    the quad in row 199-i, column j (two candidate faces); checked for every pixel."""
    mesh, point_colors = make_simple_mesh(pixels=[], color=None)
    tm = TexturedPhotogrammetryMesh(mesh, texture=point_colors, log_level="ERROR", backend=_backend(kind, request))
    ids = tm.pix2face(make_simple_camera_set(), apply_distortion=False)[0]
    i, j = np.meshgrid(np.arange(200), np.arange(200), indexing="ij")
    quad = (199 - i) * 200 + j
    assert np.array_equal(ids // 2, quad)
