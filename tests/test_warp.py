"""Distortion-warp stage: Metashape camera parsing, the lens model, the cached sampling
maps and the resampling kernels -- against golden outputs of the REAL reference (tests/golden/reference_warp.npz,
reference_cameras.npz) and, restated, the reference's own tests (tests/test_derived_cameras.py:116-415)."""
import operator
from itertools import product
from pathlib import Path

import numpy as np
import pytest

from geograypher_amd.cameras import MetashapeCameraSet, PhotogrammetryCamera
from geograypher_amd.meshes import TexturedPhotogrammetryMesh
from geograypher_amd.utils.synthetic import downward_view, make_simple_mesh
from oracle import oracle_warp

GOLDEN = Path(__file__).resolve().parent / "golden"
BACKENDS = [pytest.param("oracle", id="oracle"), pytest.param("hip", id="hip", marks=pytest.mark.gpu)]


def _backend(kind, request):
    if kind == "oracle":
        return request.getfixturevalue("oracle_backend_cls")()
    return request.getfixturevalue("hip")


def _metashape_set(tmp_path):
    xml = tmp_path / "camera.xml"
    xml.write_text((GOLDEN / "metashape_camera.xml").read_text())
    return MetashapeCameraSet(camera_file=xml, image_folder=tmp_path)


def simplify_camera(camera, image, delete=None):
    """tests/test_derived_cameras.py:98-113"""
    camera.cx = 0
    camera.cy = 0
    camera.f = 100
    camera._local_to_epsg_4978_transform = np.eye(4)
    camera.image_height = image.shape[0]
    camera.image_width = image.shape[1]
    camera.image_size = image.shape[:2]
    for param in ["b1", "b2", "k1", "k2", "k3", "k4", "p1", "p2"]:
        camera.distortion_params[param] = 0
    for key in delete or ():
        del camera.distortion_params[key]
    return camera


# ---- host side: parsing, lens model, maps ----------------------------------------------------------------------------------
def test_metashape_parse_matches_reference(tmp_path, golden_cameras):
    cams = _metashape_set(tmp_path)
    c0 = cams.cameras[0]
    assert len(cams) == 1
    assert c0.f == float(golden_cameras["f"]) and c0.cx == float(golden_cameras["cx"]) and c0.cy == float(golden_cameras["cy"])
    assert (c0.image_width, c0.image_height) == (int(golden_cameras["image_width"]), int(golden_cameras["image_height"]))
    np.testing.assert_array_equal(c0.cam_to_world_transform, golden_cameras["cam_to_world"])
    np.testing.assert_array_equal(cams.get_local_to_epsg_4978_transform(), golden_cameras["local_to_epsg_4978"])
    keys = sorted(c0.distortion_params)
    assert keys == list(golden_cameras["distortion_keys"])
    np.testing.assert_array_equal([c0.distortion_params[k] for k in keys], golden_cameras["distortion_values"])
    # tests/test_derived_cameras.py:118-136
    expected = {"b1": 0.5262024073, "b2": -0.3058334293, "k1": -0.0919367147, "k2": -0.0762807468, "k3": 0.1162639394,
                "k4": -0.0761413904, "p1": -0.0003134847, "p2": 0.0001164035}
    for k, v in c0.distortion_params.items():
        assert np.isclose(v, expected[k])


def test_ideal_to_warped_matches_reference(tmp_path, golden_cameras):
    cams = _metashape_set(tmp_path)
    wx, wy = cams.ideal_to_warped(cams.cameras[0], golden_cameras["warp_in_x"].copy(), golden_cameras["warp_in_y"].copy())
    np.testing.assert_array_equal(wx, golden_cameras["warp_out_x"])
    np.testing.assert_array_equal(wy, golden_cameras["warp_out_y"])
    cam = PhotogrammetryCamera(None, np.eye(4), 100, 0, 0, 10, 10, distortion_params={"k1": 0.1, "zz": 1.0})
    with pytest.raises(ValueError):
        cams.ideal_to_warped(cam, np.zeros(1), np.zeros(1))
    cam.distortion_params = {"k2": 0.1}
    with pytest.raises(KeyError):  # k1 is required (derived_cameras.py:181)
        cams.ideal_to_warped(cam, np.zeros(1), np.zeros(1))


@pytest.mark.parametrize("kind", BACKENDS)
@pytest.mark.parametrize("scale,ds,tol", [(1.0, 8, 0.025), (0.5, 2, 0.004)])
def test_distortion_maps_match_reference(kind, request, tmp_path, golden_warp, scale, ds, tol):
    """The forward map is bit-equal to the REAL reference's.  The inverse is the dense Newton solve on the device instead of
    the reference's griddata over every `ds`-th pixel: the two differ by the interpolation error of the reference's
    triangles -- at most 0.023 px at downsample 8 and 0.003 px at 2 on this lens (k1 = -0.05, f = 100, 97 px sensor),
    which is exactly how far the REFERENCE's inverse is from satisfying forward(inverse(p)) = p; ours satisfies it to
    1e-9 px.  Every pixel the reference can invert (inside the hull of its samples), the dense inverse can too."""
    cams = _metashape_set(tmp_path)
    n = int(golden_warp["sensor"])
    cam = simplify_camera(cams.cameras[0], np.ones((n, n, 3)))
    cam.distortion_params["k1"] = float(golden_warp["k1"])
    cams.make_distortion_map(cam, ds, scale, backend=_backend(kind, request))
    key = cams.distortion_key(cam.distortion_params, scale)
    tag = f"s{int(scale * 100)}_d{ds}"
    np.testing.assert_array_equal(cams._maps_ideal_to_warped[key], golden_warp[f"i2w_{tag}"])
    ours, ref = cams._maps_warped_to_ideal[key], golden_warp[f"w2i_{tag}"]
    assert ours.shape == ref.shape and ours.dtype == np.float64
    ref_valid, our_valid = ref[0] != -1, ours[0] != -1
    assert np.all(our_valid[ref_valid]) and our_valid.mean() > 0.9
    diff = np.abs(ours - ref)[:, ref_valid]
    assert diff.max() < tol, diff.max()
    # the defining property, which the reference's inverse meets only to `tol`: forward(inverse(p)) == p
    model = cams.distortion_model(cam)
    h = w = int(n * scale)
    ti, tj = np.meshgrid(np.arange(h), np.arange(w), indexing="ij")
    fr, fc = oracle_warp.forward_map_position(model, ours[0], ours[1], scale)
    assert np.abs(fr - ti)[our_valid].max() < 1e-9 and np.abs(fc - tj)[our_valid].max() < 1e-9
    rr, rc = oracle_warp.forward_map_position(model, ref[0], ref[1], scale)
    assert 0.1 * tol < max(np.abs(rr - ti)[ref_valid].max(), np.abs(rc - tj)[ref_valid].max()) < tol


@pytest.mark.parametrize("scale,ds", [(1.0, 8), (0.5, 2)])
def test_reference_inverse_flag_and_host_fallback_are_bit_equal_to_the_reference(tmp_path, golden_warp, scale, ds):
    """`make_distortion_map(reference_inverse=True)` -- and, on a machine without a GPU, the default -- inverts by
    `utils.indexing.inverse_map_interpolation` over every `ds`-th pixel like cameras.py:1045-1062: the map equals the
    REAL reference's bit for bit, and `warp_dewarp_pixels` (host look-ups only) works without a device."""
    import torch

    from geograypher_amd.utils.indexing import inverse_map_interpolation

    tag = f"s{int(scale * 100)}_d{ds}"
    np.testing.assert_array_equal(inverse_map_interpolation(golden_warp[f"i2w_{tag}"], ds), golden_warp[f"w2i_{tag}"])
    cams = _metashape_set(tmp_path)
    n = int(golden_warp["sensor"])
    cam = simplify_camera(cams.cameras[0], np.ones((n, n, 3)))
    cam.distortion_params["k1"] = float(golden_warp["k1"])
    key = cams.distortion_key(cam.distortion_params, scale)
    cams.make_distortion_map(cam, ds, scale, reference_inverse=True)
    np.testing.assert_array_equal(cams._maps_warped_to_ideal[key], golden_warp[f"w2i_{tag}"])
    if not torch.cuda.is_available():  # host fallback: no backend given, no GPU
        cams._maps_warped_to_ideal.clear()
        cams.make_distortion_map(cam, ds, scale)
        np.testing.assert_array_equal(cams._maps_warped_to_ideal[key], golden_warp[f"w2i_{tag}"])
        if scale == 1.0:
            pix = np.array([[10, 12], [40, 41], [80, 7]])
            got = cams.warp_dewarp_pixels(cam, pix, inversion_downsample=ds)
            np.testing.assert_array_equal(got, golden_warp[f"w2i_{tag}"][:, pix[:, 0], pix[:, 1]].T)


@pytest.mark.parametrize("kind", BACKENDS)
def test_full_model_maps_match_reference(kind, request, tmp_path, golden_warp):
    """All eight parameters of the XML's lens (k1..k4, p1, p2, b1, b2; cx, cy != 0) at scale 0.02: forward map bit-equal
    to the reference; the inverse round-trips through the full model (the reference's own inverse, griddata over a
    2 x 2 sample grid at downsample 64, is no yardstick here)."""
    cams = _metashape_set(tmp_path)
    cam = cams.cameras[0]
    cams.make_distortion_map(cam, 64, 0.02, backend=_backend(kind, request))
    key = cams.distortion_key(cam.distortion_params, 0.02)
    np.testing.assert_array_equal(cams._maps_ideal_to_warped[key], golden_warp["full_i2w_s2"])
    inv = cams._maps_warped_to_ideal[key]
    assert inv.shape == golden_warp["full_w2i_s2"].shape
    ok = inv[0] != -1
    assert ok.mean() > 0.8  # the border of the warped image is mapped from outside the ideal one
    h, w = inv.shape[1:]
    ti, tj = np.meshgrid(np.arange(h), np.arange(w), indexing="ij")
    fr, fc = oracle_warp.forward_map_position(cams.distortion_model(cam), inv[0], inv[1], 0.02)
    assert np.abs(fr - ti)[ok].max() < 1e-9 and np.abs(fc - tj)[ok].max() < 1e-9
    # the forward map of the reference and the forward model of the oracle are the same function
    r, c = np.meshgrid(np.arange(h, dtype=float), np.arange(w, dtype=float), indexing="ij")
    gr, gc = oracle_warp.forward_map_position(cams.distortion_model(cam), r, c, 0.02)
    np.testing.assert_allclose(np.stack([gr, gc]), golden_warp["full_i2w_s2"], rtol=0, atol=1e-9)


@pytest.mark.gpu
def test_device_inverse_equals_the_numpy_newton_inverse_and_round_trips_at_full_size(hip, tmp_path):
    """k_invert_distortion (analytic Jacobian) against the numpy solver (finite differences) on the XML's full lens at a
    size the CPU finishes in seconds, then the round trip forward(inverse(p)) = p at the sensor's full 5280 x 3956."""
    cams = _metashape_set(tmp_path)
    cam = cams.cameras[0]
    model = cams.distortion_model(cam)
    for scale in (0.1, 0.25):
        h, w = cam.get_image_size(scale)
        got = hip.invert_distortion(model, h, w, scale).cpu().numpy()
        want = oracle_warp.newton_inverse_map(model, h, w, scale)
        np.testing.assert_array_equal(got[0] == -1, want[0] == -1)
        np.testing.assert_allclose(got, want, rtol=0, atol=1e-8)
    h, w = cam.get_image_size(1.0)
    assert (h, w) == (3956, 5280)
    inv = hip.invert_distortion(model, h, w, 1.0)
    ok = inv[0] != -1
    assert float(ok.double().mean()) > 0.8
    inv_np = inv.cpu().numpy()
    ti, tj = np.meshgrid(np.arange(h), np.arange(w), indexing="ij")
    fr, fc = oracle_warp.forward_map_position(model, inv_np[0], inv_np[1], 1.0)
    okn = ok.cpu().numpy()
    assert np.abs(fr - ti)[okn].max() < 1e-8 and np.abs(fc - tj)[okn].max() < 1e-8
    # pixels the lens maps from outside the ideal image are `fill`; the corners of a barrel-corrected image are among them
    assert inv_np[0].min() == -1.0 or okn.all()


# ---- oracle pinned to the real flexible_inputs_warp -------------------------------------------------------------------------
CASES = [("ids", "w2i_s100_d8", 0, -1, "ids_warped"), ("ids", "i2w_s100_d8", 0, -1, "ids_dewarped"),
         ("ids_half", "w2i_s50_d2", 0, -1, "ids_half_warped"), ("mask", "w2i_s100_d8", 0, 0.0, "mask_warped"),
         ("fimg", "i2w_s100_d8", 1, 0.0, "fimg_dewarped_o1"), ("fimg", "w2i_s100_d8", 1, 0.0, "fimg_warped_o1")]


@pytest.mark.parametrize("src,mapname,order,fill,want", CASES)
def test_oracle_warp_matches_reference(golden_warp, src, mapname, order, fill, want):
    img, m, ref = golden_warp[src], golden_warp[mapname], golden_warp[want]
    got = oracle_warp.flexible_inputs_warp_reference(img, m, order, fill)
    assert got.dtype == ref.dtype and got.shape == ref.shape
    inside = oracle_warp.inside_mask(m, img.shape)
    if order == 0:
        np.testing.assert_array_equal(got[inside], ref[inside])
    else:  # bilinear: 0.18.3's "constant" mode does not interpolate towards cval in the last half pixel
        core = inside & (m[0] <= img.shape[0] - 2) & (m[1] <= img.shape[1] - 2) & (m[0] >= 1) & (m[1] >= 1)
        np.testing.assert_allclose(got[core], ref[core], rtol=0, atol=1e-12)
    assert inside.mean() > 0.5


def test_reference_float_roundtrip_corrupts_ids(golden_warp):
    """Fact 6 of SURVEY.md in the golden data: the reference's own warp returns ids that are off by one."""
    exact = oracle_warp.warp_exact(golden_warp["ids"], golden_warp["w2i_s100_d8"], 0, -1)
    ref = golden_warp["ids_warped"]
    inside = oracle_warp.inside_mask(golden_warp["w2i_s100_d8"], golden_warp["ids"].shape)
    diff = (exact - ref)[inside]
    assert set(np.unique(diff)) <= {0, 1} and 0.005 < (diff != 0).mean() < 0.2


# ---- the resampling kernels ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("kind", BACKENDS)
@pytest.mark.parametrize("src,mapname,order,fill,want", CASES)
def test_warp_image_exact_and_roundtrip(kind, request, golden_warp, src, mapname, order, fill, want):
    be = _backend(kind, request)
    img, m, ref = golden_warp[src], golden_warp[mapname], golden_warp[want]
    mt = be.upload_map(m)
    got = be.warp_image(img, mt, order=order, fill_value=fill)
    exact = oracle_warp.warp_exact(img, m, order, fill)
    assert got.dtype == img.dtype and got.shape == ref.shape
    if order == 0:
        np.testing.assert_array_equal(got, exact)  # integers gathered as integers, everywhere
        if np.issubdtype(img.dtype, np.integer) and img.dtype != np.uint8:
            rt = be.warp_image(img, mt, order=0, fill_value=fill, reference_float_roundtrip=True)
            inside = oracle_warp.inside_mask(m, img.shape)
            np.testing.assert_array_equal(rt[inside], ref[inside])  # bit for bit the reference, off-by-ones included
    else:
        np.testing.assert_allclose(got, exact, rtol=0, atol=1e-12)


@pytest.mark.parametrize("kind", BACKENDS)
def test_constant_image_shortcut(kind, request, golden_warp):
    be = _backend(kind, request)
    const = np.full((97, 97), 7, dtype=np.int64)
    got = be.warp_image(const, be.upload_map(golden_warp["w2i_s100_d8"]), order=0, fill_value=7)
    np.testing.assert_array_equal(got, golden_warp["const_warped"])


# ---- the reference's own tests, restated --------------------------------------------------------------------------------------
@pytest.fixture
def gradient():
    size = 21
    center = size // 2
    x, y = np.meshgrid(np.arange(size), np.arange(size))
    dist = np.sqrt((x - center) ** 2 + (y - center) ** 2)
    g = np.clip(1 - (dist / np.max(dist)), 0, 1)
    return (np.stack([g] * 3, axis=2) * 255).astype(np.uint8)


@pytest.mark.parametrize("kind", BACKENDS)
@pytest.mark.parametrize("w2i,k1,relationship", [(True, 0, operator.eq), (True, 10, operator.lt), (True, -10, operator.gt),
                                                 (False, 0, operator.eq), (False, 10, operator.gt), (False, -10, operator.lt)])
@pytest.mark.parametrize("downsample", [1, 2])
@pytest.mark.parametrize("grayscale", [True, False])
def test_warp_dewarp_image(kind, request, tmp_path, gradient, w2i, k1, relationship, downsample, grayscale):
    """tests/test_derived_cameras.py:138-182"""
    cameras = _metashape_set(tmp_path)
    camera = simplify_camera(cameras.cameras[0], gradient)
    camera.distortion_params["k1"] = k1
    if grayscale:
        gradient = gradient[:, :, 0]
    dewarped = cameras.warp_dewarp_image(camera, gradient, warped_to_ideal=w2i, inversion_downsample=downsample,
                                         backend=_backend(kind, request))
    assert dewarped.shape == gradient.shape and dewarped.dtype == gradient.dtype
    assert relationship(dewarped.mean(), gradient.mean())


@pytest.mark.parametrize("delete", list(__import__("itertools").combinations(["b1", "b2", "k1", "k2", "k3", "k4", "p1", "p2"], 2))[::3])
@pytest.mark.parametrize("w2i,k1,relationship", [(True, 1, operator.lt), (True, -1, operator.gt), (False, 1, operator.gt),
                                                 (False, -1, operator.lt)])
@pytest.mark.parametrize("downsample", [1, 2])
def test_dropped_parameters(request, tmp_path, gradient, delete, w2i, k1, relationship, downsample):
    """tests/test_derived_cameras.py:213-247: distortion parameters missing from the dict count as zero (every third of the
    28 pairs the reference drops; k1 is set again afterwards, as there)."""
    cameras = _metashape_set(tmp_path)
    camera = simplify_camera(cameras.cameras[0], gradient, delete=delete)
    camera.distortion_params["k1"] = k1
    dewarped = cameras.warp_dewarp_image(camera, gradient, warped_to_ideal=w2i, inversion_downsample=downsample,
                                         backend=_backend("oracle", request))
    assert dewarped.shape == gradient.shape
    assert relationship(dewarped.mean(), gradient.mean())


@pytest.mark.parametrize("kind", BACKENDS)
@pytest.mark.parametrize("k1", [1.0, 0.0, -1.0])
@pytest.mark.parametrize("w2i", [True, False])
def test_mask_image(kind, request, tmp_path, k1, w2i):
    """tests/test_derived_cameras.py:184-211 (interpolation_order=0 as pix2face uses)"""
    image = np.ones((21, 21), dtype=np.uint8)
    image[:10] = 0
    image[:, 5:] = 2
    cameras = _metashape_set(tmp_path)
    camera = simplify_camera(cameras.cameras[0], image)
    camera.distortion_params["k1"] = k1
    dewarped = cameras.warp_dewarp_image(camera, image, warped_to_ideal=w2i, inversion_downsample=2,
                                         interpolation_order=0, backend=_backend(kind, request))
    assert sorted(np.unique(dewarped)) == [0, 1, 2]


@pytest.mark.parametrize("w2i,k1,relationship", [(True, 0, operator.eq), (True, 0.5, operator.lt), (True, -0.5, operator.gt),
                                                 (False, 0, operator.eq), (False, 0.5, operator.gt), (False, -0.5, operator.lt)])
def test_warp_dewarp_pixels(tmp_path, oracle_backend_cls, w2i, k1, relationship):
    """tests/test_derived_cameras.py:251-313"""
    be = oracle_backend_cls()
    fake = np.zeros((101, 101, 3), dtype=np.uint8)
    cameras = _metashape_set(tmp_path)
    camera = simplify_camera(cameras.cameras[0], fake)
    camera.distortion_params["k1"] = k1
    pixels = np.array([[20, 20], [20, 50], [20, 80], [50, 20], [50, 80], [80, 20], [80, 50], [80, 80]])
    center = np.mean([[0, 0], fake.shape[:2]], axis=0).astype(int)
    dewarped = cameras.warp_dewarp_pixels(camera, pixels, warped_to_ideal=w2i, inversion_downsample=2, backend=be)
    assert isinstance(dewarped, np.ndarray) and dewarped.shape == pixels.shape and dewarped.dtype == float
    original = np.linalg.norm(pixels - center, axis=1)
    altered = np.linalg.norm(dewarped - center, axis=1)
    assert relationship(altered, original).all()


@pytest.mark.parametrize("kind", BACKENDS)
@pytest.mark.parametrize("render_img_scale", [0.5, 0.7, 0.9, 1.0])
def test_dewarp_pix2face(kind, request, tmp_path, render_img_scale):
    """tests/test_derived_cameras.py:339-415, complete: ideal vs warped pix2face of the simple plane."""
    mesh, point_colors = make_simple_mesh(pixels=[], color=None)
    n_faces = mesh[1].shape[0]
    be = _backend(kind, request)
    textured_mesh = TexturedPhotogrammetryMesh(mesh=mesh, texture=point_colors, log_level="ERROR", backend=be)
    cameras = _metashape_set(tmp_path)
    sensor = 2**8 + 1
    camera = simplify_camera(cameras.cameras[0], image=np.ones((sensor, sensor, 3)))
    camera.distortion_params["k1"] = -0.05
    cameras._local_to_epsg_4978_transform = np.eye(4)
    HT = downward_view(scene_width=4, focal=cameras.cameras[0].f, sensor_width=sensor)
    cameras.cameras[0].cam_to_world_transform = HT
    cameras.cameras[0].world_to_cam_transform = np.linalg.inv(HT)
    kwargs = {"cameras": cameras, "cache_folder": None, "distortion_set": cameras, "render_img_scale": render_img_scale}
    ideal = textured_mesh.pix2face(**kwargs, apply_distortion=False)
    assert len(ideal) == 1
    ideal = ideal[0]
    warped = textured_mesh.pix2face(**kwargs, apply_distortion=True)
    assert len(warped) == 1
    warped = warped[0]
    scaled_sensor = int(sensor * render_img_scale)
    for image in [ideal, warped]:
        assert isinstance(image, np.ndarray)
        assert image.dtype == np.int64
        assert image.shape == (scaled_sensor, scaled_sensor)
        assert image.min() >= -1
        assert image.max() < n_faces
        assert image.max() > 0.95 * n_faces
    for corner in product([slice(None, 10), slice(-10, None)], repeat=2):
        assert len(np.unique(ideal[corner])) > 1
        assert np.all(warped[corner] == -1)
