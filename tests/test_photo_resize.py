"""Row a5: `PhotogrammetryCamera.get_image(image_scale != 1)` -- the photo down-scale of the aggregation path
(geograypher/cameras/cameras.py:154-174 -> skimage.transform.resize, reached from meshes.py:1988).

  * oracle/oracle_resize.py against outputs of the REAL scikit-image 0.18.3 `resize` and of the >= 0.19 formulation through
    the real scipy (tests/golden/make_golden_photo_resize.py), 1e-12
  * -m gpu: gr_resize_image_f64 against the same goldens (1e-12), for uint8 / float32 / float64 inputs, at full photo size
    against the oracle, and `aggregate_projected_images(base_camera_set, aggregate_img_scale=0.25)` on FILE-BACKED photos
    against the numpy restatement of meshes.py:1987-2002, 2057-2082 fed by the golden-pinned resizer (1e-5, the north-star
    tolerance for aggregated float textures; achieved: ~1e-15)
"""
from pathlib import Path

import numpy as np
import pytest

from geograypher_amd.cameras import PhotogrammetryCamera, PhotogrammetryCameraSet
from geograypher_amd.meshes import TexturedPhotogrammetryMesh
from geograypher_amd.utils import synthetic
from oracle import oracle_c, oracle_np, oracle_resize

GOLDEN = Path(__file__).resolve().parent / "golden"
BACKENDS = [pytest.param("oracle", id="oracle"), pytest.param("hip", id="hip", marks=pytest.mark.gpu)]
PHOTO_TAGS = ["s25", "s37", "s50"]
GRAY_TAGS = ["s25", "s37", "s50", "s90"]


@pytest.fixture(scope="module")
def golden_photo():
    with np.load(GOLDEN / "reference_photo_resize.npz", allow_pickle=False) as d:
        return {k: d[k] for k in d.files}


def _backend(kind, request):
    if kind == "oracle":
        return request.getfixturevalue("oracle_backend_cls")()
    return request.getfixturevalue("hip")


def test_golden_is_from_real_skimage(golden_photo):
    assert str(golden_photo["skimage_version"]) == "0.18.3"
    assert golden_photo["photo_u8"].dtype == np.uint8 and golden_photo["photo_s25"].shape == (24, 32, 3)
    # the two scikit-image formulations agree with each other far inside the test tolerance
    for tag in PHOTO_TAGS:
        assert np.abs(golden_photo[f"photo_{tag}"] - golden_photo[f"zoom_photo_{tag}"]).max() < 1e-12


def test_gaussian_filter_restatement_is_scipy_bit_for_bit(golden_photo):
    """Step 2 of the resize is scipy.ndimage.gaussian_filter itself (both scikit-image versions call it): the oracle's
    numpy restatement of it -- weights, mirror boundary, summation order -- returns the same bits."""
    ndi = pytest.importorskip("scipy.ndimage")
    img = golden_photo["photo_u8"] / 255.0
    for sr, sc in ((1.5, 1.5), (0.8514, 0.8646), (0.0, 2.25)):
        want = ndi.gaussian_filter(img, (sr, sc, 0), mode="mirror")
        got = oracle_resize.gaussian_axis(oracle_resize.gaussian_axis(img, sr, 0), sc, 1)
        np.testing.assert_array_equal(got, want)


@pytest.mark.parametrize("kind", BACKENDS)
@pytest.mark.parametrize("tag", PHOTO_TAGS)
def test_uint8_photo_resize_matches_skimage(kind, request, golden_photo, tag):
    be = _backend(kind, request)
    photo = golden_photo["photo_u8"]
    want = golden_photo[f"photo_{tag}"]
    got = np.asarray(be.resize_image(photo, want.shape[:2]).cpu())
    assert got.shape == want.shape and got.dtype == np.float64
    np.testing.assert_allclose(got, want, rtol=0, atol=1e-12)
    np.testing.assert_allclose(got, golden_photo[f"zoom_photo_{tag}"], rtol=0, atol=1e-12)  # scikit-image >= 0.19 (pinned 0.21.0)
    # the float64 image (what get_image holds after `/ 255.0`) gives the same result as the uint8 file bytes
    got64 = np.asarray(be.resize_image(photo / 255.0, want.shape[:2]).cpu())
    np.testing.assert_allclose(got64, got, rtol=0, atol=1e-15)


@pytest.mark.parametrize("kind", BACKENDS)
@pytest.mark.parametrize("tag", GRAY_TAGS)
def test_float_image_resize_matches_skimage(kind, request, golden_photo, tag):
    be = _backend(kind, request)
    gray = golden_photo["gray_f64"]
    want = golden_photo[f"gray_{tag}"]
    got = np.asarray(be.resize_image(gray, want.shape).cpu())
    assert got.shape == want.shape
    np.testing.assert_allclose(got, want, rtol=0, atol=1e-12)
    np.testing.assert_allclose(got, golden_photo[f"zoom_gray_{tag}"], rtol=0, atol=1e-12)


@pytest.mark.parametrize("kind", BACKENDS)
def test_float32_upscale_and_identity(kind, request, golden_photo):
    be = _backend(kind, request)
    # float32 input: scikit-image computes in float32, here float64 -- equal within float32 rounding
    got = np.asarray(be.resize_image(golden_photo["rgb_f32"], (24, 40)).cpu())
    np.testing.assert_allclose(got, golden_photo["rgb32_s50"].astype(np.float64), rtol=0, atol=5e-7)
    # up-scaling: no anti-aliasing, samples beyond the border mirrored
    small = golden_photo["gray_f64"][:20, :24].copy()
    got = np.asarray(be.resize_image(small, (30, 36)).cpu())
    np.testing.assert_allclose(got, golden_photo["up_s150"], rtol=0, atol=1e-12)
    # scale 1: `image / 255.0` alone, bit for bit (cameras.py:158-159)
    photo = golden_photo["photo_u8"]
    np.testing.assert_array_equal(np.asarray(be.resize_image(photo).cpu()), photo / 255.0)
    np.testing.assert_array_equal(np.asarray(be.resize_image(small).cpu()), small)


def _write_photos(folder, n, h, w, seed=11):
    from PIL import Image

    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    files, arrays = [], []
    for v in range(n):
        base = np.stack([xx * (255.0 / w), yy * (255.0 / h), 128 + 90 * np.sin((xx + 13 * v) / 11.0) * np.cos(yy / 7.0)], axis=-1)
        img = np.clip(base + rng.normal(0, 25, size=base.shape), 0, 255).astype(np.uint8)
        path = Path(folder) / f"photo_{v:03d}.png"
        Image.fromarray(img).save(path)
        files.append(path)
        arrays.append(img)
    return files, arrays


@pytest.mark.parametrize("kind", BACKENDS)
def test_aggregate_file_backed_photos_at_scale(kind, request, tmp_path):
    """`aggregate_projected_images(base_camera_set, aggregate_img_scale=0.25)` on photos read from files -- the call
    entrypoints/aggregate_images.py:184 makes -- against the reference's arithmetic restated in numpy and fed by the
    golden-pinned resizer: meshes.py:1987-2002 (project), 2057-2067 (nansum + counts), 2069-2082 (average)."""
    from geograypher_amd.cameras.cameras import vtk_like_near_planes

    (points, faces), cams = synthetic.config1_scene()
    n, scale = 4, 0.25
    H, W = cams[0].image_height, cams[0].image_width
    files, arrays = _write_photos(tmp_path, n, H, W)
    file_cams = PhotogrammetryCameraSet(
        [PhotogrammetryCamera(files[v], cams[v].cam_to_world_transform, cams[v].f, cams[v].cx, cams[v].cy, W, H) for v in range(n)]
    )
    be = _backend(kind, request)
    mesh = TexturedPhotogrammetryMesh((points, faces), log_level="ERROR", backend=be)
    avg, info = mesh.aggregate_projected_images(file_cams, aggregate_img_scale=scale)

    h, w = int(H * scale), int(W * scale)
    lo, hi = points.min(axis=0), points.max(axis=0)
    nears = vtk_like_near_planes(np.stack([np.asarray(c.cam_to_world_transform, dtype=np.float64) for c in file_cams.cameras]),
                                 np.array([lo[0], hi[0], lo[1], hi[1], lo[2], hi[2]]))
    recs = file_cams.get_raster_records(scale, near=list(nears))
    F = faces.shape[0]
    projs = []
    for v in range(n):
        ids = oracle_c.raster(points, faces, recs[v], h, w).astype(np.int64)
        img = oracle_resize.get_image_scaled(arrays[v], scale)   # cameras.py:154-174, pinned to scikit-image above
        assert img.shape == (h, w, 3)
        projs.append(oracle_np.project_image(ids, img, F))
    summed = np.nansum(np.stack(projs), axis=0)
    counts = sum(np.any(np.isfinite(p), axis=1).astype(np.float64) for p in projs)
    summed[counts == 0] = np.nan
    with np.errstate(divide="ignore", invalid="ignore"):
        want = summed / counts[:, None]
    assert (counts > 0).mean() > 0.5
    np.testing.assert_array_equal(info["projection_counts"], counts)
    np.testing.assert_allclose(avg, want, rtol=0, atol=1e-5, equal_nan=True)        # north-star tolerance
    np.testing.assert_allclose(avg, want, rtol=0, atol=1e-12, equal_nan=True)       # what the device path achieves
    # scale 1 on the same files: uint8 crosses the link, `/ 255.0` on the device is the host's division bit for bit
    avg1, info1 = mesh.aggregate_projected_images(file_cams[0:2], aggregate_img_scale=1.0)
    recs1 = file_cams[0:2].get_raster_records(1.0, near=list(nears[:2]))
    projs1 = [oracle_np.project_image(oracle_c.raster(points, faces, recs1[v], H, W).astype(np.int64), arrays[v] / 255.0, F)
              for v in range(2)]
    s1 = np.nansum(np.stack(projs1), axis=0)
    c1 = sum(np.any(np.isfinite(p), axis=1).astype(np.float64) for p in projs1)
    s1[c1 == 0] = np.nan
    np.testing.assert_array_equal(info1["summed_projections"], s1)


@pytest.mark.gpu
def test_get_image_scaled_public_api_and_full_size(hip, tmp_path):
    """`PhotogrammetryCamera.get_image(image_scale)` itself (numpy in / numpy out through the default backend) and a
    full-size 4000 x 3000 RGB photo at the reference's example scale against the oracle."""
    files, arrays = _write_photos(tmp_path, 1, 3000, 4000, seed=3)
    cam = PhotogrammetryCamera(files[0], np.eye(4), 3000.0, 0.0, 0.0, 4000, 3000)
    got = cam.get_image(0.25, backend=hip)
    want = oracle_resize.get_image_scaled(arrays[0], 0.25)
    assert got.shape == (750, 1000, 3) and got.dtype == np.float64
    np.testing.assert_allclose(got, want, rtol=0, atol=1e-12)
    np.testing.assert_array_equal(cam.get_image(1.0), arrays[0] / 255.0)
    got37 = cam.get_image(0.37)   # default backend
    np.testing.assert_allclose(got37, oracle_resize.get_image_scaled(arrays[0], 0.37), rtol=0, atol=1e-12)
    with pytest.raises(ValueError):
        hip.resize_image(np.zeros((4, 4, 3, 2)))


def test_scaled_get_image_without_gpu_fails_loudly(tmp_path):
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    files, _ = _write_photos(tmp_path, 1, 12, 16)
    cam = PhotogrammetryCamera(files[0], np.eye(4), 10.0, 0.0, 0.0, 16, 12)
    assert cam.get_image(1.0).shape == (12, 16, 3)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        cam.get_image(0.5)
