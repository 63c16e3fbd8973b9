"""The host-image pipelines of the reference-shaped API (round 3): images cross the link in their own dtype through a pinned
double buffer and are widened on the device; the sparse index aggregation keeps its pair keys on the device and counts
them once.  Results must equal a plain numpy restatement of meshes.py:1987-2002 / 2057-2082 and derived_meshes.py:470-550
for every input dtype (CPU: the oracle backend; -m gpu: the HIP backend)."""
import numpy as np
import pytest

from geograypher_amd.cameras import PhotogrammetryCameraSet
from geograypher_amd.meshes import TexturedPhotogrammetryMesh, TexturedPhotogrammetryMeshIndexPredictions
from geograypher_amd.utils import synthetic
from oracle import oracle_c, oracle_np

BACKENDS = [pytest.param("oracle", id="oracle"), pytest.param("hip", id="hip", marks=pytest.mark.gpu)]


def _backend(kind, request):
    if kind == "oracle":
        return request.getfixturevalue("oracle_backend_cls")()
    return request.getfixturevalue("hip")


class _ImageSet(PhotogrammetryCameraSet):
    """In-memory images on top of a camera set (what bench.host_image_set builds)."""

    def __init__(self, base, images):
        self.base_camera_set, self.images, self.cameras = base, images, base.cameras
        self._local_to_epsg_4978_transform = base._local_to_epsg_4978_transform
        self._maps_ideal_to_warped, self._maps_warped_to_ideal = {}, {}
        self.image_folder = None

    def __len__(self):
        return len(self.images)

    def n_image_channels(self):
        im = np.asarray(self.images[0])
        return 1 if im.ndim == 2 else int(im.shape[-1])

    def get_subset_cameras(self, inds):
        return _ImageSet(self.base_camera_set.get_subset_cameras(inds), [self.images[i] for i in inds])

    def get_image_by_index(self, i, image_scale=1.0):
        return self.images[i]


def _scene(n_views=5):
    (points, faces), cams = synthetic.config1_scene()
    cams = cams[0:n_views]
    recs = cams.get_raster_records(0.25, near=0.05)
    h, w = cams[0].get_image_size(0.25)
    return points, faces, cams, h, w


def _want(points, faces, cams, imgs, scale=0.25):
    """numpy restatement of the reference's float path on oracle ids."""
    from geograypher_amd.cameras.cameras import vtk_like_near_planes

    lo, hi = points.min(axis=0), points.max(axis=0)
    nears = vtk_like_near_planes(np.stack([np.asarray(c.cam_to_world_transform, dtype=np.float64) for c in cams.cameras]),
                                 np.array([lo[0], hi[0], lo[1], hi[1], lo[2], hi[2]]))
    recs = cams.get_raster_records(scale, near=list(nears))
    h, w = cams[0].get_image_size(scale)
    F = faces.shape[0]
    projs = []
    for v, img in enumerate(imgs):
        ids = oracle_c.raster(points, faces, recs[v], h, w).astype(np.int64)
        flat = np.asarray(img).reshape(h, w, -1).astype(np.float64)
        projs.append(oracle_np.project_image(ids, flat, F))
    summed = np.nansum(np.stack(projs), axis=0) if len(projs) > 1 else projs[0].copy()
    counts = sum(np.any(np.isfinite(p), axis=1).astype(np.float64) for p in projs)
    summed[counts == 0] = np.nan
    with np.errstate(divide="ignore", invalid="ignore"):
        return summed / counts[:, None], counts, summed


@pytest.mark.parametrize("kind", BACKENDS)
@pytest.mark.parametrize("dtype", ["uint8", "bool", "float32", "float64", "int16", "uint16"])
def test_float_path_takes_every_image_dtype(kind, request, dtype):
    points, faces, cams, h, w = _scene()
    rng = np.random.default_rng(3)
    imgs = []
    for v in range(len(cams)):
        if dtype == "bool":
            img = rng.random((h, w, 4)) < 0.3
        elif dtype in ("float32", "float64"):
            img = rng.random((h, w, 3)).astype(dtype)
            img[rng.random((h, w)) < 0.1] = np.nan          # NaN rows: nansum counts them as 0, the face still counts
            img[v::7, :, 1] = np.inf if v == 1 else img[v::7, :, 1]
        else:
            img = rng.integers(0, 200, size=(h, w, 3)).astype(dtype)
        imgs.append(img)
    mesh = TexturedPhotogrammetryMesh((points, faces), log_level="ERROR", backend=_backend(kind, request))
    avg, info = mesh.aggregate_projected_images(_ImageSet(cams, imgs), aggregate_img_scale=0.25)
    want_avg, want_counts, want_sum = _want(points, faces, cams, imgs)
    np.testing.assert_array_equal(info["projection_counts"], want_counts)
    np.testing.assert_allclose(info["summed_projections"], want_sum, rtol=1e-12, atol=0, equal_nan=True)
    np.testing.assert_allclose(avg, want_avg, rtol=1e-12, atol=0, equal_nan=True)
    assert np.isfinite(avg).any()
    # project_images yields the per-view projections the same way
    got = list(mesh.project_images(_ImageSet(cams, imgs), aggregate_img_scale=0.25))
    assert len(got) == len(cams) and got[0].shape == (faces.shape[0], imgs[0].shape[-1]) and got[0].dtype == np.float64


@pytest.mark.parametrize("kind", BACKENDS)
def test_sparse_path_counts_once_and_survives_compaction(kind, request):
    """Device-resident pair keys: many views, ONE sort + run-length count; with a buffer too small for all of them the
    accumulator counts down in between and merges -- same CSR arrays either way, equal to the numpy restatement."""
    points, faces, cams, h, w = _scene(6)
    F, nc = faces.shape[0], 37
    rng = np.random.default_rng(5)
    imgs = []
    for v in range(len(cams)):
        img = rng.integers(0, nc, size=(h, w)).astype(np.float64)
        img[rng.random((h, w)) < 0.2] = np.nan
        imgs.append(img if v % 2 else img.astype(np.float32))
    imgs[3] = np.full((h, w), np.nan)  # a null image: skipped (check_null_image, derived_meshes.py:465)
    be = _backend(kind, request)
    mesh = TexturedPhotogrammetryMeshIndexPredictions((points, faces), log_level="ERROR", backend=be)
    avg, info = mesh.aggregate_projected_images(_ImageSet(cams, imgs), n_classes=nc, aggregate_img_scale=0.25)
    from geograypher_amd.cameras.cameras import vtk_like_near_planes

    lo, hi = points.min(axis=0), points.max(axis=0)
    nears = vtk_like_near_planes(np.stack([np.asarray(c.cam_to_world_transform, dtype=np.float64) for c in cams.cameras]),
                                 np.array([lo[0], hi[0], lo[1], hi[1], lo[2], hi[2]]))
    recs = cams.get_raster_records(0.25, near=list(nears))
    projs = [oracle_np.project_image(oracle_c.raster(points, faces, recs[v], h, w).astype(np.int64),
                                     np.asarray(imgs[v], dtype=np.float64).reshape(h, w, 1), F, check_null_image=True)
             for v in range(len(cams))]
    want_avg, want_counts, want_sum = oracle_np.aggregate_index_sparse(projs, F, nc)
    want_counts = np.asarray(want_counts).reshape(-1)
    np.testing.assert_array_equal(info["projection_counts"].toarray()[:, 0], want_counts)
    np.testing.assert_array_equal(info["summed_projections"].toarray(), want_sum)
    np.testing.assert_allclose(avg.toarray(), want_avg, rtol=0, atol=1e-15)
    if kind == "hip":
        import torch

        from geograypher_amd._hip import PairAccumulator

        counts = torch.zeros((F,), dtype=torch.int32, device=be.device)
        acc = PairAccumulator(be, nc, counts)
        acc.cap = 2 * F + 5                      # room for two views: compaction between them
        acc.keys = torch.empty((acc.cap,), dtype=torch.int64, device=be.device)
        be.upload_mesh(points.astype(np.float32), faces.astype(np.int32))
        for v in range(len(cams)):
            if np.isfinite(imgs[v]).any():
                acc.add(be.raster_face_ids(recs[v:v + 1], h, w)[0], imgs[v])
        uniq, mult = acc.finish()
        assert acc.compactions >= 2
        got = np.zeros((F, nc), dtype=np.int64)
        got[uniq // nc, uniq % nc] = mult
        np.testing.assert_array_equal(got, want_sum)
        np.testing.assert_array_equal(counts.cpu().numpy(), want_counts)
        # a value that is no class index is reported once, at the end (GR_FLAG_DEFER_CHECK), not per view
        acc2 = PairAccumulator(be, nc, torch.zeros((F,), dtype=torch.int32, device=be.device))
        acc2.add(torch.zeros((h, w), dtype=torch.int32, device=be.device), np.full((h, w), float(nc + 3)))
        with pytest.raises(IndexError):
            acc2.finish()


@pytest.mark.parametrize("kind", BACKENDS)
@pytest.mark.parametrize("null_last", [False, True])
def test_float_path_drops_a_running_nan_like_the_reference(kind, request, null_last):
    """meshes.py:2060-2062, `np.nansum([summed, projection], axis=0)` view by view: a running sum that went NaN (+inf of one view
    met -inf of the next) counts as 0 at the view after -- also when that view is a null image that check_null_image skips."""
    points, faces, cams, h, w = _scene(4)
    rng = np.random.default_rng(11)
    imgs = [np.full((h, w, 2), np.inf), np.full((h, w, 2), -np.inf), rng.normal(0, 1, (h, w, 2)), rng.normal(0, 1, (h, w, 2))]
    imgs[1][::2, :, 1] = 4.0       # channel 1: inf - inf only on the odd rows
    imgs[3][:, ::3] = np.inf       # the last view makes NaNs of its own (inf - inf never happens: view 2 is finite) ...
    imgs[2][:, ::3, 0] = -np.inf   # ... except here: -inf from view 2, +inf from view 3 -> NaN that nothing drops
    if null_last:
        imgs[3] = np.full((h, w, 2), np.nan)
    backend = _backend(kind, request)
    mesh = TexturedPhotogrammetryMesh((points, faces), backend=backend, log_level="ERROR")
    avg, info = mesh.aggregate_projected_images(_ImageSet(cams, imgs), aggregate_img_scale=0.25, check_null_image=True)
    from geograypher_amd.cameras.cameras import vtk_like_near_planes

    lo, hi = points.min(axis=0), points.max(axis=0)
    nears = vtk_like_near_planes(np.stack([np.asarray(c.cam_to_world_transform, dtype=np.float64) for c in cams.cameras]),
                                 np.array([lo[0], hi[0], lo[1], hi[1], lo[2], hi[2]]))
    recs = cams.get_raster_records(0.25, near=list(nears))
    F = faces.shape[0]
    projs = [oracle_np.project_image(oracle_c.raster(points, faces, recs[v], h, w).astype(np.int64), imgs[v], F, check_null_image=True)
             for v in range(4)]
    with np.errstate(invalid="ignore"):
        want_avg, want = oracle_np.aggregate(projs, F)
    seen = want["projection_counts"] > 0
    assert np.isnan(want["summed_projections"][seen]).any() == (not null_last)  # a NaN survives only where the LAST view makes it
    np.testing.assert_array_equal(info["projection_counts"], want["projection_counts"])
    np.testing.assert_allclose(info["summed_projections"], want["summed_projections"], rtol=1e-12, equal_nan=True)
    np.testing.assert_allclose(avg, want_avg, rtol=1e-12, equal_nan=True)
