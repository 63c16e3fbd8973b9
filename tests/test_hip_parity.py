"""-m gpu: the HIP path (through the C ABI) against the CPU oracle on the same seeded inputs -- bit-exact for face
ids / depth bits / integer votes, 1e-12 relative for float64 sums (tolerance of north_star: 1e-5) -- plus golden
fixtures of the real reference and size-independent properties at BASELINE.json's full size."""
import numpy as np
import pytest
import torch

from geograypher_amd.cameras import SegmentorPhotogrammetryCameraSet
from geograypher_amd.meshes import TexturedPhotogrammetryMesh
from geograypher_amd.predictors import ArrayLabelSegmentor
from geograypher_amd.utils import synthetic
from oracle import oracle_c, oracle_np

pytestmark = pytest.mark.gpu

# every raster test runs against each of these settings of the tuning knobs (include/geograster.h GR_OPT_*): (tile height log2,
# slots per tile [0 = exact two-pass binning], GR_OPT_VARIANT mode bits: 1 = one tile per workgroup, 4 = votes on the caller's
# stream, 16 = chains of tiles even in small launches, 128 = 48-byte entries always, 512 = the general ids kernel where the plain
# one would run, 4096 / 8192 = micro lists never / always, 16384 = no look at the first launch group (status-call protocol),
# 131072 = packed tile counters) -- every combination must give identical results (tools/fuzz_parity.py draws random ones)
VARIANTS = {"default": (5, 512, 0), "tile32_chain": (5, 512, 16), "tile64_single_full": (6, 512, 1 + 128),
            "tile32_exact_single_inline_votes": (5, 0, 1 + 4), "tile64_exact_chain": (6, 0, 16),
            "tile32_chain_full_general_ids": (5, 512, 16 + 128 + 512), "tile64_single_general_ids": (6, 512, 1 + 512),
            "tile32_micro_chain": (5, 512, 16 + 8192), "tile32_micro_single": (5, 512, 1 + 8192),
            "tile64_micro_chain": (6, 512, 16 + 8192),
            "tile32_chain_no_look_packed_counters": (5, 512, 16 + 16384 + 131072),
            "tile64_single_no_look_no_micro": (6, 512, 1 + 16384 + 4096)}


@pytest.fixture(params=list(VARIANTS), autouse=True)
def raster_variant(request, hip):
    thl, cap, var = VARIANTS[request.param]
    hip.set_option(2, thl)
    hip.set_option(6, cap)
    hip.set_option(7, var)
    yield request.param
    hip.set_option(2, 5)
    hip.set_option(6, 512)
    hip.set_option(7, 0)


def _same(a, b):
    np.testing.assert_array_equal(np.isnan(a), np.isnan(b))
    np.testing.assert_array_equal(np.nan_to_num(a, nan=-7.0), np.nan_to_num(b, nan=-7.0))


def _records(cams, scale=1.0, near=0.05):
    return cams.get_raster_records(scale, near=near)


def _check_views(hip, points, faces, recs, h, w, depth=False):
    hip.upload_mesh(points.astype(np.float32), faces.astype(np.int32))
    if depth:
        ids, dep = hip.raster_face_ids(recs, h, w, want_depth=True)
        dep = dep.cpu().numpy()
    else:
        ids = hip.raster_face_ids(recs, h, w)
    ids = ids.cpu().numpy()
    for v in range(recs.shape[0]):
        if depth:
            want, wdep = oracle_c.raster(points, faces, recs[v], h, w, want_depth=True)
            np.testing.assert_array_equal(dep[v].view(np.int32), wdep.view(np.int32), err_msg=f"depth view {v}")
        else:
            want = oracle_c.raster(points, faces, recs[v], h, w)
        bad = np.argwhere(ids[v] != want)
        assert bad.size == 0, f"view {v}: {bad.shape[0]} pixels differ, first {bad[:5].tolist()}"
    return ids


@pytest.mark.parametrize("variant", [0, 128, 8192], ids=["entries40", "entries48", "micro_lists"])
def test_faces_centred_on_tile_corners_through_the_93_pixel_boundary(hip, variant):
    """Single-pass binning derives the entries of a small face's second to fourth tiles from its first-tile entry (binning.hip
    shifted_entry), which the short form of the edge constants allows for faces below 93.75 px; larger faces over 2 x 2 tiles
    take the big-face path.  Triangles of every size from 1 to 130 px centred on the corners of the 64 x 32 tiles (all three extra
    slots), on vertical and on horizontal tile borders, at many orientations and depths, overlapping: ids and depth bits equal
    the oracle's with 40-byte entries (a face of 93 px and more makes the call fall back by itself), 48-byte entries and micro
    lists."""
    rng = np.random.default_rng(93)
    h, w, f = 448, 640, 500.0
    pts, fcs = [], []
    k = 0
    for size in list(range(1, 20)) + list(range(20, 131, 3)) + [91, 92, 93, 94, 95]:
        for rep in range(3):
            cx = 64.0 * rng.integers(1, 9) + (rng.random() - 0.5) * (0.0 if rep == 0 else 0.4 * size)
            cy = 32.0 * rng.integers(1, 13) + (rng.random() - 0.5) * (0.0 if rep == 1 else 0.4 * size)
            ang = rng.random(3) * 0.6 + np.array([0.0, 2.1, 4.2]) + rng.random() * 6.28
            r = 0.5 * size * (0.6 + 0.4 * rng.random(3))
            z = 8.0 + 4.0 * rng.random(3)      # camera-space depth of each corner: tilted faces, overlapping in depth
            px = cx + r * np.cos(ang) * (1.6 if rep == 2 else 1.0)
            py = cy + r * np.sin(ang) * (0.4 if rep == 2 else 1.0)
            for i in range(3):     # pinhole at the origin looking down -z ... the record maps (x, y, z) -> (f x / z + cx0, f y / z + cy0)
                pts.append([(px[i] - 0.5 * w) * z[i] / f, (py[i] - 0.5 * h) * z[i] / f, z[i]])
            fcs.append([k, k + 1, k + 2])
            k += 3
    points = np.asarray(pts, dtype=np.float64)
    faces = np.asarray(fcs, dtype=np.int64)
    rec = np.zeros((1, 16), dtype=np.float32)      # identity rotation, camera at the origin: q = p
    rec[0, [0, 4, 8]] = 1.0
    rec[0, 12], rec[0, 13], rec[0, 14], rec[0, 15] = f, 0.5 * w, 0.5 * h, 0.05
    hip.set_option(7, variant)
    try:
        ids = _check_views(hip, points, faces, rec, h, w, depth=True)
    finally:
        hip.set_option(7, 0)
        hip.set_option(6, 512)
    assert (ids >= 0).mean() > 0.2 and len(np.unique(ids)) > 0.8 * faces.shape[0]


def test_gl_vertex_order_bit_exact_against_the_oracle_in_the_same_order(hip):
    """GR_OPT_VERTEX_ORDER = 1 (the vertex stage in an OpenGL pipeline's order of operations): ids and depth bits equal the
    oracle's under the same switch -- C1 (all views, image-border faces), a camera inside the scene (R7 clipping through the same
    snap) --, and differ from rule R1's somewhere (the switch does something)."""
    (points, faces), cams = synthetic.config1_scene()
    recs = _records(cams)
    hip.upload_mesh(points.astype(np.float32), faces.astype(np.int32))
    base = hip.raster_face_ids(recs, 480, 640).cpu().numpy()
    hip.set_vertex_order("gl")
    try:
        ids, dep = hip.raster_face_ids(recs, 480, 640, want_depth=True)
        ids, dep = ids.cpu().numpy(), dep.cpu().numpy()
        for v in range(recs.shape[0]):
            want, wdep = oracle_c.raster(points, faces, recs[v], 480, 640, want_depth=True, vertex_order="gl")
            np.testing.assert_array_equal(ids[v], want)
            np.testing.assert_array_equal(dep[v].view(np.int32), wdep.view(np.int32))
        assert (ids != base).sum() > 0
        rng = np.random.default_rng(5)
        verts = rng.uniform(-4, 4, (150, 3)).astype(np.float32)
        tris = rng.integers(0, 150, (300, 3)).astype(np.int32)
        inside = synthetic.camera_set_from_poses([synthetic.nadir_pose(0.2, -0.1, 0.3)], f=90.0, width=128, height=96)
        rec = inside.get_raster_records(1.0, near=0.05)
        hip.upload_mesh(verts, tris)
        got = hip.raster_face_ids(rec, 96, 128).cpu().numpy()[0]
        np.testing.assert_array_equal(got, oracle_c.raster(verts, tris, rec[0], 96, 128, vertex_order="gl"))
    finally:
        hip.set_vertex_order("r1")


def test_config1_all_views_bit_exact(hip):
    (points, faces), cams = synthetic.config1_scene()
    ids = _check_views(hip, points, faces, _records(cams), 480, 640, depth=True)
    assert ids.shape == (8, 480, 640)
    assert (ids >= 0).mean() > 0.5


def test_face_order_does_not_matter(hip):
    """The library re-orders the faces internally (Morton curve of the centroids) and reports the caller's ids: a
    shuffled face list -- no spatial coherence at all -- gives the same picture, face for face, and stays bit-exact
    against the oracle run on the shuffled mesh.  Vertices with NaN / inf coordinates and a flat mesh (one axis of
    zero extent) go through the same ordering code."""
    (points, faces), cams = synthetic.config1_scene()
    recs = _records(cams)[:3]
    base = _check_views(hip, points, faces, recs, 480, 640)
    perm = np.random.default_rng(11).permutation(faces.shape[0])
    shuffled = faces[perm]
    ids = _check_views(hip, points, shuffled, recs, 480, 640)
    back = np.where(ids >= 0, perm[np.clip(ids, 0, None)], -1)  # shuffled id -> original id
    np.testing.assert_array_equal(back, base)
    flat = points.copy()
    flat[:, 2] = 0.0
    _check_views(hip, flat, shuffled, recs, 480, 640)
    broken = np.vstack([points, [[np.nan, 0.0, 0.0], [np.inf, 1.0, 2.0], [0.0, -np.inf, 0.0]]])
    extra = np.array([[0, 1, len(points)], [2, len(points) + 1, 3], [len(points) + 2, 4, 5]], dtype=faces.dtype)
    _check_views(hip, broken, np.vstack([shuffled, extra]), recs, 480, 640)


@pytest.mark.parametrize("scale", [0.25, 0.37, 1.0])
def test_ragged_sizes_and_scales(hip, scale):
    """h, w not multiples of the 64-pixel tile, and the int(H*s) truncation of cameras.py:179-200."""
    (points, faces), cams = synthetic.config1_scene()
    for c in cams.cameras:
        c.image_width, c.image_height, c.image_size = 613, 457, (457, 613)
    h, w = cams[0].get_image_size(scale)
    _check_views(hip, points, faces, _records(cams, scale), h, w)


@pytest.mark.parametrize("pitch", [0.9, 1.5, 2.6])
def test_tile_sized_faces_many_tiles_per_wave(hip, pitch):
    """Faces about as large as a tile: the 64 faces of one wave of the set-up kernel touch more distinct tiles than the wave
    has lanes (the joint tile grouping hands one tile to each lane and takes the rest one by one), and small faces (at
    most 2x2 tiles, binned by k_setup_cull) mix with big ones (k_bin_big) in the same blocks."""
    nx, ny = 24, 24
    xs, ys = np.meshgrid(np.arange(nx + 1), np.arange(ny + 1))
    rng = np.random.default_rng(int(pitch * 10))
    z = rng.uniform(0.0, 0.4, xs.shape)
    f = 300.0
    height = 30.0
    step = pitch * 64.0 * height / f  # world size of a cell that projects to `pitch` tile widths
    points = np.stack([(xs - nx / 2) * step + 0.013, (ys - ny / 2) * step * 0.5 + 0.007, z], axis=-1).reshape(-1, 3)
    idx = lambda i, j: j * (nx + 1) + i
    faces = np.array([[idx(i, j), idx(i + 1, j), idx(i + 1, j + 1)] for j in range(ny) for i in range(nx)] +
                     [[idx(i, j), idx(i + 1, j + 1), idx(i, j + 1)] for j in range(ny) for i in range(nx)])
    poses = [synthetic.nadir_pose(0.0, 0.0, height), synthetic.nadir_pose(3.0, -2.0, height, yaw_deg=23.0)]
    cams = synthetic.camera_set_from_poses(poses, f=f, width=1600, height=900)
    _check_views(hip, points, faces, _records(cams), 900, 1600)


@pytest.mark.parametrize("seed", range(3))
def test_triangle_soup_occlusion_degenerates_behind_camera(hip, seed):
    """Random overlapping triangles of every size: sub-pixel slivers, triangles larger than a tile (64-bit edge path),
    coincident depths, zero-area faces, faces behind / straddling the camera plane (the large ones among the latter pass
    right in front of the lens and cover most of the picture once they are clipped instead of dropped)."""
    rng = np.random.default_rng(seed)
    n = 3000
    centers = rng.uniform(-30, 30, (n, 1, 3)) * np.array([1, 1, 0.3])
    size = np.exp(rng.uniform(np.log(0.02), np.log(40.0), (n, 1, 1)))
    tri = centers + rng.normal(0, 1, (n, 3, 3)) * size
    tri[:50, :, 2] = 45.0                       # behind the camera (camera at z = 40 looking down)
    tri[50:100, 0, 2] = 45.0                    # straddling the camera plane -> clipped at the near plane (R7)
    tri[100:120, 2] = tri[100:120, 1]           # zero area
    tri[120:140] = tri[140:160]                 # coincident faces: lower id wins
    points = tri.reshape(-1, 3)
    faces = np.arange(3 * n).reshape(n, 3)
    poses = [synthetic.nadir_pose(0, 0, 40.0, yaw_deg=17.0 * seed, tilt_x_deg=3.0 * seed),
             synthetic.look_at((60, 10, 25), (0, 0, 0), up_hint=(0, 0, 1))]
    cams = synthetic.camera_set_from_poses(poses, f=300.0, width=333, height=251)
    ids = _check_views(hip, points, faces, _records(cams, near=0.5), 251, 333, depth=True)
    assert len(np.unique(ids)) >= 3


def test_clipping_near_plane_and_guard_band(hip):
    """R7.  A ground plane of two 1 km triangles seen from 2 m above it, looking at the horizon: every vertex is either
    behind the camera or far outside the guard band, the clipped faces fill the picture below the horizon, and the
    depth of the bottom-centre pixel is the analytic one.  Then a camera in the middle of a dense terrain (faces pass
    behind and beside the lens at every distance), and a camera whose near plane cuts the C1 plane obliquely."""
    pts = np.array([[-500, -500, 0], [500, -500, 0], [500, 500, 0], [-500, 500, 0]], dtype=np.float64)
    quad = np.array([[0, 1, 2], [0, 2, 3]])
    pose = synthetic.look_at((0.0, 0.0, 2.0), (0.0, 100.0, 2.0), up_hint=(0, 0, 1))
    cams = synthetic.camera_set_from_poses([pose], f=300.0, width=320, height=240)
    recs = _records(cams, near=0.1)
    hip.upload_mesh(pts.astype(np.float32), quad.astype(np.int32))
    ids, dep = hip.raster_face_ids(recs, 240, 320, want_depth=True)
    want, wdep = oracle_c.raster(pts, quad, recs[0], 240, 320, want_depth=True)
    np.testing.assert_array_equal(ids[0].cpu().numpy(), want)
    np.testing.assert_array_equal(dep[0].cpu().numpy().view(np.int32), wdep.view(np.int32))
    rows = np.nonzero((want >= 0).any(axis=1))[0]
    assert rows[0] == 121 and rows[-1] == 239 and (want[121:] >= 0).all() and (want[:120] == -1).all()
    assert abs(wdep[239, 160] - 2.0 * 300.0 / 119.5) < 1e-3

    (points, faces), _ = synthetic.config1_scene()
    poses = [synthetic.look_at((1.0, 2.0, 0.9), (30.0, 20.0, 0.0), up_hint=(0, 0, 1)),
             synthetic.look_at((-3.0, 0.5, 0.7), (0.0, 40.0, 5.0), up_hint=(0, 0, 1)),
             synthetic.nadir_pose(0.0, 0.0, 0.8, tilt_x_deg=70.0)]
    cams = synthetic.camera_set_from_poses(poses, f=250.0, width=333, height=251)
    for near in (0.05, 0.7):
        ids = _check_views(hip, points, faces, _records(cams, near=near), 251, 333, depth=True)
        assert (ids >= 0).mean() > 0.3


@pytest.mark.parametrize("seed", range(12))
def test_random_stress(hip, seed):
    """Randomised scenes across the corner cases of the pipeline: image sizes from 1 pixel to non-multiples of the tile,
    hundreds to thousands of entries per tile (entry chunks beyond one per thread, single-pass segments overflowing into
    the exact path), edges parallel to the axes (A == 0 / B == 0), faces far larger than the image, cameras inside
    the scene's bounding box, principal points off-centre."""
    rng = np.random.default_rng(1000 + seed)
    h, w = [(1, 1), (3, 70), (65, 33), (64, 64), (97, 131), (200, 257)][seed % 6]
    n = [50, 400, 3000, 9000][seed % 4]
    if seed % 3 == 0:  # axis-aligned lattice of small quads plus noise-free coordinates -> exact ties and A/B == 0
        g = int(np.sqrt(n / 2)) + 1
        xs, ys = np.meshgrid(np.linspace(-3, 3, g + 1), np.linspace(-3, 3, g + 1))
        points = np.stack([xs.ravel(), ys.ravel(), np.zeros(xs.size)], axis=1)
        faces = synthetic.grid_faces(g + 1, g + 1)
    else:
        centers = rng.uniform(-4, 4, (n, 1, 3)) * np.array([1, 1, 0.2])
        size = np.exp(rng.uniform(np.log(0.01), np.log(30.0 if seed % 2 else 0.3), (n, 1, 1)))
        points = (centers + rng.normal(0, 1, (n, 3, 3)) * size).reshape(-1, 3)
        faces = np.arange(3 * n).reshape(n, 3)
    z_cam = [6.0, 1.0, 0.05][seed % 3]  # the last one sits inside the scene's bounding box
    poses = [synthetic.nadir_pose(rng.uniform(-1, 1), rng.uniform(-1, 1), z_cam, yaw_deg=rng.uniform(0, 360),
                                  tilt_x_deg=rng.uniform(-20, 20), tilt_y_deg=rng.uniform(-20, 20)) for _ in range(3)]
    cams = synthetic.camera_set_from_poses(poses, f=float(max(h, w)) * rng.uniform(0.3, 2.0), width=w, height=h)
    for c in cams.cameras:
        c.cx, c.cy = rng.uniform(-5, 5), rng.uniform(-5, 5)
    recs = cams.get_raster_records(1.0, near=0.02, principal_point="intrinsics" if seed % 2 else "center")
    _check_views(hip, points, faces, recs, h, w, depth=True)


def test_empty_view_and_single_face(hip):
    points = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], dtype=np.float64)
    faces = np.array([[0, 1, 2]])
    look_away = synthetic.nadir_pose(500, 500, 10.0)
    look_at = synthetic.nadir_pose(0.3, 0.3, 2.0)
    cams = synthetic.camera_set_from_poses([look_away, look_at], f=200.0, width=130, height=70)
    ids = _check_views(hip, points, faces, _records(cams), 70, 130)
    assert np.all(ids[0] == -1) and set(np.unique(ids[1])) == {-1, 0}


@pytest.mark.parametrize("batch", [64, 32, 5, 1])
def test_many_views_in_one_call_cross_batch_boundary(hip, batch):
    """More views than one launch group: results must not depend on the batching."""
    hip.set_option(3, batch)
    (points, faces), cams = synthetic.config1_scene()
    poses = [synthetic.nadir_pose(3.0 * k - 30, 2.0 * k - 20, 35.0 + k, yaw_deg=11.0 * k) for k in range(70)]
    cams = synthetic.camera_set_from_poses(poses, f=260.0, width=320, height=200)
    try:
        _check_views(hip, points, faces, _records(cams), 200, 320)
        # the fused aggregation path runs through the same pipelining
        recs = _records(cams)
        ids = hip.raster_face_ids(recs, 200, 320)
        labels = np.stack([synthetic.synthetic_labels(ids[v].cpu().numpy(), v, 3) for v in range(len(cams))])
        v1, c1 = hip.new_vote_buffers(3)
        hip.project_labels(ids, labels, 3, v1, c1)
        v2, c2 = hip.new_vote_buffers(3)
        hip.raster_project_labels(recs, labels, 3, v2, c2)
        assert torch.equal(v1, v2) and torch.equal(c1, c2)
    finally:
        hip.set_option(3, 64)


def test_single_pass_binning_overflow_learns_or_falls_back(hip):
    """Eight entry slots per tile are far too few for the C1 views: the overflow is reported with the size the image
    needs and the retry (single-pass binning with segments of that size) must reproduce the oracle.  20 000 faces stacked
    in ONE tile need more slots than a segment may have: that retry bins exactly (two-pass), same results."""
    (points, faces), cams = synthetic.config1_scene()
    hip.set_option(6, 8)
    ids = _check_views(hip, points, faces, _records(cams), 480, 640)
    assert hip.last_stats["overflow"] == 0 and (ids >= 0).mean() > 0.5
    hip.set_option(6, 512)
    n = 20000
    rng = np.random.default_rng(5)
    centre = rng.uniform(-0.2, 0.2, size=(n, 1, 3)) * np.array([1.0, 1.0, 0.0]) + np.array([0.0, 0.0, 1.0]) * rng.uniform(-1, 0, size=(n, 1, 1))
    tri = centre + rng.uniform(-0.15, 0.15, size=(n, 3, 3)) * np.array([1.0, 1.0, 0.0])
    points2 = tri.reshape(-1, 3)
    faces2 = np.arange(3 * n).reshape(n, 3)
    cams2 = synthetic.camera_set_from_poses([synthetic.nadir_pose(0, 0, 5.0)], f=300.0, width=64, height=64)
    ids2 = _check_views(hip, points2, faces2, _records(cams2), 64, 64)
    assert hip.last_stats["overflow"] == 0 and hip.last_stats["entries"] > 16384 and (ids2 >= 0).mean() > 0.2
    hip.set_option(6, 512)


def test_bin_overflow_is_detected_and_retried(hip, raster_variant):
    """40 stacked faces that each cover every tile of a 4000 x 3000 view: 40 x 2961 bin entries exceed the initial
    list capacity (2F + 65536); the library reports the exact need and the retry must reproduce the oracle."""
    n = 40
    base = np.array([[-8, -4, 0], [8, -4, 0], [0, 12, 0]], dtype=np.float64)  # inside the +-16384 px guard band
    points = np.concatenate([base + np.array([0, 0, -0.05 * k]) for k in range(n)], axis=0)
    faces = np.arange(3 * n).reshape(n, 3)
    cams = synthetic.camera_set_from_poses([synthetic.nadir_pose(0, 0, 5.0)], f=4000.0, width=4000, height=3000)
    hip.upload_mesh(points.astype(np.float32), faces.astype(np.int32))
    recs = _records(cams)
    ids = hip.raster_face_ids(recs, 3000, 4000).cpu().numpy()
    tiles = 63 * (47 if "tile64" in raster_variant else 94)
    assert hip.last_stats["overflow"] == 0 and hip.last_stats["entries"] == n * tiles
    if "exact" in raster_variant:
        assert hip.last_stats["entry_cap"] >= n * tiles
    want = oracle_c.raster(points, faces, recs[0], 3000, 4000)
    np.testing.assert_array_equal(ids[0], want)
    assert np.all(ids[0] == 0)


# ---- aggregation ---------------------------------------------------------------------------------------------------------
def test_projection_kernels_against_reference_golden(hip, golden):
    """Kernels fed the golden pix2face images: outputs must equal the REAL reference's (tests/golden)."""
    ids, F = golden["ids"].astype(np.int32), int(golden["F"])
    hip.upload_mesh(np.zeros((F + 3, 3), dtype=np.float32), np.zeros((F, 3), dtype=np.int32))
    C = golden["onehot"].shape[-1]
    votes, counts = hip.new_vote_buffers(C)
    hip.project_labels(ids, golden["label_inds"], C, votes, counts)
    avg, summed, cnt = (t.cpu().numpy() for t in hip.finalize_votes(votes, counts))
    _same(avg, golden["agg_onehot_average"])
    _same(summed, golden["agg_onehot_summed"])
    _same(cnt, golden["agg_onehot_counts"])
    for kind in ("rgb", "scalar", "onehot"):
        imgs = golden[kind].astype(np.float64).reshape(ids.shape + (-1,))
        for v in range(ids.shape[0]):
            _same(hip.project_view(ids[v], imgs[v]).cpu().numpy(), golden[f"project_{kind}"][v])
        sums = torch.zeros((F, imgs.shape[-1]), dtype=torch.float64, device=hip.device)
        cn = torch.zeros((F,), dtype=torch.int32, device=hip.device)
        hip.project_values(ids, imgs, sums, cn)
        avg, summed, cnt = (t.cpu().numpy() for t in hip.finalize_sums(sums, cn))
        np.testing.assert_allclose(avg, golden[f"agg_{kind}_average"], rtol=1e-12, equal_nan=True)
        np.testing.assert_allclose(summed, golden[f"agg_{kind}_summed"], rtol=1e-12, equal_nan=True)
        _same(cnt, golden[f"agg_{kind}_counts"])
    _same(hip.gather_texture(ids, golden["face_texture"]).cpu().numpy(), golden["render_flat"])
    _same(hip.argmax_nonzero(golden["argmax_in"]).cpu().numpy(), golden["argmax_out_flat"])


@pytest.mark.parametrize("calls", [1, 3])
def test_running_nan_of_the_float_sums_is_dropped_like_numpy_nansum(hip, calls):
    """meshes.py:2060-2062: `summed = np.nansum([summed, projection], axis=0)` treats a NaN of the RUNNING sum as 0 too: a face
    whose sum went NaN (+inf in one view, -inf in a later one) starts again from zero at the next view -- whether that view
    shows the face or not, whether it arrives in the same call or the next; only a NaN made by the very last view survives.
    (Found by tools/fuzz_stages.py.)"""
    F, h, w, C = 7, 2, 6, 2
    hip.upload_mesh(np.zeros((3, 3), dtype=np.float32), np.zeros((F, 3), dtype=np.int32))
    inf = np.inf
    ids = np.full((6, h, w), -1, dtype=np.int32)
    img = np.zeros((6, h, w, C))
    # pixel (0, k) of view v shows face k with value val[v][k]; None: the view does not show the face
    val = [[inf, inf, inf, 1.0, inf, np.nan],
           [-inf, -inf, 2.0, -inf, None, 3.0],
           [5.0, None, -inf, inf, None, None],
           [None, None, None, None, -inf, None],
           [None, None, 7.0, None, None, inf],
           [None, None, None, None, None, -inf]]
    for v, row in enumerate(val):
        for k, x in enumerate(row):
            if x is not None:
                ids[v, 0, k] = k
                img[v, 0, k] = (x, 1.0 if x == x else np.nan)
    projs = [oracle_np.project_image(ids[v].astype(np.int64), img[v], F, neg1_is_last_face=False) for v in range(6)]
    with np.errstate(invalid="ignore"):  # inf - inf
        want_avg, want = oracle_np.aggregate(projs, F)
    # what the reference makes of it: 5 | 0 (dropped by an unseen view) | 7 | inf | NaN... checked, not assumed:
    assert want["summed_projections"][0, 0] == 5.0 and want["summed_projections"][1, 0] == 0.0
    assert np.isnan(want["summed_projections"][5, 0]) and np.isnan(want["summed_projections"][6, 0])
    sums = torch.zeros((F, C), dtype=torch.float64, device="cuda")
    cnt = torch.zeros((F,), dtype=torch.int32, device="cuda")
    step = 6 // calls
    for v0 in range(0, 6, step):
        hip.project_values(ids[v0:v0 + step], img[v0:v0 + step], sums, cnt, neg1_is_last_face=False)
    avg, summed, counts = (t.cpu().numpy() for t in hip.finalize_sums(sums, cnt))
    _same(summed, want["summed_projections"])
    _same(avg, want_avg)
    _same(counts, want["projection_counts"])


@pytest.mark.parametrize("compat", [True, False])
@pytest.mark.parametrize("C", [1, 4, 8, 9, 16, 17, 40])  # the fused vote kernel packs up to 16 classes into two registers
def test_label_votes_bit_exact_vs_oracle(hip, compat, C):
    (points, faces), cams = synthetic.config1_scene()
    F = faces.shape[0]
    hip.upload_mesh(points.astype(np.float32), faces.astype(np.int32))
    recs = _records(cams)
    ids = hip.raster_face_ids(recs, 480, 640)
    ids_np = ids.cpu().numpy()
    labels = np.stack([synthetic.synthetic_labels(ids_np[v], v, C) for v in range(len(cams))])
    votes, counts = hip.new_vote_buffers(C)
    hip.project_labels(ids, labels, C, votes, counts, neg1_is_last_face=compat)
    want_v = np.zeros((F, C), dtype=np.uint32)
    want_c = np.zeros(F, dtype=np.uint32)
    for v in range(len(cams)):
        oracle_c.project_labels(ids_np[v], labels[v], F, C, want_v, want_c, neg1_is_last_face=compat)
    np.testing.assert_array_equal(votes.cpu().numpy().view(np.uint32), want_v)
    np.testing.assert_array_equal(counts.cpu().numpy().view(np.uint32), want_c)
    assert want_c.max() > 1 and (want_v.sum(axis=1) <= want_c).all()
    # fused entry point gives the same votes
    v2, c2 = hip.new_vote_buffers(C)
    hip.raster_project_labels(recs, labels, C, v2, c2, neg1_is_last_face=compat)
    assert torch.equal(v2, votes) and torch.equal(c2, counts)


@pytest.mark.parametrize("compat", [True, False])
@pytest.mark.parametrize("shuffle", [False, True])
def test_fused_votes_with_partial_views_and_scattered_face_order(hip, compat, shuffle):
    """The fused vote kernel reads the winners only of views whose cull pass reached the workgroup's 256-face chunk of caller
    ids.  Views that see a corner of the mesh (most chunks untouched, background around it: with the compatibility flag
    background pixels vote for face F - 1, whose chunk no block may have reached), a caller face order that is shuffled (the
    faces of a 64-face block lie in more chunks than its list holds: the view's "all" word), several launch groups."""
    (points, faces), _ = synthetic.config1_scene()
    if shuffle:
        faces = faces[np.random.default_rng(5).permutation(faces.shape[0])]
    F, C = faces.shape[0], 3
    lo, hi = points.min(axis=0), points.max(axis=0)
    mid, half = 0.5 * (lo + hi), 0.5 * (hi - lo)
    poses = [synthetic.nadir_pose(mid[0] + half[0] * np.cos(0.7 * k), mid[1] + half[1] * np.sin(0.7 * k), hi[2] + 14.0 + k,
                                  yaw_deg=31.0 * k) for k in range(9)]  # above the rim of the mesh: half of every view is background
    cams = synthetic.camera_set_from_poses(poses, f=300.0, width=320, height=200)
    recs = _records(cams)
    hip.upload_mesh(points.astype(np.float32), faces.astype(np.int32))
    hip.set_option(3, 4)  # three launch groups
    try:
        ids = hip.raster_face_ids(recs, 200, 320)
        ids_np = ids.cpu().numpy()
        assert (ids_np == -1).mean() > 0.1 and (ids_np >= 0).mean() > 0.1
        labels = np.stack([synthetic.synthetic_labels(ids_np[v], v, C) for v in range(len(cams))])
        want_v = np.zeros((F, C), dtype=np.uint32)
        want_c = np.zeros(F, dtype=np.uint32)
        for v in range(len(cams)):
            oracle_c.project_labels(ids_np[v], labels[v], F, C, want_v, want_c, neg1_is_last_face=compat)
        for _ in range(2):  # twice: the winner buffers must be left clean
            v2, c2 = hip.new_vote_buffers(C)
            hip.raster_project_labels(recs, labels, C, v2, c2, neg1_is_last_face=compat)
            np.testing.assert_array_equal(v2.cpu().numpy().view(np.uint32), want_v)
            np.testing.assert_array_equal(c2.cpu().numpy().view(np.uint32), want_c)
    finally:
        hip.set_option(3, 64)


def test_end_to_end_api_matches_oracle_pipeline(hip, oracle_backend_cls):
    """TexturedPhotogrammetryMesh on HIP vs the same class on the oracle backend: pix2face, render_flat,
    project_images, aggregate (index-label fast path and float path)."""
    (points, faces), cams = synthetic.config1_scene()
    F = faces.shape[0]
    tex = np.random.default_rng(0).random((F, 3))
    m_hip = TexturedPhotogrammetryMesh((points, faces), texture=tex, log_level="ERROR", backend=hip)
    m_orc = TexturedPhotogrammetryMesh((points, faces), texture=tex, log_level="ERROR", backend=oracle_backend_cls())
    sub = cams[0:3]
    a, b = m_hip.pix2face(sub, render_img_scale=0.5, apply_distortion=False), m_orc.pix2face(sub, render_img_scale=0.5, apply_distortion=False)
    assert a.dtype == np.int64 and a.shape == (3, 240, 320)
    np.testing.assert_array_equal(a, b)
    for ra, rb in zip(m_hip.render_flat(sub, render_img_scale=0.5, apply_distortion=False),
                      m_orc.render_flat(sub, render_img_scale=0.5, apply_distortion=False)):
        _same(ra, rb)
    for rt, rb in zip(m_hip.render_flat(sub, render_img_scale=0.5, apply_distortion=False, return_tensor=True),
                      m_orc.render_flat(sub, render_img_scale=0.5, apply_distortion=False)):
        assert isinstance(rt, torch.Tensor) and rt.is_cuda
        _same(rt.cpu().numpy(), rb)
    full = m_orc.pix2face(sub, apply_distortion=False)
    labels = [synthetic.synthetic_labels(full[v], v, 4) for v in range(3)]  # full-size label images, NN-resized by 0.5
    seg = ArrayLabelSegmentor(labels, 4, filenames=[c.image_filename for c in sub.cameras])
    out_h = m_hip.aggregate_projected_images(SegmentorPhotogrammetryCameraSet(sub, seg), aggregate_img_scale=0.5)
    out_o = m_orc.aggregate_projected_images(SegmentorPhotogrammetryCameraSet(sub, seg), aggregate_img_scale=0.5)
    _same(out_h[0], out_o[0])
    _same(out_h[1]["projection_counts"], out_o[1]["projection_counts"])
    assert np.nansum(out_h[0]) > 100
    for pa, pb in zip(m_hip.project_images(SegmentorPhotogrammetryCameraSet(sub, seg), aggregate_img_scale=0.5),
                      m_orc.project_images(SegmentorPhotogrammetryCameraSet(sub, seg), aggregate_img_scale=0.5)):
        _same(pa, pb)


# ---- full size (BASELINE.json config 2/3 shapes): properties + sampled oracle views --------------------------------------
@pytest.fixture(scope="module")
def terrain():
    return synthetic.terrain_mesh()


def test_full_size_views_bit_exact_and_properties(hip, terrain):
    points, faces = terrain
    assert faces.shape[0] == 1_201_250
    cams = synthetic.config2_cameras(50)
    recs = _records(cams, near=1.0)
    hip.upload_mesh(points.astype(np.float32), faces.astype(np.int32))
    pick = [0, 23, 49]
    ids = hip.raster_face_ids(recs[pick], 3000, 4000)
    ids_np = ids.cpu().numpy()
    for k, v in enumerate(pick[:2]):
        want = oracle_c.raster(points, faces, recs[v], 3000, 4000)
        assert np.array_equal(ids_np[k], want), f"view {v} differs in {(ids_np[k] != want).sum()} pixels"
    # properties that hold at any size: ids in range, rerun idempotent, a single-view call equals the batched call,
    # votes conserve observations (every visible face is counted exactly once per view)
    assert ids_np.min() >= -1 and ids_np.max() < faces.shape[0]
    again = hip.raster_face_ids(recs[pick], 3000, 4000)
    assert torch.equal(ids, again)
    single = hip.raster_face_ids(recs[pick[2]:pick[2] + 1], 3000, 4000)
    assert torch.equal(single[0], ids[2])
    C = 4
    labels = torch.from_numpy(np.stack([synthetic.synthetic_labels(ids_np[k], k, C) for k in range(3)])).to(hip.device)
    votes, counts = hip.new_vote_buffers(C)
    hip.project_labels(ids, labels, C, votes, counts, neg1_is_last_face=False)
    n_visible = sum(len(np.setdiff1d(np.unique(ids_np[k]), [-1])) for k in range(3))
    assert int(counts.sum()) == n_visible
    assert int(votes.sum()) <= n_visible
    want_v = np.zeros((faces.shape[0], C), dtype=np.uint32)
    want_c = np.zeros(faces.shape[0], dtype=np.uint32)
    lab_np = labels.cpu().numpy()
    for k in range(3):
        oracle_c.project_labels(ids_np[k], lab_np[k], faces.shape[0], C, want_v, want_c, neg1_is_last_face=False)
    np.testing.assert_array_equal(votes.cpu().numpy().view(np.uint32), want_v)
    np.testing.assert_array_equal(counts.cpu().numpy().view(np.uint32), want_c)
