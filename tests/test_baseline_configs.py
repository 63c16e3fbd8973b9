"""-m gpu: BASELINE.json configs 3, 4 and 5 and the large-image / many-class corners of the projection kernels, at
full size, through the C ABI, against the CPU oracle.

  C3  C2 mesh, the 500-view camera grid: 130 views of it (two 64-view launch groups and a remainder) through the FUSED
      raster + projection call at 4000x3000 with 4-class labels generated on the device
  C4  2000 views sharded over 8 GPUs: one rank's 250-view shard (views i with i % 8 == rank) on one GPU equals the same
      slice of the unsharded CPU oracle, vote for vote
  C5  4 999 122-face terrain, 6000x4000, 10 classes
  keys  winner keys are pixel + 1: a 16384 x 16384 view (2^28 pixels) and 200 classes at 6000x4000
"""
import numpy as np
import pytest
import torch

from geograypher_amd.utils import synthetic
from oracle import oracle_c

pytestmark = pytest.mark.gpu

H, W = 3000, 4000


def _hash32(x):
    m = 0xFFFFFFFF
    x = x & m
    x = x ^ (x >> 16)
    x = (x * 0x7FEB352D) & m
    x = x ^ (x >> 15)
    x = (x * 0x846CA68B) & m
    return x ^ (x >> 16)


def device_labels(ids, view, n_classes, seed_face=4, seed_pix=5):
    """Device twin of synthetic.synthetic_labels (class = hash(face) mod C, 10 % flips, 1 % ignore)."""
    flat = ids.reshape(-1).to(torch.int64)
    cls = _hash32((flat & 0xFFFFFFFF) ^ seed_face) % n_classes
    pix = torch.arange(flat.numel(), dtype=torch.int64, device=ids.device)
    r = _hash32(pix * 2654435761 + view * 40503 + seed_pix)
    u = r % 1000
    cls = torch.where(u < 100, (r >> 10) % n_classes, cls)
    cls = torch.where(u >= 990, torch.full_like(cls, 255), cls)
    return cls.to(torch.uint8).reshape(ids.shape)


def test_device_label_generator_equals_the_host_one(hip):
    ids = torch.randint(-1, 50000, (97, 131), dtype=torch.int32, device=hip.device)
    for view in (0, 7):
        np.testing.assert_array_equal(device_labels(ids, view, 4).cpu().numpy(),
                                      synthetic.synthetic_labels(ids.cpu().numpy(), view, 4))


@pytest.fixture(scope="module")
def terrain():
    return synthetic.terrain_mesh()


def _oracle_votes(points, faces, recs, labels_np, views, h, w, C, compat=True, ids_check=None):
    F = faces.shape[0]
    votes = np.zeros((F, C), dtype=np.uint32)
    counts = np.zeros(F, dtype=np.uint32)
    for k, v in enumerate(views):
        want = oracle_c.raster(points, faces, recs[v], h, w)
        if ids_check is not None:
            assert np.array_equal(ids_check[k], want), f"view {v}: ids differ from the oracle"
        oracle_c.project_labels(want, labels_np[k], F, C, votes, counts, neg1_is_last_face=compat)
    return votes, counts


def test_config3_fused_aggregation_130_views(hip, terrain):
    points, faces = terrain
    F, C, n = faces.shape[0], 4, 130
    cams = synthetic.config3_cameras(n)
    recs = cams.get_raster_records(1.0, near=1.0)
    hip.upload_mesh(points.astype(np.float32), faces.astype(np.int32))
    recs_t = torch.from_numpy(recs).to(hip.device)
    labels = torch.empty((n, H, W), dtype=torch.uint8, device=hip.device)
    votes_unfused, counts_unfused = hip.new_vote_buffers(C)
    visible = 0
    sample = [0, 64, 129]  # first group, first view of the second group, last view of the remainder
    sample_ids = {}
    for c0 in range(0, n, 26):  # ids are only materialised chunk-wise (labels derive from them; the unfused reference too)
        c1 = min(c0 + 26, n)
        ids = hip.raster_face_ids(recs_t[c0:c1], H, W)
        for k in range(c1 - c0):
            labels[c0 + k] = device_labels(ids[k], c0 + k, C)
            visible += int(torch.unique(ids[k]).numel()) - int((ids[k] == -1).any())
            if c0 + k in sample:
                sample_ids[c0 + k] = ids[k].cpu().numpy()
        hip.project_labels(ids, labels[c0:c1], C, votes_unfused, counts_unfused, neg1_is_last_face=False)
    # the fused call over all 130 views: two full launch groups + a remainder of 2
    votes, counts = hip.new_vote_buffers(C)
    hip.raster_project_labels(recs_t, labels, C, votes, counts, neg1_is_last_face=False)
    assert hip.last_stats["overflow"] == 0
    assert torch.equal(votes, votes_unfused) and torch.equal(counts, counts_unfused)
    # conservation: every face a view shows is counted exactly once for that view; votes never exceed observations
    assert int(counts.to(torch.int64).sum()) == visible
    assert bool((votes.to(torch.int64).sum(dim=1) <= counts.to(torch.int64)).all())
    # additivity over any split of the views
    va, ca = hip.new_vote_buffers(C)
    hip.raster_project_labels(recs_t[:70], labels[:70], C, va, ca, neg1_is_last_face=False)
    hip.raster_project_labels(recs_t[70:], labels[70:], C, va, ca, neg1_is_last_face=False)
    assert torch.equal(va, votes) and torch.equal(ca, counts)
    # the contribution of three sampled views against the oracle, bit for bit (with the reference's -1 aliasing)
    vs, cs = hip.new_vote_buffers(C)
    idx = torch.tensor(sample, device=hip.device)
    hip.raster_project_labels(recs_t[idx], labels[idx], C, vs, cs, neg1_is_last_face=True)
    want_v, want_c = _oracle_votes(points, faces, recs, labels[idx].cpu().numpy(), sample, H, W, C, compat=True,
                                   ids_check=[sample_ids[v] for v in sample])
    np.testing.assert_array_equal(vs.cpu().numpy().view(np.uint32), want_v)
    np.testing.assert_array_equal(cs.cpu().numpy().view(np.uint32), want_c)
    assert want_c.max() >= 2 and int(want_c[-1]) >= 1  # faces seen by more than one view; background lands on the last face


def test_config4_one_rank_shard_equals_the_oracle_slice(hip, terrain):
    points, faces = terrain
    F, C, world, rank = faces.shape[0], 4, 8, 3
    cams4 = synthetic.config4_cameras()
    assert len(cams4) == 2000
    from geograypher_amd.distributed import shard_views

    mine = shard_views(len(cams4), rank, world)
    assert len(mine) == 250 and mine[0] == rank and mine[1] == rank + world
    recs_all = cams4.get_raster_records(1.0, near=1.0)
    recs = recs_all[mine]
    hip.upload_mesh(points.astype(np.float32), faces.astype(np.int32))
    recs_t = torch.from_numpy(recs).to(hip.device)
    labels = torch.empty((len(mine), H, W), dtype=torch.uint8, device=hip.device)
    want_v = np.zeros((F, C), dtype=np.uint32)
    want_c = np.zeros(F, dtype=np.uint32)
    step = 50
    for c0 in range(0, len(mine), step):
        c1 = min(c0 + step, len(mine))
        ids = hip.raster_face_ids(recs_t[c0:c1], H, W)
        for k in range(c1 - c0):
            labels[c0 + k] = device_labels(ids[k], mine[c0 + k], C)
        # the unsharded oracle restricted to this rank's slice: rasterized and projected on the CPU, view by view
        want_ids, _ = oracle_c.raster_views(points, faces, recs[c0:c1], H, W, n_threads=32)
        ids_np = ids.cpu().numpy()
        assert np.array_equal(ids_np, want_ids), f"views {mine[c0]}..{mine[c1 - 1]}: ids differ from the oracle"
        lab_np = labels[c0:c1].cpu().numpy()
        for k in range(c1 - c0):
            oracle_c.project_labels(want_ids[k], lab_np[k], F, C, want_v, want_c)
    votes, counts = hip.new_vote_buffers(C)
    hip.raster_project_labels(recs_t, labels, C, votes, counts)  # the shard in ONE fused call: 3 full launch groups + 58
    assert hip.last_stats["overflow"] == 0
    np.testing.assert_array_equal(votes.cpu().numpy().view(np.uint32), want_v)
    np.testing.assert_array_equal(counts.cpu().numpy().view(np.uint32), want_c)
    assert int(want_c.max()) > 20


def test_config5_five_million_faces(hip):
    (points, faces), cams = synthetic.config5_scene(n_views=6)
    assert faces.shape[0] == 4_999_122
    h, w, C = 4000, 6000, 10
    assert cams[0].get_image_size() == (h, w)
    F = faces.shape[0]
    recs = cams.get_raster_records(1.0, near=1.0)
    hip.upload_mesh(points.astype(np.float32), faces.astype(np.int32))
    ids = hip.raster_face_ids(recs, h, w)
    assert hip.last_stats["overflow"] == 0
    ids_np = ids.cpu().numpy()
    labels = torch.stack([device_labels(ids[v], v, C) for v in range(len(cams))])
    pick = [0, 5]
    for v in pick:
        want = oracle_c.raster(points, faces, recs[v], h, w)
        assert np.array_equal(ids_np[v], want), f"view {v} differs in {(ids_np[v] != want).sum()} pixels"
    assert ids_np.min() >= -1 and ids_np.max() < F and (ids_np >= 0).mean() > 0.5
    votes, counts = hip.new_vote_buffers(C)
    hip.raster_project_labels(recs, labels, C, votes, counts)
    vu, cu = hip.new_vote_buffers(C)
    hip.project_labels(ids, labels, C, vu, cu)
    assert torch.equal(votes, vu) and torch.equal(counts, cu)
    vs, cs = hip.new_vote_buffers(C)
    idx = torch.tensor(pick, device=hip.device)
    hip.raster_project_labels(torch.from_numpy(recs).to(hip.device)[idx], labels[idx], C, vs, cs)
    want_v, want_c = _oracle_votes(points, faces, recs, labels[idx].cpu().numpy(), pick, h, w, C)
    np.testing.assert_array_equal(vs.cpu().numpy().view(np.uint32), want_v)
    np.testing.assert_array_equal(cs.cpu().numpy().view(np.uint32), want_c)
    assert (want_v.sum(axis=0) > 0).all()  # all ten classes receive votes


def test_many_classes_and_largest_image_winner_keys(hip):
    """Winner keys are pixel + 1 in 32 bits whatever the number of classes (the label is looked up by the vote kernel):
    200 classes at 6000x4000, and the largest image the library accepts (16384 x 16384 = 2^28 pixels), fused and unfused,
    against the oracle."""
    (points, faces), _ = synthetic.config1_scene()
    F = faces.shape[0]
    hip.upload_mesh(points.astype(np.float32), faces.astype(np.int32))
    for (h, w, f, C, n) in ((4000, 6000, 3000.0, 200, 2), (16384, 16384, 8000.0, 4, 1)):
        poses = [synthetic.nadir_pose(3.0 - 4 * k, -2.0 + k, 40.0, yaw_deg=20.0 + 50 * k) for k in range(n)]
        cams = synthetic.camera_set_from_poses(poses, f=f, width=w, height=h)
        recs = cams.get_raster_records(1.0, near=0.05)
        ids = hip.raster_face_ids(recs, h, w)
        labels = torch.stack([device_labels(ids[v], v, C) for v in range(n)])
        assert int(labels[labels != 255].max()) >= min(C, 150) - 1
        vf, cf = hip.new_vote_buffers(C)
        hip.raster_project_labels(recs, labels, C, vf, cf)
        vu, cu = hip.new_vote_buffers(C)
        hip.project_labels(ids, labels, C, vu, cu)
        assert torch.equal(vf, vu) and torch.equal(cf, cu)
        want_v, want_c = _oracle_votes(points, faces, recs, labels.cpu().numpy(), list(range(n)), h, w, C,
                                       ids_check=list(ids.cpu().numpy()))
        np.testing.assert_array_equal(vf.cpu().numpy().view(np.uint32), want_v)
        np.testing.assert_array_equal(cf.cpu().numpy().view(np.uint32), want_c)
        # the face under the bottom-right pixel wins with the largest key there is: pixel index h*w - 1, key h*w
        last = int(ids[-1, -1, -1])
        assert int(want_c[last if last >= 0 else F - 1]) >= 1
        del ids, labels


def test_pix2face_host_copy_pipeline_matches_the_device_ids(hip):
    """pix2face -> int64 numpy crosses PCIe as int32 through a pinned ring while host threads widen (meshes._ids_to_host_int64):
    the numpy result must equal the device tensor of the same call, for view counts that do and do not fill the ring's
    steps, and for the short path taken by small outputs."""
    from geograypher_amd.meshes import TexturedPhotogrammetryMesh
    from geograypher_amd.meshes.meshes import _ids_to_host_int64

    (points, faces), cams = synthetic.config1_scene()
    for c in cams.cameras:
        c.image_width, c.image_height, c.image_size, c.f = 2000, 1500, (1500, 2000), 1500.0
    mesh = TexturedPhotogrammetryMesh((points, faces), log_level="ERROR")
    for n in (1, 3, 7):
        want = mesh.pix2face(cams[0:n], apply_distortion=False, return_tensor=True)
        got = mesh.pix2face(cams[0:n], apply_distortion=False)
        assert got.dtype == np.int64 and got.shape == tuple(want.shape)
        np.testing.assert_array_equal(got, want.cpu().numpy().astype(np.int64))
    t = torch.randint(-1, 1 << 30, (5, 1024, 1024), dtype=torch.int32, device="cuda")
    for step, threads in ((1, 1), (2, 3), (4, 16), (8, 5)):
        np.testing.assert_array_equal(_ids_to_host_int64(t, step, threads), t.cpu().numpy().astype(np.int64))
    small = torch.arange(-5, 95, dtype=torch.int32, device="cuda").reshape(1, 10, 10)
    np.testing.assert_array_equal(_ids_to_host_int64(small), small.cpu().numpy().astype(np.int64))


def test_cameras_inside_the_canopy_bit_exact(hip):
    """The hostile scene seen from INSIDE (the case the near plane and the guard band decide): cameras 3 m and 12 m above the
    ground between the trees, looking horizontally and 20 degrees down, near plane 5 cm.  Trunks and canopies a metre away are
    thousands of pixels wide, cross the near plane (R7: clipped, not dropped) and the +-16384 px guard band; hundreds of
    thousands of faces lie behind the camera.  Ids and depth bits must equal the oracle's at 1000 x 750, and the view must not
    be empty."""
    points, faces = synthetic.forest_scene()
    height_fn = synthetic._spectrum(1)
    poses = []
    # two cameras 2 m above the ground between the trunks (1.6-2 m from the nearest), one 30 m up just above the tree tops
    # looking across them, one 0.4 m from a trunk (the trunk fills the picture: a handful of faces, each thousands of pixels)
    for k, (x, y, agl, tilt) in enumerate([(12.0, -7.0, 3.0, 90.0), (14.9, -141.7, 2.0, 88.0), (-109.8, -29.1, 2.0, 100.0),
                                           (76.1, 11.4, 30.0, 75.0)]):
        ground = float(height_fn(np.array(x), np.array(y)))
        poses.append(synthetic.nadir_pose(x, y, ground + agl, yaw_deg=40.0 * k, tilt_x_deg=tilt, tilt_y_deg=2.0 * k))
    cams = synthetic.camera_set_from_poses(poses, f=3000.0, width=4000, height=3000)
    h, w = cams[0].get_image_size(0.25)
    recs = cams.get_raster_records(0.25, near=0.05)
    hip.upload_mesh(points.astype(np.float32), faces.astype(np.int32))
    ids, depth = hip.raster_face_ids(recs, h, w, want_depth=True)
    ids, depth = ids.cpu().numpy(), depth.cpu().numpy()
    assert hip.last_stats["overflow"] == 0
    for v in range(len(cams)):
        want, wdep = oracle_c.raster(points, faces, recs[v], h, w, want_depth=True)
        bad = np.argwhere(ids[v] != want)
        assert bad.size == 0, f"view {v}: {bad.shape[0]} pixels differ, first {bad[:5].tolist()}"
        np.testing.assert_array_equal(depth[v].view(np.int32), wdep.view(np.int32))
        assert (want >= 0).mean() > 0.5 and len(np.unique(want)) > (100 if v > 0 else 2)
        if v == 0:  # the trunk 8 cm in front of the lens: faces that cross the near plane are in the picture
            assert wdep[np.isfinite(wdep)].min() < 0.2


def test_irregular_tin_workload_full_size_and_quarter_scale(hip):
    """The bench's workload_3 (synthetic.tin_mesh: 1.2 M faces, log-normal triangle areas, Delaunay slivers, folded bumps; the
    caller's face order is spatially incoherent) under a C2 camera: ids at 4000 x 3000 and at the reference's aggregate scale
    0.25, and the fused votes of the view, bit for bit against the oracle."""
    hip.set_option(2, 5); hip.set_option(6, 512); hip.set_option(7, 0); hip.set_option(3, 64)
    pts, faces = synthetic.tin_mesh()
    F, C = faces.shape[0], 4
    cams = synthetic.config2_cameras(50)
    hip.upload_mesh(pts.astype(np.float32), faces.astype(np.int32))
    for scale in (1.0, 0.25):
        h, w = cams[0].get_image_size(scale)
        recs = cams.get_raster_records(scale, near=1.0)[[23, 24]]
        for _ in range(2):   # the second call runs with what the first one taught the library (slots per tile, micro lists)
            ids = hip.raster_face_ids(recs, h, w).cpu().numpy()
            for v in range(2):
                np.testing.assert_array_equal(ids[v], oracle_c.raster(pts, faces, recs[v], h, w))
        labels = np.stack([synthetic.synthetic_labels(ids[v], v, C) for v in range(2)])
        votes, counts = hip.new_vote_buffers(C)
        hip.raster_project_labels(recs, labels, C, votes, counts)
        want_v, want_c = np.zeros((F, C), dtype=np.uint32), np.zeros(F, dtype=np.uint32)
        for v in range(2):
            oracle_c.project_labels(ids[v], labels[v], F, C, want_v, want_c)
        np.testing.assert_array_equal(votes.cpu().numpy().view(np.uint32), want_v)
        np.testing.assert_array_equal(counts.cpu().numpy().view(np.uint32), want_c)
        assert (ids >= 0).mean() > 0.9 and want_c.sum() > 50_000
