"""-m gpu: limits and the overflow protocol of the single-pass binning, on the default kernel variant (the variant matrix
of tests/test_hip_parity.py covers the rasterization itself): the largest image the library accepts, and a fused
aggregation call whose later launch groups overflow their tile segments."""
import numpy as np
import pytest
import torch

from geograypher_amd.utils import synthetic
from oracle import oracle_c

pytestmark = pytest.mark.gpu


def _records(cams, scale=1.0, near=0.05):
    return cams.get_raster_records(scale, near=near)


NO_LOOK = 16384  # variant bit: no look at the first launch group's counts -- every overflow goes through gr_raster_status


def _lessons(h):
    """Times the last checked call was taught something: GR_EOVERFLOW retries (gr_raster_status) + times its first launch
    group was binned again after the look at its counts (an unknown mesh / image size; gr_raster_stats.rebinned_groups)."""
    return h.last_retries + h.last_stats["rebinned_groups"]


@pytest.fixture(autouse=True)
def _default_options(hip):
    hip.set_option(2, 5)
    hip.set_option(6, 512)
    hip.set_option(7, 0)
    hip.set_option(3, 64)
    yield
    hip.set_option(3, 64)
    hip.set_option(6, 512)


def test_maximum_image_size(hip):
    """16384 x 16384, the largest image the library accepts (guard band, GR_MAX_DIM): 131 072 tiles.  The fixed tile
    segments of a full launch group would not fit the scratch budget; the group shrinks instead.  One size larger is
    refused."""
    (points, faces), _ = synthetic.config1_scene()
    cams = synthetic.camera_set_from_poses([synthetic.nadir_pose(3.0, -2.0, 40.0, yaw_deg=20.0)] * 2, f=8000.0,
                                           width=16384, height=16384)
    recs = _records(cams)
    hip.upload_mesh(points.astype(np.float32), faces.astype(np.int32))
    ids = hip.raster_face_ids(recs, 16384, 16384)
    want = oracle_c.raster(points, faces, recs[0], 16384, 16384)
    assert torch.equal(ids[0].cpu(), torch.from_numpy(want)) and torch.equal(ids[1], ids[0])
    assert (want >= 0).mean() > 0.5
    with pytest.raises((ValueError, RuntimeError)):
        hip.raster_face_ids(recs, 16385, 16384)



def test_fused_call_resumes_after_a_later_launch_group_overflows(hip):
    """A fused call of three launch groups whose LAST groups overflow their tile segments (far views put many faces in a
    tile, near views few; the slots per tile are chosen between the two needs): the device must fold in exactly the groups
    in front of the first overflowed one, `views_done` must say so, and the resumed call must add the rest -- votes and
    counts equal to the unfused projection of the id images, no view twice, none dropped.  With check=False the same call
    reports nothing by itself; `raster_status()` must raise."""
    (points, faces), _ = synthetic.config1_scene()
    F, C = faces.shape[0], 3
    near = [synthetic.nadir_pose(3.0 * k - 9.0, 2.0 * k - 7.0, 22.0, yaw_deg=20.0 * k) for k in range(6)]
    far = [synthetic.nadir_pose(2.0 * k - 3.0, 1.0 - k, 130.0 + 5.0 * k, yaw_deg=15.0 * k) for k in range(5)]
    cams = synthetic.camera_set_from_poses(near + far, f=500.0, width=640, height=480)
    recs = _records(cams)
    hip.upload_mesh(points.astype(np.float32), faces.astype(np.int32))
    try:
        hip.set_option(6, 4096)
        need = []
        for v in range(len(cams)):
            hip.raster_face_ids(recs[v:v + 1], 480, 640)
            need.append(hip.last_stats["max_entries"])
        cap = (max(need[:6]) + 63) // 64 * 64
        assert cap < min(need[6:]), (need, cap)
        ids = hip.raster_face_ids(recs, 480, 640)
        ids_np = ids.cpu().numpy()
        labels = np.stack([synthetic.synthetic_labels(ids_np[v], v, C) for v in range(len(cams))])
        want_v, want_c = hip.new_vote_buffers(C)
        hip.project_labels(ids, labels, C, want_v, want_c)
        hip.set_option(3, 4)      # groups: views 0-3 | 4-7 (overflows: views 6, 7) | 8-10
        hip.set_option(6, cap)    # forgets what the context has learned
        v2, c2 = hip.new_vote_buffers(C)
        hip.raster_project_labels(recs, labels, C, v2, c2)
        assert hip.last_retries > 0 and hip.last_stats["overflow"] == 0 and hip.last_stats["views_done"] == len(cams)
        assert torch.equal(v2, want_v) and torch.equal(c2, want_c)
        # unchecked: the first group's votes only, and the status call is the one that tells
        hip.set_option(6, cap)
        v3, c3 = hip.new_vote_buffers(C)
        hip.raster_project_labels(recs, labels, C, v3, c3, check=False)
        assert hip.last_stats == {"unchecked": True}
        with pytest.raises(RuntimeError, match="overflow"):
            hip.raster_status()
        part_v, part_c = hip.new_vote_buffers(C)
        hip.project_labels(ids[:4], labels[:4], C, part_v, part_c)
        assert torch.equal(v3, part_v) and torch.equal(c3, part_c)
    finally:
        hip.set_option(3, 64)
        hip.set_option(6, 512)




def test_short_entries_fall_back_when_a_later_group_has_a_large_face(hip):
    """The single-pass binning writes 40-byte entries, which hold faces below 93 pixels.  A call whose SECOND launch group has a
    camera close to the ground (faces of several hundred pixels) must finish the first group in the short form, report the
    miss like an overflow, repeat the rest with 48-byte entries -- same ids and votes as the 48-byte form from the start --
    and remember it: the next call of the context does not retry."""
    (points, faces), _ = synthetic.config1_scene()
    C = 3
    poses = [synthetic.nadir_pose(3.0 * k - 6.0, 2.0 * k - 4.0, 40.0, yaw_deg=20.0 * k) for k in range(4)]
    poses += [synthetic.nadir_pose(1.0, -2.0, 4.5, yaw_deg=10.0), synthetic.nadir_pose(-5.0, 3.0, 35.0, yaw_deg=50.0)]
    cams = synthetic.camera_set_from_poses(poses, f=500.0, width=640, height=480)
    recs = _records(cams)
    hip.upload_mesh(points.astype(np.float32), faces.astype(np.int32))
    hip.set_option(7, 128)                     # 48-byte entries from the start
    want = hip.raster_face_ids(recs, 480, 640)
    assert hip.last_retries == 0
    for v in (0, 4):
        np.testing.assert_array_equal(want[v].cpu().numpy(), oracle_c.raster(points, faces, recs[v], 480, 640))
    labels = np.stack([synthetic.synthetic_labels(want[v].cpu().numpy(), v, C) for v in range(len(cams))])
    want_v, want_c = hip.new_vote_buffers(C)
    hip.project_labels(want, labels, C, want_v, want_c)
    hip.set_option(7, 0)
    hip.set_option(3, 4)                       # groups: views 0-3 (small faces) | 4-5 (view 4: large faces)
    hip.set_option(6, 512)                     # forgets what the context has learned
    got = hip.raster_face_ids(recs[:4], 480, 640)
    assert hip.last_retries == 0 and torch.equal(got, want[:4])          # small faces only: the short form holds them
    v2, c2 = hip.new_vote_buffers(C)
    hip.raster_project_labels(recs, labels, C, v2, c2)
    assert hip.last_retries == 1 and hip.last_stats["views_done"] == len(cams) and hip.last_stats["overflow"] == 0
    assert torch.equal(v2, want_v) and torch.equal(c2, want_c)
    got = hip.raster_face_ids(recs, 480, 640)
    assert hip.last_retries == 0 and torch.equal(got, want)              # remembered


@pytest.mark.parametrize("look", [True, False])
def test_slots_per_tile_set_by_hand_to_an_odd_size(hip, look):
    """80 slots per tile: not a whole number of 64-entry chunks, so the call starts with 48-byte entries; the C1 views put
    more than 80 entries into a tile, the call overflows, learns a segment size (a multiple of 64: 40-byte entries from then
    on) and finishes -- ids equal the oracle's either way.  The overflow is in the call's first launch group: by default the
    library sees it before the tile kernel runs and bins the group again itself; with variant bit 16384 gr_raster_status
    reports it and the caller retries."""
    (points, faces), cams = synthetic.config1_scene()
    recs = _records(cams)
    hip.upload_mesh(points.astype(np.float32), faces.astype(np.int32))
    hip.set_option(7, 0 if look else NO_LOOK)
    hip.set_option(6, 80)
    ids = hip.raster_face_ids(recs, 480, 640)
    assert hip.last_stats["max_entries"] > 80
    assert (hip.last_retries, hip.last_stats["rebinned_groups"]) == ((0, 1) if look else (1, 0))
    for v in (0, 5):
        np.testing.assert_array_equal(ids[v].cpu().numpy(), oracle_c.raster(points, faces, recs[v], 480, 640))
    again = hip.raster_face_ids(recs, 480, 640)
    assert _lessons(hip) == 0 and torch.equal(again, ids)


def _big_face_scene():
    (points, faces), _ = synthetic.config1_scene()
    poses = [synthetic.nadir_pose(1.0, -2.0, 4.5, yaw_deg=10.0)]            # a camera 4.5 m above the ground: faces of hundreds of px
    poses += [synthetic.nadir_pose(3.0 * k - 6.0, 2.0 * k - 4.0, 40.0, yaw_deg=20.0 * k) for k in range(5)]
    cams = synthetic.camera_set_from_poses(poses, f=500.0, width=640, height=480)
    return points, faces, cams, _records(cams)


@pytest.mark.parametrize("look", [False, True])
def test_short_form_miss_leaves_no_stale_entry(hip, look):
    """Round-3 advisor finding: a face the 40-byte entry form cannot hold took a list slot and wrote nothing -- the slot kept
    whatever an earlier call had left there, the fused tile kernel rasterized it and issued winner atomics with a garbage face
    id.  Debug bit 512 poisons every entry slot and row count with 0xFF before each launch group is binned: with the miss in
    the FIRST group of a fused call the result must still be the unfused projection's (the missed slot is a null entry, the
    view's tiles are not walked by the fused kernel, the retry takes 48-byte entries), and the winner scratch must be clean
    for the calls that follow.  (look=False: variant bit 16384, the protocol of rounds 1-4 in which the tile kernel runs on the
    group with the missed face; look=True, the default: the library reads the first group's counts, sees the miss before any
    tile kernel runs and starts over with 48-byte entries by itself.)"""
    points, faces, cams, recs = _big_face_scene()
    C = 3
    hip.upload_mesh(points.astype(np.float32), faces.astype(np.int32))
    hip.set_option(7, 128)
    want = hip.raster_face_ids(recs, 480, 640)
    np.testing.assert_array_equal(want[0].cpu().numpy(), oracle_c.raster(points, faces, recs[0], 480, 640))
    labels = np.stack([synthetic.synthetic_labels(want[v].cpu().numpy(), v, C) for v in range(len(cams))])
    want_v, want_c = hip.new_vote_buffers(C)
    hip.project_labels(want, labels, C, want_v, want_c)
    try:
        hip.set_option(7, 0 if look else NO_LOOK)
        hip.set_option(6, 512)      # forgets the entry form: the call starts with 40-byte entries
        hip.set_option(99, 512)     # poisoned scratch
        v2, c2 = hip.new_vote_buffers(C)
        hip.raster_project_labels(recs, labels, C, v2, c2)
        assert (hip.last_retries, hip.last_stats["rebinned_groups"]) == ((0, 1) if look else (1, 0))
        assert hip.last_stats["views_done"] == len(cams)
        assert torch.equal(v2, want_v) and torch.equal(c2, want_c)
        # nothing stale is left in the winner scratch: the same call again (no retry now) gives the same votes once more
        v3, c3 = hip.new_vote_buffers(C)
        hip.raster_project_labels(recs, labels, C, v3, c3)
        assert _lessons(hip) == 0
        assert torch.equal(v3, want_v) and torch.equal(c3, want_c)
        got = hip.raster_face_ids(recs, 480, 640)
        assert torch.equal(got, want)
    finally:
        hip.set_option(99, 0)


def test_second_context_and_new_process_start_sized(hip, tmp_path):
    """What an overflowed call taught one context is shared process-wide under the MESH SIGNATURE (face count, vertex count,
    vertex bounds) and written to the cache file: a second context for the same mesh and image starts without a retry, a
    context for ANOTHER mesh with as many faces inherits nothing, and a new process that points the library at the same file
    starts sized as well."""
    import os
    import subprocess
    import sys
    from pathlib import Path

    from geograypher_amd._hip import HipRaster, load_library

    lib = load_library()
    cache = tmp_path / "learned.txt"
    assert lib.gr_learned_cache_clear() == 0   # hermetic: nothing an earlier test (or an earlier run's file) taught the process
    assert lib.gr_learned_cache_file(str(cache).encode()) == 0
    try:
        (points, faces), cams = synthetic.config1_scene()
        far = synthetic.camera_set_from_poses([synthetic.nadir_pose(2.0 * k - 3.0, 1.0 - k, 300.0 + 5.0 * k, yaw_deg=15.0 * k)
                                               for k in range(3)], f=500.0, width=640, height=480)
        recs = _records(far)
        a = HipRaster(0)   # fresh contexts: sharing is on (the session fixture has set options by hand)
        a.upload_mesh(points.astype(np.float32), faces.astype(np.int32))
        ids_a = a.raster_face_ids(recs, 480, 640)
        assert _lessons(a) == 1 and a.last_stats["max_entries"] > 512      # the whole mesh in a few tiles
        np.testing.assert_array_equal(ids_a[1].cpu().numpy(), oracle_c.raster(points, faces, recs[1], 480, 640))
        assert cache.is_file() and len([l for l in cache.read_text().splitlines() if not l.startswith("#")]) >= 1
        b = HipRaster(0)
        b.upload_mesh(points.astype(np.float32), faces.astype(np.int32))
        ids_b = b.raster_face_ids(recs, 480, 640)
        assert _lessons(b) == 0 and torch.equal(ids_a, ids_b)
        # another mesh with the same number of faces (vertices moved): nothing inherited -> it overflows by itself
        c = HipRaster(0)
        moved = points.copy()
        moved[:, 0] += 0.125
        c.upload_mesh(moved.astype(np.float32), faces.astype(np.int32))
        c.raster_face_ids(recs, 480, 640)
        assert _lessons(c) == 1
        # a context that opted out learns for itself only
        d = HipRaster(0)
        d.set_option(8, 0)
        d.upload_mesh(points.astype(np.float32), faces.astype(np.int32))
        d.raster_face_ids(recs, 480, 640)
        assert _lessons(d) == 1
        for ctx in (a, b, c, d):
            ctx.close()
        # a NEW PROCESS with the same cache file starts sized
        child = (
            "import sys, numpy as np, torch\n"
            f"sys.path.insert(0, {str(Path(__file__).resolve().parents[1])!r})\n"
            "from geograypher_amd._hip import HipRaster\n"
            "from geograypher_amd.utils import synthetic\n"
            "(points, faces), _ = synthetic.config1_scene()\n"
            "far = synthetic.camera_set_from_poses([synthetic.nadir_pose(2.0 * k - 3.0, 1.0 - k, 300.0 + 5.0 * k, yaw_deg=15.0 * k) for k in range(3)], f=500.0, width=640, height=480)\n"
            "h = HipRaster(0)\n"
            "h.upload_mesh(points.astype(np.float32), faces.astype(np.int32))\n"
            "h.raster_face_ids(far.get_raster_records(1.0, near=0.05), 480, 640)\n"
            "print('RETRIES', h.last_retries + h.last_stats['rebinned_groups'])\n"
        )
        env = dict(os.environ, GEOGRAYPHER_AMD_CACHE=str(tmp_path))
        (tmp_path / "geograster_learned.txt").write_text(cache.read_text())
        res = subprocess.run([sys.executable, "-c", child], env=env, capture_output=True, text=True, timeout=600)
        assert res.returncode == 0, res.stderr[-2000:]
        assert "RETRIES 0" in res.stdout, res.stdout
    finally:
        lib.gr_learned_cache_file(None)
        lib.gr_learned_cache_clear()


def test_entry_memory_budget_option_shrinks_the_launch_group(hip):
    """GR_OPT_DIRECT_BUDGET_MB: with a budget of 64 MiB a launch group of C1 views holds fewer views (70 tiles x 512 slots x
    48 B = 1.7 MB per view: 37 views), results stay bit-exact; the documented scratch formula is what the library uses."""
    (points, faces), cams = synthetic.config1_scene()
    recs = np.concatenate([_records(cams)] * 8, axis=0)   # 64 views
    hip.upload_mesh(points.astype(np.float32), faces.astype(np.int32))
    want = hip.raster_face_ids(recs, 480, 640)
    try:
        hip.set_option(9, 4)     # 4 MiB: two views per launch group
        hip.set_profiling(True)
        got = hip.raster_face_ids(recs, 480, 640)
        st = hip.stage_times()
        hip.set_profiling(False)
        per_view = 10 * 15 * 512 * 48
        assert st["raster_launches"] == -(-64 // ((4 << 20) // per_view)), st
        assert torch.equal(got, want)
        with pytest.raises(ValueError):
            hip.set_option(9, 0)
    finally:
        hip.set_option(9, 24 << 10)


def test_argmax_uses_the_shared_context(hip):
    """find_argmax_nonzero_value(backend=None) takes the device's default backend: no new library context per call."""
    from geograypher_amd import _hip
    from geograypher_amd.utils.indexing import find_argmax_nonzero_value

    arr = np.array([[0.0, 2.0, 1.0], [0.0, 0.0, 0.0], [np.nan, 1.0, 0.0], [3.0, 3.0, 1.0]])
    first = find_argmax_nonzero_value(arr)
    np.testing.assert_array_equal(np.isnan(first), [False, True, True, False])
    assert first[0] == 1 and first[3] == 0
    shared = _hip.default_backend()
    created = []
    orig = _hip.HipRaster.__init__

    def counting(self, *a, **k):
        created.append(1)
        orig(self, *a, **k)

    _hip.HipRaster.__init__ = counting
    try:
        for _ in range(3):
            find_argmax_nonzero_value(arr)
            find_argmax_nonzero_value(torch.from_numpy(arr).cuda())
    finally:
        _hip.HipRaster.__init__ = orig
    assert not created and _hip.default_backend() is shared


def test_clipped_faces_outgrow_the_record_planes_of_exact_binning(hip):
    """Exact binning (slots per tile 0) keeps one record per face -- but a face that straddles the near plane or the guard band
    is clipped into up to six triangles, a record each: a small mesh AROUND the camera needs more records than it has faces.
    Found by tools/fuzz_parity.py (seed 934669: 18 faces, the call failed after four identical retries); the status call now
    reports the need, the planes grow, the retry finishes -- ids equal the oracle's."""
    # six nearly horizontal triangles at different heights, each with two vertices 400 m to the sides in front of the camera
    # (far outside the guard band) and one behind it: near plane + both side planes cut every one of them
    zs = np.linspace(-1.5, 1.0, 6)
    c = np.concatenate([np.array([[-400.0, 1.0, z], [400.0, 1.0, z], [0.0, -5.0, z + 0.2]]) for z in zs])
    faces = np.arange(18, dtype=np.int64).reshape(6, 3)
    poses = [synthetic.look_at((0.0, 0.0, 0.0), (0.0, 10.0, 0.0), up_hint=(0, 0, 1)),
             synthetic.look_at((0.3, -0.2, 0.1), (1.0, 10.0, -0.5), up_hint=(0, 0, 1))]
    cams = synthetic.camera_set_from_poses(poses, f=150.0, width=320, height=240)
    recs = _records(cams, near=0.05)
    hip.upload_mesh(c.astype(np.float32), faces.astype(np.int32))
    hip.set_option(6, 0)
    ids = hip.raster_face_ids(recs, 240, 320)
    assert hip.last_retries == 1, hip.last_stats
    for v in range(len(cams)):
        want = oracle_c.raster(c, faces, recs[v], 240, 320)
        assert (want >= 0).mean() > 0.5
        np.testing.assert_array_equal(ids[v].cpu().numpy(), want)
    again = hip.raster_face_ids(recs, 240, 320)
    assert hip.last_retries == 0 and torch.equal(again, ids)


def test_view_totals_left_to_the_status_call(hip):
    """A call of one launch group that is not fused does not add up its view totals itself (nothing on the device waits for
    them): gr_raster_status does, when asked.  The numbers must be those of the eager form (a call of several launch groups adds
    them up itself: three views per group here), also when other work ran on the stream in between, and an overflow must still
    be reported -- by the status call of an unchecked call too."""
    (points, faces), cams = synthetic.config1_scene()
    recs = _records(cams)
    hip.upload_mesh(points.astype(np.float32), faces.astype(np.int32))
    keys = ("records", "entries", "max_entries", "overflow", "views_done", "blocks")
    hip.set_option(7, 0)
    hip.set_option(6, 512)
    hip.set_option(3, 3)
    want_ids = hip.raster_face_ids(recs, 480, 640)
    want = {k: hip.last_stats[k] for k in keys}
    hip.set_option(3, 64)
    hip.set_option(6, 512)
    hip.raster_face_ids(recs, 480, 640)                       # learns (the look): the next call is an ordinary one
    ids = hip.raster_face_ids(recs, 480, 640, check=False)
    assert hip.last_stats == {"unchecked": True}
    _ = hip.gather_texture(ids[0], np.zeros((faces.shape[0], 2)))   # other work of the context on the stream
    got = hip.raster_status()
    assert {k: got[k] for k in keys} == want and torch.equal(ids, want_ids)
    assert {k: hip.raster_status()[k] for k in keys} == want        # asking twice does not count twice
    # an overflow of an unchecked call (no look: the slots are known to be too few only to us)
    hip.set_option(7, NO_LOOK)
    hip.set_option(6, 64)
    hip.raster_face_ids(recs, 480, 640, check=False)
    with pytest.raises(RuntimeError, match="overflow"):
        hip.raster_status()
    fixed = hip.raster_face_ids(recs, 480, 640)
    assert torch.equal(fixed, want_ids)
