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


def test_slots_per_tile_set_by_hand_to_an_odd_size(hip):
    """80 slots per tile: not a whole number of 64-entry chunks, so the call starts with 48-byte entries; the C1 views put
    more than 80 entries into a tile, the call overflows, learns a segment size (a multiple of 64: 40-byte entries from then
    on) and finishes -- ids equal the oracle's either way."""
    (points, faces), cams = synthetic.config1_scene()
    recs = _records(cams)
    hip.upload_mesh(points.astype(np.float32), faces.astype(np.int32))
    hip.set_option(6, 80)
    ids = hip.raster_face_ids(recs, 480, 640)
    assert hip.last_retries >= 1 and hip.last_stats["max_entries"] > 80
    for v in (0, 5):
        np.testing.assert_array_equal(ids[v].cpu().numpy(), oracle_c.raster(points, faces, recs[v], 480, 640))
    again = hip.raster_face_ids(recs, 480, 640)
    assert hip.last_retries == 0 and torch.equal(again, ids)
