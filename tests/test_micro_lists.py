"""-m gpu: micro lists (round 5).  A view whose faces are mostly at most 4 x 4 pixels -- a mesh rendered at a fraction of its photos'
resolution, the reference's operating point (AGGREGATE_IMAGE_SCALE = 0.25, examples/aggregate_predictions.ipynb:60-61) --
teaches the library to keep, for that mesh and image size, a second list per tile for such faces; the tile kernel point-samples
them one face per lane.  Results are bit-identical either way; the switch is remembered like the slots per tile."""
import numpy as np
import pytest
import torch

from geograypher_amd.utils import synthetic
from oracle import oracle_c

pytestmark = pytest.mark.gpu


def _small_face_scene():
    """A 142 x 142-vertex terrain (39 762 faces) seen from 8 cameras at 320 x 240: a face is about 2 pixels wide."""
    points, faces = synthetic.heightfield_mesh(142, 100.0, lambda x, y: 0.5 * np.sin(x / 7.0) + 0.5 * np.cos(y / 5.0), jitter=0.3, seed=1)
    poses = [synthetic.nadir_pose(0.0, 0.0, 40.0)]
    for k in range(7):
        ang = 2 * np.pi * k / 7
        poses.append(synthetic.look_at((30 * np.cos(ang), 30 * np.sin(ang), 40.0), (0.0, 0.0, 0.0), up_hint=(0, 0, 1)))
    cams = synthetic.camera_set_from_poses(poses, f=250.0, width=320, height=240)
    return points, faces, cams.get_raster_records(1.0, near=0.05)


def test_micro_lists_are_learned_persisted_and_bit_exact(tmp_path):
    from geograypher_amd._hip import HipRaster, load_library

    lib = load_library()
    cache = tmp_path / "learned.txt"
    assert lib.gr_learned_cache_clear() == 0
    assert lib.gr_learned_cache_file(str(cache).encode()) == 0
    try:
        points, faces, recs = _small_face_scene()
        want = [oracle_c.raster(points, faces, recs[v], 240, 320) for v in range(recs.shape[0])]
        a = HipRaster(0)
        a.upload_mesh(points.astype(np.float32), faces.astype(np.int32))
        # a mesh and image size nothing is known about: the first launch group is binned with ordinary lists, its counts say
        # "mostly micro faces", the library bins it again with micro lists before any tile kernel runs ...
        first = a.raster_face_ids(recs, 240, 320).cpu().numpy()
        assert a.last_retries == 0 and a.last_stats["rebinned_groups"] == 1
        lines = [l.split() for l in cache.read_text().splitlines() if not l.startswith("#")]
        assert any(len(l) == 5 and l[4] == "1" for l in lines), lines   # ... and the table says: micro lists for this image
        second = a.raster_face_ids(recs, 240, 320).cpu().numpy()     # micro lists from the start
        assert a.last_retries == 0 and a.last_stats["rebinned_groups"] == 0
        b = HipRaster(0)                                             # another context: starts with them
        b.upload_mesh(points.astype(np.float32), faces.astype(np.int32))
        third, depth = b.raster_face_ids(recs, 240, 320, want_depth=True)
        for v in range(recs.shape[0]):
            np.testing.assert_array_equal(first[v], want[v])
            np.testing.assert_array_equal(second[v], want[v])
            np.testing.assert_array_equal(third[v].cpu().numpy(), want[v])
            _, wdep = oracle_c.raster(points, faces, recs[v], 240, 320, want_depth=True)
            np.testing.assert_array_equal(depth[v].cpu().numpy().view(np.int32), wdep.view(np.int32))
        # fused aggregation through the micro lists: the same votes as from the id images
        C = 4
        labels = np.stack([synthetic.synthetic_labels(want[v], v, C) for v in range(recs.shape[0])])
        votes, counts = b.new_vote_buffers(C)
        b.raster_project_labels(recs, labels, C, votes, counts)
        want_v = np.zeros((faces.shape[0], C), dtype=np.uint32)
        want_c = np.zeros(faces.shape[0], dtype=np.uint32)
        for v in range(recs.shape[0]):
            oracle_c.project_labels(want[v], labels[v], faces.shape[0], C, want_v, want_c)
        np.testing.assert_array_equal(votes.cpu().numpy().view(np.uint32), want_v)
        np.testing.assert_array_equal(counts.cpu().numpy().view(np.uint32), want_c)
        # a full-size view of the same mesh (faces of 20 pixels) does not switch them on
        big = synthetic.camera_set_from_poses([synthetic.nadir_pose(0.0, 0.0, 40.0)], f=2500.0, width=1600, height=1200)
        a.raster_face_ids(big.get_raster_records(1.0, near=0.05), 1200, 1600)
        lines = [l.split() for l in cache.read_text().splitlines() if not l.startswith("#")]
        assert sum(1 for l in lines if len(l) == 5 and l[4] == "1") == 1, lines
        a.close(); b.close()
    finally:
        lib.gr_learned_cache_file(None)
        lib.gr_learned_cache_clear()
