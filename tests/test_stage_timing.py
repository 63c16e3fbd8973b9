"""-m gpu: the stage timing of the library (gr_set_profiling / gr_get_stage_times).  The two stages of a raster call are timed by stop
events attached to the kernel launches themselves (hipExtLaunchKernelGGL; csrc/gr_internal.hpp chain_begin / chain_stop): results must
not depend on it, and the spans must add up to what the host clock sees."""
import time

import numpy as np
import pytest
import torch

from geograypher_amd.utils import synthetic
from oracle import oracle_c

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _defaults(hip):
    hip.set_option(2, 5); hip.set_option(6, 512); hip.set_option(7, 0); hip.set_option(3, 64)
    yield
    hip.set_profiling(False)
    hip.set_option(6, 512); hip.set_option(7, 0); hip.set_option(3, 64)


@pytest.mark.parametrize("cap,var,batch", [(512, 0, 64), (0, 0, 64), (512, 0, 3), (512, 16384, 64)])
def test_results_do_not_depend_on_the_timing_events(hip, cap, var, batch):
    """ids, depth and fused votes with the spans on: single-pass and exact binning (scan / fill spans in the chain), eager view
    totals with several launch groups, the status-call protocol."""
    (points, faces), cams = synthetic.config1_scene()
    recs = cams.get_raster_records(1.0, near=0.05)
    hip.upload_mesh(points.astype(np.float32), faces.astype(np.int32))
    hip.set_option(6, cap); hip.set_option(7, var); hip.set_option(3, batch)
    hip.set_profiling(True)
    ids, dep = hip.raster_face_ids(recs, 480, 640, want_depth=True)
    ids2 = hip.raster_face_ids(recs, 480, 640)
    C = 3
    labels = np.stack([synthetic.synthetic_labels(ids[v].cpu().numpy(), v, C) for v in range(len(cams))])
    votes, counts = hip.new_vote_buffers(C)
    hip.raster_project_labels(recs, labels, C, votes, counts)
    st = hip.stage_times()
    hip.set_profiling(False)
    assert torch.equal(ids, ids2)
    want_v = np.zeros((faces.shape[0], C), dtype=np.uint32)
    want_c = np.zeros(faces.shape[0], dtype=np.uint32)
    for v in range(len(cams)):
        want, wdep = oracle_c.raster(points, faces, recs[v], 480, 640, want_depth=True)
        np.testing.assert_array_equal(ids[v].cpu().numpy(), want)
        np.testing.assert_array_equal(dep[v].cpu().numpy().view(np.int32), wdep.view(np.int32))
        oracle_c.project_labels(want, labels[v], faces.shape[0], C, want_v, want_c)
    np.testing.assert_array_equal(votes.cpu().numpy().view(np.uint32), want_v)
    np.testing.assert_array_equal(counts.cpu().numpy().view(np.uint32), want_c)
    assert st["setup_ms"] > 0 and st["raster_ms"] > 0 and st["views"] >= 3 * len(cams)
    if cap == 0:
        assert st["scan_ms"] > 0 and st["fill_ms"] > 0


def test_spans_add_up_to_the_host_clock(hip):
    """Back-to-back C2-like calls: set-up + tile kernel spans of the timed calls lie within the wall time of the loop and cover
    most of it (the init kernel and its gap are the only device time outside the two spans)."""
    points, faces = synthetic.terrain_mesh(300, 150.0)
    cams = synthetic.config2_cameras(12)
    H, W = cams[0].get_image_size(0.5)
    recs = torch.from_numpy(cams.get_raster_records(0.5, near=1.0)).cuda()
    hip.upload_mesh(points.astype(np.float32), faces.astype(np.int32))
    ids = torch.empty((len(cams), H, W), dtype=torch.int32, device="cuda")
    hip.raster_face_ids(recs, H, W, out=ids, check=True)
    for _ in range(30):
        hip.raster_face_ids(recs, H, W, out=ids, check=False)
    hip.set_profiling(True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 40
    for _ in range(n):
        hip.raster_face_ids(recs, H, W, out=ids, check=False)
    torch.cuda.synchronize()
    wall_ms = (time.perf_counter() - t0) * 1e3
    st = hip.stage_times()
    hip.set_profiling(False)
    spans = st["setup_ms"] + st["raster_ms"]
    assert st["raster_launches"] == n and st["views"] == n * len(cams)
    assert 0.5 * wall_ms < spans <= 1.02 * wall_ms, (spans, wall_ms, st)
