#!/usr/bin/env python3
"""tools/small_scale_rate.py -- dense mesh at a small render scale (5 M faces, 1500 x 1000): tiles receive thousands of
entries, the single-pass binning has to learn its segment size.  Prints views/s for the learned single-pass path and
for exact two-pass binning.  GPU box only."""
import json
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from geograypher_amd._hip import HipRaster
from geograypher_amd.utils import synthetic
from oracle import oracle_c

points, faces = synthetic.terrain_mesh(1582, 800.0)
cams = synthetic.survey_cameras(5, 4, 15.0, 18.0, agl=150.0, f=4500.0, width=6000, height=4000, seed=6)
H, W = 1000, 1500
recs_np = cams.get_raster_records(0.25, near=1.0)
hip = HipRaster(0)
hip.upload_mesh(points.astype(np.float32), faces.astype(np.int32))
recs = torch.from_numpy(recs_np).cuda()
out = {}
for name, cap in (("single_pass_learned", 512), ("exact_two_pass", 0)):
    hip.set_option(6, cap)
    ids = hip.raster_face_ids(recs, H, W)  # learns / sizes
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        hip.raster_face_ids(recs, H, W, out=ids, check=False)
    torch.cuda.synchronize()
    out[name + "_views_per_s"] = round(5 * recs.shape[0] / (time.perf_counter() - t0), 1)
    out[name + "_max_entries_per_tile_or_view"] = hip.raster_status()["max_entries"]
    want = oracle_c.raster(points, faces, recs_np[3], H, W)
    assert np.array_equal(ids[3].cpu().numpy(), want), name
print(json.dumps(out))
