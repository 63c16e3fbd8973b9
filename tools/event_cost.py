#!/usr/bin/env python3
"""tools/event_cost.py -- what the library's HIP-event spans (gr_set_profiling) cost a C2 step: 50-view pix2face calls back to back,
timed with the spans on and off, alternated.  GPU box only."""
import json
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from geograypher_amd._hip import HipRaster
from geograypher_amd.utils import synthetic

scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
points, faces = synthetic.terrain_mesh()
cams = synthetic.config2_cameras(50)
H, W = cams[0].get_image_size(scale)
recs = torch.from_numpy(cams.get_raster_records(scale, near=1.0)).cuda()
hip = HipRaster(0)
hip.upload_mesh(points.astype(np.float32), faces.astype(np.int32))
ids = torch.empty((50, H, W), dtype=torch.int32, device="cuda")
hip.raster_face_ids(recs, H, W, out=ids, check=True)
for _ in range(20):
    hip.raster_face_ids(recs, H, W, out=ids, check=False)
res = {False: [], True: []}
for rep in range(6):
    for prof in ((False, True) if rep % 2 == 0 else (True, False)):
        hip.set_profiling(prof)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(60):
            hip.raster_face_ids(recs, H, W, out=ids, check=False)
        torch.cuda.synchronize()
        res[prof].append((time.perf_counter() - t0) / 60 * 1e3)
        if prof:
            hip.stage_times()
        hip.set_profiling(False)
print(json.dumps({"image": f"{W}x{H}", "ms_per_step_spans_off": [round(x, 4) for x in res[False]], "ms_per_step_spans_on": [round(x, 4) for x in res[True]],
                  "median_off": round(float(np.median(res[False])), 4), "median_on": round(float(np.median(res[True])), 4)}))
hip.set_profiling(True)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(60):
    hip.raster_face_ids(recs, H, W, out=ids, check=False)
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / 60 * 1e3
st = hip.stage_times()
hip.set_profiling(False)
print(json.dumps({"wall_ms_per_step_of_these_calls": round(wall, 4), "spans_ms_per_step": round((st["setup_ms"] + st["raster_ms"]) / st["raster_launches"], 4), "stage_us_per_view": {k: round(st[k] / st["views"] * 1e3, 3) for k in ("setup_ms", "scan_ms", "raster_ms")}, "launches": st["raster_launches"]}))
