#!/usr/bin/env python3
"""tools/prof_c5.py [views] [reps] -- pix2face on BASELINE config 5 (5 M faces, 6000x4000) and nothing else, for rocprofv3 --pmc
passes (no torch kernels)."""
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from geograypher_amd._hip import HipRaster
from geograypher_amd.utils import synthetic

nv = int(sys.argv[1]) if len(sys.argv) > 1 else 20
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
(points, faces), cams = synthetic.config5_scene(n_views=nv)
hip = HipRaster(0)
hip.upload_mesh(points.astype(np.float32), faces.astype(np.int32))
recs = torch.from_numpy(cams.get_raster_records(1.0, near=1.0)[:nv]).cuda()
ids = torch.empty((nv, 4000, 6000), dtype=torch.int32, device="cuda")
hip.raster_face_ids(recs, 4000, 6000, out=ids, check=True)
for _ in range(reps):
    hip.raster_face_ids(recs, 4000, 6000, out=ids, check=False)
torch.cuda.synchronize()
print("done", hip.raster_status())
