#!/usr/bin/env python3
"""tools/prof_forest.py [reps] -- pix2face on the hostile workload (terrain + 20 000 trees, 20 oblique views) at 4000x3000 and at
1000x750 and nothing else, for rocprofv3 kernel-trace / --pmc passes (no torch kernels).  The first call of each size sizes the
tile segments (and, at quarter scale, may switch micro lists on); the traced calls after it run sized."""
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from geograypher_amd._hip import HipRaster
from geograypher_amd.utils import synthetic

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 4
points, faces = synthetic.forest_scene()
cams = synthetic.oblique_cameras(20)
hip = HipRaster(0)
hip.upload_mesh(points.astype(np.float32), faces.astype(np.int32))
for scale in (1.0, 0.25):
    h, w = cams[0].get_image_size(scale)
    recs = torch.from_numpy(cams.get_raster_records(scale, near=1.0)).cuda()
    ids = torch.empty((recs.shape[0], h, w), dtype=torch.int32, device="cuda")
    for _ in range(2):
        hip.raster_face_ids(recs, h, w, out=ids, check=True)
    for _ in range(reps):
        hip.raster_face_ids(recs, h, w, out=ids, check=False)
    torch.cuda.synchronize()
    print("scale", scale, hip.raster_status())
    del ids
