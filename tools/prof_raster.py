#!/usr/bin/env python3
"""tools/prof_raster.py [variant_bits] [tile_h_log2] [views] [reps] -- run only the pix2face pipeline on the C2 workload
(for rocprofv3 --pmc passes: no torch kernels, no CPU baseline)."""
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from geograypher_amd._hip import HipRaster
from geograypher_amd.utils import synthetic

kernel = int(sys.argv[1]) if len(sys.argv) > 1 else 1
thl = int(sys.argv[2]) if len(sys.argv) > 2 else 5
nv = int(sys.argv[3]) if len(sys.argv) > 3 else 32
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
points, faces = synthetic.terrain_mesh()
cams = synthetic.config2_cameras(50)
recs = torch.from_numpy(cams.get_raster_records(1.0, near=1.0)[:nv]).cuda()
hip = HipRaster(0)
hip.set_option(2, thl)
hip.set_option(7, kernel)  # GR_OPT_VARIANT
hip.upload_mesh(points.astype(np.float32), faces.astype(np.int32))
ids = torch.empty((nv, 3000, 4000), dtype=torch.int32, device="cuda")
for _ in range(reps):
    hip.raster_face_ids(recs, 3000, 4000, out=ids, check=False)
torch.cuda.synchronize()
print("done", hip.raster_status())
