#!/usr/bin/env python3
"""tools/prof_pipeline.py [views] [reps] [fused_views] [extra variant bits] -- the pix2face pipeline (C2 views) and the fused aggregation (C3 views,
votes on the caller's stream) and nothing else, for rocprofv3 --pmc passes: rocprofv3's counter collection does not survive
torch's own elementwise kernels (segfault inside at::native::gpu_kernel_impl) nor the library's side stream, so the labels
come from the host and no torch kernel runs."""
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from geograypher_amd._hip import HipRaster
from geograypher_amd.utils import synthetic

nv = int(sys.argv[1]) if len(sys.argv) > 1 else 50
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
nf = int(sys.argv[3]) if len(sys.argv) > 3 else 64
H, W, C = 3000, 4000, 4
points, faces = synthetic.terrain_mesh()
hip = HipRaster(0)
xvar = int(sys.argv[4]) if len(sys.argv) > 4 else 0
hip.set_option(7, 4 | xvar)  # GR_OPT_VARIANT: fused votes on the caller's stream (+ the variant under study)
hip.upload_mesh(points.astype(np.float32), faces.astype(np.int32))
recs = torch.from_numpy(synthetic.config2_cameras(50).get_raster_records(1.0, near=1.0)[:nv]).cuda()
ids = torch.empty((nv, H, W), dtype=torch.int32, device="cuda")
hip.raster_face_ids(recs, H, W, out=ids, check=True)
for _ in range(reps):
    hip.raster_face_ids(recs, H, W, out=ids, check=False)
torch.cuda.synchronize()
print("pix2face done", hip.raster_status())
if nf > 0:
    recs3 = torch.from_numpy(synthetic.config3_cameras(nf).get_raster_records(1.0, near=1.0)).cuda()
    rng = np.random.default_rng(0)
    labels = torch.from_numpy(rng.integers(0, C, size=(nf, H, W), dtype=np.uint8)).cuda()
    votes = torch.from_numpy(np.zeros((faces.shape[0], C), dtype=np.int32)).cuda()
    counts = torch.from_numpy(np.zeros((faces.shape[0],), dtype=np.int32)).cuda()
    hip.raster_project_labels(recs3, labels, C, votes, counts, check=True)
    for _ in range(reps):
        hip.raster_project_labels(recs3, labels, C, votes, counts, check=False)
    torch.cuda.synchronize()
    print("fused done", hip.raster_status())
