#!/usr/bin/env python3
"""tools/api_rate.py -- throughput of the reference-shaped Python API (numpy in / numpy out, PCIe included) on the C2
workload: pix2face -> (n,h,w) int64 numpy, render_flat -> (h,w,C) float64 numpy, aggregate with host label images."""
import json
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from geograypher_amd.cameras import SegmentorPhotogrammetryCameraSet
from geograypher_amd.meshes import TexturedPhotogrammetryMesh
from geograypher_amd.predictors import ArrayLabelSegmentor
from geograypher_amd.utils import synthetic

points, faces = synthetic.terrain_mesh()
cams = synthetic.config2_cameras(16)
tex = (np.arange(faces.shape[0]) % 4).astype(float)
mesh = TexturedPhotogrammetryMesh((points, faces), texture=tex, IDs_to_labels={i: str(i) for i in range(4)}, log_level="ERROR")
ids = mesh.pix2face(cams[0:2], apply_distortion=False)  # warm up (upload, scratch)
out = {}
t0 = time.perf_counter(); ids = mesh.pix2face(cams, apply_distortion=False); dt = time.perf_counter() - t0
out["pix2face_numpy_int64_views_per_s_first_call"] = round(len(cams) / dt, 1)  # includes pinning 1.5 GB of host memory
del ids  # the pinned block goes back to torch's host allocator and is reused by the next call
t0 = time.perf_counter(); ids = mesh.pix2face(cams, apply_distortion=False); dt = time.perf_counter() - t0
out["pix2face_numpy_int64_views_per_s"] = round(len(cams) / dt, 1)
t0 = time.perf_counter(); t = mesh.pix2face(cams, apply_distortion=False, return_tensor=True); import torch; torch.cuda.synchronize(); dt = time.perf_counter() - t0
out["pix2face_tensor_views_per_s"] = round(len(cams) / dt, 1)
t0 = time.perf_counter(); n = sum(1 for _ in mesh.render_flat(cams, apply_distortion=False)); dt = time.perf_counter() - t0
out["render_flat_numpy_f64_views_per_s"] = round(n / dt, 1)
labels = [synthetic.synthetic_labels(ids[v], v, 4) for v in range(len(cams))]
seg = SegmentorPhotogrammetryCameraSet(cams, ArrayLabelSegmentor(labels, 4, filenames=[c.image_filename for c in cams.cameras]))
mesh.aggregate_projected_images(seg)
t0 = time.perf_counter(); avg, info = mesh.aggregate_projected_images(seg); dt = time.perf_counter() - t0
out["aggregate_host_uint8_labels_views_per_s"] = round(len(cams) / dt, 1)
print(json.dumps(out))
