import json, sys, time
import numpy as np
sys.path.insert(0, ".")
from geograypher_amd.cameras import SegmentorPhotogrammetryCameraSet
from geograypher_amd.meshes import TexturedPhotogrammetryMesh
from geograypher_amd.predictors import ArrayLabelSegmentor
from geograypher_amd.utils import synthetic
points, faces = synthetic.terrain_mesh()
cams = synthetic.config2_cameras(48)
mesh = TexturedPhotogrammetryMesh((points, faces), log_level="ERROR")
ids = mesh.pix2face(cams[0:8], apply_distortion=False)
labels = [synthetic.synthetic_labels(ids[v % 8], v, 4) for v in range(len(cams))]
seg = SegmentorPhotogrammetryCameraSet(cams, ArrayLabelSegmentor(labels, 4, filenames=[c.image_filename for c in cams.cameras]))
mesh.aggregate_projected_images(seg)
out = {}
for nt in (None, 4, 8, 16, 32):
    kw = {} if nt is None else {"loader_threads": nt}
    t0 = time.perf_counter(); avg, info = mesh.aggregate_projected_images(seg, **kw); dt = time.perf_counter() - t0
    out[f"threads_{nt}"] = round(len(cams) / dt, 1)
print(json.dumps(out))
