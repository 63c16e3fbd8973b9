#!/usr/bin/env python3
"""tools/two_rank_gpu_check.py -- the product's distributed aggregation with the HIP backend at world size 2 on ONE GPU.

Launch from a shell (no GPU call in the launching process):
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 tools/two_rank_gpu_check.py
Both ranks use cuda:0 and a gloo group (RCCL refuses two ranks on one device; the reduce then goes through host memory:
geograypher_amd.distributed._all_reduce_sum).  Every rank runs TexturedPhotogrammetryMesh.aggregate_projected_images(
distributed=True) over its share of the views -- labels (fused path) and float images (general path) -- and rank 0 compares
with the single-process result of the same call.  Not a substitute for a run on several GPUs: it shows that two contexts of
the library on one device, the view sharding, and the single reduce give the single-process result bit for bit."""
import json
import os
import sys
import time
from pathlib import Path

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    torch.cuda.set_device(0)
    from geograypher_amd._hip import HipRaster
    from geograypher_amd.cameras import PhotogrammetryCameraSet, SegmentorPhotogrammetryCameraSet
    from geograypher_amd.meshes import TexturedPhotogrammetryMesh
    from geograypher_amd.predictors import ArrayLabelSegmentor
    from geograypher_amd.utils import synthetic

    points, faces = synthetic.terrain_mesh()
    cams = synthetic.config2_cameras(50)[0:10]
    scale, C = 1.0, 4
    mesh = TexturedPhotogrammetryMesh((points, faces), log_level="ERROR", backend=HipRaster(0))
    ids = mesh.pix2face(cams, render_img_scale=scale, apply_distortion=False)
    labels = [synthetic.synthetic_labels(ids[v], v, C) for v in range(len(cams))]
    seg_set = SegmentorPhotogrammetryCameraSet(cams, ArrayLabelSegmentor(labels, C, filenames=[c.image_filename for c in cams.cameras]))
    t0 = time.time()
    avg, info = mesh.aggregate_projected_images(seg_set, aggregate_img_scale=scale, distributed=True)
    t_lab = time.time() - t0

    h, w = ids.shape[1:]

    class ImageSet(PhotogrammetryCameraSet):
        def get_image_by_index(self, index, image_scale=1.0):
            # keyed by the camera, not by its position in the (sub)set a rank holds
            rng = np.random.default_rng(1000 + int(str(self.cameras[index].image_filename).split("_")[-1].split(".")[0]))
            img = rng.random((h, w, 3)).astype(np.float32)
            img[rng.random((h, w)) < 0.05] = np.nan
            return img

    img_set = ImageSet(cams.cameras, local_to_epsg_4978_transform=np.eye(4))
    favg, finfo = mesh.aggregate_projected_images(img_set, aggregate_img_scale=scale, distributed=True, apply_distortion=False)
    out = {"rank": rank, "world": world, "device": torch.cuda.get_device_name(0), "views": len(cams), "image": f"{w}x{h}",
           "faces": int(faces.shape[0]), "labels_call_s": round(t_lab, 3)}
    if rank == 0:
        avg1, info1 = mesh.aggregate_projected_images(seg_set, aggregate_img_scale=scale, distributed=False)
        favg1, finfo1 = mesh.aggregate_projected_images(img_set, aggregate_img_scale=scale, distributed=False, apply_distortion=False)
        same = lambda a, b: bool(np.array_equal(np.isnan(a), np.isnan(b)) and np.array_equal(np.nan_to_num(a), np.nan_to_num(b)))
        out["labels_equal_single_process"] = same(avg, avg1) and bool(np.array_equal(info["projection_counts"], info1["projection_counts"]))
        out["float_counts_equal_single_process"] = bool(np.array_equal(finfo["projection_counts"], finfo1["projection_counts"]))
        s, s1 = finfo["summed_projections"], finfo1["summed_projections"]
        m = np.isfinite(s1)
        out["float_sums_max_rel_diff"] = float(np.max(np.abs(s[m] - s1[m]) / np.maximum(np.abs(s1[m]), 1e-300))) if m.any() else 0.0
        out["faces_observed"] = int(np.sum(info1["projection_counts"] > 0))
        ok = out["labels_equal_single_process"] and out["float_counts_equal_single_process"] and out["float_sums_max_rel_diff"] < 1e-12
        out["ok"] = bool(ok)
        print(json.dumps(out), flush=True)
    dist.barrier()
    dist.destroy_process_group()
    return 0 if rank != 0 or out["ok"] else 1


if __name__ == "__main__":
    sys.exit(main())
