#!/usr/bin/env python3
"""tools/png_labels_rate.py [views] [loader_threads ...] -- views per second of aggregate_projected_images fed by
LookUpSegmentor from class-index PNG files on disk (what entrypoints/aggregate_images.py does): PNG decode on host threads,
pinned staging, fused raster + votes.  GPU box only; writes the PNGs under a temp folder."""
import json
import sys
import tempfile
import time
from pathlib import Path

import numpy as np
from PIL import Image

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from geograypher_amd.cameras import SegmentorPhotogrammetryCameraSet
from geograypher_amd.meshes import TexturedPhotogrammetryMesh
from geograypher_amd.predictors import LookUpSegmentor
from geograypher_amd.utils import synthetic


def main():
    nv = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    threads = [int(x) for x in sys.argv[2:]] or [8, 32, 64]
    points, faces = synthetic.terrain_mesh()
    cams = synthetic.config3_cameras(nv)
    mesh = TexturedPhotogrammetryMesh((points, faces), log_level="ERROR")
    ids = mesh.pix2face(cams[0:4], apply_distortion=False)
    out = {}
    with tempfile.TemporaryDirectory() as d:
        base, lookup = Path(d) / "images", Path(d) / "labels"
        lookup.mkdir(parents=True)
        labs = [synthetic.synthetic_labels(ids[v % 4], v, 6) for v in range(4)]
        for i, cam in enumerate(cams.cameras):
            cam.image_filename = Path(base / f"{i}.png")
            Image.fromarray(labs[i % 4]).save(lookup / f"{i}.png", compress_level=1)
        seg = SegmentorPhotogrammetryCameraSet(cams, LookUpSegmentor(base, lookup, num_classes=6))
        mesh.aggregate_projected_images(seg, loader_threads=8)
        for nt in threads:
            t0 = time.perf_counter()
            avg, info = mesh.aggregate_projected_images(seg, loader_threads=nt)
            dt = time.perf_counter() - t0
            out[f"loader_threads_{nt}"] = round(nv / dt, 1)
        t0 = time.perf_counter()
        mesh.aggregate_projected_images(seg)
        out["default"] = round(nv / (time.perf_counter() - t0), 1)
    print(json.dumps({"workload": f"aggregate_projected_images from {nv} class-index PNGs 4000x3000 (LookUpSegmentor)", "views_per_s": out}))


if __name__ == "__main__":
    main()
