#!/usr/bin/env python3
"""tools/save_renders_rate.py [views] [writer_threads ...] -- views per second of TexturedPhotogrammetryMesh.save_renders
(row f2: raster + fused gather/uint8 cast on the GPU, asynchronous copy into a pinned ring, deflate-TIFF writers on host
threads) on the C2 workload, 4000 x 3000, one-channel discrete texture.  GPU box only; writes under a temp folder."""
import json
import sys
import tempfile
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from geograypher_amd.cameras.cameras import PhotogrammetryCameraSet
from geograypher_amd.meshes import TexturedPhotogrammetryMesh
from geograypher_amd.utils import synthetic


def main():
    nv = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    threads = [int(x) for x in sys.argv[2:]] or [1, 8, 16]
    points, faces = synthetic.terrain_mesh()
    cams = synthetic.config2_cameras(50)[:nv]
    cams.image_folder = Path("/synthetic")
    tex = (synthetic.hash32(np.arange(faces.shape[0])) % 5).astype(np.float64)
    mesh = TexturedPhotogrammetryMesh((points, faces), texture=tex, IDs_to_labels={i: str(i) for i in range(5)}, log_level="ERROR")
    out = {}
    for fmt in ("tif", "npy"):
        for nt in threads:
            with tempfile.TemporaryDirectory() as d:
                # warm-up with every view: the pinned ring (2 x writer_threads slots) is allocated once and handed back by
                # torch's caching host allocator; a 4-view warm-up left its allocation (0.4 ms per MB) inside the timed call
                mesh.save_renders(cams, output_folder=d, apply_distortion=False, writer_threads=nt, save_as_npy=fmt == "npy")
                t0 = time.perf_counter()
                mesh.save_renders(cams, output_folder=d, apply_distortion=False, writer_threads=nt, save_as_npy=fmt == "npy")
                dt = time.perf_counter() - t0
                size = sum(f.stat().st_size for f in Path(d).rglob("*") if f.is_file()) / 1e6
            out[f"{fmt}_threads_{nt}"] = {"views_per_s": round(nv / dt, 2), "MB_written": round(size, 1)}
    print(json.dumps({"workload": f"save_renders, {nv} C2 views 4000x3000, discrete 1-channel texture, cast_to_uint8", **out}))


if __name__ == "__main__":
    main()
