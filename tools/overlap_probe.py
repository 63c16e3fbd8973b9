#!/usr/bin/env python3
"""tools/overlap_probe.py [c2|c2q] -- can the set-up stage of one launch group hide behind the tile kernel of another?  GPU box only.

Two libgeograster contexts on ONE GPU (independent scratch), each on a torch stream of its own, fed alternately with halves of the
C2 views: the device is free to run the set-up kernels of one context beside the tile kernel of the other.  Compared with one context
rasterizing all views on one stream (the product's shape today).  Prints views/s for both."""
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from geograypher_amd._hip import HipRaster
from geograypher_amd.utils import synthetic


def main():
    wl = sys.argv[1] if len(sys.argv) > 1 else "c2"
    scale = 0.25 if wl == "c2q" else 1.0
    pts, faces = synthetic.terrain_mesh()
    cams = synthetic.config2_cameras(50)
    h, w = cams[0].get_image_size(scale)
    recs = torch.from_numpy(cams.get_raster_records(scale, near=1.0)).cuda()
    n = recs.shape[0]
    one = HipRaster(0)
    one.upload_mesh(pts.astype(np.float32), faces.astype(np.int32))
    ids = torch.empty((n, h, w), dtype=torch.int32, device="cuda")
    for _ in range(2):
        one.raster_face_ids(recs, h, w, out=ids, check=True)

    def timed(fn, reps=40):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return n * reps / (time.perf_counter() - t0)

    base = timed(lambda: one.raster_face_ids(recs, h, w, out=ids, check=False))
    out = {"workload": wl, "one_context_views_per_s": round(base, 1)}
    for parts in (2, 4):
        ctxs, streams = [], []
        step = (n + parts - 1) // parts
        for k in range(parts):
            c = HipRaster(0)
            c.upload_mesh(pts.astype(np.float32), faces.astype(np.int32))
            ctxs.append(c)
            streams.append(torch.cuda.Stream())
        for k, c in enumerate(ctxs):
            with torch.cuda.stream(streams[k]):
                for _ in range(2):
                    c.raster_face_ids(recs[k * step:(k + 1) * step], h, w, out=ids[k * step:(k + 1) * step], check=True)
        torch.cuda.synchronize()

        def run():
            for k, c in enumerate(ctxs):
                with torch.cuda.stream(streams[k]):
                    c.raster_face_ids(recs[k * step:(k + 1) * step], h, w, out=ids[k * step:(k + 1) * step], check=False)

        out[f"{parts}_contexts_{parts}_streams_views_per_s"] = round(timed(run), 1)
        want = one.raster_face_ids(recs, h, w).clone()
        run()
        torch.cuda.synchronize()
        out[f"{parts}_contexts_equal"] = bool(torch.equal(want, ids))
        del ctxs
    print(out)


if __name__ == "__main__":
    main()
