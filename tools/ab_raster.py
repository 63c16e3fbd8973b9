#!/usr/bin/env python3
"""tools/ab_raster.py -- A/B the tile-kernel variants and launch batch on the C2 workload in ONE process
(interleaved, HIP-event stage times per view).  GPU box only."""
import json
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from geograypher_amd._hip import HipRaster
from geograypher_amd.utils import synthetic

H, W = 3000, 4000


def main():
    nv = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    points, faces = synthetic.terrain_mesh()
    cams = synthetic.config2_cameras(50)
    recs = torch.from_numpy(cams.get_raster_records(1.0, near=1.0)[:nv]).cuda()
    hip = HipRaster(0)
    hip.upload_mesh(points.astype(np.float32), faces.astype(np.int32))
    ids = torch.empty((nv, H, W), dtype=torch.int32, device="cuda")
    ref = None
    # (name, tile_h_log2, batch, debug mask, single-pass slots per tile)
    variants = [("direct512_b64", 5, 64, 0, 512), ("exact_b64", 5, 64, 0, 0), ("direct512_b32", 5, 32, 0, 512),
                ("tile64_direct1024", 6, 64, 0, 1024), ("noscan", 5, 64, 1, 512), ("nostore", 5, 64, 2, 512),
                ("notri", 5, 64, 4, 512)]
    results = {}
    for rep in range(3):
        for name, thl, b, dbg, cap in variants:
            hip.set_option(2, thl); hip.set_option(3, b); hip.set_option(99, dbg); hip.set_option(6, cap)
            hip.raster_face_ids(recs, H, W, out=ids, check=True)
            if ref is None:
                ref = ids.clone()
            elif dbg == 0:
                assert torch.equal(ref, ids), name
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(6):
                hip.raster_face_ids(recs, H, W, out=ids, check=False)
            torch.cuda.synchronize()
            wall = (time.perf_counter() - t0) / 6 / nv * 1e3
            hip.set_profiling(True)
            for _ in range(3):
                hip.raster_face_ids(recs, H, W, out=ids, check=False)
            st = hip.stage_times()
            hip.set_profiling(False)
            per = {k2: round(st[k2] / st["views"] * 1e3, 2) for k2 in ("setup_ms", "scan_ms", "fill_ms", "raster_ms")}
            per["wall_us_per_view"] = round(wall * 1e3, 2)
            stt = hip.raster_status()
            per["entries_per_view"] = stt["entries"] / nv
            per["max_entries"] = stt["max_entries"]
            results.setdefault(name, []).append(per)
    for name, runs in results.items():
        print(name, json.dumps(runs[-1]))


if __name__ == "__main__":
    main()
