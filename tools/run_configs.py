#!/usr/bin/env python3
"""tools/run_configs.py [c3_views] [c5_views] -- BASELINE.json configs 3 and 5 on one GPU (reduced view counts by
default), with a sampled oracle check.  Prints one JSON object per config.  GPU box only."""
import json
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from bench import device_labels
from geograypher_amd._hip import HipRaster
from geograypher_amd.utils import synthetic
from oracle import oracle_c


def run(name, points, faces, cams, H, W, C, hip, check_views=(0,)):
    F = faces.shape[0]
    recs_np = cams.get_raster_records(1.0, near=1.0)
    nv = recs_np.shape[0]
    hip.upload_mesh(points.astype(np.float32), faces.astype(np.int32))
    recs = torch.from_numpy(recs_np).cuda()
    chunk = 50
    votes, counts = hip.new_vote_buffers(C)
    # labels are a function of the face ids: generate per chunk (untimed), then time the fused aggregation
    t_fused = 0.0
    t_raster = 0.0
    for c0 in range(0, nv, chunk):
        r = recs[c0:c0 + chunk]
        n = r.shape[0]
        ids = torch.empty((n, H, W), dtype=torch.int32, device="cuda")
        hip.raster_face_ids(r, H, W, out=ids, check=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        hip.raster_face_ids(r, H, W, out=ids, check=False)
        torch.cuda.synchronize()
        t_raster += time.perf_counter() - t0
        for v in check_views:
            if c0 <= v < c0 + n:
                want = oracle_c.raster(points, faces, recs_np[v], H, W)
                assert np.array_equal(ids[v - c0].cpu().numpy(), want), f"{name}: view {v} differs from the oracle"
        labels = torch.stack([device_labels(ids[k], c0 + k, C) for k in range(n)])
        del ids
        if c0 == 0:  # untimed first call: scratch allocation (winner keys) for this mesh / image size
            v0, k0 = hip.new_vote_buffers(C)
            hip.raster_project_labels(r, labels, C, v0, k0, check=False)
            del v0, k0
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        hip.raster_project_labels(r, labels, C, votes, counts, check=False)
        torch.cuda.synchronize()
        t_fused += time.perf_counter() - t0
        del labels
    avg, summed, cnt = hip.finalize_votes(votes, counts)
    st = hip.raster_status()
    out = {
        "config": name, "faces": F, "views": nv, "image": [H, W], "classes": C,
        "raster_views_per_s": round(nv / t_raster, 1), "raster_mpix_per_s": round(nv / t_raster * H * W / 1e6, 1),
        "aggregate_views_per_s": round(nv / t_fused, 1), "aggregate_mpix_per_s": round(nv / t_fused * H * W / 1e6, 1),
        "faces_observed": int((cnt > 0).sum()), "max_views_per_face": int(cnt.max()),
        "entries_per_view_last_chunk": round(st["entries"] / min(chunk, nv), 1), "oracle_checked_views": list(check_views),
    }
    print(json.dumps(out))


def main():
    c3_views = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    c5_views = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    hip = HipRaster(0)
    if c3_views > 0:
        points, faces = synthetic.terrain_mesh()
        cams = synthetic.config3_cameras(c3_views)
        run("C3 (1.2M faces, 4000x3000, 4 classes)", points, faces, cams, 3000, 4000, 4, hip, check_views=(0, c3_views // 2))
    if c5_views > 0:
        points, faces = synthetic.terrain_mesh(1582, 800.0)
        cams = synthetic.survey_cameras(50, 40, 15.0, 18.0, agl=150.0, f=4500.0, width=6000, height=4000, seed=6)[:c5_views]
        run("C5 (5M faces, 6000x4000, 10 classes)", points, faces, cams, 4000, 6000, 10, hip, check_views=(c5_views // 2,))


if __name__ == "__main__":
    main()
