#!/usr/bin/env python3
"""tools/ab_aggregate.py [views] [reps] [var ...] -- wall-clock rate of the fused aggregation call (BASELINE config 3 shape:
C2 mesh, 4000x3000, 4 classes) for GR_OPT_VARIANT values, in one process; votes must be identical.  GPU box only."""
import json
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from geograypher_amd._hip import HipRaster
from geograypher_amd.utils import synthetic

H, W, C = 3000, 4000, 4


def main():
    nv = int(sys.argv[1]) if len(sys.argv) > 1 else 500
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    variants = [int(x) for x in sys.argv[3:]] or [0, 4]
    points, faces = synthetic.terrain_mesh()
    cams = synthetic.config3_cameras() if hasattr(synthetic, "config3_cameras") else synthetic.config2_cameras(nv)
    recs = torch.from_numpy(cams.get_raster_records(1.0, near=1.0)[:nv]).cuda()
    nv = recs.shape[0]
    hip = HipRaster(0)
    hip.upload_mesh(points.astype(np.float32), faces.astype(np.int32))
    labels = torch.randint(0, C, (64, H, W), dtype=torch.uint8, device="cuda").repeat((nv + 63) // 64, 1, 1)[:nv]
    ref = None
    for var in variants:
        hip.set_option(7, var)
        votes, counts = hip.new_vote_buffers(C)
        hip.raster_project_labels(recs, labels, C, votes, counts, ids_out=None, check=True)
        if ref is None:
            ref = (votes.clone(), counts.clone())
        else:
            assert torch.equal(ref[0], votes) and torch.equal(ref[1], counts), var
        ts = []
        for _ in range(reps):
            votes.zero_(); counts.zero_()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            hip.raster_project_labels(recs, labels, C, votes, counts, ids_out=None, check=False)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        ts.sort()
        print(json.dumps({"var": var, "views": nv, "views_per_s": round(nv / ts[len(ts) // 2], 1), "best": round(nv / ts[0], 1)}))


if __name__ == "__main__":
    main()
