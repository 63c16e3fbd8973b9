#!/usr/bin/env python3
"""tools/ab_fused.py -- stage times of the fused raster + projection call on the C2 workload with the timing-only
ablation masks of GR_OPT_DEBUG (GPU box only)."""
import json
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from geograypher_amd._hip import HipRaster
from geograypher_amd.utils import synthetic

H, W, C = 3000, 4000, 4


def main():
    nv = int(sys.argv[1]) if len(sys.argv) > 1 else 50
    points, faces = synthetic.terrain_mesh()
    cams = synthetic.config2_cameras(50)
    recs = torch.from_numpy(cams.get_raster_records(1.0, near=1.0)[:nv]).cuda()
    hip = HipRaster(0)
    hip.upload_mesh(points.astype(np.float32), faces.astype(np.int32))
    labels = torch.randint(0, C, (nv, H, W), dtype=torch.uint8, device="cuda")
    votes, counts = hip.new_vote_buffers(C)
    variants = [("full", 0), ("no_atomics", 8), ("no_label_loads", 16), ("neither", 24), ("no_triangles", 4)]
    for rep in range(2):
        for name, dbg in variants:
            hip.set_option(99, dbg)
            votes.zero_(); counts.zero_()
            hip.raster_project_labels(recs, labels, C, votes, counts, ids_out=None, check=False)
            hip.set_profiling(True)
            for _ in range(3):
                hip.raster_project_labels(recs, labels, C, votes, counts, ids_out=None, check=False)
            st = hip.stage_times()
            hip.set_profiling(False)
            if rep == 1:
                print(name, json.dumps({k: round(st[k] / st["views"] * 1e3, 2) for k in ("setup_ms", "raster_ms", "vote_ms")}))


if __name__ == "__main__":
    main()
