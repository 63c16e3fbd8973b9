#!/usr/bin/env python3
"""tools/ab_forest.py [variant ...] -- stage times of pix2face on the hostile workload (C2 terrain + 20 000 trees, cameras
tilted 30-45 degrees) at full and quarter resolution.  variant = name:var[:dbg[:thl[:cap[:batch]]]].  GPU box only."""
import json
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from geograypher_amd._hip import HipRaster
from geograypher_amd.utils import synthetic


def main():
    specs = sys.argv[1:] or ["base:0"]
    pts, faces = synthetic.forest_scene()
    cams = synthetic.oblique_cameras(20)
    hip = HipRaster(0)
    hip.upload_mesh(pts.astype(np.float32), faces.astype(np.int32))
    for scale in (1.0, 0.25):
        h, w = cams[0].get_image_size(scale)
        recs = torch.from_numpy(cams.get_raster_records(scale, near=1.0)).cuda()
        ids = torch.empty((len(cams), h, w), dtype=torch.int32, device="cuda")
        ref = None
        for sp in specs:
            parts = sp.split(":")
            nums = [int(x) for x in parts[1:]] + [None] * 5
            var, dbg, thl, cap, batch = nums[0] or 0, nums[1] or 0, nums[2] or 5, 512 if nums[3] is None else nums[3], nums[4] or 64
            hip.set_option(2, thl); hip.set_option(6, cap); hip.set_option(7, var); hip.set_option(99, dbg); hip.set_option(3, batch)
            hip.raster_face_ids(recs, h, w, out=ids, check=True)
            retries, st0 = hip.last_retries, dict(hip.last_stats)
            if dbg == 0:
                if ref is None:
                    ref = ids.clone()
                else:
                    assert torch.equal(ref, ids), parts[0]
            for _ in range(2):
                hip.raster_face_ids(recs, h, w, out=ids, check=False)
            hip.set_profiling(True)
            for _ in range(5):
                hip.raster_face_ids(recs, h, w, out=ids, check=False)
            st = hip.stage_times()
            hip.set_profiling(False)
            per = {k: round(st[k] / st["views"] * 1e3, 1) for k in ("setup_ms", "scan_ms", "fill_ms", "raster_ms")}
            print(json.dumps({"variant": parts[0], "image": f"{w}x{h}", **per, "us_per_view": round(sum(per.values()), 1),
                              "entries_per_view": round(st0["entries"] / len(cams)), "max_entries_per_tile": st0["max_entries"],
                              "records_per_view": round(st0["records"] / len(cams)), "retries": retries}))
    hip.set_option(2, 5); hip.set_option(6, 512); hip.set_option(7, 0); hip.set_option(99, 0); hip.set_option(3, 64)


if __name__ == "__main__":
    main()
