#!/usr/bin/env python3
"""tools/stamp_tile.py -- diagnostic build of the tile kernel (GR_OPT_VARIANT bit 256): where a wave of the C2 workload
spends its life (s_memtime stamps, cycles per wave).  Shares, not run time: the stamps' waits forbid overlaps."""
import ctypes, sys
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from geograypher_amd._hip import HipRaster
from geograypher_amd.utils import synthetic

nv = 50
points, faces = synthetic.terrain_mesh()
recs = torch.from_numpy(synthetic.config2_cameras(50).get_raster_records(1.0, near=1.0)[:nv]).cuda()
hip = HipRaster(0)
hip.upload_mesh(points.astype(np.float32), faces.astype(np.int32))
ids = torch.empty((nv, 3000, 4000), dtype=torch.int32, device="cuda")
hip.lib.gr_debug_stamps.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
buf = (ctypes.c_ulonglong * 8)()
for var in [int(x) for x in (sys.argv[1:] or ["266"])]:
    hip.set_option(7, var)
    for _ in range(3):
        hip.raster_face_ids(recs, 3000, 4000, out=ids, check=False)
    hip.lib.gr_debug_stamps(hip._ctx, buf, 1)
    hip.set_profiling(True)
    for _ in range(4):
        hip.raster_face_ids(recs, 3000, 4000, out=ids, check=False)
    st = hip.stage_times()
    hip.set_profiling(False)
    hip.lib.gr_debug_stamps(hip._ctx, buf, 1)
    n = max(buf[6], 1)
    names = ["load_wait", "barrier_A", "raster", "barrier_B", "store_issue", "store_done"]
    print(var, "raster us/view", round(st["raster_ms"] / st["views"] * 1e3, 2), "waves", buf[6],
          {k: round(buf[i] / n) for i, k in enumerate(names)}, "sum", round(sum(buf[i] for i in range(6)) / n))
