#!/usr/bin/env python3
"""tools/tile_phases.py [c2|c5|forest] [views] [fused] -- where a wave of k_raster_tile spends its life, phase by phase.  GPU box only.

Runs the DIAGNOSTIC build of the library (csrc/libgeograster_stamps.so: -DGR_STAMPS, built here if missing), whose tile kernel
reads the shader clock at its phase boundaries (raster_tile.hip, GR_STAMP) and sums the cycles of every phase over all waves.
Prints one JSON line: cycles per tile visit and share of the wave lifetime per phase, the kernel's HIP-event time per view
and the product of waves x lifetime against it (how many waves a CU holds on average).  The stamps cost a few per cent
(every stamp waits for the wave's outstanding LDS operations): use the shares, not the absolute time."""
import ctypes
import json
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from geograypher_amd import build as gbuild

lib_path = gbuild.CSRC / "libgeograster_stamps.so"
if not lib_path.is_file() or lib_path.stat().st_mtime < max(p.stat().st_mtime for p in gbuild.SOURCES + gbuild.HEADERS):
    gbuild.build_variant("stamps", ["GR_STAMPS"])
os.environ["GEOGRAYPHER_AMD_LIB"] = str(lib_path)

import numpy as np
import torch

from geograypher_amd._hip import HipRaster
from geograypher_amd.utils import synthetic

PHASES = ["prologue", "fill+stage", "barrier(fill)", "items chunk0", "later chunks", "barrier(items)", "epilogue", "barrier(tiles)",
          "empty tile"]


def main():
    wl = sys.argv[1] if len(sys.argv) > 1 else "c2"
    nv = int(sys.argv[2]) if len(sys.argv) > 2 else (20 if wl == "c5" else 50)
    fused = len(sys.argv) > 3 and sys.argv[3] == "fused"
    scale = 1.0
    if wl == "c5":
        (points, faces), cams = synthetic.config5_scene(n_views=nv)
    elif wl.startswith("forest"):
        points, faces = synthetic.forest_scene()
        cams = synthetic.oblique_cameras(20)
        nv = min(nv, 20)
        scale = 0.25 if wl.endswith("25") else 1.0
    else:
        points, faces = synthetic.terrain_mesh()
        cams = synthetic.config2_cameras(50)
        scale = 0.25 if wl.endswith("q") else 1.0   # c2q: config 2 at render_img_scale 0.25
    H, W = cams[0].get_image_size(scale)
    recs = torch.from_numpy(cams.get_raster_records(scale, near=1.0)[:nv]).cuda()
    hip = HipRaster(0)
    hip.upload_mesh(points.astype(np.float32), faces.astype(np.int32))
    ids = None if fused else torch.empty((nv, H, W), dtype=torch.int32, device="cuda")
    labels = torch.randint(0, 4, (nv, H, W), dtype=torch.uint8, device="cuda") if fused else None
    votes, counts = hip.new_vote_buffers(4)

    def run(check):
        if fused:
            hip.raster_project_labels(recs, labels, 4, votes, counts, check=check)
        else:
            hip.raster_face_ids(recs, H, W, out=ids, check=check)

    run(True)
    for _ in range(3):
        run(False)
    read = hip.lib.gr_debug_read_stamps
    read.restype = ctypes.c_int
    read.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint64)]
    buf = (ctypes.c_uint64 * 16)()
    assert read(hip._ctx, buf) == 0   # clear
    items = hip.lib.gr_debug_read_item_stats
    items.restype = ctypes.c_int
    items.argtypes = [ctypes.POINTER(ctypes.c_uint64)]
    ibuf = (ctypes.c_uint64 * 4)()
    assert items(ibuf) == 0           # clear
    reps = 5
    hip.set_profiling(True)
    for _ in range(reps):
        run(False)
    st = hip.stage_times()
    hip.set_profiling(False)
    assert read(hip._ctx, buf) == 0
    cyc = [int(x) for x in buf]
    assert items(ibuf) == 0
    it = [int(x) for x in ibuf]
    waves, visits = cyc[15], cyc[14]
    total = sum(cyc[:9])
    raster_us_per_view = st["raster_ms"] / st["views"] * 1e3
    kernel_s = st["raster_ms"] * 1e-3
    out = {
        "workload": wl, "views": nv, "fused": fused, "image": f"{W}x{H}", "waves": waves, "tile_visits_x_waves": visits,
        "raster_us_per_view": round(raster_us_per_view, 2),
        "cycles_per_tile_visit": {PHASES[k]: round(cyc[k] / max(visits, 1), 1) for k in range(9)},
        "share": {PHASES[k]: round(cyc[k] / max(total, 1), 4) for k in range(9)},
        "items_per_view": round(it[0] / (reps * nv)), "empty_span_items_share": round(it[1] / max(it[0], 1), 4),
        "pixels_per_nonempty_item": round(it[2] / max(it[0] - it[1], 1), 2), "batches_per_view": round(it[3] / (reps * nv)),
        "items_per_batch": round(it[0] / max(it[3], 1), 1),
        "wave_lifetime_cycles_per_tile": round(total / max(visits, 1), 1),
        # sum of wave lifetimes / (kernel time x 256 CUs): resident waves per CU if the clock were 100 MHz x s_memtime ticks
        "wave_cycles_total": total,
        # shader clock under load: wave lifetimes in shader cycles (s_memtime) over the same in 10 ns ticks (s_memrealtime)
        "shader_clock_GHz": round(cyc[12] / max(cyc[13], 1) * 0.1, 3),
        # waves resident per CU on average = sum of wave lifetimes (real time) / (kernel time x 256 CUs)
        "mean_resident_waves_per_cu": round(cyc[13] * 1e-8 / (kernel_s * 256), 2),
        "mean_resident_waves_per_cu_at_2p4GHz": round(total / (kernel_s * 2.4e9 * 256), 2),
    }
    print(json.dumps(out))


if __name__ == "__main__":
    main()
