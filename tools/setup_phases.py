#!/usr/bin/env python3
"""tools/setup_phases.py [c2|c2q|c5|forest|forestq] [views] [-D defines, comma separated] -- where a wave of k_setup_cull spends its life.  GPU box only.

Runs a DIAGNOSTIC build of the library (csrc/libgeograster_sstamps.so: -DGR_STAMPS [+ the given defines], built here if
missing or stale), whose set-up kernel reads the shader clock at its phase boundaries (binning.hip, GR_SSTAMP: every stamp first
waits for the wave's outstanding memory operations, so a latency is charged to the phase that waited for it) and sums the
cycles per phase over all waves.  Prints one JSON line: cycles per block iteration and share per phase, block iterations
per view, mean resident waves per CU.  The stamps serialise the wave's memory operations: use the shares as a map of where
the latencies are, not as the product kernel's time."""
import ctypes
import json
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from geograypher_amd import build as gbuild

defs = [d for d in (sys.argv[3] if len(sys.argv) > 3 else "").split(",") if d and d != "0"]
bits = ",".join(defs)
tag = "sstamps" + "".join(c if c.isalnum() else "_" for c in bits)
lib_path = gbuild.CSRC / f"libgeograster_{tag}.so"
if not lib_path.is_file() or lib_path.stat().st_mtime < max(p.stat().st_mtime for p in gbuild.SOURCES + gbuild.HEADERS):
    gbuild.build_variant(tag, ["GR_STAMPS"] + defs)
os.environ["GEOGRAYPHER_AMD_LIB"] = str(lib_path)

import numpy as np
import torch

from geograypher_amd._hip import HipRaster
from geograypher_amd.utils import synthetic

PHASES = ["block loads", "transform + face set-up", "clip list + tile groups", "counter atomics", "entries of small faces",
          "big faces / records"]


def main():
    wl = sys.argv[1] if len(sys.argv) > 1 else "c2"
    nv = int(sys.argv[2]) if len(sys.argv) > 2 else (20 if wl in ("c5", "forest", "forestq") else 50)
    scale = 0.25 if wl.endswith("q") else 1.0
    if wl == "c5":
        (points, faces), cams = synthetic.config5_scene(n_views=nv)
    elif wl.startswith("forest"):
        points, faces = synthetic.forest_scene()
        cams = synthetic.oblique_cameras(20)
        nv = min(nv, 20)
    else:
        points, faces = synthetic.terrain_mesh()
        cams = synthetic.config2_cameras(50)
    H, W = cams[0].get_image_size(scale)
    recs = torch.from_numpy(cams.get_raster_records(scale, near=1.0)[:nv]).cuda()
    hip = HipRaster(0)
    hip.upload_mesh(points.astype(np.float32), faces.astype(np.int32))
    ids = torch.empty((nv, H, W), dtype=torch.int32, device="cuda")
    hip.raster_face_ids(recs, H, W, out=ids, check=True)
    for _ in range(3):
        hip.raster_face_ids(recs, H, W, out=ids, check=False)
    read = hip.lib.gr_debug_read_setup_stamps
    read.restype = ctypes.c_int
    read.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint64)]
    buf = (ctypes.c_uint64 * 16)()
    assert read(hip._ctx, buf) == 0   # clear
    reps = 5
    hip.set_profiling(True)
    for _ in range(reps):
        hip.raster_face_ids(recs, H, W, out=ids, check=False)
    st = hip.stage_times()
    hip.set_profiling(False)
    assert read(hip._ctx, buf) == 0
    cyc = [int(x) for x in buf]
    waves, iters = cyc[15], cyc[14]
    total = sum(cyc[:6])
    kernel_s = st["setup_ms"] * 1e-3
    out = {
        "workload": wl, "views": nv, "image": f"{W}x{H}", "defines": bits, "setup_us_per_view": round(st["setup_ms"] / st["views"] * 1e3, 2),
        "waves_per_view": round(waves / (reps * nv), 1), "block_iterations_per_view": round(iters / (reps * nv), 1),
        "cycles_per_block_iteration": {PHASES[k]: round(cyc[k] / max(iters, 1), 1) for k in range(6)},
        "share": {PHASES[k]: round(cyc[k] / max(total, 1), 4) for k in range(6)},
        "stamped_cycles_per_iteration": round(total / max(iters, 1), 1),
        "wave_lifetime_cycles": round(cyc[12] / max(waves, 1), 1),
        "shader_clock_GHz": round(cyc[12] / max(cyc[13], 1) * 0.1, 3),
        "mean_resident_waves_per_cu": round(cyc[13] * 1e-8 / (kernel_s * 256), 2),
    }
    print(json.dumps(out))


if __name__ == "__main__":
    main()
