#!/usr/bin/env python3
"""tools/ab_kernel.py [views] [reps] [variant ...] -- interleaved A/B of tile-kernel variants on the C2 workload, plain
(pix2face) and fused (raster + label projection), in ONE process.  GPU box only.

A variant is `name:var[:dbg[:thl[:cap[:ldspad[:kt[:batch[:pfd]]]]]]]` (GR_OPT_VARIANT bits, GR_OPT_DEBUG test-hook mask, tile height log2, slots per tile).
Every variant with dbg == 0 must reproduce the first variant's ids and votes bit for bit.  Prints one JSON line per
variant: median HIP-event stage times in microseconds per view.
"""
import json
import os
import statistics
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from geograypher_amd._hip import HipRaster
from geograypher_amd.utils import synthetic

H, W, C = 3000, 4000, int(os.environ.get("AB_CLASSES", "4"))   # label classes of the fused leg
DEFAULT = ["base:0"]
# c2: 1.2 M faces, 4000 x 3000; c5: 5 M faces, 6000 x 4000 (20 views); c2q: c2 at render_img_scale 0.25 (1000 x 750);
# forest / forestq: the hostile workload (terrain + 20 000 trees, 20 oblique views) at scale 1 / 0.25
WORKLOAD = os.environ.get("AB_WORKLOAD", "c2")


def main():
    nv = int(sys.argv[1]) if len(sys.argv) > 1 else 50
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    specs = sys.argv[3:] or DEFAULT
    variants = []
    for sp in specs:
        parts = sp.split(":")
        name = parts[0]
        nums = [int(x) for x in parts[1:]] + [None] * 8
        variants.append((name, nums[0] or 0, nums[1] or 0, nums[2] or 5, 512 if nums[3] is None else nums[3], nums[4] or 0, nums[5] or 4, nums[6] or 64, nums[7] or 2048))
    global H, W
    if WORKLOAD == "c5":
        H, W = 4000, 6000
        (points, faces), cams = synthetic.config5_scene(n_views=max(nv, 1))
    elif WORKLOAD in ("forest", "forestq"):
        points, faces = synthetic.forest_scene()
        cams = synthetic.oblique_cameras(20)
        nv = min(nv, 20)
    else:
        points, faces = synthetic.terrain_mesh()
        cams = synthetic.config2_cameras(50)
    scale = 0.25 if WORKLOAD.endswith("q") else 1.0
    if scale != 1.0:
        H, W = cams[0].get_image_size(scale)
    recs = torch.from_numpy(cams.get_raster_records(scale, near=1.0)[:nv]).cuda()
    hip = HipRaster(0)
    hip.upload_mesh(points.astype(np.float32), faces.astype(np.int32))
    ids = torch.empty((nv, H, W), dtype=torch.int32, device="cuda")
    torch.manual_seed(0)
    labels = torch.randint(0, C, (nv, H, W), dtype=torch.uint8, device="cuda")
    votes, counts = hip.new_vote_buffers(C)
    ref_ids = ref_votes = ref_counts = None
    acc = {v[0]: {"plain": [], "fused": []} for v in variants}

    def setopt(var, dbg, thl, cap, ldspad=0, kt=4, batch=64, pfd=2048):
        hip.set_option(3, batch)
        hip.set_option(98, ldspad)
        hip.set_option(2, thl)
        hip.set_option(6, cap)
        hip.set_option(7, var)
        hip.set_option(99, dbg)

    for rep in range(reps + 1):
        for name, var, dbg, thl, cap, ldspad, kt, batch, pfd in variants:
            setopt(var, dbg, thl, cap, ldspad, kt, batch, pfd)
            if rep == 0:  # correctness + sizing pass
                hip.raster_face_ids(recs, H, W, out=ids, check=True)
                votes.zero_(); counts.zero_()
                hip.raster_project_labels(recs, labels, C, votes, counts, ids_out=None, check=True)
                if dbg == 0:
                    if ref_ids is None:
                        ref_ids, ref_votes, ref_counts = ids.clone(), votes.clone(), counts.clone()
                    else:
                        assert torch.equal(ref_ids, ids), f"{name}: ids differ from {variants[0][0]}"
                        assert torch.equal(ref_votes, votes) and torch.equal(ref_counts, counts), f"{name}: votes differ"
                continue
            # setopt() made the context forget the slots per tile it had learned: the first warm-up call handles the overflow
            # again (quarter-scale and forest views need more than the default 512), the timed ones run sized
            hip.raster_face_ids(recs, H, W, out=ids, check=True)
            hip.raster_face_ids(recs, H, W, out=ids, check=False)
            hip.set_profiling(True)
            for _ in range(4):
                hip.raster_face_ids(recs, H, W, out=ids, check=False)
            st = hip.stage_times()
            hip.set_profiling(False)
            acc[name]["plain"].append({k: st[k] / st["views"] * 1e3 for k in ("setup_ms", "scan_ms", "fill_ms", "raster_ms")})
            hip.raster_project_labels(recs, labels, C, votes, counts, ids_out=None, check=True)
            hip.set_profiling(True)
            for _ in range(4):
                hip.raster_project_labels(recs, labels, C, votes, counts, ids_out=None, check=False)
            st = hip.stage_times()
            hip.set_profiling(False)
            acc[name]["fused"].append({k: st[k] / st["views"] * 1e3 for k in ("setup_ms", "raster_ms", "vote_ms")})
    setopt(0, 0, 5, 512)
    checksum = None
    if os.environ.get("AB_CHECKSUM"):  # for A/B of library builds (tools/ab_libs.py): a digest of the reference ids and votes
        import hashlib

        h = hashlib.sha256()
        h.update(ref_ids.cpu().numpy().tobytes())
        h.update(ref_votes.cpu().numpy().tobytes())
        h.update(ref_counts.cpu().numpy().tobytes())
        checksum = h.hexdigest()[:16]
    for name, var, dbg, thl, cap, ldspad, kt, batch, pfd in variants:
        out = {"variant": name, "checksum": checksum, "var": var, "dbg": dbg, "thl": thl, "cap": cap, "ldspad": ldspad, "kt": kt, "batch": batch, "pfd": pfd}
        for kind in ("plain", "fused"):
            runs = acc[name][kind]
            out[kind] = {k: round(statistics.median(r[k] for r in runs), 2) for k in runs[0]}
            out[kind]["raster_min"] = round(min(r["raster_ms"] for r in runs), 2)
        print(json.dumps(out))


if __name__ == "__main__":
    main()
