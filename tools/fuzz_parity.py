#!/usr/bin/env python3
"""tools/fuzz_parity.py <seconds> [first_seed] [big] -- randomised differential campaign: the HIP path against the C oracle (oracle/)
on scenes the test-suite's fixed seeds do not reach.  GPU box only; a checker like the tests (the product never calls the oracle).

Every iteration draws, from its seed: a scene (triangle soup of log-uniform sizes, an axis-aligned lattice with exact ties, a
jittered height field, or a mix; optional zero-area / coincident / behind-the-camera faces; shuffled face order), 1-6 cameras
(nadir with tilt, oblique look-at, inside the scene's bounding box; off-centre principal points), an image size from 1 x 1 to
~700 x 500 (odd widths included), a near plane, and a setting of the tuning knobs (tile height, slots per tile incl. exact
binning and tiny segments that overflow, GR_OPT_VARIANT bits, views per launch group).  Checked bit for bit: face ids, depth
bits, and -- when the scene has at least one face -- label votes and counts of the fused aggregation (both background
conventions).  `big`: images of 500-2000 pixels a side and scenes of up to 300 000 faces (launches large enough for the chained
kernels to be chosen by themselves).  Prints one line per failure with the seed (re-run: `fuzz_parity.py 1 <seed>`), and a JSON
summary."""
import json
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from geograypher_amd._hip import HipRaster
from geograypher_amd.utils import synthetic
from oracle import oracle_c

# the mode bits of GR_OPT_VARIANT (include/geograster.h): 4096 / 8192 micro lists never / always, 16384 no look at the first
# launch group's counts (the overflow protocol of rounds 1-4)
VARIANT_BITS = [1, 4, 16, 128, 512, 4096, 8192, 16384, 131072]


BIG = False


def scene(rng):
    kind = rng.integers(0, 4)
    parts_p, parts_f, off = [], [], 0

    def add(points, faces):
        nonlocal off
        parts_p.append(points)
        parts_f.append(faces + off)
        off += points.shape[0]

    if kind in (0, 3):  # soup
        n = int(np.exp(rng.uniform(np.log(20), np.log(150000 if BIG else 20000))))
        spread = rng.uniform(2, 30)
        centers = rng.uniform(-spread, spread, (n, 1, 3)) * np.array([1, 1, rng.uniform(0.02, 0.5)])
        size = np.exp(rng.uniform(np.log(0.005), np.log(rng.choice([0.3, 5.0, 60.0])), (n, 1, 1)))
        tri = centers + rng.normal(0, 1, (n, 3, 3)) * size
        k = min(n // 10, 40)
        if k and rng.random() < 0.5:
            tri[:k, 2] = tri[:k, 1]                      # zero area
            tri[k:2 * k] = tri[2 * k:3 * k]              # coincident faces: the lower id wins
        add(tri.reshape(-1, 3), np.arange(3 * n).reshape(n, 3))
    if kind in (1, 3):  # axis-aligned lattice: exact ties, a == 0 / b == 0 edges
        g = int(rng.integers(3, 380 if BIG else 120))
        ext = rng.uniform(1, 20)
        xs, ys = np.meshgrid(np.linspace(-ext, ext, g + 1), np.linspace(-ext, ext, g + 1))
        add(np.stack([xs.ravel(), ys.ravel(), np.full(xs.size, rng.uniform(-0.5, 0.5))], axis=1), synthetic.grid_faces(g + 1, g + 1))
    if kind == 2:  # jittered height field
        g = int(rng.integers(8, 390 if BIG else 200))
        ext = rng.uniform(5, 60)
        xs, ys = np.meshgrid(np.linspace(-ext, ext, g), np.linspace(-ext, ext, g))
        cell = 2 * ext / (g - 1)
        xs = xs + rng.uniform(-0.3, 0.3, xs.shape) * cell
        ys = ys + rng.uniform(-0.3, 0.3, ys.shape) * cell
        zs = rng.uniform(0.2, 3.0) * np.sin(xs / rng.uniform(2, 9)) + rng.uniform(0.2, 3.0) * np.cos(ys / rng.uniform(2, 9))
        add(np.stack([xs.ravel(), ys.ravel(), zs.ravel()], axis=1), synthetic.grid_faces(g, g))
    points = np.concatenate(parts_p)
    faces = np.concatenate(parts_f)
    if rng.random() < 0.5:
        faces = faces[rng.permutation(faces.shape[0])]
    return points, faces


def cameras(rng, points, w, h):
    lo, hi = points.min(axis=0), points.max(axis=0)
    mid, half = 0.5 * (lo + hi), 0.5 * (hi - lo) + 1e-3
    poses = []
    for _ in range(int(rng.integers(1, 7))):
        mode = rng.integers(0, 3)
        if mode == 0:
            z = hi[2] + np.exp(rng.uniform(np.log(0.05), np.log(4 * half[:2].max() + 1)))
            poses.append(synthetic.nadir_pose(mid[0] + rng.uniform(-1, 1) * half[0], mid[1] + rng.uniform(-1, 1) * half[1], z,
                                              yaw_deg=rng.uniform(0, 360), tilt_x_deg=rng.uniform(-30, 30), tilt_y_deg=rng.uniform(-30, 30)))
        elif mode == 1:
            eye = mid + rng.normal(0, 1, 3) * half * 2 + np.array([0, 0, half[:2].max() * rng.uniform(0.2, 2)])
            target = mid + rng.uniform(-1, 1, 3) * half
            if np.linalg.norm(eye - target) < 1e-3:
                eye = eye + 1.0
            poses.append(synthetic.look_at(tuple(eye), tuple(target), up_hint=(0, 0, 1)))
        else:  # inside the bounding box
            eye = mid + rng.uniform(-0.8, 0.8, 3) * half
            target = mid + rng.uniform(-1, 1, 3) * (half + 1.0) + np.array([0.0, 0.0, -1.0])
            poses.append(synthetic.look_at(tuple(eye), tuple(target), up_hint=(0, 0, 1)))
    cams = synthetic.camera_set_from_poses(poses, f=float(max(h, w)) * np.exp(rng.uniform(np.log(0.25), np.log(3.0))), width=w, height=h)
    for c in cams.cameras:
        c.cx, c.cy = rng.uniform(-0.1, 0.1) * w, rng.uniform(-0.1, 0.1) * h
    return cams


def one(hip, seed):
    rng = np.random.default_rng(seed)
    points, faces = scene(rng)
    if BIG:
        h, w = int(rng.integers(500, 1500)), int(rng.integers(500, 2000))
    elif rng.random() < 0.15:
        h, w = [(1, 1), (2, 3), (3, 70), (65, 33), (64, 64), (33, 257)][int(rng.integers(0, 6))]
    else:
        h, w = int(rng.integers(8, 500)), int(rng.integers(8, 700))
    cams = cameras(rng, points, w, h)
    pp = "intrinsics" if rng.random() < 0.5 else "center"
    recs = cams.get_raster_records(1.0, near=float(np.exp(rng.uniform(np.log(0.01), np.log(2.0)))), principal_point=pp)
    # round 6: the vertex stage in an OpenGL pipeline's order of operations (GR_OPT_VERTEX_ORDER; needs the centred principal point)
    order = "gl" if (pp == "center" and rng.random() < 0.3) else "r1"
    thl = int(rng.choice([5, 6]))
    cap = int(rng.choice([0, 64, 128, 512, 512, 512, 2048]))
    var = 0
    for b in VARIANT_BITS:
        if rng.random() < 0.2:
            var |= b
    batch = int(rng.choice([64, 64, 5, 2, 1]))
    hip.set_option(2, thl); hip.set_option(6, cap); hip.set_option(7, var); hip.set_option(3, batch)
    hip.set_vertex_order(order)
    info = {"seed": seed, "vertex_order": order, "faces": int(faces.shape[0]), "views": int(recs.shape[0]), "image": f"{w}x{h}", "thl": thl, "cap": cap,
            "var": var, "batch": batch}
    hip.upload_mesh(points.astype(np.float32), faces.astype(np.int32))
    ids, dep = hip.raster_face_ids(recs, h, w, want_depth=True)
    info["lessons"] = (int(hip.last_retries), int(hip.last_stats.get("rebinned_groups", 0)))
    ids2 = hip.raster_face_ids(recs, h, w)  # the ids-only kernels
    ids_np, dep_np = ids.cpu().numpy(), dep.cpu().numpy()
    bad = []
    if not torch.equal(ids, ids2):
        bad.append("ids-only call differs from ids + depth call")
    for v in range(recs.shape[0]):
        want, wdep = oracle_c.raster(points, faces, recs[v], h, w, want_depth=True, vertex_order=order)
        if not np.array_equal(ids_np[v], want):
            bad.append(f"view {v}: {int((ids_np[v] != want).sum())} ids differ")
        elif not np.array_equal(dep_np[v].view(np.int32), wdep.view(np.int32)):
            bad.append(f"view {v}: depth bits differ")
    F, C = faces.shape[0], int(rng.choice([1, 2, 3, 4, 5, 7, 8, 9, 11, 15, 16, 17, 23, 40]))   # both packed registers of k_vote_labels, and beyond
    compat = bool(rng.random() < 0.5)
    labels = np.stack([synthetic.synthetic_labels(ids_np[v], v + seed, C) for v in range(recs.shape[0])])
    want_v = np.zeros((F, C), dtype=np.uint32)
    want_c = np.zeros(F, dtype=np.uint32)
    for v in range(recs.shape[0]):
        oracle_c.project_labels(ids_np[v], labels[v], F, C, want_v, want_c, neg1_is_last_face=compat)
    v2, c2 = hip.new_vote_buffers(C)
    hip.raster_project_labels(recs, labels, C, v2, c2, neg1_is_last_face=compat)
    if not (np.array_equal(v2.cpu().numpy().view(np.uint32), want_v) and np.array_equal(c2.cpu().numpy().view(np.uint32), want_c)):
        bad.append(f"fused votes differ (C={C}, compat={compat})")
    info["pixels"] = int(recs.shape[0]) * h * w
    info["covered"] = float((ids_np >= 0).mean())
    return info, bad


def main():
    global BIG
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
    BIG = len(sys.argv) > 3 and sys.argv[3] == "big"
    hip = HipRaster(0)
    t0 = time.time()
    n = views = pixels = faces = retried = rebinned = 0
    failures = []
    while time.time() - t0 < budget:
        try:
            info, bad = one(hip, seed)
        except Exception as e:  # an error return of the library is a finding too
            info, bad = {"seed": seed}, [f"exception: {type(e).__name__}: {e}"]
        if bad:
            failures.append({**info, "problems": bad})
            print("FAIL", json.dumps(failures[-1]), flush=True)
        n += 1
        views += info.get("views", 0); pixels += info.get("pixels", 0); faces += info.get("faces", 0)
        retried += 1 if info.get("lessons", (0, 0))[0] else 0; rebinned += 1 if info.get("lessons", (0, 0))[1] else 0
        seed += 1
    print(json.dumps({"scenes": n, "views": views, "pixels": pixels, "faces": faces, "failures": len(failures), "first_calls_retried": retried,
                      "first_calls_with_a_rebinned_first_group": rebinned,
                      "first_seed": seed - n, "seconds": round(time.time() - t0, 1)}))
    return 1 if failures else 0


if __name__ == "__main__":
    sys.exit(main())
