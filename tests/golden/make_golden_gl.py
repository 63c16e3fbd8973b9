"""tests/golden/make_golden_gl.py -- run in the BUILD CONTAINER only (python tests/golden/make_golden_gl.py [--full-size]).

Golden face-id images of the raster stage (geograypher/meshes/meshes.py:1776-1836) from two REAL third-party OpenGL
rasterizers present in this image (tests/golden/gl_raster.py): Mesa 23.2.1 llvmpipe (8 sub-pixel bits; the software GL
family of the reference's own Dockerfile:6-13) and Google SwiftShader 4.1 (ES 3.0, 4 sub-pixel bits, the coarsest grid
OpenGL allows).  The GPU box has neither: it receives the `.npz` files written here; tests/test_gl_pin.py compares the
oracle (CPU suite) and the HIP kernels (-m gpu) with them.

Fixtures (all ids int32, -1 = background, rows top-down like the reference's screenshot):
  reference_gl_c1.npz     C1 (9 800 faces), all 8 views 640x480: llvmpipe, llvmpipe with the VTK-style single-matrix vertex
                          transform, SwiftShader; the reference's own 80 000-triangle test plane (utils/test_utils.py:10-129)
                          through its nadir camera at 200x200
  reference_gl_scaled.npz C2 mesh views 0 / 23 and the hostile forest views 3 / 11 at render_img_scale = 0.25 (1000x750),
                          both implementations
  reference_gl_clip.npz   the near-plane / guard-band scenes of tests/test_hip_parity.py::test_clipping_near_plane_and_
                          guard_band with the clipping range VTK's ResetCameraClippingRange would pick
  reference_gl_c2_full.npz  ONE BASELINE-size view: C2 (1 201 250 faces), view 23, 4000 x 3000, llvmpipe -- the whole id image,
                          run-length encoded (rle_encode below: 1 MB)
--full-size additionally renders C2 / C5 / forest views at full size and LOGS the comparison with the oracle
(profiles/r05_gl_pin.log); nothing of that is stored.
"""
from __future__ import annotations

import argparse
import sys
import time
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
ROOT = HERE.parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(HERE))

from gl_raster import GLRasterizer  # noqa: E402

from geograypher_amd.utils import synthetic  # noqa: E402
from tests.gl_pin_scenes import clip_scenes, vtk_ranges  # noqa: E402


def rle_encode(ids):
    """(h, w) int32 ids -> run lengths (uint16, runs longer than 65535 split) and the id of every run as the difference to the
    run before (int32): a face-id image is piecewise constant along its rows."""
    flat = np.asarray(ids, dtype=np.int64).reshape(-1)
    start = np.flatnonzero(np.diff(flat, prepend=flat[0] - 1))
    lengths = np.diff(np.append(start, flat.size))
    reps = (lengths + 65534) // 65535                       # pieces of at most 65535 per run
    vals = np.repeat(flat[start], reps)
    piece = np.full(vals.size, 65535, dtype=np.int64)
    last = np.cumsum(reps) - 1
    piece[last] = lengths - (reps - 1) * 65535
    return piece.astype(np.uint16), np.diff(vals, prepend=0).astype(np.int32)


def info_array(glr):
    return np.array([f"{k}={v}" for k, v in glr.info.items()])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--full-size", action="store_true")
    args = ap.parse_args()
    backends = {name: GLRasterizer(name) for name in ("llvmpipe", "swiftshader")}
    for name, glr in backends.items():
        print(name, glr.info)

    # ---- C1 ---------------------------------------------------------------------------------------------------------------
    (points, faces), cams = synthetic.config1_scene()
    recs = cams.get_raster_records(1.0, near=0.05)
    out = {}
    for name, glr in backends.items():
        glr.upload_mesh(points, faces)
        out[f"{name}_ids"] = np.stack([glr.render_ids(recs[v], 480, 640) for v in range(len(recs))])
        out[f"{name}_info"] = info_array(glr)
    out["llvmpipe_vtk_matrix_ids"] = np.stack(
        [backends["llvmpipe"].render_ids(recs[v], 480, 640, transform="vtk_matrix") for v in range(len(recs))])
    out["c1_records"] = recs
    # the reference's own test fixture: one mesh interval per pixel, every pixel centre ON a quad diagonal
    (spoints, sfaces), _ = synthetic.make_simple_mesh([], 255)
    scams = synthetic.make_simple_camera_set()
    near, far = vtk_ranges(scams, spoints)[0]
    srec = scams.get_raster_records(1.0, near=near)
    for name, glr in backends.items():
        glr.upload_mesh(spoints, sfaces)
        out[f"{name}_simple_ids"] = glr.render_ids(srec[0], 200, 200, far=far)
    out["simple_record"] = srec[0]
    out["simple_far"] = np.float64(far)
    np.savez_compressed(HERE / "reference_gl_c1.npz", **out)

    # ---- C2 and forest at scale 0.25 -----------------------------------------------------------------------------------------
    out = {}
    tpoints, tfaces = synthetic.terrain_mesh()
    fpoints, ffaces = synthetic.forest_scene()
    c2recs = synthetic.config2_cameras(50).get_raster_records(0.25, near=1.0)
    forecs = synthetic.oblique_cameras(20).get_raster_records(0.25, near=1.0)
    for name, glr in backends.items():
        glr.upload_mesh(tpoints, tfaces)
        out[f"{name}_c2_ids"] = np.stack([glr.render_ids(c2recs[v], 750, 1000) for v in (0, 23)])
        glr.upload_mesh(fpoints, ffaces)
        out[f"{name}_forest_ids"] = np.stack([glr.render_ids(forecs[v], 750, 1000) for v in (3, 11)])
        out[f"{name}_info"] = info_array(glr)
    out["c2_records"] = c2recs[[0, 23]]
    out["forest_records"] = forecs[[3, 11]]
    np.savez_compressed(HERE / "reference_gl_scaled.npz", **out)

    # ---- clipping ------------------------------------------------------------------------------------------------------------
    out = {}
    for scene, pts, fcs, cset, h, w in clip_scenes():
        ranges = vtk_ranges(cset, pts)
        crecs = cset.get_raster_records(1.0, near=[r[0] for r in ranges])
        for name, glr in backends.items():
            glr.upload_mesh(pts, fcs)
            out[f"{name}_{scene}_ids"] = np.stack(
                [glr.render_ids(crecs[v], h, w, far=ranges[v][1]) for v in range(len(crecs))])
        out[f"{scene}_records"] = crecs
        out[f"{scene}_far"] = np.array([r[1] for r in ranges])
    np.savez_compressed(HERE / "reference_gl_clip.npz", **out)
    # ---- one BASELINE-size view: C2 view 23 at 4000 x 3000, llvmpipe (the reference Dockerfile's GL) ----------------------------
    glr = backends["llvmpipe"]
    glr.upload_mesh(tpoints, tfaces)
    rec = synthetic.config2_cameras(50).get_raster_records(1.0, near=1.0)[23]
    run_len, val_delta = rle_encode(glr.render_ids(rec, 3000, 4000))
    np.savez_compressed(HERE / "reference_gl_c2_full.npz", run_len=run_len, val_delta=val_delta, record=rec, view=np.int32(23),
                        h=np.int32(3000), w=np.int32(4000), llvmpipe_info=info_array(glr))
    for f in ("reference_gl_c1.npz", "reference_gl_scaled.npz", "reference_gl_clip.npz", "reference_gl_c2_full.npz"):
        print(f, (HERE / f).stat().st_size // 1024, "KiB")

    if args.full_size:
        full_size_log(backends)


def compare(gl_ids, points, faces, rec, h, w, bits, oracle_c):
    """Counts of one view: differing pixels, differing pixels that carry an id of the oracle's 3x3 neighbourhood, and the
    disagreements on the pixels the envelope calls implementation-independent at delta = 2^-bits + 2e-3."""
    orc = oracle_c.raster(points, faces, rec, h, w)
    diff = gl_ids != orc
    pad = np.pad(orc, 1, mode="edge")
    near = np.zeros_like(diff)
    for dy in range(3):
        for dx in range(3):
            near |= pad[dy:dy + h, dx:dx + w] == gl_ids
    cls, env_ids, straddle = oracle_c.envelope(points, faces, rec, h, w, delta=2.0 ** -bits + 2e-3)
    indep = cls != 2
    want = np.where(cls == 1, env_ids, -1)
    return {"pixels": h * w, "differ": int(diff.sum()), "differ_with_neighbour_id": int((diff & near).sum()),
            "independent": int(indep.sum()), "independent_disagree": int((indep & (gl_ids != want)).sum()),
            "straddlers": int(straddle)}


def full_size_log(backends):
    from oracle import oracle_c

    log = ROOT / "profiles" / "r05_gl_pin.log"
    lines = []

    def say(s):
        print(s, flush=True)
        lines.append(s)

    for name, glr in backends.items():
        say(f"# {name}: " + ", ".join(f"{k}={v}" for k, v in glr.info.items()))
    tpoints, tfaces = synthetic.terrain_mesh()
    jobs = [("C2 4000x3000", tpoints, tfaces, synthetic.config2_cameras(50).get_raster_records(1.0, near=1.0), (0, 23, 49), 3000, 4000)]
    fpoints, ffaces = synthetic.forest_scene()
    jobs.append(("forest 4000x3000", fpoints, ffaces, synthetic.oblique_cameras(20).get_raster_records(1.0, near=1.0), (3, 11), 3000, 4000))
    (p5, f5), c5 = synthetic.config5_scene(n_views=60)
    jobs.append(("C5 6000x4000", p5, f5, c5.get_raster_records(1.0, near=1.0), (57,), 4000, 6000))
    for title, pts, fcs, recs, views, h, w in jobs:
        for name, glr in backends.items():
            bits = glr.info["GL_SUBPIXEL_BITS"]
            glr.upload_mesh(pts, fcs)
            for v in views:
                t0 = time.time()
                ids = glr.render_ids(recs[v], h, w)
                dt = time.time() - t0
                c = compare(ids, pts, fcs, recs[v], h, w, bits, oracle_c)
                say(f"{title} view {v} {name} ({dt:.1f} s render): identical {100 * (1 - c['differ'] / c['pixels']):.4f} % "
                    f"({c['differ']} differ, {c['differ_with_neighbour_id']} of them carry a 3x3-neighbour id); "
                    f"implementation-independent at {bits} bits {100 * c['independent'] / c['pixels']:.3f} % of the pixels, "
                    f"disagreements there: {c['independent_disagree']}; near-plane straddlers {c['straddlers']}")
    log.write_text("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
