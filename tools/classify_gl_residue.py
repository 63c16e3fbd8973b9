#!/usr/bin/env python3
"""tools/classify_gl_residue.py [--render] -- WHY do the oracle and Mesa llvmpipe differ on 0.003 % of the pixels?  (build container
or anywhere: no GPU; `--render` needs the GL stack of the build container.)

For every pixel of the committed GL goldens (tests/golden/reference_gl_c1.npz: C1, 8 views; reference_gl_scaled.npz: C2 views 0 / 23
at render_img_scale 0.25; with --render also one freshly rendered C2 view at 4000 x 3000) where the rule-set oracle and llvmpipe
return different faces -- A the oracle's, B llvmpipe's -- the vertex stage of llvmpipe is restated in numpy float32, op for op as
Mesa 23.2's draw module JITs it (src/gallium/auxiliary/draw/draw_llvm.c generate_viewport; src/gallium/drivers/llvmpipe/
lp_setup_tri.c subpixel_snap):

    clip   = vertex shader:  x_c = P_x q_x,  y_c = P_y q_y,  w = q_z            (q: camera-space float32, as handed to GL)
    rw     = 1 / w                                       (one IEEE division)
    ndc    = x_c * rw                                    (one multiply)
    win    = fmuladd(ndc, scale, translate)              (llvm.fmuladd: FUSED on a host with FMA3, two roundings otherwise)
    fixed  = lrintf(256 (win - 0.5))                     (round half to even; pixel centres at integers of that grid)

against rule R1 of DESIGN.md (s = c + (f q) (1 / q_z), X = floor(256 s + 0.5), pixel centres at 256 j + 128).  Both feed the same
exact integer edge functions with a top-left rule, so a pixel changes hands only where a vertex lands on a different 1/256 step.
Each differing pixel is then decided again with llvmpipe's vertices and classed:

    (a)  snap: with the GL-order vertices the rule-set picks B -- the vertex transform's rounding decides the pixel
    (b)  depth tie: A and B both cover the pixel under both vertex sets and their 24-bit window depths are equal (GL_LEQUAL lets
         the LATER face of the draw order win; rule R5 the lower id)
    (d)  tie convention: the pixel centre lies exactly ON an exactly horizontal snapped edge shared by A and B (both models agree
         on the vertices).  Until round 6 rule R3 gave such a pixel to the triangle BELOW the edge (top-left rule, rows top-down);
         Mesa gives it to the triangle ABOVE (its top-left rule lives in GL's bottom-up window space: st_atom_rasterizer.c
         bottom_edge_rule).  R3 now follows Mesa: the class is empty on the current tree (7 of 66 pixels before)
    (e)  viewport clipping: A or B crosses the border of the image.  llvmpipe's draw module clips such a face against the view
         volume (new vertices on the border, snapped again), the rule-set rasterizes it whole inside its guard band (R1): a pixel
         within a quarter of a 1/256 px step of the shared edge can change hands
    (c)  other

Prints the table (committed as profiles/r06_gl_residue.txt)."""
import argparse
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from geograypher_amd.utils import synthetic  # noqa: E402
from oracle import oracle_c  # noqa: E402

f32 = np.float32


def camera_space(points, rec):
    """q = R^T (p - t) in float32, every operation rounded individually (rule R1; what gl_raster.py hands to GL)."""
    p = np.asarray(points, dtype=f32)
    R, t = rec[0:9].reshape(3, 3).astype(f32), rec[9:12].astype(f32)
    d = p - t[None, :]
    q = np.empty_like(d)
    for c in range(3):
        q[:, c] = (R[0, c] * d[:, 0] + R[1, c] * d[:, 1]) + R[2, c] * d[:, 2]
    return q


def snap_rule_r1(q, rec):
    """Snapped vertices of rule R1, frame: pixel (j, i) has its centre at (256 j + 128, 256 i + 128)."""
    fe, cx, cy = f32(rec[12]), f32(rec[13]), f32(rec[14])
    iz = f32(1.0) / q[:, 2]
    sx = cx + (fe * q[:, 0]) * iz
    sy = cy + (fe * q[:, 1]) * iz
    X = np.floor(sx * f32(256.0) + f32(0.5)).astype(np.int64)
    Y = np.floor(sy * f32(256.0) + f32(0.5)).astype(np.int64)
    return X, Y, iz


def fma32(a, b, c):
    """float32 fused multiply-add through float64 (the product of two float32 is exact in float64)."""
    return (a.astype(np.float64) * np.float64(b) + np.float64(c)).astype(f32)


def snap_llvmpipe(q, rec, h, w, fused=True):
    """Snapped vertices as llvmpipe's pipeline produces them, mapped into R1's frame (rows top-down)."""
    fe = f32(rec[12])
    px, py = f32(2.0 * float(fe) / w), f32(-2.0 * float(fe) / h)     # the `proj` uniform of tests/golden/gl_raster.py
    xc, yc = px * q[:, 0], py * q[:, 1]
    rw = f32(1.0) / q[:, 2]
    xn, yn = xc * rw, yc * rw
    sxv, syv = f32(w / 2.0), f32(h / 2.0)                             # viewport scale = translate = size / 2
    if fused:
        xw, yw = fma32(xn, sxv, sxv), fma32(yn, syv, syv)
    else:
        xw, yw = xn * sxv + sxv, yn * syv + syv
    Xf = np.rint((xw - f32(0.5)) * f32(256.0)).astype(np.int64)      # lrintf: round half to even, like np.rint
    Yf = np.rint((yw - f32(0.5)) * f32(256.0)).astype(np.int64)
    # llvmpipe's grid has pixel centres at integers x 256, rows bottom-up; R1's at 256 j + 128, rows top-down
    return Xf + 128, 256 * h - 128 - Yf


def covers(X, Y, tri, Px, Py):
    """Rule R3 for one face (vertex ids `tri`) at the pixel centre (Px, Py): exact integers, top-left rule."""
    x = [int(X[k]) for k in tri]
    y = [int(Y[k]) for k in tri]
    area2 = (x[1] - x[0]) * (y[2] - y[0]) - (x[2] - x[0]) * (y[1] - y[0])
    if area2 == 0:
        return False
    if area2 < 0:
        x[1], x[2], y[1], y[2] = x[2], x[1], y[2], y[1]
    for k in range(3):
        a, b = k, (k + 1) % 3
        dx, dy = x[b] - x[a], y[b] - y[a]
        E = dx * (Py - y[a]) - dy * (Px - x[a])
        top_left = dy < 0 or (dy == 0 and dx < 0)   # R3: left and bottom edges own their pixels (Mesa's convention, rows top-down)
        if E < 0 or (E == 0 and not top_left):
            return False
    return True


def on_horizontal_edge(X, Y, tri, Px, Py):
    """the pixel centre lies exactly on an exactly horizontal snapped edge of the face"""
    for k in range(3):
        a, b = tri[k], tri[(k + 1) % 3]
        if int(Y[a]) == int(Y[b]) == Py and min(int(X[a]), int(X[b])) <= Px <= max(int(X[a]), int(X[b])):
            return True
    return False


def crosses_viewport(X, Y, tri, h, w):
    xs, ys = [int(X[k]) for k in tri], [int(Y[k]) for k in tri]
    return min(xs) < 0 or max(xs) > 256 * w or min(ys) < 0 or max(ys) > 256 * h


def depth24(X, Y, iz, tri, Px, Py, near, far):
    """The 24-bit window depth llvmpipe would store for the face at the pixel, to the accuracy a tie test needs: window z is
    affine in 1/z_eye; interpolated through the plane of the three vertices (float64 here)."""
    x = np.array([X[k] for k in tri], dtype=np.float64)
    y = np.array([Y[k] for k in tri], dtype=np.float64)
    z = np.array([iz[k] for k in tri], dtype=np.float64)
    M = np.stack([x - x[0], y - y[0]], axis=1)[1:]
    try:
        g = np.linalg.solve(M, (z - z[0])[1:])
    except np.linalg.LinAlgError:
        return None
    izp = z[0] + g[0] * (Px - x[0]) + g[1] * (Py - y[0])
    A, B = (far + near) / (far - near), -2.0 * far * near / (far - near)
    zn = A + B * izp                       # z_ndc = (A z + B) / z
    return int(np.floor((0.5 * zn + 0.5) * 16777215.0 + 0.5))


def classify(name, points, faces, rec, h, w, gl_ids, far=None):
    want = oracle_c.raster(points, faces, rec, h, w)
    diff = np.argwhere(want != gl_ids)
    q = camera_space(points, rec)
    X1, Y1, iz = snap_rule_r1(q, rec)
    out = {"scene": name, "pixels": h * w, "differ": int(diff.shape[0])}
    near = float(rec[15])
    if far is None:
        far = max(2.0 * float(np.nanmax(q[:, 2])), 10.0 * near)
    for fused in (True, False):
        Xg, Yg = snap_llvmpipe(q, rec, h, w, fused=fused)
        moved = (Xg != X1) | (Yg != Y1)
        tag = "fused" if fused else "unfused"
        out[f"vertices_on_another_step_{tag}"] = round(float(moved.mean()), 5)
        out[f"largest_step_difference_{tag}"] = int(max(np.abs(Xg - X1).max(), np.abs(Yg - Y1).max()))
        a = b = c = d = e = 0
        for i, j in diff:
            Px, Py = 256 * int(j) + 128, 256 * int(i) + 128
            A, B = int(want[i, j]), int(gl_ids[i, j])
            cov = {}
            for label, (X, Y) in (("r1", (X1, Y1)), ("gl", (Xg, Yg))):
                for fname, fid in (("A", A), ("B", B)):
                    cov[(label, fname)] = fid >= 0 and covers(X, Y, faces[fid], Px, Py)
            if cov[("gl", "B")] and not cov[("gl", "A")]:
                a += 1                                    # with llvmpipe's vertices the rule-set gives the pixel to B
                continue
            if B < 0 and not cov[("gl", "A")]:
                a += 1                                    # ... or to nobody (background), as llvmpipe does
                continue
            if A >= 0 and B >= 0 and cov[("gl", "A")] and cov[("gl", "B")]:
                zA = depth24(Xg, Yg, iz, faces[A], Px, Py, near, far)
                zB = depth24(Xg, Yg, iz, faces[B], Px, Py, near, far)
                if zA is not None and zB is not None and abs(zA - zB) <= 1:
                    b += 1
                    continue
            if on_horizontal_edge(Xg, Yg, faces[A], Px, Py) if A >= 0 else False:
                d += 1
                continue
            if crosses_viewport(Xg, Yg, faces[A] if A >= 0 else faces[B], h, w) or (B >= 0 and crosses_viewport(Xg, Yg, faces[B], h, w)):
                e += 1
                continue
            c += 1
        out[f"class_d_tie_convention_{tag}"], out[f"class_e_viewport_clip_{tag}"] = d, e
        out[f"class_a_snap_{tag}"], out[f"class_b_depth_tie_{tag}"], out[f"class_c_other_{tag}"] = a, b, c
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--render", action="store_true", help="also render one 4000 x 3000 C2 view with llvmpipe (build container)")
    args = ap.parse_args()
    rows = []
    (points, faces), _ = synthetic.config1_scene()
    g = np.load(ROOT / "tests" / "golden" / "reference_gl_c1.npz")
    for v in range(8):
        rows.append(classify(f"C1 view {v} 640x480", points, faces, g["c1_records"][v], 480, 640, g["llvmpipe_ids"][v]))
    tp, tf = synthetic.terrain_mesh()
    g2 = np.load(ROOT / "tests" / "golden" / "reference_gl_scaled.npz")
    for k, v in enumerate((0, 23)):
        rows.append(classify(f"C2 view {v} 1000x750", tp, tf, g2["c2_records"][k], 750, 1000, g2["llvmpipe_c2_ids"][k]))
    full = ROOT / "tests" / "golden" / "reference_gl_c2_full.npz"
    if full.is_file():
        g3 = np.load(full)
        ids = np.repeat(np.cumsum(g3["val_delta"].astype(np.int64)), g3["run_len"].astype(np.int64)).astype(np.int32).reshape(3000, 4000)
        rows.append(classify(f"C2 view {int(g3['view'])} 4000x3000", tp, tf, g3["record"], 3000, 4000, ids))
    elif args.render:
        sys.path.insert(0, str(ROOT / "tests" / "golden"))
        from gl_raster import GLRasterizer

        glr = GLRasterizer("llvmpipe")
        glr.upload_mesh(tp, tf)
        rec = synthetic.config2_cameras(50).get_raster_records(1.0, near=1.0)[23]
        rows.append(classify("C2 view 23 4000x3000", tp, tf, rec, 3000, 4000, glr.render_ids(rec, 3000, 4000)))
    keys = list(rows[0].keys())
    print("# oracle (rule-set R0-R7) against Mesa 23.2.1 llvmpipe: every differing pixel decided again with llvmpipe's vertex transform")
    print("# restated in numpy float32 (tools/classify_gl_residue.py).  fused / unfused: the viewport transform as one FMA (a host with")
    print("# FMA3: this one) or as multiply + add.")
    for r in rows:
        print(" | ".join(f"{k}={r[k]}" for k in keys))
    for tag in ("fused", "unfused"):
        tot = sum(r["differ"] for r in rows)
        a = sum(r[f"class_a_snap_{tag}"] for r in rows)
        b = sum(r[f"class_b_depth_tie_{tag}"] for r in rows)
        c = sum(r[f"class_c_other_{tag}"] for r in rows)
        d = sum(r[f"class_d_tie_convention_{tag}"] for r in rows)
        e = sum(r[f"class_e_viewport_clip_{tag}"] for r in rows)
        print(f"# total ({tag} viewport): {tot} differing pixels: (a) snap {a} = {100.0 * a / max(tot, 1):.1f} %, (b) depth tie {b}, "
              f"(d) tie convention {d}, (e) viewport clipping {e}, (c) other {c}")
    return 0


if __name__ == "__main__":
    sys.exit(main())
