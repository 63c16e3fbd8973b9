"""tools/forest_items.py [forest|c2] -- (face, tile) entries, scanline work items (row mode, column mode, the smaller of the
two per entry), pixels drawn and their distribution over tiles, for one view of the hostile workload or of C2.  numpy only;
numbers quoted in DESIGN.md section 5 (the forest has 13 x the work items of C2 and a depth complexity of 10.5)."""
import sys, numpy as np
from pathlib import Path; sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from geograypher_amd.utils import synthetic
which = sys.argv[1] if len(sys.argv) > 1 else 'forest'
if which == 'forest':
    pts, faces = synthetic.forest_scene(); cams = synthetic.oblique_cameras(20); views=(3,)
    scales = (1.0, 0.25)
else:
    pts, faces = synthetic.terrain_mesh(); cams = synthetic.survey_cameras(10, 5, 40.0, 60.0, seed=3); views=(7,)
    scales = (1.0,)
TW, TH = 64, 32
for scale in scales:
    h, w = cams[0].get_image_size(scale)
    recs = cams.get_raster_records(scale, near=1.0)
    for vi in views:
        cam = recs[vi].astype(np.float32)
        R = cam[:9].reshape(3, 3); t = cam[9:12]; fe, cx, cy, near = cam[12:16]
        d = pts.astype(np.float32) - t
        q = d @ R
        valid = q[:, 2] > near
        iz = 1.0 / np.where(valid, q[:, 2], 1)
        sx = cx + fe * q[:, 0] * iz; sy = cy + fe * q[:, 1] * iz
        valid &= (np.abs(sx) < 16384) & (np.abs(sy) < 16384)
        X = np.floor(sx * 256 + 0.5).astype(np.int64); Y = np.floor(sy * 256 + 0.5).astype(np.int64)
        f = faces
        ok = valid[f].all(1)
        X0, X1, X2 = X[f[:, 0]], X[f[:, 1]], X[f[:, 2]]
        Y0, Y1, Y2 = Y[f[:, 0]], Y[f[:, 1]], Y[f[:, 2]]
        area = (X1 - X0) * (Y2 - Y0) - (X2 - X0) * (Y1 - Y0)
        ok &= area != 0
        Xmin = np.minimum(np.minimum(X0, X1), X2); Xmax = np.maximum(np.maximum(X0, X1), X2)
        Ymin = np.minimum(np.minimum(Y0, Y1), Y2); Ymax = np.maximum(np.maximum(Y0, Y1), Y2)
        jmin = np.maximum((Xmin - 128 + 255) >> 8, 0); jmax = np.minimum((Xmax - 128) >> 8, w - 1)
        imin = np.maximum((Ymin - 128 + 255) >> 8, 0); imax = np.minimum((Ymax - 128) >> 8, h - 1)
        ok &= (jmin <= jmax) & (imin <= imax)
        idx = np.nonzero(ok)[0]
        jmin, jmax, imin, imax = jmin[idx], jmax[idx], imin[idx], imax[idx]
        # approximate per-row span width of the triangle: area / height
        A = np.abs(area[idx]) / 65536.0 / 2
        tx0, tx1, ty0, ty1 = jmin >> 6, jmax >> 6, imin >> 5, imax >> 5
        ntx = tx1 - tx0 + 1; nty = ty1 - ty0 + 1
        n_pairs = ntx * nty
        fi = np.repeat(np.arange(len(idx)), n_pairs)
        off = np.arange(n_pairs.sum()) - np.repeat(np.cumsum(n_pairs) - n_pairs, n_pairs)
        tx = tx0[fi] + off % ntx[fi]; ty = ty0[fi] + off // ntx[fi]
        rows = np.minimum(imax[fi], ty * TH + TH - 1) - np.maximum(imin[fi], ty * TH) + 1
        cols = np.minimum(jmax[fi], tx * TW + TW - 1) - np.maximum(jmin[fi], tx * TW) + 1
        # approx covered pixels in this tile: face area * (tile-bbox area / bbox area)
        bw = (jmax - jmin + 1)[fi]; bh = (imax - imin + 1)[fi]
        pix = A[fi] * (rows * cols) / (bw * bh)
        # dead entries: triangle misses tile though bbox touches -- approximate: pix < 0.02*rows*cols and face is big
        TX = (w + TW - 1) // TW
        tile = ty * TX + tx
        T = TX * ((h + TH - 1) // TH)
        print(f"{which} scale {scale} view {vi}: records {len(idx)} entries(bbox) {len(fi)} tiles {T}")
        print(f"   items row-mode {rows.sum()}  col-mode {cols.sum()}  min-mode {np.minimum(rows, cols).sum()}  pixels~ {pix.sum():.0f} ({pix.sum()/(h*w):.2f} x image)")
        print(f"   mean rows {rows.mean():.1f} cols {cols.mean():.1f}; pix/row-item {pix.sum()/rows.sum():.2f}; pix/min-item {pix.sum()/np.minimum(rows,cols).sum():.2f}")
        ept = np.bincount(tile, minlength=T); ipt = np.bincount(tile, weights=rows, minlength=T); mpt = np.bincount(tile, weights=np.minimum(rows, cols), minlength=T)
        print(f"   entries/tile mean {ept.mean():.0f} max {ept.max()} p99 {np.percentile(ept,99):.0f}; row-items/tile mean {ipt.mean():.0f} max {ipt.max():.0f}; min-items/tile mean {mpt.mean():.0f} max {mpt.max():.0f}")
        tall = rows > 2 * cols
        print(f"   entries with rows > 2*cols: {tall.mean():.2f} holding {rows[tall].sum()/rows.sum():.2f} of row-items")
        np.save(f"/tmp/ipt_{which}_{scale}.npy", ipt)
