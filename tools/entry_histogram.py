#!/usr/bin/env python3
"""tools/entry_histogram.py -- CPU statistics (numpy only): how large are the (face, tile) entries of a view, and how much of the
tile kernel's WORK (scanline items = rows of entries) sits in the small ones?  For the hostile forest and BASELINE config 2, at
render_img_scale 1 and 0.25.  The question behind the micro lists: an entry whose part of the face's pixel
bounding box in its tile is at most 2 x 2 / 4 x 4 / 8 x 8 pixels could be point-sampled by one lane instead of being cut into
scanline items -- worth it where such entries carry the items.  Output committed as profiles/r05_entry_histogram.txt."""
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from geograypher_amd.utils import synthetic


def stats(name, pts, faces, cams, scale, views):
    h, w = cams[0].get_image_size(scale)
    recs = cams.get_raster_records(scale, near=1.0)
    for vi in views:
        cam = recs[vi].astype(np.float32)
        R = cam[:9].reshape(3, 3); t = cam[9:12]; fe, cx, cy, near = cam[12:16]
        q = (pts.astype(np.float32) - t) @ R
        valid = q[:, 2] > near
        iz = 1.0 / np.where(valid, q[:, 2], 1)
        sx = cx + fe * q[:, 0] * iz; sy = cy + fe * q[:, 1] * iz
        valid &= (np.abs(sx) < 16384) & (np.abs(sy) < 16384)
        X = np.floor(sx * 256 + 0.5).astype(np.int64); Y = np.floor(sy * 256 + 0.5).astype(np.int64)
        f = faces
        ok = valid[f].all(1)
        X0, X1, X2 = X[f[:, 0]], X[f[:, 1]], X[f[:, 2]]
        Y0, Y1, Y2 = Y[f[:, 0]], Y[f[:, 1]], Y[f[:, 2]]
        ok &= ((X1 - X0) * (Y2 - Y0) - (X2 - X0) * (Y1 - Y0)) != 0
        Xmin = np.minimum(np.minimum(X0, X1), X2); Xmax = np.maximum(np.maximum(X0, X1), X2)
        Ymin = np.minimum(np.minimum(Y0, Y1), Y2); Ymax = np.maximum(np.maximum(Y0, Y1), Y2)
        jmin = np.maximum((Xmin - 128 + 255) >> 8, 0); jmax = np.minimum((Xmax - 128) >> 8, w - 1)
        imin = np.maximum((Ymin - 128 + 255) >> 8, 0); imax = np.minimum((Ymax - 128) >> 8, h - 1)
        ok &= (jmin <= jmax) & (imin <= imax)
        jmin, jmax, imin, imax = jmin[ok], jmax[ok], imin[ok], imax[ok]
        whole = ((jmax - jmin < 4) & (imax - imin < 4)).mean()     # faces whose WHOLE box is at most 4 x 4: the statistic of K1
        TW, TH = 64, 32
        tot = 0; rows_tot = 0
        ent = {2: 0, 4: 0, 8: 0}; rows = {2: 0, 4: 0, 8: 0}
        tx0, tx1, ty0, ty1 = jmin >> 6, jmax >> 6, imin >> 5, imax >> 5
        for dy in range(int((ty1 - ty0).max()) + 1):
            for dx in range(int((tx1 - tx0).max()) + 1):
                sel = (tx0 + dx <= tx1) & (ty0 + dy <= ty1)
                if not sel.any():
                    continue
                tx = tx0[sel] + dx; ty = ty0[sel] + dy
                nj = np.minimum(jmax[sel], tx * TW + TW - 1) - np.maximum(jmin[sel], tx * TW) + 1
                ni = np.minimum(imax[sel], ty * TH + TH - 1) - np.maximum(imin[sel], ty * TH) + 1
                tot += int(sel.sum()); rows_tot += int(ni.sum())
                for k in ent:
                    m = (nj <= k) & (ni <= k)
                    ent[k] += int(m.sum()); rows[k] += int(ni[m].sum())
        print(f"{name:7s} scale {scale:4.2f} view {vi:2d}: faces {int(ok.sum()):7d} (whole box <= 4x4: {100 * whole:5.1f} %)  entries {tot:8d}  items {rows_tot:9d} | "
              f"entries with box <= 2x2 / 4x4 / 8x8: {100 * ent[2] / tot:5.1f} / {100 * ent[4] / tot:5.1f} / {100 * ent[8] / tot:5.1f} % | "
              f"their share of the ITEMS: {100 * rows[2] / rows_tot:5.1f} / {100 * rows[4] / rows_tot:5.1f} / {100 * rows[8] / rows_tot:5.1f} %")


def main():
    fp, ff = synthetic.forest_scene(); fc = synthetic.oblique_cameras(20)
    tp, tf = synthetic.terrain_mesh(); tc = synthetic.config2_cameras(50)
    for s in (1.0, 0.25):
        stats("forest", fp, ff, fc, s, (3, 11))
        stats("C2", tp, tf, tc, s, (0, 23))


if __name__ == "__main__":
    main()
