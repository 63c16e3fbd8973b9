"""tools/forest_stats.py -- CPU statistics of the hostile workload (terrain + 20 000 trees, oblique views): surviving faces,
faces by tile footprint, faces whose bounding box holds at most 2 x 2 pixel centres and cover none (a cull that was considered:
it would drop 0.01 % of the faces at 4000x3000, 4 % at 1000x750).  numpy only; numbers quoted in DESIGN.md section 5."""
import sys, numpy as np
from pathlib import Path; sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from geograypher_amd.utils import synthetic
pts, faces = synthetic.forest_scene()
cams = synthetic.oblique_cameras(20)
for scale in (1.0, 0.25):
    h, w = cams[0].get_image_size(scale)
    recs = cams.get_raster_records(scale, near=1.0)
    for vi in (3, 11):
        cam = recs[vi].astype(np.float32)
        R = cam[:9].reshape(3, 3); t = cam[9:12]; fe, cx, cy, near = cam[12:16]
        d = pts.astype(np.float32) - t
        q = d @ R  # q_c = sum_r R[r][c] d_r
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
        n = ok.sum()
        nj = (jmax - jmin + 1)[ok]; ni = (imax - imin + 1)[ok]
        tx0, tx1 = (jmin >> 6)[ok], (jmax >> 6)[ok]; ty0, ty1 = (imin >> 5)[ok], (imax >> 5)[ok]
        ntile = (tx1 - tx0 + 1) * (ty1 - ty0 + 1)
        small = (tx1 - tx0 <= 1) & (ty1 - ty0 <= 1)
        print(f"scale {scale} view {vi}: records {n}, small {small.sum()} entries_small {ntile[small].sum()}, big {(~small).sum()} bbox_entries_big {ntile[~small].sum()}")
        tiny = (nj <= 2) & (ni <= 2)
        print(f"   bbox<=2x2 px centres: {tiny.sum()} ({tiny.mean():.3f}); 1x1: {((nj==1)&(ni==1)).sum()}; <=4x4 {((nj<=4)&(ni<=4)).sum()}; <=8x8 {((nj<=8)&(ni<=8)).sum()}")
        # exact coverage for tiny faces
        idx = np.nonzero(ok)[0][tiny]
        s = np.sign(area[idx])
        cov = np.zeros(len(idx), bool)
        for di in range(2):
            for dj in range(2):
                px = (jmin[idx] + dj) * 256 + 128; py = (imin[idx] + di) * 256 + 128
                inb = (jmin[idx] + dj <= jmax[idx]) & (imin[idx] + di <= imax[idx])
                def E(xa, ya, xb, yb):
                    dx = xb - xa; dy = yb - ya
                    e = (dx * (py - ya) - dy * (px - xa)) * s
                    dxs, dys = dx * s, dy * s
                    tl = (dys < 0) | ((dys == 0) & (dxs > 0))
                    return (e > 0) | ((e == 0) & tl)
                c = E(X0[idx], Y0[idx], X1[idx], Y1[idx]) & E(X1[idx], Y1[idx], X2[idx], Y2[idx]) & E(X2[idx], Y2[idx], X0[idx], Y0[idx])
                cov |= c & inb
        print(f"   tiny faces covering no pixel centre: {(~cov).sum()} of {len(idx)}  -> records after tiny cull {n - (~cov).sum()}")
        # entries per tile distribution (bbox-based)
