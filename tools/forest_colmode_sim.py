"""Model of the tile kernel's VALU cost per view: row items only vs per-entry choice of row / column items.
Batch of 64 consecutive items: cost = FIXED + STEP * max over lanes of ceil(span/2).  Entries in list order ~ Morton order."""
import sys, numpy as np
from pathlib import Path; sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from geograypher_amd.utils import synthetic
which = sys.argv[1] if len(sys.argv) > 1 else 'c2'
if which == 'forest':
    pts, faces = synthetic.forest_scene(); cams = synthetic.oblique_cameras(20); vi = 3
else:
    pts, faces = synthetic.terrain_mesh(); cams = synthetic.survey_cameras(10, 5, 40.0, 60.0, seed=3); vi = 7
TW, TH = 64, 32
h, w = cams[0].get_image_size(1.0)
cam = cams.get_raster_records(1.0, near=1.0)[vi].astype(np.float32)
R = cam[:9].reshape(3, 3); t = cam[9:12]; fe, cx, cy, near = cam[12:16]
q = (pts.astype(np.float32) - t) @ R
valid = q[:, 2] > near
iz = 1.0 / np.where(valid, q[:, 2], 1)
sx = cx + fe * q[:, 0] * iz; sy = cy + fe * q[:, 1] * iz
valid &= (np.abs(sx) < 16384) & (np.abs(sy) < 16384)
X = np.floor(sx * 256 + 0.5).astype(np.int64); Y = np.floor(sy * 256 + 0.5).astype(np.int64)
f = faces
ok = valid[f].all(1)
X0, X1, X2 = X[f[:, 0]], X[f[:, 1]], X[f[:, 2]]; Y0, Y1, Y2 = Y[f[:, 0]], Y[f[:, 1]], Y[f[:, 2]]
area = (X1 - X0) * (Y2 - Y0) - (X2 - X0) * (Y1 - Y0)
ok &= area != 0
jmin = np.maximum((np.minimum(np.minimum(X0, X1), X2) - 128 + 255) >> 8, 0); jmax = np.minimum((np.maximum(np.maximum(X0, X1), X2) - 128) >> 8, w - 1)
imin = np.maximum((np.minimum(np.minimum(Y0, Y1), Y2) - 128 + 255) >> 8, 0); imax = np.minimum((np.maximum(np.maximum(Y0, Y1), Y2) - 128) >> 8, h - 1)
ok &= (jmin <= jmax) & (imin <= imax)
idx = np.nonzero(ok)[0]
c = pts[f[idx]].mean(1)
def spread(x):
    x = x.astype(np.uint64) & 0xFFFF
    x = (x | (x << 8)) & 0x00FF00FF; x = (x | (x << 4)) & 0x0F0F0F0F; x = (x | (x << 2)) & 0x33333333; x = (x | (x << 1)) & 0x55555555
    return x
lo = pts.min(0); ext = pts.max(0) - lo
mort = spread(np.clip((c[:, 0] - lo[0]) / ext[0] * 65535, 0, 65535)) | (spread(np.clip((c[:, 1] - lo[1]) / ext[1] * 65535, 0, 65535)) << 1)
TX = (w + TW - 1) // TW; TY = (h + TH - 1) // TH
rng = np.random.default_rng(0)
tiles = rng.choice(TX * TY, size=int(sys.argv[2]) if len(sys.argv) > 2 else 60, replace=False)
FIXED, STEP = 91.0, 8.0
def batch_cost(steps_per_item, fixed):
    n = len(steps_per_item)
    cost = 0.0
    for b in range(0, n, 64):
        cost += fixed + STEP * steps_per_item[b:b + 64].max()
    return cost, (n + 63) // 64
tot = {'row': 0.0, 'min_mixed': 0.0, 'min_split': 0.0}; nb = {'row': 0, 'min_mixed': 0, 'min_split': 0}; items = {'row': 0, 'min': 0}
for tile in tiles:
    ty, tx = divmod(tile, TX); px0, py0 = tx * TW, ty * TH
    m = (jmin[idx] <= px0 + TW - 1) & (jmax[idx] >= px0) & (imin[idx] <= py0 + TH - 1) & (imax[idx] >= py0)
    e = np.nonzero(m)[0]
    if len(e) == 0: continue
    e = e[np.argsort(mort[e], kind='stable')]
    fi = idx[e]
    gx = (np.arange(TW) + px0) * 256 + 128; gy = (np.arange(TH) + py0) * 256 + 128
    GX, GY = np.meshgrid(gx, gy)
    ins = (GX[None] <= (w - 1) * 256 + 128) & (GY[None] <= (h - 1) * 256 + 128)
    s = np.sign(area[fi])[:, None, None]
    def E(xa, ya, xb, yb):
        dx = (xb - xa)[:, None, None]; dy = (yb - ya)[:, None, None]
        ev = (dx * (GY[None] - ya[:, None, None]) - dy * (GX[None] - xa[:, None, None])) * s
        tl = (dy * s < 0) | ((dy * s == 0) & (dx * s > 0))
        return (ev > 0) | ((ev == 0) & tl)
    cov = E(X0[fi], Y0[fi], X1[fi], Y1[fi]) & E(X1[fi], Y1[fi], X2[fi], Y2[fi]) & E(X2[fi], Y2[fi], X0[fi], Y0[fi]) & ins
    r0 = np.maximum(imin[fi], py0) - py0; r1 = np.minimum(imax[fi], py0 + TH - 1) - py0
    c0 = np.maximum(jmin[fi], px0) - px0; c1 = np.minimum(jmax[fi], px0 + TW - 1) - px0
    rowsteps, minsteps, mode = [], [], []
    split_row, split_col = [], []
    for k in range(len(fi)):
        rs = (cov[k, r0[k]:r1[k] + 1, :].sum(1) + 1) // 2   # steps per row item
        cs = (cov[k, :, c0[k]:c1[k] + 1].sum(0) + 1) // 2   # steps per column item
        rowsteps.append(rs)
        if len(cs) < len(rs):
            minsteps.append(cs); split_col.append(cs)
        else:
            minsteps.append(rs); split_row.append(rs)
    rs_all = np.concatenate(rowsteps); ms_all = np.concatenate(minsteps)
    items['row'] += len(rs_all); items['min'] += len(ms_all)
    cst, n = batch_cost(rs_all, FIXED); tot['row'] += cst; nb['row'] += n
    cst, n = batch_cost(ms_all, FIXED + 6 + 5.4); tot['min_mixed'] += cst; nb['min_mixed'] += n  # +6 per item, +1 per step
    for lst in (split_row, split_col):
        if lst:
            cst, n = batch_cost(np.concatenate(lst), FIXED); tot['min_split'] += cst; nb['min_split'] += n
print(which, 'items row', items['row'], 'min', items['min'], 'ratio', items['min'] / items['row'])
for k in tot: print(f"  {k:10s} cost {tot[k]:12.0f} batches {nb[k]:7d}  vs row {tot[k] / tot['row']:.3f}  per batch {tot[k]/nb[k]:.1f}")
