"""tools/forest_hiz_sim.py [scale] [block] [refresh] -- what a hierarchical-z cull of tile entries could remove on the hostile
workload: per sampled tile the entries are rasterized in numpy in four list orders (random, Morton, ideal front-to-back, blocks
by depth) and an entry is skipped when every block x block region its bounding box touches is already covered by nearer keys
(refreshed every `refresh` entries).  8-22 % of the work items at 4000x3000: not pursued."""
import sys, numpy as np
from pathlib import Path; sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from geograypher_amd.utils import synthetic
pts, faces = synthetic.forest_scene(); cams = synthetic.oblique_cameras(20)
TW, TH = 64, 32
scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
SB = int(sys.argv[2]) if len(sys.argv) > 2 else 8   # sub-block size
REFRESH = int(sys.argv[3]) if len(sys.argv) > 3 else 256
h, w = cams[0].get_image_size(scale)
recs = cams.get_raster_records(scale, near=1.0)
vi = 3
cam = recs[vi].astype(np.float32)
R = cam[:9].reshape(3, 3); t = cam[9:12]; fe, cx, cy, near = cam[12:16]
d = pts.astype(np.float32) - t
q = d @ R
valid = q[:, 2] > near
iz = (1.0 / np.where(valid, q[:, 2], 1)).astype(np.float64)
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
# morton order of centroid in plan (x,y)
c = pts[f[idx]].mean(1)
def spread(x):
    x = x.astype(np.uint64) & 0xFFFF
    x = (x | (x << 8)) & 0x00FF00FF; x = (x | (x << 4)) & 0x0F0F0F0F; x = (x | (x << 2)) & 0x33333333; x = (x | (x << 1)) & 0x55555555
    return x
lo = pts.min(0); ext = pts.max(0) - lo
u0 = np.clip((c[:, 0] - lo[0]) / ext[0] * 65535, 0, 65535); u1 = np.clip((c[:, 1] - lo[1]) / ext[1] * 65535, 0, 65535)
mort = spread(u0) | (spread(u1) << 1)
zf = iz[f[idx]]  # (n,3) inverse depth; larger = nearer
znear = zf.max(1)
TX = (w + TW - 1) // TW; TY = (h + TH - 1) // TH
rng = np.random.default_rng(0)
tiles = rng.choice(TX * TY, size=40, replace=False)
tot = {k: [0, 0] for k in ('random', 'morton', 'front2back', 'blockdepth')}
# block depth order: blocks of 64 morton-consecutive faces sorted by mean depth
order_m = np.argsort(mort, kind='stable')
blk_of = np.empty(len(idx), np.int64); blk_of[order_m] = np.arange(len(idx)) // 64
blk_depth = np.zeros(blk_of.max() + 1); np.maximum.at(blk_depth, blk_of, znear)
for tile in tiles:
    ty, tx = divmod(tile, TX)
    px0, py0 = tx * TW, ty * TH
    m = (jmin[idx] <= px0 + TW - 1) & (jmax[idx] >= px0) & (imin[idx] <= py0 + TH - 1) & (imax[idx] >= py0)
    e = np.nonzero(m)[0]
    if len(e) == 0: continue
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
    # depth plane
    z0, z1, z2 = zf[e, 0], zf[e, 1], zf[e, 2]
    a2 = area[fi].astype(np.float64)
    A = ((z1 - z0) * (Y2[fi] - Y0[fi]) - (z2 - z0) * (Y1[fi] - Y0[fi])) / a2
    B = ((z2 - z0) * (X1[fi] - X0[fi]) - (z1 - z0) * (X2[fi] - X0[fi])) / a2
    Z = z0[:, None, None] + A[:, None, None] * (GX[None] - X0[fi][:, None, None]) + B[:, None, None] * (GY[None] - Y0[fi][:, None, None])
    rows_e = np.minimum(imax[fi], py0 + TH - 1) - np.maximum(imin[fi], py0) + 1
    # sub-block rect of each entry's bbox in tile
    bx0 = (np.maximum(jmin[fi], px0) - px0) // SB; bx1 = (np.minimum(jmax[fi], px0 + TW - 1) - px0) // SB
    by0 = (np.maximum(imin[fi], py0) - py0) // SB; by1 = (np.minimum(imax[fi], py0 + TH - 1) - py0) // SB
    zn = znear[e] * (1 + 2.0**-8)
    orders = {'random': rng.permutation(len(e)), 'morton': np.argsort(mort[e], kind='stable'), 'front2back': np.argsort(-znear[e]),
              'blockdepth': np.lexsort((mort[e], -blk_depth[blk_of[e]]))}
    for name, od in orders.items():
        zb = np.zeros((TH, TW))
        hz = np.zeros((TH // SB, TW // SB))
        culled_rows = 0
        for k, i in enumerate(od):
            if k % REFRESH == 0 and k > 0:
                hz = zb.reshape(TH // SB, SB, TW // SB, SB).min(axis=(1, 3))
            if k >= REFRESH and hz[by0[i]:by1[i] + 1, bx0[i]:bx1[i] + 1].min() > zn[i]:
                culled_rows += rows_e[i]
                continue
            zb = np.where(cov[i] & (Z[i] > zb), Z[i], zb)
        tot[name][0] += culled_rows; tot[name][1] += rows_e.sum()
for k, (a, b) in tot.items():
    print(f"scale {scale} SB {SB} refresh {REFRESH}: {k:12s} culled {a/b:.3f} of {b} row-items")
