"""Synthetic scenes for tests and benchmarks (the reference's counterparts: geograypher/utils/test_utils.py:10-156 and
geograypher/utils/example_data.py:30-112, restated in numpy without pyvista).

`make_simple_camera_set` / `downward_view` / `make_simple_mesh` reproduce the fixtures of the reference's own hot-path
tests in closed form; `config_*` build the BASELINE.json workloads with the seeds fixed in SURVEY.md section 8(d).
"""
from __future__ import annotations

from itertools import product
from pathlib import Path
from typing import List

import numpy as np

from geograypher_amd.cameras.cameras import PhotogrammetryCamera, PhotogrammetryCameraSet


# ---- the reference's test fixtures (utils/test_utils.py) ----------------------------------------------------------------
def downward_view(scene_width, focal, sensor_width):
    """Camera-to-world transform of a nadir camera that exactly frames a `scene_width` square (test_utils.py:42-66)."""
    return np.array(
        [
            [1, 0, 0, 0],
            [0, -1, 0, 0],
            [0, 0, -1, scene_width * focal / sensor_width],
            [0, 0, 0, 1],
        ],
        dtype=np.float64,
    )


def make_simple_camera_set(focal=100, sensor=200) -> PhotogrammetryCameraSet:
    """One nadir camera above the simple mesh, one mesh interval per pixel (test_utils.py:10-39)."""
    return PhotogrammetryCameraSet(
        cameras=[
            PhotogrammetryCamera(
                image_filename=None,
                cam_to_world_transform=downward_view(scene_width=4, focal=focal, sensor_width=sensor),
                f=focal,
                cx=0,
                cy=0,
                image_width=sensor,
                image_height=sensor,
                local_to_epsg_4978_transform=np.eye(4),
            )
        ],
        local_to_epsg_4978_transform=np.eye(4),
    )


def pixel_idx(vector: np.ndarray, i: int, j: int, stride: int, color, buffer: int = 0) -> None:
    """Colour the mesh vertices around image pixel (i, j) (test_utils.py:132-156)."""
    spread = range(-buffer, 2 + buffer)
    for di, dj in product(spread, spread):
        vector[(stride - i - di) * stride + (j + dj)] = color


def make_simple_mesh(pixels, color, background: int = 50, buffer: int = 0):
    """The 201 x 201-vertex plane of side 4 (80 000 triangles) with per-vertex colours (test_utils.py:69-129).

    pyvista's `Plane(i_resolution=200, j_resolution=200)` lays vertex k = row*201 + col at
    (-2 + 0.02 col, -2 + 0.02 row, 0); `triangulate()` splits each quad in two.
    Returns ((points, faces), point_colors)."""
    N = 201
    lin = -2.0 + 0.02 * np.arange(N)
    xx, yy = np.meshgrid(lin, lin)  # row-major: row index -> y
    points = np.stack([xx.ravel(), yy.ravel(), np.zeros(N * N)], axis=1)
    faces = grid_faces(N, N)
    point_colors = np.full((N * N, 3), fill_value=background, dtype=np.uint8)
    for pixel in pixels:
        pixel_idx(point_colors, *pixel, stride=N, color=color, buffer=buffer)
    return (points, faces), point_colors


# ---- generic builders ----------------------------------------------------------------------------------------------------
def grid_faces(n_rows: int, n_cols: int) -> np.ndarray:
    """Two triangles per grid quad, (2 (n_rows-1)(n_cols-1), 3) int64, faces ordered quad by quad, row by row."""
    r, c = np.meshgrid(np.arange(n_rows - 1), np.arange(n_cols - 1), indexing="ij")
    v00 = (r * n_cols + c).ravel()
    v01 = v00 + 1
    v10 = v00 + n_cols
    v11 = v10 + 1
    tri_a = np.stack([v00, v01, v11], axis=1)
    tri_b = np.stack([v00, v11, v10], axis=1)
    return np.stack([tri_a, tri_b], axis=1).reshape(-1, 3).astype(np.int64)


def heightfield_mesh(n_side: int, extent: float, height_fn, jitter: float = 0.0, seed: int = 0, noise: float = 0.0,
                     noise_seed: int = 0):
    """Regular n_side x n_side heightfield over [-extent/2, extent/2]^2. jitter: fraction of a cell (uniform xy)."""
    lin = np.linspace(-extent / 2, extent / 2, n_side)
    xx, yy = np.meshgrid(lin, lin)
    if jitter > 0:
        rng = np.random.default_rng(seed)
        cell = extent / (n_side - 1)
        xx = xx + rng.uniform(-jitter, jitter, xx.shape) * cell
        yy = yy + rng.uniform(-jitter, jitter, yy.shape) * cell
    zz = height_fn(xx, yy)
    if noise > 0:
        zz = zz + np.random.default_rng(noise_seed).normal(0.0, noise, zz.shape)
    points = np.stack([xx.ravel(), yy.ravel(), zz.ravel()], axis=1)
    return points, grid_faces(n_side, n_side)


def look_at(position, target, up_hint=(0.0, 1.0, 0.0)) -> np.ndarray:
    """cam_to_world for a camera at `position` looking at `target` (+X right, +Y down, +Z forward)."""
    position = np.asarray(position, dtype=np.float64)
    z = np.asarray(target, dtype=np.float64) - position
    z = z / np.linalg.norm(z)
    up = np.asarray(up_hint, dtype=np.float64)
    x = np.cross(z, up)  # right-handed with y down: x = z x up ... y = z x x
    if np.linalg.norm(x) < 1e-9:
        x = np.cross(z, np.array([1.0, 0.0, 0.0]))
    x = x / np.linalg.norm(x)
    y = np.cross(z, x)
    T = np.eye(4)
    T[:3, 0], T[:3, 1], T[:3, 2], T[:3, 3] = x, y, z, position
    return T


def nadir_pose(x, y, z, yaw_deg=0.0, tilt_x_deg=0.0, tilt_y_deg=0.0) -> np.ndarray:
    """Downward-looking camera (image x along world +x for yaw 0) with yaw about the vertical and small tilts."""
    base = np.array([[1, 0, 0], [0, -1, 0], [0, 0, -1]], dtype=np.float64)  # as downward_view
    cy_, sy_ = np.cos(np.deg2rad(yaw_deg)), np.sin(np.deg2rad(yaw_deg))
    Rz = np.array([[cy_, -sy_, 0], [sy_, cy_, 0], [0, 0, 1]])
    a, b = np.deg2rad(tilt_x_deg), np.deg2rad(tilt_y_deg)
    Rx = np.array([[1, 0, 0], [0, np.cos(a), -np.sin(a)], [0, np.sin(a), np.cos(a)]])
    Ry = np.array([[np.cos(b), 0, np.sin(b)], [0, 1, 0], [-np.sin(b), 0, np.cos(b)]])
    T = np.eye(4)
    T[:3, :3] = Rz @ base @ Rx @ Ry
    T[:3, 3] = (x, y, z)
    return T


def camera_set_from_poses(poses: List[np.ndarray], f, width, height, name_prefix="/synthetic/view") -> PhotogrammetryCameraSet:
    cams = [
        PhotogrammetryCamera(Path(f"{name_prefix}_{i:05d}.png"), T, f=f, cx=0.0, cy=0.0, image_width=width,
                             image_height=height, local_to_epsg_4978_transform=np.eye(4))
        for i, T in enumerate(poses)
    ]
    return PhotogrammetryCameraSet(cams, local_to_epsg_4978_transform=np.eye(4))


# ---- BASELINE.json configs ----------------------------------------------------------------------------
def _spectrum(seed: int, amps=(12.0, 6.0, 3.0, 1.5), wavelengths=(200.0, 100.0, 50.0, 25.0)):
    rng = np.random.default_rng(seed)
    phi = rng.uniform(0, 2 * np.pi, len(amps))
    psi = rng.uniform(0, 2 * np.pi, len(amps))
    ks = [2 * np.pi / wl for wl in wavelengths]

    def height(x, y):
        z = np.zeros_like(x, dtype=np.float64)
        for a, k, p, q in zip(amps, ks, phi, psi):
            z = z + a * np.sin(k * x + p) * np.cos(k * y + q)
        return z

    return height


def config1_scene():
    """C1: 71 x 71 jittered plane (9 800 faces) + 8 pinhole cameras 640 x 480, f = 500 px."""
    points, faces = heightfield_mesh(
        71, 100.0, lambda x, y: 0.5 * np.sin(x / 7.0) + 0.5 * np.cos(y / 5.0), jitter=0.3, seed=0
    )
    poses = [nadir_pose(0.0, 0.0, 40.0)]
    for k in range(7):
        ang = 2 * np.pi * k / 7
        poses.append(look_at((30 * np.cos(ang), 30 * np.sin(ang), 40.0), (0.0, 0.0, 0.0), up_hint=(0, 0, 1)))
    return (points, faces), camera_set_from_poses(poses, f=500.0, width=640, height=480)


def terrain_mesh(n_side: int = 776, extent: float = 400.0):
    """C2/C3/C4 mesh: 776 x 776 heightfield -> 1 201 250 faces (V = 602 176); C5: n_side=1582, extent=800."""
    return heightfield_mesh(n_side, extent, _spectrum(1), noise=0.05, noise_seed=2)


def survey_cameras(nx: int, ny: int, dx: float, dy: float, agl: float = 120.0, f: float = 3000.0, width: int = 4000,
                   height: int = 3000, tilt_sigma_deg: float = 5.0, seed: int = 3, altitudes=None) -> PhotogrammetryCameraSet:
    """Lawn-mower grid of nadir cameras (yaw alternates 0/180 per line, Gaussian tilt) above the terrain."""
    rng = np.random.default_rng(seed)
    height_fn = _spectrum(1)
    poses = []
    alts = [agl] if altitudes is None else list(altitudes)
    for alt in alts:
        for iy in range(ny):
            for ix in range(nx):
                x = (ix - (nx - 1) / 2) * dx
                y = (iy - (ny - 1) / 2) * dy
                ground = float(height_fn(np.array(x), np.array(y)))
                tilt = rng.normal(0.0, tilt_sigma_deg, 2)
                poses.append(nadir_pose(x, y, ground + alt, yaw_deg=180.0 * (iy % 2), tilt_x_deg=tilt[0], tilt_y_deg=tilt[1]))
    return camera_set_from_poses(poses, f=f, width=width, height=height)


def config2_cameras(n_views: int = 50, **kw) -> PhotogrammetryCameraSet:
    """C2: 50 cameras 4000 x 3000, f = 3000 px, 120 m AGL, 10 x 5 grid, 40 m x 60 m spacing, tilt N(0, 5 deg)."""
    cams = survey_cameras(10, 5, 40.0, 60.0, **kw)
    return cams[:n_views] if n_views < len(cams) else cams


def config3_cameras(n_views: int = 500, **kw) -> PhotogrammetryCameraSet:
    """C3: 500 cameras, 25 x 20 grid, 16 m x 20 m spacing, same intrinsics."""
    cams = survey_cameras(25, 20, 16.0, 20.0, **kw)
    return cams[:n_views] if n_views < len(cams) else cams


def hash32(x: np.ndarray) -> np.ndarray:
    """Counter-based 32-bit mixer (lowbias32) used to derive synthetic labels reproducibly on any device."""
    x = np.asarray(x, dtype=np.uint64) & np.uint64(0xFFFFFFFF)
    x ^= x >> np.uint64(16)
    x = (x * np.uint64(0x7FEB352D)) & np.uint64(0xFFFFFFFF)
    x ^= x >> np.uint64(15)
    x = (x * np.uint64(0x846CA68B)) & np.uint64(0xFFFFFFFF)
    x ^= x >> np.uint64(16)
    return x.astype(np.uint32)


def synthetic_labels(ids: np.ndarray, view: int, n_classes: int = 4, seed_face: int = 4, seed_pix: int = 5) -> np.ndarray:
    """C3 label image for one view from its face-id image: class = hash(face ^ seed) mod C, 10 % per-pixel flips,
    1 % ignore (255).  Pure function of (face id, view, pixel) so host and device generators agree."""
    ids = np.asarray(ids)
    flat = ids.reshape(-1).astype(np.int64)
    cls = (hash32((flat & 0xFFFFFFFF) ^ seed_face) % np.uint32(n_classes)).astype(np.uint8)
    pix = np.arange(flat.size, dtype=np.uint64)
    r = hash32(pix * np.uint64(2654435761) + np.uint64(view) * np.uint64(40503) + np.uint64(seed_pix))
    u = r % np.uint32(1000)
    flip = u < 100
    cls = np.where(flip, ((r >> np.uint32(10)) % np.uint32(n_classes)).astype(np.uint8), cls)
    cls = np.where(u >= 990, np.uint8(255), cls)
    return cls.reshape(ids.shape).astype(np.uint8)


# ---- hostile workload: terrain + trees (restates geograypher/utils/example_data.py:30-112 without pyvista) ---------------
def _cylinder(cx, cy, z0, radius, height, resolution=10):
    """Closed cylinder like `pv.Cylinder(..., resolution=10).triangulate()` of example_data.py:58-66: two n-gon caps
    (triangle fans) and 2n side triangles.  Returns (points (2n,3), faces (4n-4,3))."""
    ang = 2 * np.pi * np.arange(resolution) / resolution
    ring = np.stack([cx + radius * np.cos(ang), cy + radius * np.sin(ang)], axis=1)
    pts = np.concatenate([np.c_[ring, np.full(resolution, z0)], np.c_[ring, np.full(resolution, z0 + height)]], axis=0)
    n = resolution
    k = np.arange(n)
    k1 = (k + 1) % n
    side = np.concatenate([np.stack([k, k1, n + k1], axis=1), np.stack([k, n + k1, n + k], axis=1)], axis=0)
    fan = np.arange(1, n - 1)
    bottom = np.stack([np.zeros(n - 2, dtype=np.int64), fan + 1, fan], axis=1)
    top = np.stack([np.full(n - 2, n), n + fan, n + fan + 1], axis=1)
    return pts, np.concatenate([side, bottom, top], axis=0).astype(np.int64)


def _cone(cx, cy, z0, radius, height, resolution=12):
    """Cone with its apex up, like the `pv.Cone(..., resolution=12).triangulate()` of example_data.py:70-80: n side
    triangles and an n-gon base (triangle fan).  Returns (points (n+1,3), faces (2n-2,3))."""
    n = resolution
    ang = 2 * np.pi * np.arange(n) / n
    ring = np.stack([cx + radius * np.cos(ang), cy + radius * np.sin(ang), np.full(n, z0)], axis=1)
    pts = np.concatenate([ring, [[cx, cy, z0 + height]]], axis=0)
    k = np.arange(n)
    side = np.stack([k, (k + 1) % n, np.full(n, n)], axis=1)
    fan = np.arange(1, n - 1)
    base = np.stack([np.zeros(n - 2, dtype=np.int64), fan + 1, fan], axis=1)
    return pts, np.concatenate([side, base], axis=0).astype(np.int64)


def forest_scene(n_trees: int = 20000, n_side: int = 776, extent: float = 400.0, seed: int = 7):
    """The C2 terrain with `n_trees` trees on it: a trunk (10-gon cylinder, 36 triangles) carrying a canopy (12-gon
    cone, 22 triangles), tree heights 8-25 m, canopy radii 1.5-4 m, uniformly scattered (seeded).  What the heightfield
    alone does not have: depth complexity > 1, triangles from sub-pixel slivers to tile-sized canopy sides, and tiles whose
    entry lists differ by two orders of magnitude once the cameras are tilted.  Returns (points, faces)."""
    rng = np.random.default_rng(seed)
    points, faces = terrain_mesh(n_side, extent)
    height_fn = _spectrum(1)
    xy = rng.uniform(-0.48 * extent, 0.48 * extent, size=(n_trees, 2))
    tall = rng.uniform(8.0, 25.0, n_trees)
    rad = rng.uniform(1.5, 4.0, n_trees)
    ground = height_fn(xy[:, 0], xy[:, 1])
    cp, cf = _cylinder(0.0, 0.0, 0.0, 1.0, 1.0)
    kp, kf = _cone(0.0, 0.0, 0.0, 1.0, 1.0)
    # trunks: radius 0.3 m, 40 % of the tree height, sunk 0.5 m into the ground; canopies on top of them
    trunk_pts = cp[None, :, :] * np.stack([np.full(n_trees, 0.3), np.full(n_trees, 0.3), 0.4 * tall + 0.5], axis=1)[:, None, :]
    trunk_pts = trunk_pts + np.stack([xy[:, 0], xy[:, 1], ground - 0.5], axis=1)[:, None, :]
    can_pts = kp[None, :, :] * np.stack([rad, rad, 0.6 * tall], axis=1)[:, None, :]
    can_pts = can_pts + np.stack([xy[:, 0], xy[:, 1], ground + 0.4 * tall], axis=1)[:, None, :]
    v0 = points.shape[0]
    nt, nc = cp.shape[0], kp.shape[0]
    trunk_faces = cf[None, :, :] + (v0 + nt * np.arange(n_trees))[:, None, None]
    v1 = v0 + nt * n_trees
    can_faces = kf[None, :, :] + (v1 + nc * np.arange(n_trees))[:, None, None]
    all_points = np.concatenate([points, trunk_pts.reshape(-1, 3), can_pts.reshape(-1, 3)], axis=0)
    all_faces = np.concatenate([faces, trunk_faces.reshape(-1, 3), can_faces.reshape(-1, 3)], axis=0).astype(np.int64)
    return all_points, all_faces


# ---- a realistic third workload: an irregular TIN (what BASELINE config 2 calls "Example-data Metashape mesh") ------------------
def tin_mesh(seed: int = 11, extent: float = 400.0, n_points: int = 600_000, sigma: float = 1.0, overhang_share: float = 0.05,
             cell: float = 2.0):
    """A photogrammetric-style triangulated irregular network over the C2 terrain spectrum: about `2 n_points` faces
    (1.2 M by default), vertex density varying log-normally (sigma = `sigma` of the log) over a smooth random field -- a
    dense-cloud mesh is fine where the scene has texture and coarse where it has none --, Delaunay connectivity (slivers,
    valences from 3 to 12, triangle areas over two orders of magnitude), and leaning bumps on `overhang_share` of the area
    whose downhill side folds under itself (depth complexity 2-3 from above: building eaves, canopy edges).  Blue-noise-like
    points: every `cell` x `cell` m cell holds an n x n jittered sub-grid, n from the density field (a white-noise point set
    would make every second Delaunay triangle a sliver; Metashape's are not).  Face order: Delaunay's (spatially incoherent,
    like a file written by a mesher).  Deterministic in `seed`.  Returns (points (V,3) float64, faces (F,3) int64)."""
    from scipy.spatial import Delaunay

    rng = np.random.default_rng(seed)
    n_cells = int(round(extent / cell))
    cx = (np.arange(n_cells) + 0.5) * cell - extent / 2
    gx, gy = np.meshgrid(cx, cx)
    # smooth log-density field: four random plane waves of 60-250 m wavelength, normalised to unit variance
    g = np.zeros_like(gx)
    for _ in range(4):
        lam, ang, ph = rng.uniform(60.0, 250.0), rng.uniform(0, 2 * np.pi), rng.uniform(0, 2 * np.pi)
        g += np.sin(2 * np.pi / lam * (gx * np.cos(ang) + gy * np.sin(ang)) + ph)
    g = (g - g.mean()) / g.std()
    dens = np.exp(sigma * g)
    dens *= n_points / dens.sum()                          # expected points per cell
    n_sub = np.clip(np.rint(np.sqrt(dens)), 1, 12).astype(np.int64)
    # rescale once so that the total lands near n_points despite the rounding
    n_sub = np.clip(np.rint(np.sqrt(dens * n_points / float((n_sub ** 2).sum()))), 1, 12).astype(np.int64)
    pts = []
    for n in np.unique(n_sub):
        ci, cj = np.nonzero(n_sub == n)
        k = np.arange(n)
        ox, oy = np.meshgrid((k + 0.5) / n, (k + 0.5) / n)
        jit = rng.uniform(-0.35 / n, 0.35 / n, size=(ci.size, n * n, 2))
        x = cx[cj][:, None] - cell / 2 + (ox.ravel()[None, :] + jit[..., 0]) * cell
        y = cx[ci][:, None] - cell / 2 + (oy.ravel()[None, :] + jit[..., 1]) * cell
        pts.append(np.stack([x.ravel(), y.ravel()], axis=1))
    uv = np.concatenate(pts, axis=0)
    uv = uv[rng.permutation(uv.shape[0])]                  # vertex order: a mesher's, not the generator's
    faces = Delaunay(uv).simplices.astype(np.int64)
    z = _spectrum(1)(uv[:, 0], uv[:, 1]) + rng.normal(0.0, 0.03, uv.shape[0])
    xyz = np.stack([uv[:, 0], uv[:, 1], z], axis=1)
    # leaning bumps: height h exp(-r^2 / 2 s^2), the whole bump sheared sideways by 1.6 x its height -- where the shear's
    # gradient along the lean direction is below -1 the surface folds under itself
    area, covered = extent * extent, 0.0
    while covered < overhang_share * area:
        c = rng.uniform(-0.45 * extent, 0.45 * extent, 2)
        s_b, ang = rng.uniform(2.0, 6.0), rng.uniform(0, 2 * np.pi)
        h_b = s_b * rng.uniform(1.0, 2.5)                  # steep enough to fold: the shear's gradient reaches -0.97 h / s
        d = uv - c
        r2 = (d * d).sum(axis=1)
        near = r2 < (3.0 * s_b) ** 2
        b = h_b * np.exp(-r2[near] / (2 * s_b * s_b))
        xyz[near, 2] += b
        xyz[near, 0] += 1.6 * b * np.cos(ang)
        xyz[near, 1] += 1.6 * b * np.sin(ang)
        covered += np.pi * (2.0 * s_b) ** 2
    return xyz, faces


def depth_complexity(points, faces, rec, h, w):
    """Mean number of mesh layers under a covered pixel of one view (camera record `rec`): the summed screen
    area of the faces that lie wholly inside the image over the number of pixels they can cover (float64, no clipping: an
    estimate for reporting, not part of any parity claim)."""
    R = np.asarray(rec[:9], dtype=np.float64).reshape(3, 3)
    t = np.asarray(rec[9:12], dtype=np.float64)
    q = (np.asarray(points, dtype=np.float64) - t) @ R
    ok = q[:, 2] > max(float(rec[15]), 1e-9)
    with np.errstate(divide="ignore", invalid="ignore"):
        sx = rec[13] + rec[12] * q[:, 0] / q[:, 2]
        sy = rec[14] + rec[12] * q[:, 1] / q[:, 2]
    inside = ok & (sx >= 0) & (sx <= w) & (sy >= 0) & (sy <= h)
    f = np.asarray(faces)
    keep = inside[f].all(axis=1)
    a, b, c = f[keep, 0], f[keep, 1], f[keep, 2]
    area2 = np.abs((sx[b] - sx[a]) * (sy[c] - sy[a]) - (sx[c] - sx[a]) * (sy[b] - sy[a]))
    g = 64   # coverage at 64-pixel granularity (every covered cell of that size holds a vertex) is enough for a ratio
    cover = np.zeros((h // g + 1, w // g + 1), dtype=bool)
    for k in (a, b, c):
        cover[np.clip((sy[k] // g).astype(np.int64), 0, h // g), np.clip((sx[k] // g).astype(np.int64), 0, w // g)] = True
    covered_px = min(cover.sum() * float(g * g), float(h) * w)
    return float(0.5 * area2.sum() / max(covered_px, 1.0))


def oblique_cameras(n_views: int = 20, tilt_range=(30.0, 45.0), agl: float = 120.0, f: float = 3000.0, width: int = 4000,
                    height: int = 3000, seed: int = 8, extent: float = 400.0) -> PhotogrammetryCameraSet:
    """Cameras over the central part of the terrain, tilted 30-45 degrees off nadir with random headings."""
    rng = np.random.default_rng(seed)
    height_fn = _spectrum(1)
    poses = []
    for _ in range(n_views):
        x, y = rng.uniform(-0.25 * extent, 0.25 * extent, 2)
        ground = float(height_fn(np.array(x), np.array(y)))
        poses.append(nadir_pose(x, y, ground + agl, yaw_deg=rng.uniform(0, 360), tilt_x_deg=rng.uniform(*tilt_range),
                                tilt_y_deg=rng.uniform(-5, 5)))
    return camera_set_from_poses(poses, f=f, width=width, height=height)


def config4_cameras(**kw) -> PhotogrammetryCameraSet:
    """C4: 2000 views = the C3 grid at four altitudes (90 / 110 / 130 / 150 m AGL); view i belongs to GPU i mod world."""
    return survey_cameras(25, 20, 16.0, 20.0, altitudes=(90.0, 110.0, 130.0, 150.0), **kw)


def config5_scene(n_views: int = 2000):
    """C5: 1582 x 1582 heightfield over 800 m (4 999 122 faces), cameras 6000 x 4000, f = 4500 px, 150 m AGL, 50 x 40 grid
    (15 m x 18 m spacing), tilt N(0, 5 deg) with seed 6."""
    points, faces = terrain_mesh(1582, 800.0)
    cams = survey_cameras(50, 40, 15.0, 18.0, agl=150.0, f=4500.0, width=6000, height=4000, seed=6)
    return (points, faces), (cams[:n_views] if n_views < len(cams) else cams)
