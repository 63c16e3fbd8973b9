"""oracle/oracle_resize.py -- CPU restatement of the PHOTO resize of the aggregation path.  TEST INFRASTRUCTURE: only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this; the product never does.

What it restates: `PhotogrammetryCamera.get_image(image_scale != 1)` (geograypher/cameras/cameras.py:154-174):
`imread`, uint8 -> `/ 255.0`, then `skimage.transform.resize(image, (int(h * s), int(w * s)))` with scikit-image's defaults --
the algorithm lives in the third-party dependency scikit-image (pinned 0.21.0, poetry.lock; 0.18.3 is what this container
can run).  Published algorithm (skimage/transform/_warps.py `resize`, both versions):

  1. factors = n_in / n_out per axis (channel axis: 1);  sigma = max(0, (factors - 1) / 2)
  2. anti-aliasing: scipy.ndimage.gaussian_filter(image, sigma, mode="mirror") -- separable, axis 0 first, kernel radius
     int(4 sigma + 0.5), weights exp(-x^2 / (2 sigma^2)) normalised to sum 1, boundary d c b | a b c d | c b a
  3. order-1 sampling at the half-pixel centres  coord = (i + 0.5) * factor - 0.5  (0.18: skimage `warp` with an exactly
     metric affine map, mode "reflect" = the same mirror boundary; >= 0.19: scipy.ndimage.zoom(order=1, mode="mirror",
     grid_mode=True) -- the same positions, the same two-tap weights)
  4. clip to the input range (a no-op for order 1 up to rounding).

PINNED: tests/test_photo_resize.py checks both functions below against tests/golden/reference_photo_resize.npz -- outputs of
the REAL scikit-image 0.18.3 `resize` and of the >= 0.19 formulation through the real scipy (make_golden_photo_resize.py) --
to 1e-12.
"""
from __future__ import annotations

import numpy as np


def gaussian_weights(sigma: float, truncate: float = 4.0) -> np.ndarray:
    """scipy.ndimage._filters._gaussian_kernel1d (order 0): radius int(truncate * sigma + 0.5)."""
    radius = int(truncate * float(sigma) + 0.5)
    x = np.arange(-radius, radius + 1)
    phi = np.exp(-0.5 / (sigma * sigma) * x**2)
    return phi / phi.sum()


def mirror_index(i: np.ndarray, n: int) -> np.ndarray:
    """Index of the sample that position i (any integer) reads under the "mirror" boundary (d c b | a b c d | c b a)."""
    if n == 1:
        return np.zeros_like(i)
    period = 2 * (n - 1)
    i = np.mod(i, period)
    return np.where(i >= n, period - i, i)


def gaussian_axis(image: np.ndarray, sigma: float, axis: int) -> np.ndarray:
    """One axis of scipy.ndimage.gaussian_filter(mode="mirror"), in numpy: symmetric correlation, centre tap first and the
    pairs from the outside in (the summation order of scipy's NI_Correlate1D for symmetric kernels)."""
    if not sigma > 1e-15:
        return image
    w = gaussian_weights(sigma)
    r = (w.size - 1) // 2
    n = image.shape[axis]
    idx = np.arange(n)
    take = lambda off: np.take(image, mirror_index(idx + off, n), axis=axis)
    out = take(0) * w[r]
    for j in range(r, 0, -1):
        out = out + (take(-j) + take(j)) * w[r - j]
    return out


def resize_antialias(image: np.ndarray, out_hw) -> np.ndarray:
    """skimage.transform.resize(image, out_hw) for a float image (H, W) or (H, W, C): float64 result of shape
    (h, w[, C])."""
    img = np.asarray(image, dtype=np.float64)
    H, W = img.shape[:2]
    h, w = int(out_hw[0]), int(out_hw[1])
    fr, fc = H / h, W / w
    img = gaussian_axis(img, max(0.0, (fr - 1) / 2), 0)
    img = gaussian_axis(img, max(0.0, (fc - 1) / 2), 1)
    r = (np.arange(h) + 0.5) * fr - 0.5
    c = (np.arange(w) + 0.5) * fc - 0.5
    r0, c0 = np.floor(r), np.floor(c)
    dr, dc = r - r0, c - c0
    r0i, r1i = mirror_index(r0.astype(np.int64), H), mirror_index(np.ceil(r).astype(np.int64), H)
    c0i, c1i = mirror_index(c0.astype(np.int64), W), mirror_index(np.ceil(c).astype(np.int64), W)
    shape_r = (h, 1) + (1,) * (img.ndim - 2)
    shape_c = (1, w) + (1,) * (img.ndim - 2)
    dr, dc = dr.reshape(shape_r), dc.reshape(shape_c)
    top = (1 - dc) * img[r0i][:, c0i] + dc * img[r0i][:, c1i]
    bottom = (1 - dc) * img[r1i][:, c0i] + dc * img[r1i][:, c1i]
    return (1 - dr) * top + dr * bottom


def get_image_scaled(raw: np.ndarray, image_scale: float) -> np.ndarray:
    """cameras.py:154-174 on an image as read from its file: uint8 -> / 255.0, then the resize."""
    img = raw / 255.0 if raw.dtype == np.uint8 else raw
    if image_scale == 1.0:
        return img
    return resize_antialias(img, (int(img.shape[0] * image_scale), int(img.shape[1] * image_scale)))
