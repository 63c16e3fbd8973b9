"""oracle/oracle_warp.py -- TEST INFRASTRUCTURE ONLY.

numpy/scipy restatement of the reference's distortion-warp stage:

    flexible_inputs_warp_reference   geograypher/utils/image.py:72-126 -- rescale to [0,1], skimage.transform.warp with a
                                     coordinate map (= scipy.ndimage.map_coordinates, the pinned scikit-image 0.21.0 uses
                                     mode "grid-constant"), clip, rescale back, truncate to the input dtype
    warp_exact                       the same resampling WITHOUT the float round trip (what the product returns by
                                     default: integers are gathered as integers)

PARITY STATUS: pinned.  tests/golden/reference_warp.npz holds outputs of the REAL flexible_inputs_warp (run with
scikit-image 0.18.3 by tests/golden/make_golden_warp.py); tests/test_warp.py checks this restatement against them bit
for bit wherever the sample position lies inside the input (the two scikit-image versions differ only within half a
pixel outside it).
"""
import numpy as np
from scipy import ndimage as ndi


def _map_coordinates(image2d, inverse_map, order, cval):
    return ndi.map_coordinates(image2d, inverse_map, prefilter=order > 1, mode="grid-constant", order=order, cval=cval)


def flexible_inputs_warp_reference(input_image, inverse_map, interpolation_order=None, fill_value=0.0):
    """utils/image.py:72-126, line for line."""
    input_image = np.atleast_3d(input_image)
    input_min = min(np.min(input_image), fill_value)
    input_max = max(np.max(input_image), fill_value)
    min_max_range = input_max - input_min
    if min_max_range == 0:
        return np.full_like(np.squeeze(input_image), fill_value=fill_value)
    initial_dtype = input_image.dtype
    input_image = (input_image.astype(float) - input_min) / min_max_range
    rescaled_fill_value = (float(fill_value) - input_min) / min_max_range
    output_image = np.zeros(inverse_map.shape[1:] + (input_image.shape[2],))
    for channel in range(input_image.shape[2]):
        img = input_image[:, :, channel]
        warped = _map_coordinates(img, inverse_map, interpolation_order, rescaled_fill_value)
        # skimage _clip_warp_output(clip=True): clip to the input range, widened to keep cval
        min_val, max_val = img.min(), img.max()
        if not (min_val <= rescaled_fill_value <= max_val):
            min_val, max_val = min(min_val, rescaled_fill_value), max(max_val, rescaled_fill_value)
        output_image[:, :, channel] = np.clip(warped, min_val, max_val)
    output_image = ((output_image * min_max_range) + input_min).astype(initial_dtype)
    return np.squeeze(output_image)


def warp_exact(input_image, inverse_map, interpolation_order=0, fill_value=0.0):
    """Resample without the rescale round trip; result cast to the input dtype by truncation."""
    img3 = np.atleast_3d(input_image)
    if max(np.max(img3), fill_value) - min(np.min(img3), fill_value) == 0:
        return np.full_like(np.squeeze(img3), fill_value=fill_value)
    out = np.zeros(inverse_map.shape[1:] + (img3.shape[2],))
    for ch in range(img3.shape[2]):
        out[:, :, ch] = _map_coordinates(img3[:, :, ch].astype(float), inverse_map, interpolation_order, float(fill_value))
    return np.squeeze(out.astype(img3.dtype))


def inside_mask(inverse_map, in_shape):
    """Output pixels whose sample position lies inside [0, n-1] of the input (where scipy's "constant" and
    "grid-constant" boundary modes agree)."""
    r, c = inverse_map
    return (r >= 0) & (r <= in_shape[0] - 1) & (c >= 0) & (c <= in_shape[1] - 1)


# ---- inverse of the lens model (checker of gr_invert_distortion_f64) -----------------------------------------------------
def metashape_forward(params, xpix, ypix):
    """derived_cameras.py:163-208 restated: ideal pinhole pixel -> distorted image pixel.
    params: dict with f, cx, cy, image_width, image_height and any of k1..k4, p1, p2, b1, b2."""
    f, W, H = params["f"], params["image_width"], params["image_height"]
    x = (xpix - W / 2.0) / f
    y = (ypix - H / 2.0) / f
    g = params.get
    r2 = x * x + y * y
    radial = 1 + g("k1", 0) * r2 + g("k2", 0) * r2**2 + g("k3", 0) * r2**3 + g("k4", 0) * r2**4
    xd = x * radial + (g("p1", 0) * (r2 + 2 * x * x) + 2 * g("p2", 0) * x * y)
    yd = y * radial + (g("p2", 0) * (r2 + 2 * y * y) + 2 * g("p1", 0) * x * y)
    return W / 2.0 + params["cx"] + xd * f + xd * g("b1", 0) + yd * g("b2", 0), H / 2.0 + params["cy"] + yd * f


def forward_map_position(params, rows, cols, scale):
    """Where the (possibly fractional) pixel (rows, cols) of the IDEAL image of scale `scale` lands in the warped image of
    the same scale -- the construction of cameras.py:1012-1043: at scale 1 the model is evaluated at the pixel index
    itself, otherwise at the original-resolution position of the scaled pixel's centre, (index + 0.5) / scale."""
    if np.isclose(scale, 1.0):
        u, v = metashape_forward(params, cols, rows)
        return v, u
    u, v = metashape_forward(params, (cols + 0.5) / scale, (rows + 0.5) / scale)
    return v * scale, u * scale


def newton_inverse_map(params, h, w, scale=1.0, iters=12, fill=-1.0, tol=1e-9):
    """(2, h, w) map: for every pixel (i, j) of the warped image, the fractional pixel (row, col) of the ideal image that
    the lens model sends there; `fill` where that position lies outside the ideal image or the iteration does not
    converge.  Newton's method with a FINITE-DIFFERENCE Jacobian (the device kernel uses the analytic one)."""
    ti, tj = np.meshgrid(np.arange(h, dtype=np.float64), np.arange(w, dtype=np.float64), indexing="ij")
    r, c = ti.copy(), tj.copy()
    eps = 1e-4
    for _ in range(iters):
        fr, fc = forward_map_position(params, r, c, scale)
        fr_r, fc_r = forward_map_position(params, r + eps, c, scale)
        fr_c, fc_c = forward_map_position(params, r, c + eps, scale)
        a, b = (fr_r - fr) / eps, (fr_c - fr) / eps
        cc, d = (fc_r - fc) / eps, (fc_c - fc) / eps
        er, ec = fr - ti, fc - tj
        det = a * d - b * cc
        with np.errstate(divide="ignore", invalid="ignore"):
            dr, dc = (d * er - b * ec) / det, (a * ec - cc * er) / det
        dr, dc = np.nan_to_num(dr), np.nan_to_num(dc)
        r, c = r - np.clip(dr, -h, h), c - np.clip(dc, -w, w)
    fr, fc = forward_map_position(params, r, c, scale)
    ok = (np.abs(fr - ti) < tol * max(h, w)) & (np.abs(fc - tj) < tol * max(h, w))
    ok &= (r >= 0) & (r <= h - 1) & (c >= 0) & (c <= w - 1)
    return np.stack([np.where(ok, r, fill), np.where(ok, c, fill)], axis=0)
