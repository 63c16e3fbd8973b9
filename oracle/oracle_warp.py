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
