"""Post-aggregation helpers (reference: geograypher/utils/indexing.py:9-32)."""
import numpy as np


def find_argmax_nonzero_value(array, keepdims: bool = False, axis: int = 1, backend=None):
    """Per-row argmax with NaN for rows that sum to zero or hold a non-finite value.

    Same contract as the reference (utils/indexing.py:9-32).  `array` may be a numpy array or a device tensor;
    the arithmetic runs in the HIP kernel `k_argmax_nonzero` through `backend` (a `HipRaster`); a numpy result is
    returned for numpy input, a tensor for tensor input.
    """
    import torch

    from geograypher_amd._hip import HipRaster

    if axis not in (1, -1) or getattr(array, "ndim", 2) != 2:
        raise ValueError("the device implementation reduces the last axis of a 2-D (F, C) array")
    is_tensor = isinstance(array, torch.Tensor)
    if backend is None:
        backend = HipRaster(array.device.index if is_tensor and array.is_cuda else None)
    out = backend.argmax_nonzero(array)
    if keepdims:
        out = out[:, None]
    return out if is_tensor else out.cpu().numpy()


def inverse_map_interpolation(ijmap: np.ndarray, downsample: int = 1, fill: int = -1) -> np.ndarray:
    """Invert a (2, H, W) sampling map (destination pixel -> source position) by scattered linear interpolation.

    Same construction as the reference (utils/indexing.py:87-150): the mapped positions of every `downsample`-th grid
    pixel are the scattered samples, the grid indices they came from are the values, and `scipy.interpolate.griddata`
    (Qhull Delaunay + barycentric interpolation) resamples them on the regular grid; `fill` outside the convex hull.
    This is one-time host work per distortion key (cameras.py:995-1062), cached by the camera set.
    """
    from scipy.interpolate import griddata

    H, W = ijmap.shape[1:]
    igrid, jgrid = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    grid_coords = np.stack([igrid.ravel(), jgrid.ravel()], axis=1)
    if downsample > 1:
        ds = slice(None, None, downsample)
        sample_y = np.stack([igrid[ds, ds].ravel(), jgrid[ds, ds].ravel()], axis=1)
        sample_x = np.stack([ijmap[0][ds, ds].ravel(), ijmap[1][ds, ds].ravel()], axis=1)
    else:
        sample_y = grid_coords.copy()
        sample_x = np.stack([ijmap[0].ravel(), ijmap[1].ravel()], axis=1)
    inv_i = griddata(sample_x, sample_y[:, 0], grid_coords, method="linear", fill_value=fill)
    inv_j = griddata(sample_x, sample_y[:, 1], grid_coords, method="linear", fill_value=fill)
    return np.stack([inv_i.reshape(H, W), inv_j.reshape(H, W)], axis=0)
