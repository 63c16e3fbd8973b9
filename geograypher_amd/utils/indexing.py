"""Post-aggregation helpers (reference: geograypher/utils/indexing.py:9-32) and the host-side inversion of a sampling
map (reference: geograypher/utils/indexing.py:87-150)."""
import numpy as np


def find_argmax_nonzero_value(array, keepdims: bool = False, axis: int = 1, backend=None):
    """Per-row argmax with NaN for rows that sum to zero or hold a non-finite value.

    Same contract as the reference (utils/indexing.py:9-32).  `array` may be a numpy array or a device tensor;
    the arithmetic runs in the HIP kernel `k_argmax_nonzero` through `backend` (a `HipRaster`); a numpy result is
    returned for numpy input, a tensor for tensor input.
    """
    import torch

    from geograypher_amd._hip import default_backend

    if axis not in (1, -1) or getattr(array, "ndim", 2) != 2:
        raise ValueError("the device implementation reduces the last axis of a 2-D (F, C) array")
    is_tensor = isinstance(array, torch.Tensor)
    if backend is None:  # the shared context of the device (never a new context, stats buffers and library handle per call)
        backend = default_backend(array.device.index if is_tensor and array.is_cuda else None)
    out = backend.argmax_nonzero(array)
    if keepdims:
        out = out[:, None]
    return out if is_tensor else out.cpu().numpy()


def inverse_map_interpolation(ijmap: np.ndarray, downsample: int = 1, fill: int = -1) -> np.ndarray:
    """Invert a (2, H, W) sampling map of the kind `skimage.transform.warp` takes: `ijmap[:, i, j]` is where destination
    pixel (i, j) samples the source; the result holds, at every integer source position, the destination position that
    samples it -- by piecewise-linear interpolation over the Delaunay triangulation of the (every `downsample`-th) known
    samples, `fill` outside their convex hull.  Public in the reference (utils/indexing.py:87-150) and kept for callers of
    it; HOST code (scipy / Qhull: minutes at survey resolution).  The projection path itself inverts the lens model
    densely on the device instead (`gr_invert_distortion_f64`); `make_distortion_map(reference_inverse=True)` selects
    this function for parity with the reference's down-sampled inverse.

    One triangulation serves both output channels (the reference triangulates twice; same triangles, same weights, same
    numbers)."""
    from scipy.interpolate import LinearNDInterpolator

    ijmap = np.asarray(ijmap)
    if ijmap.ndim != 3 or ijmap.shape[0] != 2:
        raise ValueError(f"sampling map must be (2, H, W), got {ijmap.shape}")
    H, W = ijmap.shape[1:]
    step = max(int(downsample), 1)
    dest = np.mgrid[0:H, 0:W]                                   # (2, H, W): the destination grid (i, j)
    known_at = ijmap[:, ::step, ::step].reshape(2, -1).T         # source positions we have a destination for
    known_dest = dest[:, ::step, ::step].reshape(2, -1).T.astype(np.float64)
    interp = LinearNDInterpolator(known_at, known_dest, fill_value=fill)
    return interp(dest.reshape(2, -1).T).T.reshape(2, H, W)
