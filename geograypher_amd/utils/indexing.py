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
