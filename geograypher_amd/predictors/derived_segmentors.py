"""Segmentors that feed the aggregation loop (reference: geograypher/predictors/derived_segmentors.py:32-51)."""
from pathlib import Path

import numpy as np

from geograypher_amd.constants import PATH_TYPE
from geograypher_amd.predictors.segmentor import Segmentor


def _nearest_resize(image: np.ndarray, out_hw) -> np.ndarray:
    """Nearest-neighbour resize with pixel-centre sampling (skimage.transform.resize(order=0) convention)."""
    h, w = image.shape[:2]
    oh, ow = out_hw
    rows = np.clip(np.floor((np.arange(oh) + 0.5) * (h / oh)).astype(np.int64), 0, h - 1)
    cols = np.clip(np.floor((np.arange(ow) + 0.5) * (w / ow)).astype(np.int64), 0, w - 1)
    return image[rows][:, cols]


def _float_rescaled(inds: np.ndarray) -> np.ndarray:
    """What `skimage.transform.resize` returns for an unsigned-integer image when `preserve_range` is not given -- the
    reference's call, derived_segmentors.py:44-50: float64 `index * (1 / dtype_max)` in [0, 1] (skimage.util.dtype)."""
    if inds.dtype.kind != "u":
        raise NotImplementedError(f"reference_float_rescale is defined for unsigned integer index images, got {inds.dtype}")
    return np.multiply(inds, 1.0 / np.iinfo(inds.dtype).max, dtype=np.float64)


def _float_rescaled_indices(inds: np.ndarray) -> np.ndarray:
    """The class every pixel of `_float_rescaled(inds)` selects in `inds_to_one_hot`, as a uint8 index image: only 0.0 and
    1.0 equal a class index, i.e. index 0 stays class 0, the dtype's maximum (255) becomes class 1, and every other index
    matches nothing (255 here: an all-False one-hot row that still counts as an observation)."""
    if inds.dtype.kind != "u":
        raise NotImplementedError(f"reference_float_rescale is defined for unsigned integer index images, got {inds.dtype}")
    out = np.full(inds.shape, 255, dtype=np.uint8)
    out[inds == 0] = 0
    out[inds == np.iinfo(inds.dtype).max] = 1
    return out


class LookUpSegmentor(Segmentor):
    """Reads `<lookup_folder>/<path of the image relative to base_folder>.png` as a class-index image.

    At `image_scale != 1` the index image is resized with nearest-neighbour sampling of the INDICES (pixel-centre
    convention; equal to scikit-image >= 0.19 / `scipy.ndimage.zoom(order=0, grid_mode=True)`, tie scales included:
    tests/golden/make_golden_resize.py).  The reference does NOT get that: its `resize(image, ..., order=0)` without
    `preserve_range` returns floats in [0, 1], so its one-hot image keeps class 0, turns index 255 into class 1 and drops
    every other class (derived_segmentors.py:44-50).  `reference_float_rescale=True` reproduces that bit for bit.
    """

    thread_safe_lookup = True  # stateless file look-ups: the aggregation input pipeline may decode several at once

    def __init__(self, base_folder, lookup_folder, num_classes=10, reference_float_rescale: bool = False, decoded_cache=None):
        self.base_folder = Path(base_folder)
        self.lookup_folder = lookup_folder
        self.num_classes = num_classes
        self.reference_float_rescale = reference_float_rescale
        # None: every look-up decodes its PNG (the reference's behaviour).  True / a folder: the decoded -- and, at
        # image_scale != 1, resized -- index image is kept as an uncompressed .npy keyed by (path, mtime, size, scale) and later
        # passes memory-map it (utils/decoded_cache.py)
        self.decoded_cache = decoded_cache

    def segment_image_indices(self, image: np.ndarray, filename: PATH_TYPE, image_scale: float):
        from geograypher_amd.utils.decoded_cache import cached_decode

        relative_path = Path(filename).relative_to(self.base_folder)
        lookup_path = Path(self.lookup_folder, relative_path).with_suffix(".png")

        def decode():
            from PIL import Image

            with Image.open(lookup_path) as im:
                inds = np.asarray(im)
            if image_scale != 1:
                inds = _nearest_resize(inds, (int(inds.shape[0] * image_scale), int(inds.shape[1] * image_scale)))
                if self.reference_float_rescale:
                    inds = _float_rescaled_indices(inds)
            return inds

        tag = f"label-indices|scale={float(image_scale):.8f}|float_rescale={int(bool(self.reference_float_rescale))}"
        return cached_decode(lookup_path, self.decoded_cache, decode, tag)

    def segment_image(self, image: np.ndarray, filename: PATH_TYPE, image_scale: float):
        if self.reference_float_rescale and image_scale != 1:
            from PIL import Image

            relative_path = Path(filename).relative_to(self.base_folder)
            with Image.open(Path(self.lookup_folder, relative_path).with_suffix(".png")) as im:
                inds = np.asarray(im)
            inds = _nearest_resize(inds, (int(inds.shape[0] * image_scale), int(inds.shape[1] * image_scale)))
            return self.inds_to_one_hot(_float_rescaled(inds), num_classes=self.num_classes)
        inds = self.segment_image_indices(image, filename=filename, image_scale=image_scale)
        return self.inds_to_one_hot(inds, num_classes=self.num_classes)


class ArrayLabelSegmentor(Segmentor):
    """In-memory class-index images keyed by view order or filename: the synthetic-data twin of LookUpSegmentor
    used by tests and bench (no PNG decode, no file system)."""

    def __init__(self, label_images, num_classes: int, filenames=None, reference_float_rescale: bool = False):
        self.label_images = label_images
        self.num_classes = num_classes
        self.reference_float_rescale = reference_float_rescale
        self._by_name = None if filenames is None else {str(f): i for i, f in enumerate(filenames)}
        self._cursor = 0
        # keyed by filename: stateless, several look-ups may run at once; keyed by call order: strictly sequential
        self.thread_safe_lookup = self._by_name is not None

    def _lookup(self, filename):
        if self._by_name is not None and filename is not None and str(filename) in self._by_name:
            return self.label_images[self._by_name[str(filename)]]
        raise KeyError(f"no label image registered for {filename}")

    def segment_image_indices(self, image, filename=None, image_scale: float = 1):
        inds = np.asarray(self._lookup(filename))
        if image_scale != 1:
            inds = _nearest_resize(inds, (int(inds.shape[0] * image_scale), int(inds.shape[1] * image_scale)))
            if self.reference_float_rescale:
                inds = _float_rescaled_indices(inds)
        return inds

    def segment_image(self, image, filename=None, image_scale: float = 1):
        if self.reference_float_rescale and image_scale != 1:
            inds = np.asarray(self._lookup(filename))
            inds = _nearest_resize(inds, (int(inds.shape[0] * image_scale), int(inds.shape[1] * image_scale)))
            return self.inds_to_one_hot(_float_rescaled(inds), self.num_classes)
        return self.inds_to_one_hot(
            self.segment_image_indices(image, filename=filename, image_scale=image_scale), self.num_classes
        )
