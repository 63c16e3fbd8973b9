"""`Segmentor` base class: defines the FORMAT of the label image handed to `project_images`.

Mirror of geograypher/predictors/segmentor.py:6-69.  `inds_to_one_hot` gives the (h, w, C) bool image the reference
scatters onto faces; a label equal to `ignore_ind` (255) or >= num_classes yields an all-False row, which still
counts as an observation of the face (meshes.py:2064-2067).
"""
import typing

import numpy as np


class Segmentor:
    def __init__(self, num_classes=None):
        self.num_classes = num_classes

    def setup(self, **kwargs) -> None:
        pass

    def segment_image(self, image: np.ndarray, **kwargs):
        raise NotImplementedError("Abstract base class")

    def segment_image_batch(self, images: typing.List[np.ndarray], **kwargs):
        return [self.segment_image(image, **kwargs) for image in images]

    @staticmethod
    def inds_to_one_hot(
        inds_image: np.ndarray,
        num_classes: typing.Union[int, None] = None,
        ignore_ind: int = 255,
    ) -> np.ndarray:
        """(m, n) integer image -> (m, n, num_classes) bool one-hot (reference: segmentor.py:37-69)."""
        if num_classes is None:
            # The reference computes max(inds_image) + 1 here; its masking of `ignore_ind` is a no-op comparison
            # (segmentor.py:55), so an image containing 255 yields 256 channels.  Reproduced as is.
            num_classes = int(np.max(inds_image)) + 1
        inds_image = np.asarray(inds_image)
        classes = np.arange(num_classes).reshape((1,) * inds_image.ndim + (num_classes,))
        return inds_image[..., None] == classes
