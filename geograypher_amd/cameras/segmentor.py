"""`SegmentorPhotogrammetryCameraSet`: a camera set whose images are per-pixel class predictions.

Mirror of geograypher/cameras/segmentor.py:10-76 -- the label-image supplier of `aggregate_projected_images`.
In addition to the reference's (h, w, C) bool one-hot image (`get_image_by_index`), `get_label_index_image` hands
the aggregation fast path the (h, w) uint8 class-index image the one-hot was made from (12 MB instead of 48 MB per
4000x3000 view), which is what the HIP vote kernel consumes.
"""
import typing
from copy import copy

import numpy as np

from geograypher_amd.cameras.cameras import PhotogrammetryCameraSet
from geograypher_amd.predictors.segmentor import Segmentor


class SegmentorPhotogrammetryCameraSet(PhotogrammetryCameraSet):
    def __init__(
        self,
        base_camera_set: PhotogrammetryCameraSet,
        segmentor: Segmentor,
        dont_load_base_image: bool = True,
    ):
        """Wrap a camera set so that its images are the segmentor's output (reference: segmentor.py:11-31)."""
        self.base_camera_set = base_camera_set
        self.segmentor = segmentor
        self.dont_load_base_image = dont_load_base_image
        self.cameras = self.base_camera_set.cameras
        self._local_to_epsg_4978_transform = self.base_camera_set._local_to_epsg_4978_transform
        self._maps_ideal_to_warped = {}
        self._maps_warped_to_ideal = {}
        self.image_folder = getattr(self.base_camera_set, "image_folder", None)

    def _raw(self, index: int, image_scale: float):
        if self.dont_load_base_image:
            return None
        return self.base_camera_set.get_image_by_index(index, image_scale)

    def get_image_by_index(self, index: int, image_scale: float = 1) -> np.ndarray:
        """reference: segmentor.py:33-42"""
        image_filename = self.base_camera_set.get_image_filename(index, absolute=True)
        return self.segmentor.segment_image(
            self._raw(index, image_scale), filename=image_filename, image_scale=image_scale
        )

    def get_label_index_image(self, index: int, image_scale: float = 1):
        """(h, w) uint8 class indices for view `index`, or None when the segmentor cannot provide them."""
        fn = getattr(self.segmentor, "segment_image_indices", None)
        if fn is None:
            return None
        image_filename = self.base_camera_set.get_image_filename(index, absolute=True)
        return fn(self._raw(index, image_scale), filename=image_filename, image_scale=image_scale)

    def get_raw_image_by_index(self, index: int, image_scale: float = 1) -> np.ndarray:
        return self.base_camera_set.get_image_by_index(index=index, image_scale=image_scale)

    def get_subset_cameras(self, inds: typing.List[int]):
        """The set restricted to views `inds` (reference: segmentor.py:49-55, which deep-copies the whole wrapper -- segmentor
        included -- once per call).  Here: a shallow copy of the wrapper around the BASE set's own subset (cameras copied there,
        as the reference's base class does) with the SAME segmentor object -- look-up configuration or a model, neither of
        which a view subset should duplicate -- and the same distortion-map caches (keyed by the lens parameters: a map built
        for a subset serves the whole set)."""
        subset = copy(self)
        subset.base_camera_set = self.base_camera_set.get_subset_cameras(inds)
        subset.cameras = subset.base_camera_set.cameras
        return subset

    def n_image_channels(self) -> int:
        return self.segmentor.num_classes

    def get_subset_with_valid_segmentation(self) -> "SegmentorPhotogrammetryCameraSet":
        """The views whose segmentation can be produced (reference: segmentor.py:60-76: whatever the segmentor raises for a
        view -- typically a missing prediction file -- drops that view).  The class-index image is asked for where the segmentor
        has one (a quarter of the one-hot image's bytes, no per-class loop)."""
        probe = self.get_label_index_image if hasattr(self.segmentor, "segment_image_indices") else self.get_image_by_index

        def produces_a_segmentation(i: int) -> bool:
            try:
                probe(i)
            except Exception:
                return False
            return True

        return self.get_subset_cameras([i for i in range(len(self)) if produces_a_segmentation(i)])
