from geograypher_amd.cameras.cameras import (
    PhotogrammetryCamera,
    PhotogrammetryCameraSet,
    vtk_like_near_plane,
    vtk_like_near_planes,
)
from geograypher_amd.cameras.segmentor import SegmentorPhotogrammetryCameraSet
from geograypher_amd.cameras.derived_cameras import MetashapeCameraSet

__all__ = [
    "PhotogrammetryCamera",
    "PhotogrammetryCameraSet",
    "SegmentorPhotogrammetryCameraSet",
    "MetashapeCameraSet",
    "vtk_like_near_plane",
]
