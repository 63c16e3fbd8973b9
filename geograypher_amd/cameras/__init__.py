from geograypher_amd.cameras.cameras import PhotogrammetryCamera, PhotogrammetryCameraSet, vtk_like_near_plane
from geograypher_amd.cameras.segmentor import SegmentorPhotogrammetryCameraSet

__all__ = [
    "PhotogrammetryCamera",
    "PhotogrammetryCameraSet",
    "SegmentorPhotogrammetryCameraSet",
    "vtk_like_near_plane",
]
