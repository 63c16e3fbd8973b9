"""Defaults that appear in the reference's signatures on the hot path.

Mirrors geograypher/constants.py:18 (CACHE_FOLDER), :27 (NULL_TEXTURE_INT_VALUE), :106-113 (EXAMPLE_INTRINSICS).
Only the constants the image<->mesh projection path touches are restated.
"""
from pathlib import Path
from typing import Union

PATH_TYPE = Union[str, Path]

# geograypher/constants.py:18 -- kept so pix2face(save_to_cache=..., cache_folder=...) keeps its signature.
CACHE_FOLDER = Path(Path.home(), ".cache", "geograypher")

# geograypher/constants.py:27
NULL_TEXTURE_INT_VALUE = 0

# geograypher/constants.py:106-113
EXAMPLE_INTRINSICS = {
    "f": 1000,
    "cx": 0,
    "cy": 0,
    "image_width": 800,
    "image_height": 600,
    "distortion_params": {},
}

# Name of the CRS the reference meshes live in (EPSG:4978); kept as a plain string because pyproj is not a
# dependency of the projection path (the CRS hand-over itself is out of scope).
EARTH_CENTERED_EARTH_FIXED_CRS = "EPSG:4978"

# geograypher/constants.py:16 (default output root of save_renders)
VIS_FOLDER = Path(Path(__file__).parent, "..", "vis").resolve()
