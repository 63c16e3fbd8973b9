from geograypher_amd.meshes.meshes import LocalMesh, TexturedPhotogrammetryMesh
from geograypher_amd.meshes.derived_meshes import (
    TexturedPhotogrammetryMeshChunked,
    TexturedPhotogrammetryMeshIndexPredictions,
)

__all__ = [
    "TexturedPhotogrammetryMesh",
    "TexturedPhotogrammetryMeshChunked",
    "TexturedPhotogrammetryMeshIndexPredictions",
    "LocalMesh",
]
