from geograypher_amd.meshes.meshes import LocalMesh, TexturedPhotogrammetryMesh

__all__ = ["TexturedPhotogrammetryMesh", "LocalMesh"]
