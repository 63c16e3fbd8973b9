"""geograypher_amd -- MI355X-native implementation of geograypher's image<->mesh projection hot path.

Public surface (same names as the reference package):
    geograypher_amd.cameras   PhotogrammetryCamera, PhotogrammetryCameraSet, SegmentorPhotogrammetryCameraSet
    geograypher_amd.meshes    TexturedPhotogrammetryMesh  (pix2face, render_flat, project_images,
                              aggregate_projected_images / aggregate_viewpoints)
    geograypher_amd.predictors Segmentor, LookUpSegmentor
The arithmetic lives in csrc/libgeograster.so (hand-written HIP for gfx950) behind include/geograster.h.
"""
__version__ = "0.1.0"
