"""Derived mesh classes of the reference, on the GPU path.

Mirror of geograypher/meshes/derived_meshes.py for the two variants whose work is the projection path:

* `TexturedPhotogrammetryMeshChunked` (derived_meshes.py:23-317) exists in the reference because the VTK render cost
  grows with the mesh: it clusters the cameras (KMeans), crops a sub-mesh per cluster and renders chunk by chunk.
  Here every view is frustum-culled on the device per 256-face block and binned per tile, so the whole mesh is
  rendered at once and the class only keeps the reference's constructor/method signatures: results are those of the
  un-chunked class (a superset of what a 125 m-buffered chunk can see).
* `TexturedPhotogrammetryMeshIndexPredictions` (derived_meshes.py:414-550): aggregation of single-channel class-index
  images with many classes into scipy CSR arrays; the per-view projection, pair emission, radix sort and
  run-length count run on the device (`gr_project_index_pairs`, `gr_count_pairs`).
"""
import typing

import numpy as np

from geograypher_amd.cameras.cameras import PhotogrammetryCamera, PhotogrammetryCameraSet
from geograypher_amd.meshes.meshes import TexturedPhotogrammetryMesh, _torch, tqdm

CHUNKED_MESH_BUFFER_DIST_METERS = 125  # geograypher/constants.py:130


class TexturedPhotogrammetryMeshChunked(TexturedPhotogrammetryMesh):
    """Drop-in for the reference's chunked class; `n_clusters`, `buffer_dist_meters` and `vis_clusters` are accepted and
    ignored (the GPU path needs no chunking)."""

    def _say_unchunked(self, what, n_clusters, buffer_dist_meters, vis_clusters):
        """One log line per call: the chunking arguments are accepted for signature compatibility and have no effect."""
        self.logger.info(
            f"{what}: n_clusters={n_clusters}, buffer_dist_meters={buffer_dist_meters}, vis_clusters={vis_clusters} are "
            "ignored -- the GPU path culls and bins the whole mesh per view, no camera clustering or sub-mesh cropping; "
            "views are processed in CAMERA order (the reference goes cluster by cluster, derived_meshes.py:206, 281)"
        )

    def render_flat(self, cameras, batch_size: int = 1, render_img_scale: float = 1, n_clusters: int = 8,
                    buffer_dist_meters: float = CHUNKED_MESH_BUFFER_DIST_METERS, vis_clusters: bool = False,
                    **pix2face_kwargs):
        """reference: derived_meshes.py:153-220.  NOTE: the reference yields the renders cluster by cluster, i.e. in
        KMeans cluster order; here they come in camera order."""
        self._say_unchunked("render_flat", n_clusters, buffer_dist_meters, vis_clusters)
        yield from super().render_flat(cameras, batch_size=batch_size, render_img_scale=render_img_scale,
                                       **pix2face_kwargs)

    def aggregate_projected_images(self, cameras, batch_size: int = 1, aggregate_img_scale: float = 1,
                                   n_clusters: int = 8, buffer_dist_meters: float = CHUNKED_MESH_BUFFER_DIST_METERS,
                                   vis_clusters: bool = False, **kwargs):
        """reference: derived_meshes.py:222-317 (same return structure as the base class)."""
        self._say_unchunked("aggregate_projected_images", n_clusters, buffer_dist_meters, vis_clusters)
        return super().aggregate_projected_images(cameras, batch_size=batch_size,
                                                  aggregate_img_scale=aggregate_img_scale, **kwargs)


class TexturedPhotogrammetryMeshIndexPredictions(TexturedPhotogrammetryMesh):
    def aggregate_projected_images(
        self,
        cameras: typing.Union[PhotogrammetryCamera, PhotogrammetryCameraSet],
        n_classes: int,
        batch_size: int = 1,
        aggregate_img_scale: float = 1,
        return_all: bool = False,
        **kwargs,
    ):
        """Sparse aggregation of class-index images (reference: derived_meshes.py:415-550).

        Every image is (h, w) or (h, w, 1) float with NaN where nothing was predicted and a class index elsewhere.
        Returns `(average (F, n_classes) scipy CSR, {"projection_counts": CSR (F,1) int, "summed_projections": CSR
        (F, n_classes) int[, "all_projections"]})` exactly like the reference.
        """
        from scipy.sparse import csr_array

        if len(cameras) == 0 or batch_size > len(cameras):
            raise IndexError("list index out of range")
        torch = _torch()
        n_faces = self.faces.shape[0]
        kwargs.pop("check_null_image", None)
        all_projections = [] if return_all else None
        counts = torch.zeros((n_faces,), dtype=torch.int32, device=self.backend.device)
        # the (face, class) pair keys of all views stay on the device; ONE sort + run-length count at the end
        acc = self.backend.new_pair_accumulator(n_classes, counts, neg1_is_last_face=self.neg1_is_last_face)
        gen = self._iter_view_inputs(cameras, batch_size, aggregate_img_scale, True, kwargs)
        for _, ids, img, n_channels in tqdm(gen, total=len(cameras), desc="Aggregating projected viewpoints"):
            if return_all:
                if img is None:
                    all_projections.append(np.full((n_faces, n_channels), fill_value=np.nan))
                else:
                    all_projections.append(
                        self.backend.project_view(ids, img, neg1_is_last_face=self.neg1_is_last_face).cpu().numpy()
                    )
            if img is None:  # null image: nothing to project (check_null_image=True, derived_meshes.py:465)
                continue
            if img.shape[-1] != 1:
                raise ValueError("index predictions must be single-channel images")
            acc.add(ids, img[..., 0])
        uniq, mult = acc.finish()
        summed_vals = mult.astype(int)
        rows, cols = uniq // n_classes, uniq % n_classes
        summed_projections = csr_array((summed_vals, (rows, cols)), shape=(n_faces, n_classes), dtype=int)
        cnt = counts.cpu().numpy().astype(int)
        seen = np.nonzero(cnt)[0]
        projection_counts = csr_array((cnt[seen], (seen, np.zeros_like(seen))), shape=(n_faces, 1), dtype=int)
        info = {"projection_counts": projection_counts, "summed_projections": summed_projections}
        if return_all:
            info["all_projections"] = all_projections
        reciprocal = csr_array(
            (np.reciprocal(projection_counts.data.astype(float)), projection_counts.indices, projection_counts.indptr),
            shape=projection_counts.shape,
        )
        average_projections = summed_projections.multiply(reciprocal)
        return average_projections, info
