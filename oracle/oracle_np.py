"""oracle/oracle_np.py -- TEST INFRASTRUCTURE ONLY.

numpy restatement of the reference's numpy stages on the projection path, one function per reference block:

    render_flat_gather           geograypher/meshes/meshes.py:1921-1937
    project_image                geograypher/meshes/meshes.py:1990-2002
    aggregate                    geograypher/meshes/meshes.py:2044-2084
    inds_to_one_hot              geograypher/predictors/segmentor.py:37-69
    find_argmax_nonzero_value    geograypher/utils/indexing.py:9-32
    get_image_size               geograypher/cameras/cameras.py:179-200
    view_parameters              geograypher/cameras/cameras.py:446-477

PARITY STATUS: pinned.  tests/golden/reference_numpy_stages.npz holds inputs and outputs of the REAL reference
functions (imported from /root/reference under stubbed third-party modules by tests/golden/make_golden.py);
tests/test_oracle_golden.py checks every function below against them.
"""
import numpy as np


def get_image_size(image_height, image_width, image_scale=1.0):
    """cameras.py:179-200 -- truncation, not rounding."""
    return (int(image_height * image_scale), int(image_width * image_scale))


def view_parameters(cam_to_world, f, image_height, focal_dist=10):
    """cameras.py:446-477 -- position, look-at point, up vector, vertical field of view (degrees)."""
    T = np.asarray(cam_to_world, dtype=np.float64)
    position = T[:3, 3]
    look = position + T[:3, :3] @ np.array((0, 0, focal_dist))
    up = T[:3, :3] @ np.array((0, -1, 0))
    fov = np.rad2deg(2 * np.arctan((image_height / 2) / f))
    return position, look, up, fov


def render_flat_gather(pix2face, face_texture):
    """meshes.py:1921-1937"""
    img_shape = pix2face.shape[:2]
    texture_dim = face_texture.shape[1]
    flat = pix2face.flatten()
    mesh_pixel_inds = np.where(flat != -1)[0]
    rendered = np.full((flat.shape[0], texture_dim), fill_value=np.nan)
    rendered[mesh_pixel_inds] = face_texture[flat[mesh_pixel_inds]]
    return rendered.reshape(img_shape + (texture_dim,))


def project_image(pix2face, img, n_faces, check_null_image=False, neg1_is_last_face=True):
    """meshes.py:1990-2002: textured_faces[flat_pix2face] = flat_img (numpy fancy assignment: the LAST occurrence
    of a repeated index wins; index -1 addresses the last face).  neg1_is_last_face=False drops background pixels
    first -- the behaviour the reference's TODO asks for, not what it does."""
    n_channels = 1 if img.ndim == 2 else img.shape[-1]
    textured_faces = np.full((n_faces, n_channels), fill_value=np.nan)
    if not check_null_image or np.any(np.isfinite(img)):
        flat_img = np.reshape(img, (img.shape[0] * img.shape[1], -1))
        flat_pix2face = pix2face.flatten()
        if not neg1_is_last_face:
            keep = flat_pix2face != -1
            flat_pix2face, flat_img = flat_pix2face[keep], flat_img[keep]
        textured_faces[flat_pix2face] = flat_img
    return textured_faces


def aggregate(projections, n_faces, return_all=False):
    """meshes.py:2044-2084 over an iterable of per-view (F,C) projections."""
    projection_counts = np.zeros(n_faces)
    summed_projection = None
    all_projections = []
    for projection_for_image in projections:
        if return_all:
            all_projections.append(projection_for_image)
        if summed_projection is None:
            summed_projection = projection_for_image.astype(float)
        else:
            summed_projection = np.nansum([summed_projection, projection_for_image], axis=0)
        projected_faces = np.any(np.isfinite(projection_for_image), axis=1).astype(int)
        projection_counts += projected_faces
    no_projections = projection_counts == 0
    summed_projection[no_projections] = np.nan
    info = {"projection_counts": projection_counts, "summed_projections": summed_projection}
    if return_all:
        info["all_projections"] = all_projections
    with np.errstate(divide="ignore", invalid="ignore"):
        average = np.divide(summed_projection, np.expand_dims(projection_counts, 1))
    return average, info


def inds_to_one_hot(inds_image, num_classes):
    """predictors/segmentor.py:58-69"""
    one_hot = np.zeros((inds_image.shape[0], inds_image.shape[1], num_classes), dtype=bool)
    for i in range(num_classes):
        one_hot[..., i] = inds_image == i
    return one_hot


def find_argmax_nonzero_value(array, keepdims=False, axis=1):
    """utils/indexing.py:9-32"""
    argmax = np.argmax(array, axis=axis, keepdims=keepdims).astype(float)
    zero_sum_mask = np.sum(array, axis=axis) == 0
    infinite_mask = np.any(~np.isfinite(array), axis=axis)
    argmax[np.logical_or(zero_sum_mask, infinite_mask)] = np.nan
    return argmax


def render_postprocess_uint8(rendered, null_value=0):
    """meshes.py:2325-2337 -- values that cannot be represented as uint8 become the null value, then cast + squeeze."""
    rendered = np.array(rendered, dtype=float, copy=True)
    with np.errstate(invalid="ignore"):
        mask = np.logical_or.reduce([rendered < 0, rendered > 255, np.logical_not(np.isfinite(rendered))])
    rendered[mask] = null_value
    return np.squeeze(rendered.astype(np.uint8))


def aggregate_index_sparse(projections, n_faces, n_classes):
    """derived_meshes.py:470-550 over an iterable of per-view (F,1) projections (NaN = no prediction), dense output:
    (average (F, n_classes), counts (F, 1), summed (F, n_classes))."""
    counts = np.zeros((n_faces, 1), dtype=int)
    summed = np.zeros((n_faces, n_classes), dtype=int)
    for proj in projections:
        inds = np.nonzero(np.isfinite(np.squeeze(proj)))[0]
        if len(inds) == 0:
            continue
        counts[inds, 0] += 1
        classes = proj[inds, 0].astype(int)
        summed[inds, classes] += 1
    with np.errstate(divide="ignore", invalid="ignore"):
        recip = np.where(counts > 0, 1.0 / np.where(counts > 0, counts, 1), 0.0)
    return summed * recip, counts, summed
