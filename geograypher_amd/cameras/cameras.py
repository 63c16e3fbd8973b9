"""Camera records of the projection path: `PhotogrammetryCamera` and `PhotogrammetryCameraSet`.

Host-side mirror of geograypher/cameras/cameras.py (same names, arguments and error behaviour) restricted to what
`pix2face` / `render_flat` / `project_images` / `aggregate_projected_images` touch.  No pyvista: where the reference
builds a `pv.Camera` (cameras.py:446-477) this module exposes the same view parameters as plain numbers and packs
them into the 16-float record the HIP rasterizer consumes (`include/geograster.h`).
"""
from __future__ import annotations

import hashlib
import json
import os
from copy import deepcopy
from pathlib import Path
from typing import Dict, List, Optional, Tuple, Union

import numpy as np

from geograypher_amd.constants import EXAMPLE_INTRINSICS, PATH_TYPE


def _imread(filename) -> np.ndarray:
    """Image file -> numpy array (the reference uses skimage.io.imread, cameras.py:157)."""
    from PIL import Image

    with Image.open(filename) as im:
        return np.array(im)  # a writable copy (PIL hands out a read-only view)


def vtk_like_clipping_range(cam_to_world: np.ndarray, bounds: np.ndarray, tolerance: float = 0.001):
    """(near, far) clipping distances the reference's renderer would pick for this camera and mesh bounds.

    The pyvista path never sets a clipping range (cameras.py:446-477), so VTK derives one from the actor bounds
    (`vtkRenderer::ResetCameraClippingRange`): the range of the 8 bounding-box corners along the view direction,
    widened by 1 % plus half its extent, with the near plane kept at >= `tolerance` x far (0.001 for depth buffers
    deeper than 16 bit).  Restated from the published VTK algorithm; VTK is not runnable here (parity unpinned).  The far
    plane lies beyond every corner of the bounds by construction, so only the near plane can ever cut the mesh.
    bounds: (xmin, xmax, ymin, ymax, zmin, zmax) in the cameras' local frame.
    """
    forward = cam_to_world[:3, 2]
    position = cam_to_world[:3, 3]
    xs, ys, zs = bounds[0:2], bounds[2:4], bounds[4:6]
    dists = [float(np.dot(forward, np.array([x, y, z]) - position)) for x in xs for y in ys for z in zs]
    near, far = min(dists), max(max(dists), 1e-18)
    near = max(near, 0.0)
    expansion = 0.5
    near = 0.99 * near - (far - near) * expansion
    far = 1.01 * far + (far - near) * expansion
    if near >= far:
        near = 0.01 * far
    if near < tolerance * far:
        near = tolerance * far
    return float(near), float(far)


def vtk_like_near_plane(cam_to_world: np.ndarray, bounds: np.ndarray, tolerance: float = 0.001) -> float:
    """The near distance of `vtk_like_clipping_range` (the only one of the two planes that can cut the mesh)."""
    return vtk_like_clipping_range(cam_to_world, bounds, tolerance)[0]


def vtk_like_near_planes(cam_to_worlds: np.ndarray, bounds: np.ndarray, tolerance: float = 0.001) -> np.ndarray:
    """`vtk_like_near_plane` for N cameras at once: cam_to_worlds (N,4,4) -> (N,) near distances."""
    T = np.asarray(cam_to_worlds, dtype=np.float64)
    corners = np.array([[x, y, z] for x in bounds[0:2] for y in bounds[2:4] for z in bounds[4:6]], dtype=np.float64)
    d = np.einsum("nk,nck->nc", T[:, :3, 2], corners[None, :, :] - T[:, None, :3, 3])
    near = np.maximum(d.min(axis=1), 0.0)
    far = np.maximum(d.max(axis=1), 1e-18)
    near = 0.99 * near - (far - near) * 0.5
    far = 1.01 * far + (far - near) * 0.5
    near = np.where(near >= far, 0.01 * far, near)
    return np.where(near < tolerance * far, tolerance * far, near)


class PhotogrammetryCamera:
    def __init__(
        self,
        image_filename: PATH_TYPE,
        cam_to_world_transform: np.ndarray,
        f: float,
        cx: float,
        cy: float,
        image_width: int,
        image_height: int,
        distortion_params: Dict[str, float] = {},
        lon_lat: Union[None, Tuple[float, float]] = None,
        local_to_epsg_4978_transform: Union[np.ndarray, None] = None,
    ):
        """One camera pose + intrinsics as determined by photogrammetry (reference: cameras.py:55-102).

        Args:
            image_filename: the image used for reconstruction (may be None for synthetic cameras)
            cam_to_world_transform: 4x4 camera-to-world (chunk-local) transform; camera frame +X right, +Y down,
                +Z forward
            f: focal length in pixels
            cx, cy: principal point in pixels from the image centre
            image_width, image_height: sensor size in pixels
            distortion_params: lens distortion coefficients (Metashape names)
            lon_lat: optional (lon, lat)
            local_to_epsg_4978_transform: 4x4 chunk-local -> EPSG:4978
        """
        self.image_filename = image_filename
        self.cam_to_world_transform = cam_to_world_transform
        self.world_to_cam_transform = np.linalg.inv(cam_to_world_transform)
        self.f = f
        self.cx = cx
        self.cy = cy
        self.image_width = image_width
        self.image_height = image_height
        self.distortion_params = distortion_params
        self._local_to_epsg_4978_transform = local_to_epsg_4978_transform
        self.lon_lat = (None, None) if lon_lat is None else lon_lat
        self.image_size = (image_height, image_width)
        self.image = None
        self.cache_image = False

    # -- identity ------------------------------------------------------------------------------------------------
    def get_camera_hash(self, include_image_hash: bool = False):
        """sha256 of the camera geometry (reference: cameras.py:104-134; same JSON layout, same digest)."""
        camera_settings = {
            "transform": np.asarray(self.cam_to_world_transform).tolist(),
            "f": self.f,
            "cx": self.cx,
            "cy": self.cy,
            "image_width": self.image_width,
            "image_height": self.image_height,
            "distortion_params": self.distortion_params,
            "lon_lat": self.lon_lat,
        }
        if include_image_hash:
            camera_settings["image_filename"] = str(self.image_filename)
        data = json.dumps(camera_settings, sort_keys=True)
        return hashlib.sha256(data.encode("utf-8")).hexdigest()

    def get_camera_properties(self):
        """reference: cameras.py:136-152"""
        return {
            "focal_length": self.f,
            "principal_point_x": self.cx,
            "principal_point_y": self.cy,
            "image_height": self.image_height,
            "image_width": self.image_width,
            "distortion_params": self.distortion_params,
            "world_to_cam_transform": self.world_to_cam_transform,
        }

    # -- image access --------------------------------------------------------------------------------------------
    def get_image_native(self) -> np.ndarray:
        """The image as its file holds it (`imread`, cameras.py:157; cached like the reference's float image when
        `cache_image` is set).  The device paths take this: a uint8 photo crosses the link as uint8 and is divided by 255 and
        down-scaled there (`HipRaster.resize_image`)."""
        native = getattr(self, "_image_native", None)
        if native is None:
            from geograypher_amd.utils.decoded_cache import cached_decode

            # decoded_cache (set by the camera set / aggregate_projected_images(decoded_cache=...)): the decoded photo as an
            # uncompressed .npy keyed by (path, mtime, size), memory-mapped by later passes; None: decode every time
            native = cached_decode(self.image_filename, getattr(self, "decoded_cache", None),
                                   lambda: _imread(self.image_filename), "photo-native")
            if self.cache_image:
                self._image_native = native
        return native

    def get_image(self, image_scale: float = 1.0, backend=None) -> np.ndarray:
        """reference: cameras.py:154-174 -- uint8 images are returned as float in [0, 1]; `image_scale != 1` resizes with
        scikit-image's `resize` defaults (anti-aliasing Gaussian + order 1).  The resize runs on the device
        (`gr_resize_image_f64`, pinned to the real scikit-image's output in tests/test_photo_resize.py); there is no CPU
        resizer in the product: without a GPU a scaled image raises RuntimeError.  The scaled image is float64 whatever the
        file's dtype (scikit-image >= 0.19, the reference's pinned 0.21, returns float32 for a float32 FILE; photos are uint8
        and come back float64 there too): documented, values within float32 rounding of that."""
        if self.image is None:
            native = self.get_image_native()
            image = native / 255.0 if native.dtype == np.uint8 else native
            # `cache_image` keeps ONE copy per photo: the file's own array (a uint8 photo: an eighth of its float image, and
            # what the device paths upload; `/ 255.0` is redone per call) -- the reference caches the float image
            if self.cache_image and native.dtype != np.uint8:
                self.image = image             # the file's array itself
                self._image_native = None
        else:
            native = None
            image = self.image
        if image_scale != 1.0:
            if backend is None:
                from geograypher_amd._hip import default_backend

                backend = default_backend()
            out_hw = (int(image.shape[0] * image_scale), int(image.shape[1] * image_scale))
            src = native if native is not None else image  # the file dtype when it is at hand (uint8: an eighth of the bytes)
            resized = backend.resize_image(src, out_hw, divide_by_255=src.dtype == np.uint8)
            image = resized.cpu().numpy() if hasattr(resized, "cpu") else np.asarray(resized)
        return image

    def get_image_filename(self):
        return self.image_filename

    def get_image_size(self, image_scale=1.0):
        """(h, w) in pixels, truncated like the reference (cameras.py:179-200)."""
        if self.image_size is not None:
            pass
        elif self.image is not None:
            self.image_size = self.image.shape[:2]
        else:
            self.image_size = self.get_image().shape[:2]
        return (int(self.image_size[0] * image_scale), int(self.image_size[1] * image_scale))

    def get_local_to_epsg_4978_transform(self) -> np.ndarray:
        """reference: cameras.py:311-326"""
        return self._local_to_epsg_4978_transform

    # -- view ----------------------------------------------------------------------------------------------------
    def get_view_parameters(self, focal_dist: float = 10) -> dict:
        """The numbers the reference loads into a pyvista camera (cameras.py:446-477): position, focal point,
        view-up and the VERTICAL field of view in degrees.  The principal point is not part of this view."""
        T = np.asarray(self.cam_to_world_transform, dtype=np.float64)
        position = T[:3, 3]
        focal_point = position + T[:3, :3] @ np.array((0, 0, focal_dist), dtype=np.float64)
        up = T[:3, :3] @ np.array((0, -1, 0), dtype=np.float64)
        view_angle = np.rad2deg(2 * np.arctan((self.image_height / 2) / self.f))
        return {"position": position, "focal_point": focal_point, "up": up, "view_angle": view_angle}

    def get_raster_record(
        self, image_scale: float = 1.0, near: float = 1e-3, principal_point: str = "center", origin=None,
        focal_scaling: str = "scaled",
    ) -> np.ndarray:
        """Pack this view into the 16-float record of include/geograster.h.

        The pinhole that a pyvista camera with the parameters above realises in an (h, w) window is
        u = w/2 + f_eff X/Z, v = h/2 + f_eff Y/Z with f_eff = (h/2)/tan(view_angle/2) = f*h/image_height
        (cameras.py:469-475, meshes.py:1801, 1820-1822).  principal_point="center" reproduces that (cx, cy ignored
        exactly as the reference's pyvista path does); "intrinsics" places it at (w/2 + cx*s, h/2 + cy*s), the
        convention of the reference's PyTorch3D plugin (derived_meshes.py:772-780) scaled to the window.

        origin: float64 point subtracted from the camera position before the fp32 cast (the mesh class subtracts the
            same point from the vertices: ECEF-magnitude coordinates keep their metres, VTK does the same with its
            automatic VBO shift).
        focal_scaling: "scaled" = f * h / image_height (what the pyvista camera's vertical view angle gives in an
            (h, w) window); "unscaled" reproduces the PyTorch3D plugin, which hands the FULL-resolution focal length and
            principal point to a down-scaled image (derived_meshes.py:686-692, 772-780: `focal_length=camera.f`,
            `image_size` scaled) -- a bug-compatibility switch, only meaningful with principal_point="intrinsics".
        """
        h, w = self.get_image_size(image_scale)
        T = np.asarray(self.cam_to_world_transform, dtype=np.float64)
        if focal_scaling == "scaled":
            f_eff = float(self.f) * h / float(self.image_height)
        elif focal_scaling == "unscaled":
            f_eff = float(self.f)
        else:
            raise ValueError(f"focal_scaling must be 'scaled' or 'unscaled', not {focal_scaling!r}")
        if principal_point == "center":
            cxp, cyp = w / 2.0, h / 2.0
        elif principal_point == "intrinsics":
            k = 1.0 if focal_scaling == "unscaled" else h / float(self.image_height)
            cxp = w / 2.0 + float(self.cx) * k
            cyp = h / 2.0 + float(self.cy) * k
        else:
            raise ValueError(f"principal_point must be 'center' or 'intrinsics', not {principal_point!r}")
        rec = np.empty(16, dtype=np.float32)
        rec[0:9] = T[:3, :3].reshape(9)
        rec[9:12] = T[:3, 3] if origin is None else T[:3, 3] - np.asarray(origin, dtype=np.float64)
        rec[12] = f_eff
        rec[13] = cxp
        rec[14] = cyp
        rec[15] = near
        return rec


class PhotogrammetryCameraSet:
    def __init__(
        self,
        cameras: Union[None, PhotogrammetryCamera, List[PhotogrammetryCamera]] = None,
        cam_to_world_transforms: Optional[List[np.ndarray]] = None,
        intrinsic_params_per_sensor_type: Dict[int, Dict[str, float]] = {0: EXAMPLE_INTRINSICS},
        image_filenames: Optional[List[PATH_TYPE]] = None,
        lon_lats: Optional[List[Union[None, Tuple[float, float]]]] = None,
        image_folder: Optional[PATH_TYPE] = None,
        sensor_IDs: Optional[List[int]] = None,
        validate_images: bool = False,
        local_to_epsg_4978_transform: np.ndarray = np.eye(4),
    ):
        """A set of cameras in one chunk-local frame (reference: cameras.py:661-781, same arguments).

        Raises:
            ValueError: if the number of sensor IDs differs from the number of transforms.
        """
        self._local_to_epsg_4978_transform = local_to_epsg_4978_transform
        self._maps_ideal_to_warped = {}
        self._maps_warped_to_ideal = {}

        if cameras is not None:
            if isinstance(cameras, PhotogrammetryCamera):
                self.image_folder = None if cameras.image_filename is None else Path(cameras.image_filename).parent
                cameras = [cameras]
            else:
                names = [str(cam.image_filename) for cam in cameras if cam.image_filename is not None]
                self.image_folder = Path(os.path.commonpath(names)) if len(names) == len(cameras) and names else None
            self.cameras = cameras
            return

        n_transforms = len(cam_to_world_transforms)
        if image_filenames is None:
            image_filenames = [None] * n_transforms
        if sensor_IDs is None and len(intrinsic_params_per_sensor_type) == 1:
            sensor_IDs = [list(intrinsic_params_per_sensor_type.keys())[0]] * n_transforms
        elif len(sensor_IDs) != n_transforms:
            raise ValueError(
                f"Number of sensor_IDs ({len(sensor_IDs)}) is different than the number of transforms ({n_transforms})"
            )
        if lon_lats is None:
            lon_lats = [None] * n_transforms

        self.cam_to_world_transforms = cam_to_world_transforms
        self.intrinsic_params_per_sensor_type = intrinsic_params_per_sensor_type
        self.image_filenames = image_filenames
        self.lon_lats = lon_lats
        self.sensor_IDs = sensor_IDs
        self.image_folder = image_folder

        if validate_images:
            missing_images, invalid_mask = self.find_missing_images()
            if len(missing_images) > 0:
                print(f"Deleting {len(missing_images)} missing images")
                keep = [i for i, bad in enumerate(invalid_mask) if not bad]
                self.image_filenames = [self.image_filenames[i] for i in keep]
                self.cam_to_world_transforms = [self.cam_to_world_transforms[i] for i in keep]
                self.sensor_IDs = [self.sensor_IDs[i] for i in keep]
                self.lon_lats = [self.lon_lats[i] for i in keep]

        self.cameras = []
        for image_filename, cam_to_world_transform, sensor_ID, lon_lat in zip(
            self.image_filenames, self.cam_to_world_transforms, self.sensor_IDs, self.lon_lats
        ):
            sensor_params = self.intrinsic_params_per_sensor_type[sensor_ID]
            if sensor_params is None:  # sensor without a full calibration
                continue
            self.cameras.append(
                PhotogrammetryCamera(
                    image_filename,
                    cam_to_world_transform,
                    lon_lat=lon_lat,
                    local_to_epsg_4978_transform=local_to_epsg_4978_transform,
                    **sensor_params,
                )
            )

    # -- container -----------------------------------------------------------------------------------------------
    def __deepcopy__(self, memo):
        """`get_subset_cameras` deep-copies the set (cameras.py:861-864); device-resident sampling maps and the GPU
        context they belong to are shared, never duplicated."""
        new = self.__class__.__new__(self.__class__)
        memo[id(self)] = new
        for k, v in self.__dict__.items():
            new.__dict__[k] = v if k == "_maps_device" else deepcopy(v, memo)
        return new

    def __len__(self):
        return self.n_cameras()

    def __getitem__(self, slice):
        subset_cameras = self.cameras[slice]
        if isinstance(subset_cameras, PhotogrammetryCamera):
            return subset_cameras
        return PhotogrammetryCameraSet(
            subset_cameras, local_to_epsg_4978_transform=self._local_to_epsg_4978_transform
        )

    def get_image_folder(self):
        return self.image_folder

    def find_missing_images(self):
        invalid_mask = [not Path(image_file).is_file() for image_file in self.image_filenames]
        invalid_images = [f for f, bad in zip(self.image_filenames, invalid_mask) if bad]
        return invalid_images, invalid_mask

    def n_cameras(self) -> int:
        return len(self.cameras)

    def n_image_channels(self) -> int:
        return 3

    def get_subset_cameras(self, inds: List[int]):
        """reference: cameras.py:861-864 (a copy of the set holding only `inds`; IndexError when out of range)."""
        subset_camera_set = deepcopy(self)
        subset_camera_set.cameras = [subset_camera_set[i] for i in inds]
        return subset_camera_set

    def get_image_by_index(self, index: int, image_scale: float = 1.0) -> np.ndarray:
        return self[index].get_image(image_scale=image_scale)

    # May `get_image_by_index` of this set run on the mesh class's loader thread while the caller's thread rasterizes?  False
    # by default: a subclass (a segmentor set around a stateful or GPU segmentor, a user's own look-up) must say so itself.
    # The plain file-backed set needs no flag: the mesh class recognises it by its un-overridden methods and decodes ahead.
    thread_safe_lookup = False

    def set_decoded_cache(self, spec) -> None:
        """Opt in to (or, with None, out of) the decoded-input cache for this set's photos (utils/decoded_cache.py): True --
        `CACHE_FOLDER/decoded`, the reference's cache root (constants.py:18) --, or a folder.  Passes after the first
        memory-map the decoded photos instead of decoding them again."""
        for cam in self.cameras:
            cam.decoded_cache = spec

    def get_native_image_by_index(self, index: int) -> np.ndarray:
        """The image of camera `index` in its file dtype, for the device input pipeline of project_images /
        aggregate_projected_images: `get_image_by_index(i, s)` == resize(native / 255 if uint8 else native, s), with the
        division and the resize done on the device."""
        return self[index].get_image_native()

    def get_image_filename(self, index: Union[int, None], absolute=True):
        """reference: cameras.py:883-909"""
        if index is None:
            return [self.get_image_filename(i, absolute=absolute) for i in range(len(self.cameras))]
        filename = self.cameras[index].get_image_filename()
        if absolute:
            return Path(filename)
        return Path(filename).relative_to(self.get_image_folder())

    def get_local_to_epsg_4978_transform(self):
        """reference: cameras.py:911-926"""
        return self._local_to_epsg_4978_transform

    def get_raster_records(
        self, image_scale: float = 1.0, near: Union[float, List[float]] = 1e-3, principal_point: str = "center",
        origin=None, focal_scaling: str = "scaled",
    ) -> np.ndarray:
        """(N,16) float32 records for the HIP rasterizer; all cameras must share one image size."""
        nears = [near] * len(self.cameras) if np.isscalar(near) else list(near)
        sizes = {cam.get_image_size(image_scale) for cam in self.cameras}
        if len(sizes) > 1:
            raise ValueError("Not all cameras have the same image size")
        return np.stack(
            [cam.get_raster_record(image_scale, nr, principal_point, origin, focal_scaling)
             for cam, nr in zip(self.cameras, nears)], axis=0
        )

    # -- distortion (the warp stage itself is the "next" row f1 of SURVEY.md section 8) -------------------------------
    def distortion_key(self, parameters: Dict[str, float], image_scale: float = 1.0) -> str:
        """reference: cameras.py:968-993 (8-decimal repeatable key)"""
        keys = sorted(parameters.keys())
        strings = [f"{key}:{parameters[key]:.8f}" for key in keys] + [f"image_scale:{image_scale:.8f}"]
        return "|".join(strings)

    def make_distortion_map(self, camera: PhotogrammetryCamera, inversion_downsample: int = 8, image_scale: float = 1.0,
                            backend=None, reference_inverse: bool = False) -> None:
        """Build and cache the two sampling maps of a distortion key (reference: cameras.py:995-1062).

        `_maps_ideal_to_warped[key]` (2, H, W): for every pixel of the IDEAL image, where it lands in the warped image
        (`ideal_to_warped` evaluated on the pixel grid; for a down-scaled image the model is evaluated at the original
        pixel positions of the scaled pixel centres and the result is scaled) -- bit-equal to the reference's.
        `_maps_warped_to_ideal[key]`: its inverse.  The reference inverts by scattered-data interpolation
        (scipy griddata) on every `inversion_downsample`-th pixel, minutes of host time at full resolution; here every
        pixel solves the lens model with Newton's method on the device (`gr_invert_distortion_f64`, float64, residual
        below 1e-9 px): the DENSE inverse, from which the reference's down-sampled one differs by its own interpolation
        error (0.022 px at downsample 8, 0.003 px at 2 on the reference's test lens; tests/test_warp.py) -- enough to move
        about 0.01 % of the nearest-neighbour samples of an id image.  One-time work per key.

        reference_inverse=True inverts the way the reference does instead -- `utils.indexing.inverse_map_interpolation` over
        every `inversion_downsample`-th pixel, on the host, bit-equal to the reference's map -- for parity runs.  The same
        host inversion is what a machine without a GPU gets (and only then is `inversion_downsample` used by default):
        building the maps and `warp_dewarp_pixels` need no device; the image warps themselves do.
        """
        im_h, im_w = camera.image_size
        if np.isclose(image_scale, 1.0):
            h_range = np.arange(im_h)
            w_range = np.arange(im_w)
        else:
            start, step = 1 / (2 * image_scale), 1 / image_scale
            h_range = np.arange(start, im_h, step)[: int(im_h * image_scale)]
            w_range = np.arange(start, im_w, step)[: int(im_w * image_scale)]
        rows, cols = np.meshgrid(h_range, w_range, indexing="ij")
        warp_cols, warp_rows = self.ideal_to_warped(camera, cols, rows)
        if not np.isclose(image_scale, 1.0):
            warp_cols = warp_cols * image_scale
            warp_rows = warp_rows * image_scale
        dkey = self.distortion_key(camera.distortion_params, image_scale)
        self._maps_ideal_to_warped[dkey] = np.stack([warp_rows, warp_cols], axis=0)
        self._maps_device = getattr(self, "_maps_device", {})
        self._maps_device.pop((dkey, True), None)
        self._maps_device.pop((dkey, False), None)
        if backend is None and not reference_inverse:
            import torch

            if torch.cuda.is_available():
                from geograypher_amd._hip import default_backend

                backend = default_backend()
        if reference_inverse or backend is None:
            from geograypher_amd.utils.indexing import inverse_map_interpolation

            self._maps_warped_to_ideal[dkey] = inverse_map_interpolation(self._maps_ideal_to_warped[dkey], inversion_downsample)
            return
        inv = backend.invert_distortion(self.distortion_model(camera), len(h_range), len(w_range), image_scale)
        self._maps_device[(dkey, False)] = (backend, inv)  # already where the warp kernels want it
        self._maps_warped_to_ideal[dkey] = inv.cpu().numpy()

    def distortion_model(self, camera: PhotogrammetryCamera) -> Dict[str, float]:
        """Parameters of the camera's lens model in the layout `gr_invert_distortion_f64` takes -- only derived sets
        know a distortion model (as for `ideal_to_warped`, cameras.py:1064-1090)."""
        raise NotImplementedError(f"distortion_model not implemented for {self.__class__}.")

    def ideal_to_warped(self, camera: PhotogrammetryCamera, xpix: np.ndarray, ypix: np.ndarray):
        """reference: cameras.py:1064-1090 -- only derived sets know a distortion model."""
        raise NotImplementedError(f"ideal_to_warped not implemented for {self.__class__}.")

    def warp_dewarp_image(
        self,
        camera: PhotogrammetryCamera,
        input_image,
        fill_value: float = 0.0,
        inversion_downsample: int = 8,
        interpolation_order: int = 1,
        warped_to_ideal: bool = True,
        image_scale: float = 1.0,
        backend=None,
        reference_float_roundtrip: bool = False,
    ):
        """Apply (ideal->warped) or undo (warped->ideal) a camera's lens distortion on an image
        (reference: cameras.py:1092-1156 + utils/image.py:72-126).

        The cached (2, H, W) map gives, for every output pixel, the position to sample in the input; the resampling
        itself (nearest neighbour for `interpolation_order=0`, bilinear for 1; `fill_value` outside the input) is a
        HIP gather kernel (`gr_warp_nearest_i32` / `gr_warp_f64`).  Input may be a numpy array (numpy is returned,
        same dtype) or a device tensor (a tensor is returned).

        reference_float_roundtrip: the reference pushes every image -- integer face-id images included -- through a
            float rescale to [0,1] and back before truncating to the input dtype (utils/image.py:102, 123), which
            returns ids that are off by one for a few percent of the pixels.  False (default) gathers integers exactly;
            True reproduces the reference's arithmetic bit for bit (order 0).
        """
        dkey = self.distortion_key(camera.distortion_params, image_scale)
        if dkey not in self._maps_ideal_to_warped:  # a set without a lens model raises NotImplementedError here
            self.make_distortion_map(camera, inversion_downsample, image_scale, backend=backend)
        inverse_map = self._maps_ideal_to_warped[dkey] if warped_to_ideal else self._maps_warped_to_ideal[dkey]
        if backend is None:
            from geograypher_amd._hip import default_backend

            backend = default_backend()
        self._maps_device = getattr(self, "_maps_device", {})
        mkey = (dkey, bool(warped_to_ideal))
        if mkey not in self._maps_device or self._maps_device[mkey][0] is not backend:
            self._maps_device[mkey] = (backend, backend.upload_map(inverse_map))
        return backend.warp_image(
            input_image, self._maps_device[mkey][1], order=interpolation_order, fill_value=fill_value,
            reference_float_roundtrip=reference_float_roundtrip,
        )

    def warp_dewarp_pixels(self, camera: PhotogrammetryCamera, pixels: np.ndarray, inversion_downsample: int = 8,
                           warped_to_ideal: bool = True, backend=None):
        """(N,2) integer (i,j) pixels -> their float positions in the other image (reference: cameras.py:1158-1205)."""
        dkey = self.distortion_key(camera.distortion_params)
        if dkey not in self._maps_ideal_to_warped:
            self.make_distortion_map(camera, inversion_downsample, backend=backend)
        rowmap, colmap = self._maps_warped_to_ideal[dkey] if warped_to_ideal else self._maps_ideal_to_warped[dkey]
        rows = rowmap[pixels[:, 0], pixels[:, 1]]
        cols = colmap[pixels[:, 0], pixels[:, 1]]
        return np.stack([rows, cols], axis=0).T
