"""`TexturedPhotogrammetryMesh`: the image<->mesh projection path of geograypher on MI355X.

Host-side mirror of the hot-path slice of geograypher/meshes/meshes.py (1631-2084): same method names, arguments,
return types and error behaviour, so `render_labels` / `aggregate_images`-style callers run unchanged.  All per-pixel
and per-face work happens in hand-written HIP kernels behind the C ABI of include/geograster.h:

    pix2face                     -> gr_raster_face_ids           (replaces the VTK render of meshes.py:1776-1836)
    render_flat                  -> + gr_gather_texture_f64      (meshes.py:1921-1937)
    project_images               -> + gr_project_view_f64        (meshes.py:1987-2002)
    aggregate_projected_images   -> + gr_project_labels_u8 / gr_project_values_f64 + gr_finalize_*  (2044-2084)

There is no CPU implementation in this package: without the HIP extension or a GPU the constructor of the backend
raises.  (`backend=` exists so tests can drive this host logic against the CPU oracle; the product never does.)
"""
from __future__ import annotations

import hashlib
import logging
import sys
import contextlib
import typing
from pathlib import Path

import numpy as np

from geograypher_amd.cameras.cameras import (
    PhotogrammetryCamera,
    PhotogrammetryCameraSet,
    vtk_like_near_planes,
)
from geograypher_amd.constants import (
    CACHE_FOLDER,
    EARTH_CENTERED_EARTH_FIXED_CRS,
    NULL_TEXTURE_INT_VALUE,
    PATH_TYPE,
    VIS_FOLDER,
)

try:  # tqdm is optional: the reference wraps its view loops in it (meshes.py:2049-2053)
    from tqdm import tqdm
except Exception:  # pragma: no cover

    def tqdm(it, **_):
        return it


class LocalMesh:
    """A mesh expressed in the cameras' chunk-local frame: what `get_mesh_in_cameras_coords` returns in place of
    the reference's transformed `pv.PolyData` (meshes.py:1641-1676)."""

    def __init__(self, points: np.ndarray, faces: np.ndarray, key: typing.Optional[str] = None):
        self.points = points
        self.faces = faces
        self.key = key

    @property
    def n_faces(self):
        return int(self.faces.shape[0])

    def bounds(self) -> np.ndarray:
        if getattr(self, "_bounds", None) is None:
            lo, hi = self.points.min(axis=0), self.points.max(axis=0)
            self._bounds = np.array([lo[0], hi[0], lo[1], hi[1], lo[2], hi[2]], dtype=np.float64)
        return self._bounds

    def origin(self) -> np.ndarray:
        """float64 point the rasterizer's fp32 frame is centred on.  The kernels compute p - t in fp32, so coordinates
        of ECEF magnitude (6.4e6 m: fp32 ulp 0.5 m) must be re-centred before the cast -- VTK does the same with its
        automatic VBO shift/scale.  Meshes whose coordinates are already small (|x| < 4096: ulp <= 0.5 mm) keep the
        frame they have; otherwise the origin is the bounding-box centre rounded to whole metres."""
        if getattr(self, "_origin", None) is None:
            b = self.bounds()
            finite = np.all(np.isfinite(b))
            if finite and np.max(np.abs(b)) >= 4096.0:
                self._origin = np.round(np.array([(b[0] + b[1]) / 2, (b[2] + b[3]) / 2, (b[4] + b[5]) / 2]))
            else:
                self._origin = np.zeros(3)
        return self._origin


def _parse_mesh(mesh) -> typing.Tuple[np.ndarray, np.ndarray]:
    """Accept (points, faces), a pyvista-like object (.points/.faces), or a .npz file with those two arrays."""
    if isinstance(mesh, (str, Path)):
        path = Path(mesh)
        if path.suffix != ".npz":
            raise NotImplementedError(
                f"Mesh file loading is limited to .npz (points, faces) here; {path.suffix} readers (pyvista) are "
                "outside the projection path (SURVEY.md section 8)"
            )
        with np.load(path) as data:
            return _parse_mesh((data["points"], data["faces"]))
    if isinstance(mesh, (tuple, list)) and len(mesh) == 2:
        points, faces = mesh
    elif hasattr(mesh, "points") and hasattr(mesh, "faces"):
        points, faces = mesh.points, mesh.faces
    else:
        raise TypeError("mesh must be (points, faces), an object with .points/.faces, or a .npz path")
    points = np.asarray(points)
    faces = np.asarray(faces)
    if faces.ndim == 1:  # pyvista's padded layout [3, a, b, c, 3, ...]
        faces = faces.reshape(-1, 4)
        if not np.all(faces[:, 0] == 3):
            raise ValueError("only triangular faces are supported")
        faces = faces[:, 1:4]
    if points.ndim != 2 or points.shape[1] != 3 or faces.ndim != 2 or faces.shape[1] != 3:
        raise ValueError(f"expected (V,3) points and (F,3) faces, got {points.shape} and {faces.shape}")
    return points, faces.astype(np.int64, copy=True)


class TexturedPhotogrammetryMesh:
    def __init__(
        self,
        mesh,
        input_CRS=EARTH_CENTERED_EARTH_FIXED_CRS,
        downsample_target: float = 1.0,
        texture: typing.Union[PATH_TYPE, np.ndarray, None] = None,
        texture_column_name: typing.Union[PATH_TYPE, None] = None,
        IDs_to_labels: typing.Union[PATH_TYPE, dict, None] = None,
        shift: typing.Union[np.ndarray, None] = None,
        ROI=None,
        ROI_buffer_meters: float = 0,
        log_level: str = "INFO",
        device: typing.Optional[int] = None,
        backend=None,
        neg1_is_last_face: bool = True,
        devices: typing.Optional[typing.Sequence[int]] = None,
        vertex_order: str = "r1",
    ):
        """A textured mesh that renders to / aggregates from camera views on an MI355X.

        Same leading arguments as the reference constructor (meshes.py:55-156).  `mesh` is `(points (V,3), faces
        (F,3))`, any object with `.points`/`.faces` (pyvista layout accepted) or a `.npz` path.  The coordinates are
        interpreted in `input_CRS`; only EPSG:4978 (the frame the reference reprojects every mesh into,
        meshes.py:1659) is accepted because CRS reprojection is outside the projection path.  Decimation, ROI
        cropping and vector/raster texture files are likewise outside it and raise NotImplementedError.

        Extra keyword arguments (defaults keep reference behaviour):
            device: GPU index for the HIP backend (default: current torch device).
            devices: several GPU indices, e.g. `[0, 1, 2, 3, 4, 5, 6, 7]`: `aggregate_projected_images` deals its views round-robin to
                one libgeograster context + host thread per entry (the mesh is uploaded to each), adds the per-device partial
                votes (or float sums + counts) on the first device and finalises there -- the unchanged single-process caller
                (entrypoints/aggregate_images.py:146-184) uses the whole node.  Everything else runs on `devices[0]`.  The same
                index may appear more than once (contexts are independent).  Default None: one device, `device`.
            backend: test hook -- an object with the `HipRaster` interface (or a list of them: one per entry of `devices`).
            vertex_order: "r1" (default) -- the rule-set's own vertex stage --, or "gl": perspective divide,
                viewport transform and sub-pixel snap in the order of operations of an OpenGL pipeline, op for op what Mesa's
                llvmpipe (the software GL of the reference's Dockerfile) executes behind the camera transform
                (`GR_OPT_VERTEX_ORDER`).  The two put a few per cent of a view's vertices on neighbouring 1/256 px steps, which
                decides 0.004 % of its pixels; with "gl" a C2 view differs from a real llvmpipe render on 14 of 12 000 000 pixels
                (`profiles/r06_gl_residue.txt`).  Needs the pyvista camera's principal point (`principal_point="center"`).
            neg1_is_last_face: reproduce meshes.py:1998-2001, where background pixels (-1) index the LAST face
                during projection.  True matches the reference's aggregated textures on every face.
        """
        if downsample_target != 1.0:
            raise NotImplementedError("mesh decimation is outside the projection path (meshes.py:215-226)")
        if ROI is not None:
            raise NotImplementedError("ROI cropping is outside the projection path (meshes.py:646-731)")
        if input_CRS is not None and str(input_CRS).upper().replace(" ", "") not in ("EPSG:4978",):
            raise NotImplementedError(
                f"input_CRS={input_CRS!r}: only EPSG:4978 meshes are accepted (CRS reprojection needs pyproj and is "
                "outside the projection path)"
            )
        self.downsample_target = downsample_target
        self.CRS = input_CRS
        self.texture = None
        self.vertex_texture = None
        self.face_texture = None
        self.IDs_to_labels = None
        self.neg1_is_last_face = neg1_is_last_face

        self.logger = logging.getLogger(f"mesh_{id(self)}")
        self.logger.setLevel(log_level)
        if not self.logger.hasHandlers():
            self.logger.addHandler(logging.StreamHandler(stream=sys.stdout))

        self.logger.info("Loading mesh")
        points, faces = _parse_mesh(mesh)
        if shift is not None:
            points = points + np.asarray(shift).reshape(1, 3)
        self.points = points
        self.faces = faces

        if devices is not None:
            devices = [int(d) for d in devices]
            if len(devices) == 0:
                raise ValueError("devices must name at least one GPU")
            if device is not None and int(device) != devices[0]:
                raise ValueError(f"device={device} contradicts devices[0]={devices[0]}")
            device = devices[0]
        backends = None
        if isinstance(backend, (list, tuple)):
            backends = list(backend)
            if devices is not None and len(backends) != len(devices):
                raise ValueError(f"{len(backends)} backends for {len(devices)} devices")
            backend = backends[0]
        self._device_index = device
        self._devices = devices
        self._backend = backend
        self._backends = backends
        self._uploaded = {}   # id(backend) -> (points, faces) it holds
        if vertex_order not in ("r1", "gl"):
            raise ValueError(f"vertex_order must be 'r1' or 'gl', got {vertex_order!r}")
        self.vertex_order = vertex_order

        self.logger.info("Loading texture")
        if isinstance(IDs_to_labels, (str, Path)):
            import json

            with open(IDs_to_labels, "r") as file:
                IDs_to_labels = {int(k): v for k, v in json.load(file).items()}
        self.IDs_to_labels = IDs_to_labels
        if isinstance(texture, (str, Path)):
            if Path(texture).suffix != ".npy":
                raise NotImplementedError("only .npy texture files are read here (meshes.py:533-644 is out of scope)")
            texture = np.load(texture)
        if texture is not None:
            self.set_texture(np.asarray(texture))

    # -- backend -------------------------------------------------------------------------------------------------
    @property
    def backend(self):
        if self._backend is None:
            from geograypher_amd._hip import HipRaster

            self._backend = HipRaster(self._device_index)  # raises when the extension or the GPU is missing
        self._apply_vertex_order(self._backend)
        return self._backend

    def _apply_vertex_order(self, backend):
        if getattr(backend, "vertex_order", "r1") != self.vertex_order:
            if not hasattr(backend, "set_vertex_order"):
                raise NotImplementedError(f"backend {type(backend).__name__} has no vertex_order switch")
            backend.set_vertex_order(self.vertex_order)

    @property
    def backends(self):
        """One backend per entry of `devices` (the first is `self.backend`); a single-device mesh has one."""
        if self._backends is None:
            if not self._devices or len(self._devices) == 1:
                self._backends = [self.backend]
            else:
                from geograypher_amd._hip import HipRaster

                self._backends = [self.backend] + [HipRaster(d) for d in self._devices[1:]]
        for b in self._backends:
            self._apply_vertex_order(b)
        return self._backends

    # -- texture (reference: meshes.py:325-531) ------------------------------------------------------------------
    def standardize_texture(self, texture_array: np.ndarray):
        if texture_array.ndim == 1:
            texture_array = np.expand_dims(texture_array, axis=1)
        elif texture_array.ndim != 2:
            raise ValueError(f"Input texture should have 1 or 2 dimensions but instead has {texture_array.ndim}")
        return texture_array

    def is_discrete_texture(self):
        return self.IDs_to_labels is not None

    def get_IDs_to_labels(self):
        return self.IDs_to_labels

    def set_texture(self, texture_array, is_vertex_texture: typing.Union[bool, None] = None, delete_existing=True):
        """reference: meshes.py:476-531 (same inference rules and errors)."""
        texture_array = self.standardize_texture(np.asarray(texture_array))
        if is_vertex_texture is None:
            n_values = texture_array.shape[0]
            n_faces = self.faces.shape[0]
            n_verts = self.points.shape[0]
            if n_verts == n_faces:
                raise ValueError(
                    "Cannot infer whether texture should be applied to vertices of faces because the number is the same"
                )
            elif n_values == n_verts:
                is_vertex_texture = True
            elif n_values == n_faces:
                is_vertex_texture = False
            else:
                raise ValueError(
                    f"The number of elements in the texture ({n_values}) did not match the number of faces "
                    f"({n_faces}) or vertices ({n_verts})"
                )
        if is_vertex_texture:
            self.vertex_texture = texture_array
            if delete_existing:
                self.face_texture = None
        else:
            self.face_texture = texture_array
            if delete_existing:
                self.vertex_texture = None

    def vert_to_face_texture(self, vert_IDs, discrete=True):
        """reference: meshes.py:947-987.  Continuous textures: mean of the three vertex rows.  The discrete branch
        of the reference votes with an UNSEEDED random tie-break (utils/numeric.py:622-659); here ties go to the
        smallest value so results are reproducible."""
        if vert_IDs is None:
            raise ValueError("None")
        vert_IDs = np.squeeze(vert_IDs)
        if vert_IDs.ndim != 1 and discrete:
            raise ValueError(
                f"Can only perform discrete conversion with one dimensional array but instead had {vert_IDs.ndim}"
            )
        values_per_face = vert_IDs[self.faces]
        if not discrete:
            return np.mean(values_per_face, axis=1)
        a, b, c = values_per_face[:, 0], values_per_face[:, 1], values_per_face[:, 2]
        out = np.where((b == c) & np.isfinite(b), b, np.fmin(np.fmin(a, b), c))
        out = np.where(((a == b) | (a == c)) & np.isfinite(a), a, out)
        return out

    def get_texture(self, request_vertex_texture: typing.Union[bool, None] = None, try_verts_faces_conversion=True):
        """reference: meshes.py:337-378"""
        if self.vertex_texture is None and self.face_texture is None:
            return
        if request_vertex_texture is None:
            if self.vertex_texture is not None and self.face_texture is not None:
                raise ValueError("Ambigious which texture is requested, set request_vertex_texture appropriately")
            request_vertex_texture = self.vertex_texture is not None
        if request_vertex_texture:
            if self.vertex_texture is not None:
                return self.standardize_texture(self.vertex_texture)
            raise NotImplementedError("face -> vertex texture conversion is outside the projection path")
        if self.face_texture is not None:
            return self.standardize_texture(self.face_texture)
        elif try_verts_faces_conversion:
            face_texture = self.vert_to_face_texture(self.vertex_texture, discrete=self.is_discrete_texture())
            self.set_texture(face_texture, is_vertex_texture=False)
            return self.face_texture
        raise ValueError("Face texture not present and conversion was not requested")

    # -- geometry ------------------------------------------------------------------------------------------------
    def get_mesh_hash(self):
        """sha256 over the vertex bytes and pyvista's padded int64 face layout (reference: meshes.py:1631-1639)."""
        hasher = hashlib.sha256()
        hasher.update(np.ascontiguousarray(self.points).tobytes())
        padded = np.concatenate([np.full((self.faces.shape[0], 1), 3, dtype=np.int64), self.faces], axis=1)
        hasher.update(np.ascontiguousarray(padded).tobytes())
        return hasher.hexdigest()

    def get_mesh_in_cameras_coords(self, cameras, inplace: bool = False) -> typing.Optional[LocalMesh]:
        """Mesh in the chunk-local frame of `cameras` (reference: meshes.py:1641-1676): x_local = inv(T) x_ecef in
        float64, with T = cameras.get_local_to_epsg_4978_transform()."""
        T = cameras.get_local_to_epsg_4978_transform()
        T = np.eye(4) if T is None else np.asarray(T, dtype=np.float64)
        key = hashlib.sha1(T.tobytes()).hexdigest()
        cache = self.__dict__.setdefault("_local_mesh_cache", {})
        if not inplace and key in cache and cache[key][0] is self.points and cache[key][1].faces is self.faces:
            return cache[key][1]  # the reference transforms the whole mesh again for every call (meshes.py:1659-1666)
        epsg_4978_to_camera = np.linalg.inv(T)
        pts = np.asarray(self.points, dtype=np.float64)
        local = pts @ epsg_4978_to_camera[:3, :3].T + epsg_4978_to_camera[:3, 3]
        mesh = LocalMesh(local, self.faces, key=key)
        if not inplace:
            cache.clear()
            cache[key] = (self.points, mesh)
        if inplace:
            self.points = local
            self.CRS = None
            return None
        return mesh

    def _ensure_uploaded(self, mesh: LocalMesh, backend=None):
        """Upload `mesh` to `backend` (default: the first) unless that device already holds exactly these arrays.  The cache
        holds strong references and compares identity (`is`) of the point and face arrays: a different array -- also one
        that happens to reuse the id() of a freed temporary -- is always uploaded again."""
        backend = self.backend if backend is None else backend
        held = self._uploaded.get(id(backend))
        if held is None or held[0] is not mesh.points or held[1] is not mesh.faces:
            origin = mesh.origin()
            pts = np.asarray(mesh.points, dtype=np.float64)
            if np.any(origin != 0.0):
                pts = pts - origin
            backend.upload_mesh(pts.astype(np.float32), mesh.faces.astype(np.int32))
            self._uploaded[id(backend)] = (mesh.points, mesh.faces)

    # -- pix2face ------------------------------------------------------------------------------------------------
    def _raster_records(self, cameras, mesh, render_img_scale, near=None, principal_point="center", focal_scaling="scaled",
                        backend=None):
        """Upload the local mesh if needed and pack the (N,16) camera records; returns (records, (h, w))."""
        if isinstance(cameras, PhotogrammetryCamera):
            cameras = PhotogrammetryCameraSet([cameras], local_to_epsg_4978_transform=cameras._local_to_epsg_4978_transform)
        if mesh is None:
            mesh = self.get_mesh_in_cameras_coords(cameras)
        elif not isinstance(mesh, LocalMesh):
            pts, fcs = _parse_mesh(mesh)
            mesh = LocalMesh(np.asarray(pts, dtype=np.float64), fcs)
        if self.vertex_order == "gl" and principal_point != "center":
            raise ValueError('vertex_order="gl" restates an OpenGL viewport: it needs principal_point="center" (the pyvista camera)')
        self._ensure_uploaded(mesh, backend)
        if near is None:
            near = vtk_like_near_planes(
                np.stack([np.asarray(c.cam_to_world_transform, dtype=np.float64) for c in cameras.cameras]), mesh.bounds()
            )
        records = cameras.get_raster_records(render_img_scale, near=near, principal_point=principal_point,
                                             origin=mesh.origin(), focal_scaling=focal_scaling)
        return records, cameras.cameras[0].get_image_size(render_img_scale)

    def _pix2face_device(self, cameras, mesh, render_img_scale, near=None, principal_point="center", focal_scaling="scaled"):
        """(N,h,w) int32 device tensor of face ids for a camera or camera set."""
        records, (h, w) = self._raster_records(cameras, mesh, render_img_scale, near=near, principal_point=principal_point,
                                               focal_scaling=focal_scaling)
        return self.backend.raster_face_ids(records, h, w)

    def pix2face(
        self,
        cameras: typing.Union[PhotogrammetryCamera, PhotogrammetryCameraSet],
        mesh=None,
        render_img_scale: float = 1,
        save_to_cache: bool = False,
        cache_folder: typing.Union[None, PATH_TYPE] = CACHE_FOLDER,
        distortion_set: typing.Optional[PhotogrammetryCameraSet] = None,
        apply_distortion: bool = True,
        return_tensor: bool = False,
        near: typing.Union[None, float, typing.List[float]] = None,
        principal_point: str = "center",
        focal_scaling: str = "scaled",
    ):
        """Face hit by the ray through each pixel, per camera (reference: meshes.py:1678-1856).

        Returns an int64 numpy array of shape (h, w) for a single camera or (n_cameras, h, w) for a camera set, with
        -1 where no face is visible and (h, w) = (int(H*scale), int(W*scale)).  With `return_tensor=True` the int32
        device tensor is returned instead (no host copy).  `save_to_cache` / `cache_folder` are accepted for API
        compatibility and unused, as in the reference's own GPU plugin (derived_meshes.py:665-668): rasterizing on
        the GPU is faster than reading a cached array from disk.

        `principal_point="intrinsics"` + `focal_scaling="unscaled"` reproduces the reference's PyTorch3D plugin
        (derived_meshes.py:686-692, 772-780), including its use of the full-resolution focal length on a down-scaled
        image; the defaults reproduce the pyvista path (cameras.py:446-477).
        """
        if distortion_set is None and apply_distortion:
            self.logger.warning("Distortion requested but no distortion parameters provided. Skipping")
            apply_distortion = False
        single = isinstance(cameras, PhotogrammetryCamera)
        if not single and not isinstance(cameras, PhotogrammetryCameraSet):
            raise TypeError()
        ids = self._pix2face_device(cameras, mesh, render_img_scale, near=near, principal_point=principal_point,
                                    focal_scaling=focal_scaling)
        if apply_distortion:
            # reference: meshes.py:1842-1854.  A base camera set raises NotImplementedError here, as the reference does
            cams = [cameras] if single else cameras.cameras
            torch = _torch()
            warped = []
            for i, cam in enumerate(cams):
                # the id image stays on the device: tensor in, tensor out (gr_warp_nearest_i32)
                out_i = distortion_set.warp_dewarp_image(
                    camera=cam,
                    input_image=ids[i],
                    warped_to_ideal=False,
                    fill_value=-1,
                    interpolation_order=0,
                    image_scale=render_img_scale,
                    backend=self.backend,
                )
                warped.append(out_i if isinstance(out_i, torch.Tensor) else torch.as_tensor(np.asarray(out_i)))
            out = torch.stack([w.to(torch.int32) for w in warped], dim=0)
            if return_tensor:
                return out[0] if single else out
            out = _ids_to_host_int64(out)
            return out[0] if single else out
        if return_tensor:
            return ids[0] if single else ids
        out = _ids_to_host_int64(ids)  # int64 like the reference (meshes.py:1804)
        return out[0] if single else out

    # -- render_flat ---------------------------------------------------------------------------------------------
    def render_flat(
        self,
        cameras: typing.Union[PhotogrammetryCamera, PhotogrammetryCameraSet],
        batch_size: int = 1,
        render_img_scale: float = 1,
        return_camera: bool = False,
        return_tensor: bool = False,
        **pix2face_kwargs,
    ):
        """Generator: the face texture seen from each camera, (h, w, C) float64 with NaN where no face is visible
        (reference: meshes.py:1858-1942, including its batch arithmetic).  `return_tensor=True` yields the renders as
        device tensors instead (no 96 MB host copy per 4000 x 3000 channel: the numpy path is bound by the link)."""
        mesh = self.get_mesh_in_cameras_coords(cameras)
        if isinstance(cameras, PhotogrammetryCamera):
            cameras = PhotogrammetryCameraSet([cameras], local_to_epsg_4978_transform=cameras._local_to_epsg_4978_transform)
        elif not isinstance(cameras, PhotogrammetryCameraSet):
            raise TypeError()
        face_texture = self.get_texture(request_vertex_texture=False, try_verts_faces_conversion=True)
        face_texture = np.asarray(face_texture, dtype=np.float64)
        tex_dev = None

        batch_stop = max(len(cameras) - batch_size + 1, 1)
        for batch_start in range(0, batch_stop, batch_size):
            batch_cameras = cameras[batch_start : batch_start + batch_size]
            batch_pix2face = self.pix2face(
                cameras=batch_cameras, mesh=mesh, render_img_scale=render_img_scale, return_tensor=True,
                **pix2face_kwargs,
            )
            if isinstance(batch_pix2face, np.ndarray):  # distortion applied on the host
                batch_pix2face = self.backend._dev(batch_pix2face.astype(np.int32), _torch().int32)
            if tex_dev is None:
                tex_dev = self.backend._dev(face_texture, _torch().float64)
            rendered = self.backend.gather_texture(batch_pix2face, tex_dev)
            if not return_tensor:
                rendered = _to_host(rendered)
            for i in range(rendered.shape[0]):
                if return_camera:
                    yield (rendered[i], batch_cameras[i])
                else:
                    yield rendered[i]

    # -- project_images ------------------------------------------------------------------------------------------
    def _iter_view_inputs(self, cameras, batch_size, aggregate_img_scale, check_null_image, pix2face_kwargs, loader_threads=None):
        """Shared view loop of project_images / aggregate_projected_images (reference: meshes.py:1970-1996):
        yields (view index, ids (h,w) int32 device tensor, image (h,w,C) device tensor or None for a null image,
        n_channels).  Trailing cameras that do not fill a batch are dropped, as in the reference (1976-1977).

        Input pipeline: the reference converts every image to float64 on the host and the first versions here uploaded that
        (a 4000 x 3000 RGB photo: 36 MB of uint8 blown up to 288 MB, copied from pageable memory, synchronously).  Now the
        image crosses the link in ITS OWN dtype (bool one-hot masks and uint8 photos are 8x smaller than float64) through a
        pinned double buffer; a loader thread fetches and stages view i + 1 (`get_image_by_index`, row blocks copied by a
        small pool) while the device works on view i; the consumers widen to float64 on the device, which is exact for every
        dtype numpy widens exactly."""
        mesh = self.get_mesh_in_cameras_coords(cameras)
        torch = _torch()
        from concurrent.futures import ThreadPoolExecutor

        on_gpu = self.backend.device.type == "cuda"
        batch_stop = max(len(cameras) - batch_size + 1, 1)
        order = [b + i for b in range(0, batch_stop, batch_size) for i in range(batch_size)]
        slots = [None, None]        # pinned staging tensors
        slot_free = [None, None]    # event after the last copy out of the slot
        # File-backed photos of a plain camera set (cameras.py:154-174 not overridden): the image is staged in its FILE dtype
        # and `get_image`'s own arithmetic -- / 255.0 for uint8, skimage's resize for aggregate_img_scale != 1 -- runs on the
        # device (HipRaster.resize_image): a 4000 x 3000 RGB photo crosses the link as 36 MB of uint8 instead of a float64
        # image, and no CPU resizer exists in the product.
        native_files = (
            type(cameras).get_image_by_index is PhotogrammetryCameraSet.get_image_by_index
            and len(cameras.cameras) > 0
            and all(type(cam).get_image is PhotogrammetryCamera.get_image for cam in cameras.cameras)
            and hasattr(self.backend, "resize_image")
        )
        # a loader thread stages view i + 1 while the device works on view i -- only where look-ups are known to be
        # independent (file reads; a camera set or segmentor that says so): an arbitrary segmentor may run its own GPU work
        # or keep state that must not run beside pix2face on another thread
        threaded = native_files or bool(
            getattr(cameras, "thread_safe_lookup", False)
            or getattr(getattr(cameras, "segmentor", None), "thread_safe_lookup", False)
        )
        native = {"uint8": torch.uint8, "int8": torch.int8, "int16": torch.int16, "int32": torch.int32, "int64": torch.int64,
                  "float16": torch.float16, "float32": torch.float32, "float64": torch.float64}

        # file-backed photos: the DECODE (PNG / JPEG inflate, one core per image) is what binds, so a window of the next views
        # is decoded ahead by a pool of threads (PIL releases the interpreter lock while it decodes); staging and upload of
        # view i + 1 still overlap the device work on view i
        import os as _os

        n_dec = int(loader_threads or min(16, _os.cpu_count() or 1)) if native_files else 0
        decoded = {}

        def decode_ahead(pool, pos):
            for p in range(pos, min(pos + max(n_dec, 1), len(order))):
                if p not in decoded:
                    decoded[p] = pool.submit(cameras.get_native_image_by_index, order[p])

        def stage(k, pos, copy_pool):
            """-> (host tensor, n_channels, resize target or None)"""
            resize_to = None
            if native_files:
                decode_ahead(dec_pool, pos)
                img = np.asarray(decoded.pop(pos).result())
                out_hw = (int(img.shape[0] * aggregate_img_scale), int(img.shape[1] * aggregate_img_scale))
                if aggregate_img_scale != 1.0 or img.dtype == np.uint8:
                    resize_to = (out_hw, img.dtype == np.uint8)  # target size, `/ 255.0` first (cameras.py:158-159)
            else:
                img = np.asarray(cameras.get_image_by_index(order[pos], aggregate_img_scale))
            n_channels = 1 if img.ndim == 2 else img.shape[-1]
            flat = np.reshape(img, (img.shape[0], img.shape[1], -1))
            if flat.dtype == np.bool_:
                flat = flat.view(np.uint8)
            tdtype = native.get(flat.dtype.name)
            if tdtype is None:  # anything else takes the reference's route: float64 on the host
                flat, tdtype = flat.astype(np.float64), torch.float64
            if not flat.flags.writeable:  # a memory-mapped cache entry: torch wants a writable array
                flat = np.array(flat)
            if not on_gpu:
                return torch.from_numpy(np.ascontiguousarray(flat)), n_channels, resize_to
            if slot_free[k] is not None:
                slot_free[k].synchronize()  # the copy that read this slot two views ago is done
            if slots[k] is None or slots[k].shape != flat.shape or slots[k].dtype != tdtype:
                try:
                    slots[k] = torch.empty(flat.shape, dtype=tdtype, pin_memory=True)
                except RuntimeError:  # no pinned memory left: pageable upload
                    return torch.from_numpy(np.ascontiguousarray(flat)), n_channels, resize_to
            dst = slots[k].numpy()
            rows = flat.shape[0]
            if flat.nbytes >= (32 << 20) and rows >= 8:  # numpy releases the interpreter lock inside large copies
                step = (rows + 7) // 8
                list(copy_pool.map(lambda r0: np.copyto(dst[r0:r0 + step], flat[r0:r0 + step]), range(0, rows, step)))
            else:
                np.copyto(dst, flat)
            return slots[k], n_channels, resize_to

        class _Now:  # a finished "future": staging on the caller's thread
            def __init__(self, value):
                self._value = value

            def result(self):
                return self._value

        with ThreadPoolExecutor(max_workers=1) as loader, ThreadPoolExecutor(max_workers=8) as copy_pool, \
                ThreadPoolExecutor(max_workers=max(n_dec, 1)) as dec_pool:
            def submit(k, pos):
                if pos >= len(order):
                    return None
                return loader.submit(stage, k, pos, copy_pool) if threaded else None

            pending = submit(0, 0)
            pos = 0
            for batch_start in range(0, batch_stop, batch_size):
                batch_inds = list(range(batch_start, batch_start + batch_size))
                batch_cameras = cameras.get_subset_cameras(batch_inds)
                batch_pix2face = self.pix2face(
                    cameras=batch_cameras, mesh=mesh, render_img_scale=aggregate_img_scale, return_tensor=True,
                    **pix2face_kwargs,
                )
                if isinstance(batch_pix2face, np.ndarray):  # distortion applied on the host
                    batch_pix2face = self.backend._dev(batch_pix2face.astype(np.int32), torch.int32)
                for i in range(batch_pix2face.shape[0]):
                    if pending is None:  # not threaded: the view is fetched when it is consumed, like the reference does
                        pending = _Now(stage(pos & 1, pos, copy_pool))
                    host, n_channels, resize_to = pending.result()
                    k = pos & 1
                    pos += 1
                    pending = submit(pos & 1, pos)
                    if on_gpu:
                        dev_img = host.to(self.backend.device, non_blocking=True)
                        if host.is_pinned():
                            slot_free[k] = torch.cuda.Event()
                            slot_free[k].record(torch.cuda.current_stream(self.backend.device))
                    else:
                        dev_img = host
                    if resize_to is not None:  # get_image's own arithmetic, on the device (cameras.py:158-172)
                        dev_img = self.backend.resize_image(dev_img, resize_to[0], divide_by_255=resize_to[1])
                    if check_null_image and dev_img.is_floating_point() and not bool(torch.isfinite(dev_img).any()):
                        yield batch_start + i, batch_pix2face[i], None, n_channels
                        continue
                    yield batch_start + i, batch_pix2face[i], dev_img, n_channels

    def project_images(
        self,
        cameras: typing.Union[PhotogrammetryCamera, PhotogrammetryCameraSet],
        batch_size: int = 1,
        aggregate_img_scale: float = 1,
        check_null_image: bool = False,
        **pix2face_kwargs,
    ):
        """Generator: per-face projection of each camera's image, (F, C) float64 with NaN for unseen faces
        (reference: meshes.py:1944-2002).  Per view the LAST pixel (row-major) mapped to a face provides its value;
        with `neg1_is_last_face` background pixels address the last face exactly like numpy's index -1 does."""
        n_faces = self.faces.shape[0]
        loader_threads = pix2face_kwargs.pop("loader_threads", None)
        for _, ids, img, n_channels in self._iter_view_inputs(
            cameras, batch_size, aggregate_img_scale, check_null_image, pix2face_kwargs, loader_threads
        ):
            if img is None:
                yield np.full((n_faces, n_channels), fill_value=np.nan)
            else:
                yield _to_host(self.backend.project_view(ids, img, neg1_is_last_face=self.neg1_is_last_face))

    @staticmethod
    @contextlib.contextmanager
    def _decoded_cache_scope(cameras, spec):
        """`decoded_cache=` of project_images / aggregate_projected_images: for the duration of the call the camera set's photos
        and its segmentor's label files go through the decoded-input cache (utils/decoded_cache.py); None changes nothing."""
        if spec is None:
            yield
            return
        segmentor = getattr(cameras, "segmentor", None)
        base = getattr(cameras, "base_camera_set", cameras)
        before_seg = getattr(segmentor, "decoded_cache", None) if hasattr(segmentor, "decoded_cache") else None
        before_cams = [getattr(cam, "decoded_cache", None) for cam in getattr(base, "cameras", [])]
        try:
            if hasattr(segmentor, "decoded_cache"):
                segmentor.decoded_cache = spec
            for cam in getattr(base, "cameras", []):
                cam.decoded_cache = spec
            yield
        finally:
            if hasattr(segmentor, "decoded_cache"):
                segmentor.decoded_cache = before_seg
            for cam, old in zip(getattr(base, "cameras", []), before_cams):
                cam.decoded_cache = old

    # -- aggregate_projected_images ------------------------------------------------------------------------------
    def aggregate_projected_images(
        self,
        cameras: typing.Union[PhotogrammetryCamera, PhotogrammetryCameraSet],
        batch_size: int = 1,
        aggregate_img_scale: float = 1,
        return_all: bool = False,
        distributed: bool = False,
        decoded_cache=None,
        **kwargs,
    ):
        """`_aggregate_projected_images` (below: the reference's method) with one more opt-in keyword, `decoded_cache`: None
        (default) -- every pass decodes the label PNGs / photos it reads, like the reference --; True or a folder -- decoded
        inputs are kept as uncompressed `.npy` files keyed by (path, mtime, size, scale) under `CACHE_FOLDER/decoded` (the
        reference's cache root, constants.py:18; its own `save_to_cache` precedent: meshes.py:1759-1770, 1838-1840) or the
        folder, and the second and later passes memory-map them: the host side of a pass is bound by PNG / JPEG inflate
        otherwise (bench `io`)."""
        with self._decoded_cache_scope(cameras, decoded_cache):
            return self._aggregate_projected_images(cameras, batch_size=batch_size, aggregate_img_scale=aggregate_img_scale,
                                                    return_all=return_all, distributed=distributed, **kwargs)

    def _aggregate_projected_images(
        self,
        cameras: typing.Union[PhotogrammetryCamera, PhotogrammetryCameraSet],
        batch_size: int = 1,
        aggregate_img_scale: float = 1,
        return_all: bool = False,
        distributed: bool = False,
        **kwargs,
    ):
        """Average the images of many cameras per face (reference: meshes.py:2004-2084).

        Returns `(average (F,C), {"projection_counts": (F,), "summed_projections": (F,C)[, "all_projections"]})`,
        all float64, NaN for faces no view observed.

        Fast path: when `cameras` can supply class-index images (`get_label_index_image`, e.g. a
        `SegmentorPhotogrammetryCameraSet`) and `return_all` is False, face ids and vote histograms stay on the GPU:
        per view one winner pass + one vote pass in uint32 (bit-exact, order independent), one finalize at the end.
        `distributed=True` (inside an initialised torch.distributed job): this rank handles views rank::world and the
        per-face votes (index labels: [F x (C+1)] int32, bit-identical for every world size) or sums + counts (float
        images: [F x (C+1)] float64) are added with ONE all-reduce (RCCL over xGMI) before finalising.
        """
        from geograypher_amd import distributed as dist_utils

        if len(cameras) == 0 or batch_size > len(cameras):
            raise IndexError("list index out of range")  # what the reference's get_subset_cameras raises here
        rank, world = dist_utils.rank_world() if distributed else (0, 1)
        torch = _torch()
        n_faces = self.faces.shape[0]
        check_null_image = bool(kwargs.pop("check_null_image", False))

        batch_stop = max(len(cameras) - batch_size + 1, 1)
        view_inds = [i for s in range(0, batch_stop, batch_size) for i in range(s, s + batch_size)]

        loader_threads = kwargs.pop("loader_threads", None)
        unknown = [k for k in kwargs if k not in _PIX2FACE_KWARGS]
        if unknown:
            raise TypeError(f"pix2face() got an unexpected keyword argument {unknown[0]!r}")
        # The fused path rasterizes with the plain pinhole records.  Whatever else pix2face honours -- a distortion set
        # (unless explicitly switched off), an explicit mesh -- goes through pix2face itself: ids per chunk from
        # `pix2face(return_tensor=True, **kwargs)`, then the unfused winner + vote kernels.
        wants_warp = kwargs.get("distortion_set") is not None and kwargs.get("apply_distortion", True)
        fused_ok = not wants_warp and kwargs.get("mesh") is None
        label_fn = getattr(cameras, "get_label_index_image", None) if not return_all else None
        if label_fn is not None and int(cameras.n_image_channels()) > 255:
            label_fn = None  # class indices do not fit the uint8 label images of the fast path
        first_label = label_fn(view_inds[0], aggregate_img_scale) if label_fn is not None else None

        if first_label is not None:
            # ---- index-label fast path: uint32 votes on device --------------------------------------------------------
            my_inds = view_inds[rank::world]
            mesh = self.get_mesh_in_cameras_coords(cameras)
            C = int(cameras.n_image_channels())
            backends = self.backends
            labels_known = {view_inds[0]: first_label}

            def run(k):
                return self._aggregate_label_views(backends[k], cameras, my_inds[k::len(backends)], mesh, C, label_fn, labels_known,
                                                   batch_size, aggregate_img_scale, loader_threads, fused_ok, kwargs,
                                                   progress=(k == 0), own_stream=len(backends) > 1)

            if len(backends) == 1:
                votes, counts = run(0)
            else:
                # one host thread per device (the C calls and torch's copies release the interpreter lock); the partial votes
                # come home to the first device -- peer copies -- and are added there: integer sums, so the result is the
                # single-device one bit for bit, whatever the number of devices
                from concurrent.futures import ThreadPoolExecutor

                with ThreadPoolExecutor(max_workers=len(backends)) as dev_pool:
                    parts = list(dev_pool.map(run, range(len(backends))))
                votes, counts = parts[0]
                for v_k, c_k in parts[1:]:
                    votes += v_k.to(votes.device)
                    counts += c_k.to(counts.device)
            if distributed and world > 1:
                dist_utils.all_reduce_votes(votes, counts)
            avg, summed, cnt = self.backend.finalize_votes(votes, counts)
            return _to_host(avg), {
                "projection_counts": _to_host(cnt),
                "summed_projections": _to_host(summed),
            }

        # ---- general path: float images; nansum + finite-row counts accumulate on device (meshes.py:2057-2067) ----------
        single_view = len(view_inds) == 1
        shard = distributed and world > 1 and not single_view
        if shard and return_all:
            raise NotImplementedError("return_all keeps every view's projection: not available with distributed=True")
        all_projections = [] if return_all else None
        sums = counts = first = None
        n_channels = None
        multi = len(self.backends) > 1 and not return_all and not single_view
        if multi:
            # devices=[...]: this rank's views dealt round-robin to the devices, each running the per-view recurrence over its own
            # views on a host thread of its own; the partial sums and counts are added on the first device.  The float contract
            # is that of views sharded over processes (include/geograster.h, gr_project_values_f64): counts exact, sums of finite
            # inputs within 1e-12 relative of the serial result
            import copy
            from concurrent.futures import ThreadPoolExecutor

            mine = view_inds[rank::world] if shard else view_inds
            self.get_mesh_in_cameras_coords(cameras)   # the local mesh, cached before the workers ask for it

            def run(k):
                inds_k = mine[k::len(self.backends)]
                if not inds_k:
                    return None
                worker = copy.copy(self)   # same arrays, caches and upload table; its own backend
                worker._backend, worker._backends = self.backends[k], [self.backends[k]]
                bk = self.backends[k]
                ctx = contextlib.nullcontext()
                if bk.device.type == "cuda":
                    torch.cuda.set_device(bk.device)
                    ctx = torch.cuda.stream(torch.cuda.Stream(bk.device))
                with ctx:
                    s_k = c_k = None
                    nch = None
                    for _, ids, img, nch in worker._iter_view_inputs(cameras.get_subset_cameras(inds_k), 1, aggregate_img_scale,
                                                                     check_null_image, dict(kwargs), loader_threads):
                        if s_k is None:
                            s_k = torch.zeros((n_faces, nch), dtype=torch.float64, device=bk.device)
                            c_k = torch.zeros((n_faces,), dtype=torch.int32, device=bk.device)
                        if img is not None:
                            bk.project_values(ids, img, s_k, c_k, neg1_is_last_face=self.neg1_is_last_face)
                        else:
                            torch.nan_to_num_(s_k, nan=0.0, posinf=float("inf"), neginf=float("-inf"))
                    if bk.device.type == "cuda":
                        torch.cuda.current_stream(bk.device).synchronize()
                return s_k, c_k, nch

            with ThreadPoolExecutor(max_workers=len(self.backends)) as dev_pool:
                parts = [p for p in dev_pool.map(run, range(len(self.backends))) if p is not None and p[0] is not None]
            for s_k, c_k, nch in parts:
                n_channels = nch
                if sums is None:
                    sums, counts = s_k.to(self.backend.device), c_k.to(self.backend.device)
                else:
                    sums += s_k.to(sums.device)
                    counts += c_k.to(counts.device)
            gen, total = iter(()), 0
        elif shard:  # this rank's views; the per-view arithmetic does not depend on the other views
            my_cams = cameras.get_subset_cameras(view_inds[rank::world])
            gen = self._iter_view_inputs(my_cams, 1, aggregate_img_scale, check_null_image, kwargs, loader_threads) if len(my_cams) else iter(())
            total = len(my_cams)
        else:
            gen = self._iter_view_inputs(cameras, batch_size, aggregate_img_scale, check_null_image, kwargs, loader_threads)
            total = len(cameras)
        for _, ids, img, n_channels in (gen if multi else tqdm(gen, total=total, desc="Aggregating projected viewpoints")):
            if sums is None:
                sums = torch.zeros((n_faces, n_channels), dtype=torch.float64, device=self.backend.device)
                counts = torch.zeros((n_faces,), dtype=torch.int32, device=self.backend.device)
            if return_all or single_view:
                if img is None:
                    proj = np.full((n_faces, n_channels), fill_value=np.nan)
                else:
                    proj = _to_host(self.backend.project_view(ids, img, neg1_is_last_face=self.neg1_is_last_face))
                if return_all:
                    all_projections.append(proj)
                first = proj if first is None else first
            if img is not None:
                self.backend.project_values(ids, img, sums, counts, neg1_is_last_face=self.neg1_is_last_face)
            else:
                # a skipped (null) image still passes through np.nansum([summed, all-NaN projection]) in the reference
                # (meshes.py:2060-2062), which drops a NaN of the running sum (+inf met -inf) like every other view does
                torch.nan_to_num_(sums, nan=0.0, posinf=float("inf"), neginf=float("-inf"))
        if shard:
            if sums is None:  # a rank without views still takes part in the collective
                n_channels = int(np.asarray(cameras.get_image_by_index(view_inds[0], aggregate_img_scale)).reshape(
                    cameras.cameras[view_inds[0]].get_image_size(aggregate_img_scale) + (-1,)).shape[-1])
                sums = torch.zeros((n_faces, n_channels), dtype=torch.float64, device=self.backend.device)
                counts = torch.zeros((n_faces,), dtype=torch.int32, device=self.backend.device)
            dist_utils.all_reduce_sums(sums, counts)
        avg, summed, cnt = self.backend.finalize_sums(sums, counts)
        avg, summed, cnt = _to_host(avg), _to_host(summed), _to_host(cnt)
        if single_view:
            # the reference keeps the first projection as is (meshes.py:2057-2058): a NaN channel of a seen face survives
            summed = first.astype(float)
            summed[cnt == 0] = np.nan
            with np.errstate(divide="ignore", invalid="ignore"):
                avg = np.divide(summed, np.expand_dims(cnt, 1))
        info = {"projection_counts": cnt, "summed_projections": summed}
        if return_all:
            info["all_projections"] = all_projections
        return avg, info

    def _aggregate_label_views(self, backend, cameras, inds, mesh, C, label_fn, labels_known, batch_size, aggregate_img_scale,
                               loader_threads, fused_ok, kwargs, progress=True, own_stream=False):
        """The index-label fast path of aggregate_projected_images for the views `inds` on ONE backend: returns that device's
        (votes (F,C), counts (F,)) int32 tensors.  Called once for a single-device mesh, once per device -- each on a host
        thread of its own -- for `devices=[...]`."""
        torch = _torch()
        import os
        from concurrent.futures import ThreadPoolExecutor

        on_gpu = backend.device.type == "cuda"
        stream_ctx = contextlib.nullcontext()
        if on_gpu:
            torch.cuda.set_device(backend.device)   # the current device is a property of the host thread
            if own_stream:   # contexts that share a device would otherwise queue on its one default stream
                stream_ctx = torch.cuda.stream(torch.cuda.Stream(backend.device))
        with stream_ctx:
            self._ensure_uploaded(mesh, backend)
            votes, counts = backend.new_vote_buffers(C)
            if len(inds) == 0:
                return votes, counts
            # launch groups of up to 32 views; a short run is still cut into at least four chunks, so that the loader stages
            # chunk i + 1 while chunk i crosses the link (16 views in ONE chunk ran staging, copy and kernels back to back)
            chunk = max(int(batch_size), min(32, max(1, -(-len(inds) // 4))))
            chunks = [inds[c0 : c0 + chunk] for c0 in range(0, len(inds), chunk)]

            # Input pipeline (row f4).  The label images of a chunk are decoded straight into ONE pinned (n,h,w) uint8
            # staging tensor (no stack / pin copies), by `loader_threads` workers when the segmentor says its lookups
            # are thread safe (file look-ups and in-memory arrays are; an arbitrary model may not be), while the GPU
            # works on the previous chunk.
            thread_safe = bool(getattr(getattr(cameras, "segmentor", None), "thread_safe_lookup", False))
            n_workers = int(loader_threads if loader_threads is not None else (min(16, os.cpu_count() or 1) if thread_safe else 1))
            # the label images must have the size the camera records are built for (the reference fails with a shape
            # error in `textured_faces[flat_pix2face] = flat_img` otherwise, meshes.py:1998-2001)
            h0, w0 = cameras.cameras[inds[0]].get_image_size(aggregate_img_scale)

            def check_shape(shape, i):
                if tuple(shape[:2]) != (h0, w0):
                    raise ValueError(
                        f"label image of view {i} has shape {tuple(shape[:2])}, but the camera renders {(h0, w0)} at "
                        f"aggregate_img_scale={aggregate_img_scale}"
                    )

            def one_label(i):
                return labels_known[i] if i in labels_known else label_fn(i, aggregate_img_scale)

            probe = next(iter(labels_known.values()))

            def load_chunk(chunk_inds, img_pool):
                if isinstance(probe, torch.Tensor):  # the segmentor already produces tensors
                    labs = [one_label(i) for i in chunk_inds]
                    for i, lab in zip(chunk_inds, labs):
                        check_shape(lab.shape, i)
                    labs = [torch.where((lab < 0) | (lab > 255), torch.full_like(lab, 255), lab)
                            if lab.dtype != torch.uint8 else lab for lab in labs]
                    return torch.stack([lab.to(backend.device, torch.uint8) for lab in labs], dim=0)
                stage = torch.empty((len(chunk_inds), h0, w0), dtype=torch.uint8, pin_memory=on_gpu)
                view = stage.numpy()

                def fill(k):
                    lab = np.asarray(one_label(chunk_inds[k]))
                    check_shape(lab.shape, chunk_inds[k])
                    if lab.dtype != np.uint8:  # an index outside [0, 255] is no class: the ignore value, not a wrapped class
                        lab = np.where((lab < 0) | (lab > 255), 255, lab)
                    view[k] = lab  # casts to uint8 while copying

                list(img_pool.map(fill, range(len(chunk_inds))))
                return stage

            steps = range(len(chunks))
            if progress:
                steps = tqdm(steps, total=len(chunks), desc="Aggregating projected viewpoints")
            with ThreadPoolExecutor(max_workers=1) as pool, ThreadPoolExecutor(max_workers=max(n_workers, 1)) as img_pool:
                pending = pool.submit(load_chunk, chunks[0], img_pool)
                for ci in steps:
                    lab = pending.result()
                    pending = pool.submit(load_chunk, chunks[ci + 1], img_pool) if ci + 1 < len(chunks) else None
                    # the chunk's cameras only: a subset of the segmentor wrapper would deep-copy the segmentor with it
                    # (segmentor.py:49-55) -- every in-memory label image of an ArrayLabelSegmentor, per chunk
                    sub = getattr(cameras, "base_camera_set", cameras).get_subset_cameras(chunks[ci])
                    if lab.device.type != backend.device.type:
                        lab = lab.to(backend.device, non_blocking=True)
                    if fused_ok:
                        # fused: face ids stay in the rasterizer's LDS tiles, only per-face winners reach HBM
                        records, _ = self._raster_records(sub, mesh, aggregate_img_scale, backend=backend, **_raster_kwargs(kwargs))
                        backend.raster_project_labels(records, lab, C, votes, counts, neg1_is_last_face=self.neg1_is_last_face)
                    else:
                        p2f_kwargs = {k: v for k, v in kwargs.items() if k in _PIX2FACE_KWARGS}
                        p2f_kwargs.setdefault("mesh", mesh)
                        ids = self.pix2face(cameras=sub, render_img_scale=aggregate_img_scale, return_tensor=True, **p2f_kwargs)
                        if isinstance(ids, np.ndarray):
                            ids = backend._dev(ids.astype(np.int32), torch.int32)
                        elif ids.device != backend.device:
                            ids = ids.to(backend.device)
                        if tuple(ids.shape[-2:]) != (h0, w0):
                            raise ValueError(f"pix2face returned {tuple(ids.shape[-2:])} ids for {(h0, w0)} label images")
                        backend.project_labels(ids, lab, C, votes, counts, neg1_is_last_face=self.neg1_is_last_face)
            if on_gpu and own_stream:
                torch.cuda.current_stream(backend.device).synchronize()   # the partial is complete when the thread hands it back
        return votes, counts

    # the north star's name for the same method
    aggregate_viewpoints = aggregate_projected_images

    # -- save_renders ----------------------------------------------------------------
    def save_IDs_to_labels(self, savepath: PATH_TYPE):
        """reference: meshes.py:1081-1108"""
        import json

        Path(savepath).parent.mkdir(parents=True, exist_ok=True)
        if self.is_discrete_texture():
            self.logger.info(f"Saving IDs_to_labels to {str(savepath)}")
            try:
                with open(savepath, "w") as outfile_h:
                    json.dump(self.get_IDs_to_labels(), outfile_h, ensure_ascii=False, indent=4, default=str)
            except Exception:
                self.logger.warning("Could not serialize IDs_to_labels due to JSON error")
        else:
            self.logger.warning("non-discrete texture, not saving classes")

    @staticmethod
    def _resize_map(src_hw, dst_hw):
        """(2, H, W) sampling map of skimage.transform.resize: destination pixel centres mapped into the source grid,
        mirrored at the borders (skimage's default mode "reflect")."""
        (h, w), (H, W) = src_hw, dst_hw
        r = (np.arange(H) + 0.5) * (h / H) - 0.5
        c = (np.arange(W) + 0.5) * (w / W) - 0.5
        r = np.where(r < 0, -r, np.where(r > h - 1, 2 * (h - 1) - r, r))
        c = np.where(c < 0, -c, np.where(c > w - 1, 2 * (w - 1) - c, c))
        rr, cc = np.meshgrid(r, c, indexing="ij")
        return np.stack([rr, cc], axis=0)

    def save_renders(
        self,
        camera_set: PhotogrammetryCameraSet,
        render_image_scale=1.0,
        output_folder: PATH_TYPE = Path(VIS_FOLDER, "renders"),
        make_composites: bool = False,
        save_native_resolution: bool = False,
        cast_to_uint8: bool = True,
        save_as_npy: bool = False,
        uint8_value_for_null_texture: np.uint8 = NULL_TEXTURE_INT_VALUE,
        **render_kwargs,
    ):
        """Render the face texture from every camera and save one file per image (reference: meshes.py:2248-2397).

        Same arguments, same on-disk layout (`<output_folder>/<image path relative to camera_set.image_folder>` as
        deflate-compressed .tif, or .npy with `save_as_npy`; `IDs_to_labels.json` for discrete textures).  The
        post-processing of meshes.py:2312-2349 runs on the device: with `cast_to_uint8` the texture gather, the
        null/out-of-range masking and the uint8 cast are ONE kernel (`gr_gather_texture_u8`), so a 4000x3000 view
        leaves the GPU as 12 MB instead of the reference's 96 MB float64 image; the optional native-resolution
        upsampling (nearest for discrete textures, bilinear otherwise) is the warp kernel with a resize map.
        Like the reference this renders with `distortion_set=camera_set` (a camera set without a distortion model
        needs `apply_distortion=False`).  `make_composites` (matplotlib visualisation) is outside the projection path.

        Extra keywords (not forwarded to pix2face): `writer_threads` (default min(16, cores)) host threads that deflate
        and write while the GPU renders the next views -- results travel through a ring of pinned buffers, one
        asynchronous copy per view; `views_per_group` (default 8) views rasterized per device call.
        """
        from concurrent.futures import ThreadPoolExecutor

        from PIL import Image

        from geograypher_amd.utils.tiff import write_tiff_deflate

        if make_composites:
            raise NotImplementedError("composite visualisations are outside the projection path (utils/visualization.py)")
        torch = _torch()
        writer_threads = int(render_kwargs.pop("writer_threads", min(16, __import__("os").cpu_count() or 1)))
        views_per_group = int(render_kwargs.pop("views_per_group", 8))
        output_folder = Path(output_folder)
        output_folder.mkdir(parents=True, exist_ok=True)
        self.logger.info(f"Saving renders to {output_folder}")
        self.save_IDs_to_labels(Path(output_folder, "IDs_to_labels.json"))

        face_texture = np.asarray(
            self.get_texture(request_vertex_texture=False, try_verts_faces_conversion=True), dtype=np.float64
        )
        tex_dev = self.backend._dev(face_texture, torch.float64)
        mesh = self.get_mesh_in_cameras_coords(camera_set)
        discrete = self.is_discrete_texture()
        resize_maps = {}
        render_kwargs = dict(render_kwargs)
        render_kwargs.setdefault("distortion_set", camera_set)
        on_gpu = self.backend.device.type == "cuda"

        # Output paths first: a camera outside the image folder fails before any rendering, as in the reference
        output_files = []
        for camera in camera_set.cameras:
            try:
                camera_filename = Path(camera.get_image_filename()).relative_to(camera_set.image_folder)
            except (ValueError, TypeError):
                raise ValueError(
                    "Tried to find the relative path of the camera path"
                    f" ({camera.get_image_filename()}) inside of the camera set image"
                    f" folder ({camera_set.image_folder}), but failed. The tool being called"
                    " may have an 'original_image_folder' argument, which could be used to"
                    " delete the initial, mismatched portion of the camera path."
                )
            output_files.append(Path(output_folder, camera_filename))

        def write_one(host, event, output_filename):
            """Writer thread: wait for the view's copy to land in its pinned slot, finish the reference's post-processing
            (meshes.py:2325-2349) where it was not fused into the gather kernel, compress and write."""
            if event is not None:
                event.synchronize()
            rendered = host.numpy() if hasattr(host, "numpy") else host
            if rendered.dtype != np.uint8 and cast_to_uint8:
                rendered = rendered.copy()
                mask = np.logical_or.reduce([rendered < 0, rendered > 255, np.logical_not(np.isfinite(rendered))])
                rendered[mask] = uint8_value_for_null_texture
                rendered = rendered.astype(np.uint8)
            rendered = np.squeeze(rendered)
            if rendered.ndim == 3:
                rendered = rendered[..., :3]
            output_filename.parent.mkdir(parents=True, exist_ok=True)
            if save_as_npy is True:
                np.save(str(output_filename.with_suffix(".npy")), rendered)
            else:
                if cast_to_uint8 is False:
                    with np.errstate(invalid="ignore"):
                        if np.nanmax(rendered) <= np.iinfo(np.uint16).max:
                            rendered = rendered.astype(np.uint16)
                        else:
                            rendered = rendered.astype(np.uint32)
                target = output_filename.with_suffix(".tif")
                if rendered.dtype in (np.uint8, np.uint16, np.uint32) and (rendered.ndim == 2 or rendered.shape[2] == 3):
                    write_tiff_deflate(target, rendered)  # zlib releases the interpreter lock: the writers run in parallel
                else:
                    Image.fromarray(rendered).save(str(target), compression="tiff_deflate")

        # Pipeline (row f2): the GPU rasterizes / gathers a group of views and copies each result into a slot of a pinned
        # ring while `writer_threads` host threads deflate and write the views before it.  A slot is reused only after
        # the writer that read it last is done, so at most ring_slots results are in flight.
        ring_slots = max(2 * writer_threads, 2)
        ring_budget = _PINNED_LIMIT_BYTES  # pinned bytes the ring may hold: float64 (h, w, C) frames are 8 C bytes per pixel
        ring = {}
        slot_writer = {}
        futures = []
        n = len(camera_set)
        with ThreadPoolExecutor(max_workers=max(writer_threads, 1)) as pool:
            for g0 in tqdm(range(0, n, views_per_group), total=(n + views_per_group - 1) // views_per_group,
                           desc="Computing and saving renders"):
                group = list(range(g0, min(g0 + views_per_group, n)))
                sub = camera_set.get_subset_cameras(group) if len(group) > 1 else camera_set[g0]
                ids_group = self.pix2face(cameras=sub, mesh=mesh, render_img_scale=render_image_scale, return_tensor=True,
                                          **render_kwargs)
                if isinstance(ids_group, np.ndarray):
                    ids_group = self.backend._dev(ids_group.astype(np.int32), torch.int32)
                if ids_group.ndim == 2:
                    ids_group = ids_group[None]
                for k, i in enumerate(group):
                    camera = camera_set.cameras[i]
                    ids = ids_group[k]
                    native = save_native_resolution and render_image_scale != 1
                    if native:
                        key = (tuple(ids.shape), tuple(camera.get_image_size()))
                        if key not in resize_maps:
                            resize_maps[key] = self.backend.upload_map(self._resize_map(*key))
                    if cast_to_uint8 and (not native or discrete):
                        if native:  # nearest-neighbour upsampling commutes with the per-pixel gather: resize the ids
                            ids = self.backend.warp_image(ids, resize_maps[key], order=0, fill_value=-1)
                        rendered = self.backend.gather_texture_u8(ids, tex_dev, int(uint8_value_for_null_texture))
                    else:
                        rendered = self.backend.gather_texture(ids, tex_dev)  # (h, w, C) float64, NaN without a face
                        if native:
                            rendered = self.backend.warp_image(rendered, resize_maps[key], order=0 if discrete else 1,
                                                               fill_value=float("nan"))
                    frame_bytes = rendered.numel() * rendered.element_size() if hasattr(rendered, "numel") else rendered.nbytes
                    slots_now = max(2, min(ring_slots, ring_budget // max(frame_bytes, 1)))
                    slot = i % slots_now
                    if slot in slot_writer:
                        slot_writer[slot].result()  # the slot's previous view is on disk (a failed writer raises here)
                    if on_gpu and rendered.is_cuda:
                        skey = (slot, tuple(rendered.shape), rendered.dtype)
                        if ring.get(slot, (None,))[0] != skey[1:]:
                            ring.pop(slot, None)
                            try:
                                ring[slot] = (skey[1:], torch.empty(rendered.shape, dtype=rendered.dtype, pin_memory=True))
                            except RuntimeError:  # the host refuses more pinned memory: this view takes a blocking pageable copy
                                ring[slot] = None
                        if ring[slot] is not None:
                            host = ring[slot][1]
                            host.copy_(rendered, non_blocking=True)
                            event = torch.cuda.Event()
                            event.record(torch.cuda.current_stream(rendered.device))
                        else:
                            del ring[slot]
                            host, event = rendered.cpu(), None
                    else:
                        host, event = rendered, None
                    slot_writer[slot] = pool.submit(write_one, host, event, output_files[i])
                    futures.append(slot_writer[slot])
            for f in futures:
                f.result()


_FUSED_KWARGS = ("near", "principal_point", "focal_scaling")
_PIX2FACE_KWARGS = _FUSED_KWARGS + ("mesh", "save_to_cache", "cache_folder", "distortion_set", "apply_distortion")


def _raster_kwargs(kwargs: dict) -> dict:
    return {k: kwargs[k] for k in _FUSED_KWARGS if k in kwargs}


# device -> host through pinned memory.  A pageable destination moves at ~10 GB/s on the MI355X host link, a pinned one
# at the link rate; torch's caching host allocator hands the same pinned blocks back once earlier results are dropped.
_PINNED_LIMIT_BYTES = 8 << 30


def _ids_to_host_int64(ids, views_per_step: int = 4, threads: typing.Optional[int] = None) -> np.ndarray:
    """(n, h, w) int32 ids on the device -> int64 numpy, as the reference returns them (meshes.py:1804).  The PCIe link is the
    bottleneck of this call, so the ids cross it as int32 (half the bytes of a device-side widening) into a pinned ring,
    and host threads widen slot k into the result while slot k+1 is on the wire."""
    torch = _torch()
    if not ids.is_cuda or ids.dim() != 3 or ids.numel() < (1 << 21):
        return _to_host(ids.to(torch.int64))
    import os
    from concurrent.futures import ThreadPoolExecutor

    n, h, w = ids.shape
    ids = ids.contiguous()
    if threads is None:
        threads = max(1, min(16, os.cpu_count() or 1))
    step = max(1, min(int(views_per_step), n))
    try:
        ring = [torch.empty((step, h, w), dtype=torch.int32, pin_memory=True) for _ in range(2)]
    except RuntimeError:  # pinned memory exhausted
        return _to_host(ids.to(torch.int64))
    out = np.empty((n, h, w), dtype=np.int64)
    stream = torch.cuda.current_stream(ids.device)
    events = [torch.cuda.Event() for _ in range(2)]
    rows = max(1, (step * h + threads - 1) // threads)

    def widen(dst2d, src2d, r0, r1):
        np.copyto(dst2d[r0:r1], src2d[r0:r1], casting="safe")  # numpy releases the interpreter lock for this loop

    with ThreadPoolExecutor(max_workers=threads) as pool:
        jobs = [[], []]
        starts = list(range(0, n, step))
        for k, v0 in enumerate(starts):
            slot = k & 1
            for j in jobs[slot]:  # the slot's previous contents have been widened
                j.result()
            m = min(step, n - v0)
            ring[slot][:m].copy_(ids[v0 : v0 + m], non_blocking=True)
            events[slot].record(stream)
            if k >= 1:  # widen the previous slot while this one is on the wire
                ps = (k - 1) & 1
                pv0 = starts[k - 1]
                pm = min(step, n - pv0)
                events[ps].synchronize()
                src = ring[ps][:pm].numpy().reshape(pm * h, w)
                dst = out[pv0 : pv0 + pm].reshape(pm * h, w)
                jobs[ps] = [pool.submit(widen, dst, src, r0, min(r0 + rows, pm * h)) for r0 in range(0, pm * h, rows)]
        ls = (len(starts) - 1) & 1
        lv0 = starts[-1]
        lm = min(step, n - lv0)
        events[ls].synchronize()
        src = ring[ls][:lm].numpy().reshape(lm * h, w)
        dst = out[lv0 : lv0 + lm].reshape(lm * h, w)
        jobs[ls] = [pool.submit(widen, dst, src, r0, min(r0 + rows, lm * h)) for r0 in range(0, lm * h, rows)]
        for js in jobs:
            for j in js:
                j.result()
    return out


def _to_host(t) -> np.ndarray:
    """numpy copy of a tensor.  Device tensors are copied into a pinned staging tensor and returned as a numpy view of
    it (kept alive through the array's base); tensors that are already on the host are returned as they are."""
    torch = _torch()
    if not t.is_cuda:
        return t.numpy()
    nbytes = t.numel() * t.element_size()
    if nbytes == 0 or nbytes > _PINNED_LIMIT_BYTES:
        return t.cpu().numpy()
    try:
        host = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
    except RuntimeError:  # pinned memory exhausted: pageable copy
        return t.cpu().numpy()
    host.copy_(t, non_blocking=True)
    torch.cuda.current_stream(t.device).synchronize()
    return host.numpy()


def _torch():
    import torch

    return torch
