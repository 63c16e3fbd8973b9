"""ctypes binding of libgeograster (include/geograster.h) and the device backend the mesh class drives.

The library is the product: there is no CPU fallback.  Importing this module is cheap; the first use of
`HipRaster` loads `csrc/libgeograster.so` and raises `RuntimeError` when the shared object or a GPU is missing.
PyTorch-ROCm supplies device memory (tensors), streams and `torch.distributed`; no torch type crosses the C ABI,
only `tensor.data_ptr()` and the raw `hipStream_t` of the current torch stream.
"""
from __future__ import annotations

import ctypes
import threading
from pathlib import Path
from typing import Optional

import numpy as np

import os as _os

# GEOGRAYPHER_AMD_LIB: a diagnostic build of the same library (geograypher_amd.build.build_variant) instead of the product's
_LIB_PATH = Path(_os.environ.get("GEOGRAYPHER_AMD_LIB") or Path(__file__).resolve().parent / "csrc" / "libgeograster.so")
_lib = None

GR_OK = 0
GR_EOVERFLOW = -6
GR_FLAG_NEG1_IS_LAST_FACE = 1
GR_FLAG_DEFER_CHECK = 2
GR_CAM_FLOATS = 16

# every symbol include/geograster.h declares (tests check that the library exports each of them)
EXPORTED_SYMBOLS = (
    "gr_version",
    "gr_ctx_create",
    "gr_ctx_destroy",
    "gr_last_error",
    "gr_set_profiling",
    "gr_set_option",
    "gr_learned_cache_file",
    "gr_learned_cache_clear",
    "gr_get_stage_times",
    "gr_mesh_upload",
    "gr_raster_face_ids",
    "gr_raster_status",
    "gr_gather_texture_f64",
    "gr_project_labels_u8",
    "gr_project_values_f64",
    "gr_project_view_f64",
    "gr_raster_project_labels_u8",
    "gr_gather_texture_u8",
    "gr_project_index_pairs",
    "gr_count_pairs",
    "gr_warp_nearest_i32",
    "gr_warp_f64",
    "gr_invert_distortion_f64",
    "gr_resize_image_f64",
    "gr_finalize_votes",
    "gr_finalize_sums_f64",
    "gr_argmax_nonzero_f64",
)


class StageTimes(ctypes.Structure):
    _fields_ = [
        ("setup_ms", ctypes.c_float),
        ("scan_ms", ctypes.c_float),
        ("fill_ms", ctypes.c_float),
        ("raster_ms", ctypes.c_float),
        ("project_ms", ctypes.c_float),
        ("vote_ms", ctypes.c_float),
        ("gather_ms", ctypes.c_float),
        ("raster_launches", ctypes.c_int32),
        ("views", ctypes.c_int32),
    ]

    def as_dict(self):
        return {name: getattr(self, name) for name, _ in self._fields_}


class RasterStats(ctypes.Structure):
    _fields_ = [
        ("records", ctypes.c_int64),
        ("entries", ctypes.c_int64),
        ("max_entries", ctypes.c_int64),
        ("entry_cap", ctypes.c_int64),
        ("overflow", ctypes.c_int32),
        ("views_done", ctypes.c_int32),
        ("blocks", ctypes.c_int64),
        ("chunk_visits", ctypes.c_int64),
        ("rebinned_groups", ctypes.c_int64),
    ]

    def as_dict(self):
        return {name: getattr(self, name) for name, _ in self._fields_}


def library_path() -> Path:
    return _LIB_PATH


def load_library() -> ctypes.CDLL:
    """Load libgeograster.so (built in-tree by `__graft_entry__.build()` / `geograypher_amd.build`)."""
    global _lib
    if _lib is not None:
        return _lib
    if not _LIB_PATH.is_file():
        raise RuntimeError(
            f"HIP extension missing: {_LIB_PATH} not found. Build it with "
            "`python -c 'import __graft_entry__ as g; g.build()'` (hipcc --offload-arch=gfx950). "
            "geograypher_amd has no CPU fallback for the projection path."
        )
    # PyTorch-ROCm ships its own HIP runtime.  It has to be in the process BEFORE this library pulls in the system's: with the
    # library loaded first (e.g. __graft_entry__.build() and smoke() in one process) the later `import torch` leaves the
    # library's runtime without devices and gr_ctx_create answers GR_ENODEVICE.
    _torch()
    lib = ctypes.CDLL(str(_LIB_PATH))
    vp, i32, i64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64
    lib.gr_version.restype = i32
    lib.gr_version.argtypes = []
    lib.gr_ctx_create.restype = i32
    lib.gr_ctx_create.argtypes = [i32, ctypes.POINTER(vp)]
    lib.gr_ctx_destroy.restype = i32
    lib.gr_ctx_destroy.argtypes = [vp]
    lib.gr_last_error.restype = ctypes.c_char_p
    lib.gr_last_error.argtypes = [vp]
    lib.gr_set_profiling.restype = i32
    lib.gr_set_profiling.argtypes = [vp, i32]
    lib.gr_set_option.restype = i32
    lib.gr_set_option.argtypes = [vp, i32, i32]
    lib.gr_get_stage_times.restype = i32
    lib.gr_get_stage_times.argtypes = [vp, ctypes.POINTER(StageTimes)]
    lib.gr_mesh_upload.restype = i32
    lib.gr_mesh_upload.argtypes = [vp, vp, vp, i64, i64, vp]
    lib.gr_raster_face_ids.restype = i32
    lib.gr_raster_face_ids.argtypes = [vp, vp, i32, i32, i32, vp, vp, vp]
    lib.gr_raster_status.restype = i32
    lib.gr_raster_status.argtypes = [vp, ctypes.POINTER(RasterStats)]
    lib.gr_gather_texture_f64.restype = i32
    lib.gr_gather_texture_f64.argtypes = [vp, vp, i64, vp, i64, i32, vp, vp]
    lib.gr_project_labels_u8.restype = i32
    lib.gr_project_labels_u8.argtypes = [vp, vp, vp, i32, i32, i32, i32, vp, vp, i32, vp]
    lib.gr_project_values_f64.restype = i32
    lib.gr_project_values_f64.argtypes = [vp, vp, vp, i32, i32, i32, i32, vp, vp, i32, vp]
    lib.gr_project_view_f64.restype = i32
    lib.gr_project_view_f64.argtypes = [vp, vp, vp, i32, i32, i32, vp, i32, vp]
    lib.gr_raster_project_labels_u8.restype = i32
    lib.gr_raster_project_labels_u8.argtypes = [vp, vp, vp, i32, i32, i32, i32, vp, vp, vp, i32, vp]
    f64 = ctypes.c_double
    lib.gr_gather_texture_u8.restype = i32
    lib.gr_gather_texture_u8.argtypes = [vp, vp, i64, vp, i64, i32, i32, vp, vp]
    lib.gr_project_index_pairs.restype = i32
    lib.gr_project_index_pairs.argtypes = [vp, vp, vp, i32, i32, i32, i64, vp, vp, i64, vp, i32, vp]
    lib.gr_count_pairs.restype = i32
    lib.gr_count_pairs.argtypes = [vp, vp, i64, vp, vp, ctypes.POINTER(ctypes.c_int64), vp]
    lib.gr_warp_nearest_i32.restype = i32
    lib.gr_warp_nearest_i32.argtypes = [vp, vp, i32, i32, vp, vp, i32, i32, ctypes.c_int32, i32, f64, f64, vp, vp]
    lib.gr_warp_f64.restype = i32
    lib.gr_warp_f64.argtypes = [vp, vp, i32, i32, i32, vp, vp, i32, i32, i32, f64, vp, vp]
    lib.gr_invert_distortion_f64.restype = i32
    lib.gr_invert_distortion_f64.argtypes = [vp, ctypes.POINTER(f64), i32, i32, f64, i32, f64, vp, vp, vp]
    lib.gr_resize_image_f64.restype = i32
    lib.gr_resize_image_f64.argtypes = [vp, vp, i32, i32, i32, i32, i32, i32, i32, vp, vp]
    lib.gr_learned_cache_file.restype = i32
    lib.gr_learned_cache_file.argtypes = [ctypes.c_char_p]
    lib.gr_learned_cache_clear.restype = i32
    lib.gr_learned_cache_clear.argtypes = []
    lib.gr_finalize_votes.restype = i32
    lib.gr_finalize_votes.argtypes = [vp, vp, vp, i64, i32, vp, vp, vp, vp]
    lib.gr_finalize_sums_f64.restype = i32
    lib.gr_finalize_sums_f64.argtypes = [vp, vp, vp, i64, i32, vp, vp, vp]
    lib.gr_argmax_nonzero_f64.restype = i32
    lib.gr_argmax_nonzero_f64.argtypes = [vp, vp, i64, i32, vp, vp]
    _lib = lib
    _attach_learned_cache(lib)
    return lib


def _attach_learned_cache(lib):
    """What overflowed raster calls taught the library (slots per tile, entry form per mesh and image size) is kept under
    the reference's CACHE_FOLDER (constants.py:18, the default of pix2face's `cache_folder`), so that a new process starts
    with bins that fit.  GEOGRAYPHER_AMD_CACHE=<dir> moves the file, GEOGRAYPHER_AMD_CACHE=off switches persistence off."""
    import os

    from geograypher_amd.constants import CACHE_FOLDER

    where = os.environ.get("GEOGRAYPHER_AMD_CACHE", str(CACHE_FOLDER))
    if where.lower() in ("off", "0", ""):
        return
    try:
        Path(where).mkdir(parents=True, exist_ok=True)
        lib.gr_learned_cache_file(str(Path(where, "geograster_learned.txt")).encode())
    except OSError:
        pass  # a read-only home: learn per process, as before


def _torch():
    import torch

    return torch


class _StatsAccumulator:
    """Statistics of a checked raster call over its attempts: an attempt that overflowed contributes the views it
    completed (records and entries are summed over the views a call processed, so they are scaled by the completed share);
    `max_entries` is the largest per-tile (single-pass) or per-view (exact binning) count any attempt saw."""

    def __init__(self):
        self.records = 0.0
        self.entries = 0.0
        self.blocks = 0.0
        self.chunk_visits = 0.0
        self.max_entries = 0
        self.views = 0
        self.rebinned = 0
        self.last = None

    def add(self, st: "RasterStats", n_views: int, partial: bool):
        done = int(st.views_done) if partial else n_views
        share = done / max(n_views, 1)
        self.records += st.records * share
        self.entries += st.entries * share
        self.blocks += st.blocks * share
        self.chunk_visits += st.chunk_visits * share
        self.max_entries = max(self.max_entries, int(st.max_entries))
        self.views += done
        self.rebinned += int(st.rebinned_groups)
        self.last = st

    def result(self) -> dict:
        d = self.last.as_dict()
        d.update(records=int(round(self.records)), entries=int(round(self.entries)), max_entries=self.max_entries,
                 views_done=self.views, blocks=int(round(self.blocks)), chunk_visits=int(round(self.chunk_visits)),
                 rebinned_groups=self.rebinned)
        return d


class _nullcontext:
    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


class PairAccumulator:
    """Device-resident pair keys of the sparse index aggregation (derived_meshes.py:470-520).  Every `add` appends the
    keys `face * n_classes + class` of its views to one device buffer through `gr_project_index_pairs` with
    GR_FLAG_DEFER_CHECK (no synchronisation, no per-view sort, no host round trip); `finish` runs ONE radix sort +
    run-length encode over everything (`gr_count_pairs`) and returns (pair_keys, multiplicities) as int64 numpy arrays.
    A view emits at most one pair per face, so the host knows an upper bound of the fill level without asking the
    device; when the buffer could overflow it is counted down to its distinct pairs (with their multiplicities, kept on
    the host) and reused."""

    def __init__(self, backend: "HipRaster", n_classes: int, counts, neg1_is_last_face: bool = True):
        torch = _torch()
        self.b = backend
        self.n_classes = int(n_classes)
        self.counts = counts
        self.flags = (GR_FLAG_NEG1_IS_LAST_FACE if neg1_is_last_face else 0) | GR_FLAG_DEFER_CHECK
        self.cap = max(8 * backend.n_faces, 1 << 20)
        self.keys = torch.empty((self.cap,), dtype=torch.int64, device=backend.device)
        self.key_count = torch.zeros((2,), dtype=torch.int64, device=backend.device)  # {pair count, error flag} (GR_FLAG_DEFER_CHECK)
        self.bound = 0          # upper bound of the pairs in the buffer
        self.parts = []         # (keys, multiplicities) of earlier compactions, host
        self.compactions = 0

    def add(self, ids, img):
        torch = _torch()
        b = self.b
        ids_t = b._dev(ids, torch.int32)
        img_t = b._dev(img, torch.float64)
        if ids_t.ndim == 2:
            ids_t, img_t = ids_t[None], img_t[None]
        if img_t.ndim == 4 and img_t.shape[-1] == 1:
            img_t = img_t[..., 0]
        if ids_t.shape != img_t.shape:
            raise ValueError(f"ids {tuple(ids_t.shape)} and index image {tuple(img_t.shape)} differ in shape")
        n, h, w = (int(x) for x in ids_t.shape)
        if n * b.n_faces > self.cap:
            for k in range(n):
                self.add(ids_t[k], img_t[k])
            return
        if self.bound + n * b.n_faces > self.cap:
            self._compact()
        with torch.cuda.device(b.device):
            rc = b.lib.gr_project_index_pairs(
                b._ctx, ids_t.data_ptr(), img_t.contiguous().data_ptr(), n, h, w, self.n_classes, self.counts.data_ptr(),
                self.keys.data_ptr(), self.cap, self.key_count.data_ptr(), self.flags, b._stream(),
            )
        b._check(rc, "gr_project_index_pairs")
        self.bound += n * b.n_faces

    def _compact(self):
        torch = _torch()
        b = self.b
        raw, bad = (int(x) for x in self.key_count.cpu().tolist())
        if bad:
            raise IndexError(f"gr_project_index_pairs: an image value is not a class index in [0, {self.n_classes})")
        if raw > 0:
            uniq = torch.empty((raw,), dtype=torch.int64, device=b.device)
            mult = torch.empty((raw,), dtype=torch.int32, device=b.device)
            n_unique = ctypes.c_int64(0)
            with torch.cuda.device(b.device):
                rc = b.lib.gr_count_pairs(b._ctx, self.keys.data_ptr(), raw, uniq.data_ptr(), mult.data_ptr(),
                                          ctypes.byref(n_unique), b._stream())
            b._check(rc, "gr_count_pairs")
            k = int(n_unique.value)
            self.parts.append((uniq[:k].cpu().numpy(), mult[:k].cpu().numpy().astype(np.int64)))
            self.compactions += 1
        self.key_count.zero_()
        self.bound = 0

    def finish(self):
        self._compact()
        if not self.parts:
            return np.zeros(0, dtype=np.int64), np.zeros(0, dtype=np.int64)
        if len(self.parts) == 1:
            return self.parts[0]
        keys = np.concatenate([p[0] for p in self.parts])
        mult = np.concatenate([p[1] for p in self.parts])
        uniq, inv = np.unique(keys, return_inverse=True)
        return uniq, np.bincount(inv, weights=mult, minlength=uniq.size).astype(np.int64)


_default_backends = {}
_default_backends_lock = threading.Lock()


def default_backend(device: Optional[int] = None):
    """One shared `HipRaster` per (device, host thread) for callers that are not handed a backend (camera-set warps, the
    down-scale of `get_image`).  A libgeograster context is not thread safe (include/geograster.h: one context per device and
    host thread): a loader thread that resizes photos while the caller's thread rasterizes gets a context of its own."""
    torch = _torch()
    if not torch.cuda.is_available():
        raise RuntimeError(
            "geograypher_amd: no ROCm GPU visible (torch.cuda.is_available() is False). "
            "The projection path runs on MI355X only; there is no CPU fallback."
        )
    dev = torch.cuda.current_device() if device is None else int(device)
    key = (dev, threading.get_ident())
    with _default_backends_lock:
        if key not in _default_backends:
            _default_backends[key] = HipRaster(dev)
        return _default_backends[key]


class HipRaster:
    """Device backend: one libgeograster context on one GPU, operating on torch tensors.

    All methods take and return torch tensors that live on `self.device`; the Python mesh class converts to the
    numpy arrays the reference API promises only at its own boundary.
    """

    def __init__(self, device: Optional[int] = None):
        torch = _torch()
        self.lib = load_library()
        if not torch.cuda.is_available():
            raise RuntimeError(
                "geograypher_amd: no ROCm GPU visible (torch.cuda.is_available() is False). "
                "The projection path runs on MI355X only; there is no CPU fallback."
            )
        if device is None:
            device = torch.cuda.current_device()
        self.device_index = int(device)
        self.device = torch.device("cuda", self.device_index)
        handle = ctypes.c_void_p()
        rc = self.lib.gr_ctx_create(self.device_index, ctypes.byref(handle))
        if rc != GR_OK:
            raise RuntimeError(f"gr_ctx_create(device={device}) failed with code {rc}")
        self._ctx = handle
        self._verts = None
        self._faces = None
        self.last_retries = 0
        self.last_stats = {}
        self.n_faces = 0
        self.n_verts = 0
        self.vertex_order = "r1"

    # -- plumbing ------------------------------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_ctx", None):
            self.lib.gr_ctx_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _stream(self):
        torch = _torch()
        return ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _check(self, rc: int, what: str):
        if rc == GR_OK:
            return
        msg = self.lib.gr_last_error(self._ctx)
        msg = msg.decode("utf-8", "replace") if msg else ""
        if rc == -1:
            raise ValueError(f"{what}: {msg}")
        if rc == -5:
            raise IndexError(f"{what}: {msg}")
        raise RuntimeError(f"{what} failed (code {rc}): {msg}")

    def _dev(self, array, dtype):
        """numpy / tensor -> contiguous tensor of `dtype` on this device (no copy when already there)."""
        torch = _torch()
        if isinstance(array, torch.Tensor):
            return array.to(device=self.device, dtype=dtype).contiguous()
        return torch.as_tensor(np.ascontiguousarray(array)).to(device=self.device, dtype=dtype).contiguous()

    def set_profiling(self, enabled: bool):
        self._check(self.lib.gr_set_profiling(self._ctx, 1 if enabled else 0), "gr_set_profiling")

    def set_option(self, key: int, value: int):
        """Tuning knobs of include/geograster.h (GR_OPT_*): 2 tile height log2, 3 views per launch group, 6 single-pass
        slots per tile (0 = exact binning), 7 variant bits (see GR_OPT_VARIANT in the header)."""
        self._check(self.lib.gr_set_option(self._ctx, int(key), int(value)), "gr_set_option")

    def set_vertex_order(self, name: str):
        """"r1" (default): rule R1 of DESIGN.md; "gl": the perspective divide, viewport transform and snap in an OpenGL
        pipeline's order of operations (GR_OPT_VERTEX_ORDER: what Mesa's llvmpipe executes; the principal point must be the
        window centre).  The only option results depend on."""
        if name not in ("r1", "gl"):
            raise ValueError(f"vertex_order must be 'r1' or 'gl', got {name!r}")
        self.set_option(10, 1 if name == "gl" else 0)
        self.vertex_order = name

    def stage_times(self) -> dict:
        st = StageTimes()
        self._check(self.lib.gr_get_stage_times(self._ctx, ctypes.byref(st)), "gr_get_stage_times")
        return st.as_dict()

    # -- mesh ----------------------------------------------------------------------------------------------------
    def upload_mesh(self, verts, faces):
        """verts (V,3) float, faces (F,3) int in the cameras' local frame (meshes.py:1641-1676 output)."""
        torch = _torch()
        v = self._dev(verts, torch.float32)
        f = self._dev(faces, torch.int32)
        if v.ndim != 2 or v.shape[1] != 3 or f.ndim != 2 or f.shape[1] != 3:
            raise ValueError(f"mesh must be (V,3) vertices and (F,3) faces, got {tuple(v.shape)} and {tuple(f.shape)}")
        with torch.cuda.device(self.device):
            rc = self.lib.gr_mesh_upload(self._ctx, v.data_ptr(), f.data_ptr(), v.shape[0], f.shape[0], self._stream())
        self._check(rc, "gr_mesh_upload")
        self._verts, self._faces = v, f  # borrowed by the library: keep alive
        self.n_verts, self.n_faces = int(v.shape[0]), int(f.shape[0])

    # -- pix2face ------------------------------------------------------------------------------------------------
    def raster_face_ids(self, cams, h: int, w: int, out=None, want_depth: bool = False, check: bool = True):
        """cams (N,16) camera records -> ids (N,h,w) int32 tensor [, depth (N,h,w) float32].

        `check=True` (default) reads the call's status back and repeats the unfinished views when a tile overflowed its
        bin segment (`last_retries`; the statistics of all attempts are summed in `last_stats`).  `check=False` only
        enqueues the work: nothing is known about its outcome -- `last_stats` says `{"unchecked": True}` -- and a view whose
        bins overflowed is INCOMPLETE until the caller asks `raster_status()`, which raises on overflow.  Use it only for
        repeats of a call that was sized with `check=True` on the same inputs."""
        torch = _torch()
        cams_t = self._dev(cams, torch.float32)
        if cams_t.ndim != 2 or cams_t.shape[1] != GR_CAM_FLOATS:
            raise ValueError(f"camera records must be (N,{GR_CAM_FLOATS}), got {tuple(cams_t.shape)}")
        n = int(cams_t.shape[0])
        if out is None:
            out = torch.empty((n, h, w), dtype=torch.int32, device=self.device)
        elif tuple(out.shape) != (n, h, w) or out.dtype != torch.int32 or not out.is_contiguous():
            raise ValueError("out must be a contiguous int32 tensor of shape (N,h,w)")
        depth = torch.empty((n, h, w), dtype=torch.float32, device=self.device) if want_depth else None
        v0 = 0
        self.last_retries = 0
        acc = _StatsAccumulator()
        for attempt in range(4):
            with torch.cuda.device(self.device):
                rc = self.lib.gr_raster_face_ids(
                    self._ctx, cams_t[v0:].data_ptr(), n - v0, h, w, out[v0:].data_ptr(),
                    depth[v0:].data_ptr() if depth is not None else None, self._stream(),
                )
            self._check(rc, "gr_raster_face_ids")
            if not check:
                self.last_stats = {"unchecked": True}
                break
            st = RasterStats()
            rc = self.lib.gr_raster_status(self._ctx, ctypes.byref(st))
            if rc == GR_EOVERFLOW and attempt < 3:
                acc.add(st, n - v0, partial=True)
                v0 += int(st.views_done)  # the library has recorded the need; only the unfinished views are repeated
                self.last_retries += 1
                continue
            self._check(rc, "gr_raster_status")
            acc.add(st, n - v0, partial=False)
            self.last_stats = acc.result()
            break
        return (out, depth) if want_depth else out

    def raster_status(self) -> dict:
        st = RasterStats()
        self._check(self.lib.gr_raster_status(self._ctx, ctypes.byref(st)), "gr_raster_status")
        return st.as_dict()

    # -- render_flat gather --------------------------------------------------------------------------------------
    def gather_texture(self, ids, face_texture):
        """ids (...,) int32 tensor, face_texture (F,C) -> (..., C) float64 tensor, NaN where ids == -1."""
        torch = _torch()
        ids_t = self._dev(ids, torch.int32)
        tex = self._dev(face_texture, torch.float64)
        F, C = int(tex.shape[0]), int(tex.shape[1])
        out = torch.empty(tuple(ids_t.shape) + (C,), dtype=torch.float64, device=self.device)
        with torch.cuda.device(self.device):
            rc = self.lib.gr_gather_texture_f64(
                self._ctx, ids_t.data_ptr(), ids_t.numel(), tex.data_ptr(), F, C, out.data_ptr(), self._stream()
            )
        self._check(rc, "gr_gather_texture_f64")
        return out

    def gather_texture_u8(self, ids, face_texture, null_value: int = 0):
        """save_renders epilogue: ids (...,) int32, face_texture (F,C) -> (..., C) uint8 tensor (meshes.py:2325-2337)."""
        torch = _torch()
        ids_t = self._dev(ids, torch.int32)
        tex = self._dev(face_texture, torch.float64)
        F, C = int(tex.shape[0]), int(tex.shape[1])
        out = torch.empty(tuple(ids_t.shape) + (C,), dtype=torch.uint8, device=self.device)
        with torch.cuda.device(self.device):
            rc = self.lib.gr_gather_texture_u8(
                self._ctx, ids_t.data_ptr(), ids_t.numel(), tex.data_ptr(), F, C, int(null_value), out.data_ptr(),
                self._stream(),
            )
        self._check(rc, "gr_gather_texture_u8")
        return out

    def project_index_pairs(self, ids, img, n_classes: int, counts, neg1_is_last_face: bool = True):
        """Sparse index aggregation step (derived_meshes.py:470-520) for N views: ids (N,h,w) int32, img (N,h,w) float64
        with NaN = no prediction.  Accumulates counts (F,) and returns (pair_keys, multiplicities) int64 numpy arrays
        with pair key = face * n_classes + class."""
        torch = _torch()
        ids_t = self._dev(ids, torch.int32)
        img_t = self._dev(img, torch.float64)
        if ids_t.ndim == 2:
            ids_t, img_t = ids_t[None], img_t[None]
        if img_t.ndim == 4 and img_t.shape[-1] == 1:
            img_t = img_t[..., 0]
        if ids_t.shape != img_t.shape:
            raise ValueError(f"ids {tuple(ids_t.shape)} and index image {tuple(img_t.shape)} differ in shape")
        n, h, w = (int(x) for x in ids_t.shape)
        cap = n * self.n_faces
        keys = torch.empty((max(cap, 1),), dtype=torch.int64, device=self.device)
        key_count = torch.zeros((1,), dtype=torch.int64, device=self.device)
        flags = GR_FLAG_NEG1_IS_LAST_FACE if neg1_is_last_face else 0
        with torch.cuda.device(self.device):
            rc = self.lib.gr_project_index_pairs(
                self._ctx, ids_t.data_ptr(), img_t.contiguous().data_ptr(), n, h, w, int(n_classes), counts.data_ptr(),
                keys.data_ptr(), cap, key_count.data_ptr(), flags, self._stream(),
            )
        self._check(rc, "gr_project_index_pairs")
        m = int(key_count.item())
        if m == 0:
            return np.zeros(0, dtype=np.int64), np.zeros(0, dtype=np.int64)
        uniq = torch.empty((m,), dtype=torch.int64, device=self.device)
        mult = torch.empty((m,), dtype=torch.int32, device=self.device)
        n_unique = ctypes.c_int64(0)
        with torch.cuda.device(self.device):
            rc = self.lib.gr_count_pairs(self._ctx, keys.data_ptr(), m, uniq.data_ptr(), mult.data_ptr(),
                                         ctypes.byref(n_unique), self._stream())
        self._check(rc, "gr_count_pairs")
        k = int(n_unique.value)
        return uniq[:k].cpu().numpy(), mult[:k].cpu().numpy().astype(np.int64)

    def new_pair_accumulator(self, n_classes: int, counts, neg1_is_last_face: bool = True):
        """Sparse index aggregation over MANY views with the (face, class) pair keys kept on the device: `add(ids, img)` per
        view (or group of views) only enqueues work, `finish()` sorts and counts the pairs ONCE -- see `PairAccumulator`."""
        return PairAccumulator(self, n_classes, counts, neg1_is_last_face)

    # -- projection / aggregation --------------------------------------------------------------------------------
    def new_vote_buffers(self, C: int):
        torch = _torch()
        votes = torch.zeros((self.n_faces, C), dtype=torch.int32, device=self.device)  # uint32 payload
        counts = torch.zeros((self.n_faces,), dtype=torch.int32, device=self.device)
        return votes, counts

    def project_labels(self, ids, labels, C: int, votes, counts, neg1_is_last_face: bool = True):
        """ids (N,h,w) int32, labels (N,h,w) uint8 class indices; accumulates into votes (F,C), counts (F,)."""
        torch = _torch()
        ids_t = self._dev(ids, torch.int32)
        lab_t = self._dev(labels, torch.uint8)
        if ids_t.ndim == 2:
            ids_t, lab_t = ids_t[None], lab_t[None]
        if ids_t.shape != lab_t.shape:
            raise ValueError(f"ids {tuple(ids_t.shape)} and labels {tuple(lab_t.shape)} differ in shape")
        n, h, w = (int(x) for x in ids_t.shape)
        flags = GR_FLAG_NEG1_IS_LAST_FACE if neg1_is_last_face else 0
        with torch.cuda.device(self.device):
            rc = self.lib.gr_project_labels_u8(
                self._ctx, ids_t.data_ptr(), lab_t.data_ptr(), n, h, w, C, votes.data_ptr(), counts.data_ptr(), flags,
                self._stream(),
            )
        self._check(rc, "gr_project_labels_u8")

    def project_values(self, ids, img, sums, counts, neg1_is_last_face: bool = True):
        """ids (N,h,w) int32, img (N,h,w,C) float64; accumulates nansum into sums (F,C) and counts (F,)."""
        torch = _torch()
        ids_t = self._dev(ids, torch.int32)
        img_t = self._dev(img, torch.float64)
        if ids_t.ndim == 2:
            ids_t, img_t = ids_t[None], img_t[None]
        n, h, w = (int(x) for x in ids_t.shape)
        C = int(img_t.shape[-1])
        if tuple(img_t.shape) != (n, h, w, C):
            raise ValueError(f"img {tuple(img_t.shape)} does not match ids {tuple(ids_t.shape)}")
        flags = GR_FLAG_NEG1_IS_LAST_FACE if neg1_is_last_face else 0
        with torch.cuda.device(self.device):
            rc = self.lib.gr_project_values_f64(
                self._ctx, ids_t.data_ptr(), img_t.data_ptr(), n, h, w, C, sums.data_ptr(), counts.data_ptr(), flags,
                self._stream(),
            )
        self._check(rc, "gr_project_values_f64")

    def project_view(self, ids, img, neg1_is_last_face: bool = True):
        """One view of project_images: ids (h,w) int32, img (h,w,C) float64 -> (F,C) float64, NaN for unseen faces."""
        torch = _torch()
        ids_t = self._dev(ids, torch.int32)
        img_t = self._dev(img, torch.float64)
        h, w = (int(x) for x in ids_t.shape)
        C = int(img_t.shape[-1])
        if tuple(img_t.shape) != (h, w, C):
            raise ValueError(f"img {tuple(img_t.shape)} does not match ids {tuple(ids_t.shape)}")
        tex = torch.empty((self.n_faces, C), dtype=torch.float64, device=self.device)
        flags = GR_FLAG_NEG1_IS_LAST_FACE if neg1_is_last_face else 0
        with torch.cuda.device(self.device):
            rc = self.lib.gr_project_view_f64(
                self._ctx, ids_t.data_ptr(), img_t.data_ptr(), h, w, C, tex.data_ptr(), flags, self._stream()
            )
        self._check(rc, "gr_project_view_f64")
        return tex

    def raster_project_labels(self, cams, labels, C: int, votes, counts, ids_out=None, neg1_is_last_face: bool = True,
                              check: bool = True):
        """Fused pix2face + label projection for N views (aggregate_projected_images fast path): the face ids stay in
        the rasterizer's LDS tiles unless `ids_out` (N,h,w int32) is given.  Accumulates into votes / counts.
        `check` as in `raster_face_ids`: with `check=False` a launch group whose bins overflowed (and every later one) adds
        NO votes and nobody is told until `raster_status()` is asked."""
        torch = _torch()
        cams_t = self._dev(cams, torch.float32)
        lab_t = self._dev(labels, torch.uint8)
        n, h, w = (int(x) for x in lab_t.shape)
        if cams_t.shape[0] != n:
            raise ValueError(f"{cams_t.shape[0]} camera records for {n} label images")
        flags = GR_FLAG_NEG1_IS_LAST_FACE if neg1_is_last_face else 0
        v0 = 0
        self.last_retries = 0
        acc = _StatsAccumulator()
        for attempt in range(4):
            with torch.cuda.device(self.device):
                rc = self.lib.gr_raster_project_labels_u8(
                    self._ctx, cams_t[v0:].data_ptr(), lab_t[v0:].data_ptr(), n - v0, h, w, C, votes.data_ptr(),
                    counts.data_ptr(), ids_out[v0:].data_ptr() if ids_out is not None else None, flags, self._stream(),
                )
            self._check(rc, "gr_raster_project_labels_u8")
            if not check:
                self.last_stats = {"unchecked": True}
                break
            st = RasterStats()
            rc = self.lib.gr_raster_status(self._ctx, ctypes.byref(st))
            if rc == GR_EOVERFLOW and attempt < 3:
                # the votes of the first views_done views are in; the library skipped the rest on the device
                acc.add(st, n - v0, partial=True)
                v0 += int(st.views_done)
                self.last_retries += 1
                continue
            self._check(rc, "gr_raster_status")
            acc.add(st, n - v0, partial=False)
            self.last_stats = acc.result()
            break
        return ids_out

    # -- get_image(image_scale) behind the file read (row a5) -----------------------------------------------------
    _RESIZE_DTYPES = {"uint8": 0, "float32": 1, "float64": 2}

    def resize_image(self, image, out_hw=None, divide_by_255: Optional[bool] = None):
        """cameras.py:154-174 on the device: `image` ((H,W) or (H,W,C); numpy or tensor, uint8 / float32 / float64, in the dtype
        its file holds) -> float64 tensor of shape out_hw (+ C): uint8 values are divided by 255.0 (`divide_by_255`, default:
        exactly when the dtype is uint8, as get_image does), then skimage.transform.resize with its defaults
        (gr_resize_image_f64: anti-aliasing Gaussian + order-1 sampling at half-pixel centres).  out_hw None or the input
        size: the conversion alone."""
        torch = _torch()
        if isinstance(image, torch.Tensor):
            t = image.to(self.device).contiguous()
        else:
            t = torch.as_tensor(np.ascontiguousarray(image)).to(self.device)
        if t.dtype == torch.bool:
            t = t.to(torch.uint8)
        name = str(t.dtype).replace("torch.", "")
        if name not in self._RESIZE_DTYPES:
            # every other dtype: widened to float64 first, values kept (scikit-image would rescale integer types by their
            # range and truncate the filtered image to the integer type; not reproduced: documented in DESIGN.md)
            t, name = t.to(torch.float64), "float64"
        if t.ndim not in (2, 3):
            raise ValueError(f"image must be (H,W) or (H,W,C), got shape {tuple(t.shape)}")
        h_in, w_in = int(t.shape[0]), int(t.shape[1])
        C = 1 if t.ndim == 2 else int(t.shape[2])
        h_out, w_out = (h_in, w_in) if out_hw is None else (int(out_hw[0]), int(out_hw[1]))
        if divide_by_255 is None:
            divide_by_255 = name == "uint8"
        out = torch.empty((h_out, w_out) + tuple(t.shape[2:]), dtype=torch.float64, device=self.device)
        with torch.cuda.device(self.device):
            rc = self.lib.gr_resize_image_f64(self._ctx, t.data_ptr(), self._RESIZE_DTYPES[name], h_in, w_in, C,
                                              1 if divide_by_255 else 0, h_out, w_out, out.data_ptr(), self._stream())
        self._check(rc, "gr_resize_image_f64")
        return out

    # -- distortion warp (row f1) --------------------------------------------------------------------------------
    def upload_map(self, inverse_map):
        """(2, H, W) float64 sampling map (rows, cols) -> device tensor."""
        torch = _torch()
        m = self._dev(inverse_map, torch.float64)
        if m.ndim != 3 or m.shape[0] != 2:
            raise ValueError(f"sampling map must be (2, H, W), got {tuple(m.shape)}")
        return m

    LENS_PARAMS = ("f", "cx", "cy", "image_width", "image_height", "k1", "k2", "k3", "k4", "p1", "p2", "b1", "b2")

    def invert_distortion(self, params: dict, h: int, w: int, image_scale: float = 1.0, max_iters: int = 12,
                          fill: float = -1.0):
        """(2, h, w) float64 device map: for every pixel of the warped image the position to sample in the ideal image
        (gr_invert_distortion_f64: dense Newton inverse of the Metashape lens model; replaces the host griddata inversion
        of cameras.py:1045-1062)."""
        torch = _torch()
        unknown = set(params) - set(self.LENS_PARAMS)
        if unknown:
            raise ValueError(f"Unexpected distortion params found: {sorted(unknown)}")
        par = (ctypes.c_double * 13)(*[float(params.get(k, 0.0)) for k in self.LENS_PARAMS])
        out = torch.empty((2, h, w), dtype=torch.float64, device=self.device)
        with torch.cuda.device(self.device):
            rc = self.lib.gr_invert_distortion_f64(self._ctx, par, int(h), int(w), float(image_scale), int(max_iters),
                                                   float(fill), out[0].data_ptr(), out[1].data_ptr(), self._stream())
        self._check(rc, "gr_invert_distortion_f64")
        return out

    def warp_image(self, input_image, map_t, order: int = 1, fill_value: float = 0.0,
                   reference_float_roundtrip: bool = False):
        """Resample `input_image` ((I,J) or (I,J,C)) through the (2,H,W) device map: utils/image.py:72-126 on device.

        Integer images with order 0 are gathered as integers (gr_warp_nearest_i32); everything else goes through the
        float64 kernel and is cast back to the input dtype by truncation like the reference's `.astype(initial_dtype)`.
        numpy in -> numpy out, tensor in -> tensor out."""
        torch = _torch()
        is_tensor = isinstance(input_image, torch.Tensor)
        img = input_image if is_tensor else np.asarray(input_image)
        if img.ndim not in (2, 3):
            raise ValueError(f"image must be (I,J) or (I,J,C), got shape {tuple(img.shape)}")
        np_dtype = None if is_tensor else img.dtype
        h_out, w_out = int(map_t.shape[1]), int(map_t.shape[2])
        h_in, w_in = int(img.shape[0]), int(img.shape[1])
        is_int = (not torch.is_floating_point(img)) if is_tensor else np.issubdtype(img.dtype, np.integer) or img.dtype == bool
        # utils/image.py:86-96: an image without variation (fill included) is returned as a constant of the INPUT shape
        if is_tensor:
            vmin, vmax = float(img.min()), float(img.max())
        else:
            vmin, vmax = float(np.min(img)), float(np.max(img))
        lo, hi = min(vmin, float(fill_value)), max(vmax, float(fill_value))
        if hi - lo == 0:
            if is_tensor:
                return torch.full_like(img.squeeze(), fill_value)
            return np.full_like(np.squeeze(img), fill_value=fill_value)
        small_int = is_int and -2**31 <= lo and hi < 2**31 and float(fill_value) == int(fill_value)
        with torch.cuda.device(self.device) if self.device.type == "cuda" else _nullcontext():
            if small_int and order == 0:
                src = self._dev(img, torch.int32)
                squeeze = src.ndim == 2
                if squeeze:
                    src = src[..., None]
                outs = []
                for ch in range(src.shape[2]):
                    plane = src[..., ch].contiguous()
                    out = torch.empty((h_out, w_out), dtype=torch.int32, device=self.device)
                    rc = self.lib.gr_warp_nearest_i32(
                        self._ctx, plane.data_ptr(), h_in, w_in, map_t[0].data_ptr(), map_t[1].data_ptr(), h_out, w_out,
                        int(fill_value), 1 if reference_float_roundtrip else 0, lo, hi - lo, out.data_ptr(),
                        self._stream(),
                    )
                    self._check(rc, "gr_warp_nearest_i32")
                    outs.append(out)
                res = outs[0] if squeeze else torch.stack(outs, dim=-1)
            else:
                src = self._dev(img, torch.float64)
                squeeze = src.ndim == 2
                if squeeze:
                    src = src[..., None]
                src = src.contiguous()
                C = int(src.shape[2])
                res = torch.empty((h_out, w_out, C), dtype=torch.float64, device=self.device)
                rc = self.lib.gr_warp_f64(
                    self._ctx, src.data_ptr(), h_in, w_in, C, map_t[0].data_ptr(), map_t[1].data_ptr(), h_out, w_out,
                    int(order), float(fill_value), res.data_ptr(), self._stream(),
                )
                self._check(rc, "gr_warp_f64")
                if squeeze:
                    res = res[..., 0]
        if is_tensor:
            return res.to(img.dtype)
        return np.squeeze(res.cpu().numpy().astype(np_dtype))

    def finalize_votes(self, votes, counts):
        """(votes, counts) -> average (F,C), summed (F,C), counts (F,) float64 tensors (meshes.py:2069-2082)."""
        torch = _torch()
        F, C = int(votes.shape[0]), int(votes.shape[1])
        avg = torch.empty((F, C), dtype=torch.float64, device=self.device)
        summed = torch.empty((F, C), dtype=torch.float64, device=self.device)
        cnt = torch.empty((F,), dtype=torch.float64, device=self.device)
        with torch.cuda.device(self.device):
            rc = self.lib.gr_finalize_votes(
                self._ctx, votes.data_ptr(), counts.data_ptr(), F, C, avg.data_ptr(), summed.data_ptr(), cnt.data_ptr(),
                self._stream(),
            )
        self._check(rc, "gr_finalize_votes")
        return avg, summed, cnt

    def finalize_sums(self, sums, counts):
        torch = _torch()
        F, C = int(sums.shape[0]), int(sums.shape[1])
        avg = torch.empty((F, C), dtype=torch.float64, device=self.device)
        cnt = torch.empty((F,), dtype=torch.float64, device=self.device)
        with torch.cuda.device(self.device):
            rc = self.lib.gr_finalize_sums_f64(
                self._ctx, sums.data_ptr(), counts.data_ptr(), F, C, avg.data_ptr(), cnt.data_ptr(), self._stream()
            )
        self._check(rc, "gr_finalize_sums_f64")
        return avg, sums, cnt

    def argmax_nonzero(self, array):
        """utils/indexing.py:9-32 on device: (F,C) float64 -> (F,) float64."""
        torch = _torch()
        arr = self._dev(array, torch.float64)
        F, C = int(arr.shape[0]), int(arr.shape[1])
        out = torch.empty((F,), dtype=torch.float64, device=self.device)
        with torch.cuda.device(self.device):
            rc = self.lib.gr_argmax_nonzero_f64(self._ctx, arr.data_ptr(), F, C, out.data_ptr(), self._stream())
        self._check(rc, "gr_argmax_nonzero_f64")
        return out
