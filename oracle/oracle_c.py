"""oracle/oracle_c.py -- TEST INFRASTRUCTURE ONLY.  ctypes loader for oracle/_build/liboracle.so (oracle_raster.c)."""
import ctypes
import subprocess
from pathlib import Path

import numpy as np

_DIR = Path(__file__).resolve().parent
_SO = _DIR / "_build" / "liboracle.so"
_lib = None


def build(force: bool = False):
    """Compile the C oracle with gcc (building the checker is not using it)."""
    newest = max((_DIR / f).stat().st_mtime for f in ("oracle_raster.c", "oracle_envelope.c", "Makefile"))
    if force or not _SO.is_file() or _SO.stat().st_mtime < newest:
        subprocess.run(["make", "-C", str(_DIR), "-B" if force else "-s"], check=True, capture_output=True)
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(str(_SO))
        vp, i32, i64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64
        for name in ("orc_raster_spec", "orc_raster_fast"):
            fn = getattr(L, name)
            fn.restype = i32
            fn.argtypes = [vp, vp, i64, i64, vp, i32, i32, vp, vp, vp]
        L.orc_raster_views.restype = i32
        L.orc_raster_views.argtypes = [vp, vp, i64, i64, vp, i32, i32, i32, vp, i32]
        L.orc_project_labels.restype = i32
        L.orc_project_labels.argtypes = [vp, vp, i32, i32, i64, i32, i32, vp, vp, vp]
        f64 = ctypes.c_double
        L.orc_envelope.restype = i32
        L.orc_envelope.argtypes = [vp, vp, i64, vp, i32, i32, f64, f64, vp, vp, vp, vp, vp, vp, vp, vp]
        L.orc_set_vertex_order.restype = None
        L.orc_set_vertex_order.argtypes = [i32]
        L.orc_raster_float.restype = i32
        L.orc_raster_float.argtypes = [vp, vp, i64, vp, i32, i32, vp, vp]
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _vertex_order(name):
    if name not in ("r1", "gl"):
        raise ValueError(f"vertex_order must be 'r1' or 'gl', got {name!r}")
    lib().orc_set_vertex_order(1 if name == "gl" else 0)


def raster(verts, faces, cam, h, w, want_depth=False, spec=False, vertex_order="r1"):
    """One view: (h,w) int32 face ids [, (h,w) float32 depth].  cam: 16-float record (include/geograster.h).
    vertex_order="gl": the second half of the vertex stage in an OpenGL pipeline's order of operations (oracle_raster.c R1-GL)."""
    _vertex_order(vertex_order)
    try:
        return _raster(verts, faces, cam, h, w, want_depth, spec)
    finally:
        lib().orc_set_vertex_order(0)


def _raster(verts, faces, cam, h, w, want_depth, spec):
    verts = np.ascontiguousarray(verts, dtype=np.float32)
    faces = np.ascontiguousarray(faces, dtype=np.int32)
    cam = np.ascontiguousarray(cam, dtype=np.float32).reshape(16)
    ids = np.empty((h, w), dtype=np.int32)
    zbuf = np.empty((h, w), dtype=np.int32)
    depth = np.empty((h, w), dtype=np.float32) if want_depth else None
    fn = lib().orc_raster_spec if spec else lib().orc_raster_fast
    rc = fn(_p(verts), _p(faces), verts.shape[0], faces.shape[0], _p(cam), h, w, _p(ids),
            _p(depth) if want_depth else None, _p(zbuf))
    assert rc == 0
    return (ids, depth) if want_depth else ids


def raster_views(verts, faces, cams, h, w, n_threads=1):
    """N views: (N,h,w) int32.  Returns (ids, threads_used)."""
    verts = np.ascontiguousarray(verts, dtype=np.float32)
    faces = np.ascontiguousarray(faces, dtype=np.int32)
    cams = np.ascontiguousarray(cams, dtype=np.float32).reshape(-1, 16)
    ids = np.empty((cams.shape[0], h, w), dtype=np.int32)
    used = lib().orc_raster_views(_p(verts), _p(faces), verts.shape[0], faces.shape[0], _p(cams), cams.shape[0], h, w,
                                  _p(ids), int(n_threads))
    return ids, used


def project_labels(ids, labels, n_faces, C, votes, counts, neg1_is_last_face=True):
    """C restatement of project_images + the per-view accumulate for uint8 index labels (accumulates in place)."""
    ids = np.ascontiguousarray(ids, dtype=np.int32)
    labels = np.ascontiguousarray(labels, dtype=np.uint8)
    h, w = ids.shape
    winner = np.empty(n_faces, dtype=np.int64)
    rc = lib().orc_project_labels(_p(ids), _p(labels), h, w, n_faces, C, 1 if neg1_is_last_face else 0, _p(votes),
                                  _p(counts), _p(winner))
    assert rc == 0


ENVELOPE_DELTA = 1.0 / 256 + 2e-3   # pixels: one sub-pixel step of an 8-bit grid + the fp32 transform error at f = 4000 px
ENVELOPE_GAP = 1e-5                 # relative depth gap below which the nearer of two faces is an implementation choice


def envelope(verts, faces, cam, h, w, delta=ENVELOPE_DELTA, gap_rel=ENVELOPE_GAP):
    """Classify every pixel of one view (oracle_envelope.c::orc_envelope): returns (cls (h,w) uint8 with 0 = background
    under every conforming rasterizer, 1 = the face in `ids` under every conforming rasterizer, 2 = implementation-
    defined; ids (h,w) int32; number of faces that straddle the near plane and were not classified)."""
    verts = np.ascontiguousarray(verts, dtype=np.float32)
    faces = np.ascontiguousarray(faces, dtype=np.int32)
    cam = np.ascontiguousarray(cam, dtype=np.float32).reshape(16)
    cls = np.empty((h, w), dtype=np.uint8)
    ids = np.empty((h, w), dtype=np.int32)
    zA, zB = np.empty((h, w), dtype=np.float64), np.empty((h, w), dtype=np.float64)
    fA, sure = np.empty((h, w), dtype=np.int32), np.empty((h, w), dtype=np.uint8)
    zlo, zhi = np.empty((h, w), dtype=np.float64), np.empty((h, w), dtype=np.float64)
    n = lib().orc_envelope(_p(verts), _p(faces), faces.shape[0], _p(cam), h, w, float(delta), float(gap_rel), _p(cls),
                           _p(ids), _p(zA), _p(zB), _p(fA), _p(sure), _p(zlo), _p(zhi))
    return cls, ids, n


def raster_float(verts, faces, cam, h, w):
    """The second, independently written rasterizer (un-snapped float64, closed triangles, ties -> higher id):
    returns ((h,w) int32 ids, number of faces skipped because they reach behind the near plane)."""
    verts = np.ascontiguousarray(verts, dtype=np.float32)
    faces = np.ascontiguousarray(faces, dtype=np.int32)
    cam = np.ascontiguousarray(cam, dtype=np.float32).reshape(16)
    ids = np.empty((h, w), dtype=np.int32)
    zbuf = np.empty((h, w), dtype=np.float64)
    n = lib().orc_raster_float(_p(verts), _p(faces), faces.shape[0], _p(cam), h, w, _p(ids), _p(zbuf))
    return ids, n
