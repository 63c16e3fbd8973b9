"""tests/golden/gl_raster.py -- BUILD-CONTAINER TOOL (test infrastructure; never imported by the product, never run on the
GPU box): renders the reference's face-id image with a REAL third-party OpenGL rasterizer that happens to be in this image.

The stage restated is geograypher/meshes/meshes.py:1776-1836: face ids encoded base-256 into a uint8 colour per face
(1783-1798), an unlit flat-colour render of the mesh through the pyvista camera of cameras.py:446-477 with anti-aliasing
off (1778, 1811-1822), screenshot, decode sum(ch_i * 256**i) (1823-1829), ids > F -> -1 (1836).

Two GL implementations, both loaded with ctypes (no X server, no window system):

  "swiftshader"  Google SwiftShader 4.1 (OpenGL ES 3.0, conformance-tested), shipped inside the `kaleido` wheel:
                 libEGL.so + libGLESv2.so, an EGL pbuffer surface.  GL_SUBPIXEL_BITS = 4.
  "llvmpipe"     Mesa 23.2.1 `swrast_dri.so` (the libgl1-mesa-dri package: the software rasterizer family the reference's
                 own Dockerfile:6-13 -- libgl1 + xvfb -- renders with), driven through the DRI software-rasterizer
                 loader interface by tests/golden/drisw_loader.c; rendering goes to a framebuffer object.
                 GL_SUBPIXEL_BITS = 8.

Geometry handed to GL: one vertex per face corner (3F vertices, no index buffer: each face carries its own colour, a
`flat` varying), positions in CAMERA space computed in float32 exactly as rule R1 of DESIGN.md does (d = p - t,
q = R^T d, every operation rounded individually), so that both rasterizers start from the same numbers;
gl_Position = (2 f/w x, -2 f/h y, A z + B, z) with A, B the usual near/far mapping -- the perspective matrix VTK builds
for a camera with only a vertical view angle (principal point at the window centre, cameras.py:469-475).  Depth test on
(GL_LEQUAL, VTK's default), no culling, no dithering, no blending, clear colour white (= id 0xFFFFFF > F -> background).
GL's window origin is bottom-left: rows are flipped on read-back, which is what the -2 f/h sign accounts for.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent

# ---- constants (khronos registry values) ---------------------------------------------------------------------------------
EGL_PBUFFER_BIT, EGL_OPENGL_ES3_BIT = 0x0001, 0x0040
EGL_SURFACE_TYPE, EGL_RENDERABLE_TYPE = 0x3033, 0x3040
EGL_ALPHA_SIZE, EGL_BLUE_SIZE, EGL_GREEN_SIZE, EGL_RED_SIZE, EGL_DEPTH_SIZE = 0x3021, 0x3022, 0x3023, 0x3024, 0x3025
EGL_NONE, EGL_HEIGHT, EGL_WIDTH = 0x3038, 0x3056, 0x3057
EGL_CONTEXT_CLIENT_VERSION, EGL_OPENGL_ES_API = 0x3098, 0x30A0

GL_TRIANGLES = 0x0004
GL_DEPTH_BUFFER_BIT, GL_COLOR_BUFFER_BIT = 0x0100, 0x4000
GL_LESS, GL_LEQUAL = 0x0201, 0x0203
GL_CULL_FACE, GL_DEPTH_TEST, GL_DITHER, GL_BLEND = 0x0B44, 0x0B71, 0x0BD0, 0x0BE2
GL_MULTISAMPLE = 0x809D
GL_PACK_ALIGNMENT, GL_SUBPIXEL_BITS, GL_DEPTH_BITS, GL_MAX_VIEWPORT_DIMS = 0x0D05, 0x0D50, 0x0D56, 0x0D3A
GL_UNSIGNED_BYTE, GL_FLOAT, GL_RGBA = 0x1401, 0x1406, 0x1908
GL_VENDOR, GL_RENDERER, GL_VERSION = 0x1F00, 0x1F01, 0x1F02
GL_RGBA8, GL_DEPTH_COMPONENT24, GL_DEPTH_COMPONENT32F = 0x8058, 0x81A6, 0x8CAC
GL_ARRAY_BUFFER, GL_STATIC_DRAW = 0x8892, 0x88E4
GL_FRAGMENT_SHADER, GL_VERTEX_SHADER, GL_COMPILE_STATUS, GL_LINK_STATUS = 0x8B30, 0x8B31, 0x8B81, 0x8B82
GL_FRAMEBUFFER, GL_RENDERBUFFER, GL_COLOR_ATTACHMENT0, GL_DEPTH_ATTACHMENT = 0x8D40, 0x8D41, 0x8CE0, 0x8D00
GL_FRAMEBUFFER_COMPLETE = 0x8CD5
GL_FIRST_VERTEX_CONVENTION, GL_LAST_VERTEX_CONVENTION = 0x8E4D, 0x8E4E

_vp, _i, _u, _f = ctypes.c_void_p, ctypes.c_int, ctypes.c_uint, ctypes.c_float
_SIGS = {  # name: (restype, argtypes)
    "glGetString": (ctypes.c_char_p, [_u]), "glGetIntegerv": (None, [_u, _vp]), "glGetError": (_u, []),
    "glViewport": (None, [_i] * 4), "glClearColor": (None, [_f] * 4), "glClear": (None, [_u]),
    "glClearDepthf": (None, [_f]), "glEnable": (None, [_u]), "glDisable": (None, [_u]), "glDepthFunc": (None, [_u]),
    "glDepthRangef": (None, [_f, _f]), "glPixelStorei": (None, [_u, _i]),
    "glCreateShader": (_u, [_u]), "glShaderSource": (None, [_u, _i, _vp, _vp]), "glCompileShader": (None, [_u]),
    "glGetShaderiv": (None, [_u, _u, _vp]), "glGetShaderInfoLog": (None, [_u, _i, _vp, _vp]),
    "glCreateProgram": (_u, []), "glAttachShader": (None, [_u, _u]), "glBindAttribLocation": (None, [_u, _u, ctypes.c_char_p]),
    "glLinkProgram": (None, [_u]), "glGetProgramiv": (None, [_u, _u, _vp]), "glGetProgramInfoLog": (None, [_u, _i, _vp, _vp]),
    "glUseProgram": (None, [_u]), "glGetUniformLocation": (_i, [_u, ctypes.c_char_p]), "glUniform4f": (None, [_i] + [_f] * 4),
    "glGenBuffers": (None, [_i, _vp]), "glBindBuffer": (None, [_u, _u]), "glBufferData": (None, [_u, ctypes.c_ssize_t, _vp, _u]),
    "glDeleteBuffers": (None, [_i, _vp]),
    "glGenVertexArrays": (None, [_i, _vp]), "glBindVertexArray": (None, [_u]),
    "glEnableVertexAttribArray": (None, [_u]), "glVertexAttribPointer": (None, [_u, _i, _u, ctypes.c_ubyte, _i, _vp]),
    "glDrawArrays": (None, [_u, _i, _i]), "glFinish": (None, []), "glReadPixels": (None, [_i, _i, _i, _i, _u, _u, _vp]),
    "glGenFramebuffers": (None, [_i, _vp]), "glBindFramebuffer": (None, [_u, _u]),
    "glGenRenderbuffers": (None, [_i, _vp]), "glBindRenderbuffer": (None, [_u, _u]),
    "glRenderbufferStorage": (None, [_u, _u, _i, _i]), "glFramebufferRenderbuffer": (None, [_u, _u, _u, _u]),
    "glCheckFramebufferStatus": (_u, [_u]), "glDeleteFramebuffers": (None, [_i, _vp]), "glDeleteRenderbuffers": (None, [_i, _vp]),
    "glProvokingVertex": (None, [_u]), "glUniformMatrix4fv": (None, [_i, _i, ctypes.c_ubyte, _vp]),
}

_VS = """#version 300 es
in vec3 pos;
in vec4 col;
flat out vec4 vcol;
uniform vec4 proj;   // 2f/w, -2f/h, A, B
void main() {
  vcol = col;
  gl_Position = vec4(proj.x * pos.x, proj.y * pos.y, proj.z * pos.z + proj.w, pos.z);
}
"""
_VS_MATRIX = """#version 300 es
in vec3 pos;
in vec4 col;
flat out vec4 vcol;
uniform mat4 mcdc;   // model coordinates -> device (clip) coordinates, the one matrix VTK's vertex shader applies
void main() {
  vcol = col;
  gl_Position = mcdc * vec4(pos, 1.0);
}
"""
_FS = """#version 300 es
precision highp float;
flat in vec4 vcol;
out vec4 frag;
void main() { frag = vcol; }
"""


def swiftshader_dir() -> Path:
    import importlib.util

    spec = importlib.util.find_spec("kaleido")
    if spec is None or not spec.submodule_search_locations:
        raise RuntimeError("the kaleido wheel (which ships SwiftShader) is not installed")
    d = Path(list(spec.submodule_search_locations)[0]) / "executable" / "bin" / "swiftshader"
    if not (d / "libEGL.so").is_file():
        raise RuntimeError(f"no SwiftShader under {d}")
    return d


class _GLFunctions:
    def __init__(self, get_proc, optional=("glProvokingVertex", "glGenVertexArrays", "glBindVertexArray")):
        for name, (res, args) in _SIGS.items():
            addr = get_proc(name.encode())
            if not addr:
                if name in optional:
                    setattr(self, name, None)
                    continue
                raise RuntimeError(f"GL entry point {name} not found")
            setattr(self, name, ctypes.CFUNCTYPE(res, *args)(addr))


class GLRasterizer:
    """A headless GL context of one of the two implementations; `render_ids` restates meshes.py:1776-1836."""

    def __init__(self, backend: str = "swiftshader", max_size=(8192, 8192)):
        self.backend = backend
        self._fbo = None
        self._fbo_size = None
        if backend == "swiftshader":
            self._init_swiftshader(max_size)
        elif backend == "llvmpipe":
            self._init_drisw()
        else:
            raise ValueError(backend)
        gl = self.gl
        self.info = {
            "backend": backend,
            "GL_VENDOR": (gl.glGetString(GL_VENDOR) or b"").decode(),
            "GL_RENDERER": (gl.glGetString(GL_RENDERER) or b"").decode(),
            "GL_VERSION": (gl.glGetString(GL_VERSION) or b"").decode(),
            "GL_SUBPIXEL_BITS": self._geti(GL_SUBPIXEL_BITS),
        }
        self._prog = self._make_program(_VS)
        self._uproj = gl.glGetUniformLocation(self._prog, b"proj")
        self._prog_matrix = self._make_program(_VS_MATRIX)
        self._umcdc = gl.glGetUniformLocation(self._prog_matrix, b"mcdc")
        if gl.glGenVertexArrays is not None:
            vao = _u(0)
            gl.glGenVertexArrays(1, ctypes.byref(vao))
            gl.glBindVertexArray(vao.value)

    def _geti(self, what, n=1):
        buf = (ctypes.c_int * max(n, 4))()
        self.gl.glGetIntegerv(what, buf)
        return buf[0] if n == 1 else list(buf[:n])

    # -- SwiftShader over EGL ----------------------------------------------------------------------------------------------
    def _init_swiftshader(self, max_size):
        d = swiftshader_dir()
        self._gles = ctypes.CDLL(str(d / "libGLESv2.so"), mode=ctypes.RTLD_GLOBAL)
        egl = self._egl = ctypes.CDLL(str(d / "libEGL.so"), mode=ctypes.RTLD_GLOBAL)
        egl.eglGetDisplay.restype, egl.eglGetDisplay.argtypes = _vp, [_vp]
        egl.eglInitialize.argtypes = [_vp, _vp, _vp]
        egl.eglChooseConfig.argtypes = [_vp, _vp, _vp, _i, _vp]
        egl.eglCreatePbufferSurface.restype, egl.eglCreatePbufferSurface.argtypes = _vp, [_vp, _vp, _vp]
        egl.eglCreateContext.restype, egl.eglCreateContext.argtypes = _vp, [_vp, _vp, _vp, _vp]
        egl.eglMakeCurrent.argtypes = [_vp, _vp, _vp, _vp]
        egl.eglGetProcAddress.restype, egl.eglGetProcAddress.argtypes = _vp, [ctypes.c_char_p]
        egl.eglBindAPI.argtypes = [_u]
        dpy = egl.eglGetDisplay(None)
        major, minor = _i(0), _i(0)
        if not dpy or not egl.eglInitialize(dpy, ctypes.byref(major), ctypes.byref(minor)):
            raise RuntimeError("eglInitialize failed")
        egl.eglBindAPI(EGL_OPENGL_ES_API)
        attrs = (_i * 15)(EGL_SURFACE_TYPE, EGL_PBUFFER_BIT, EGL_RENDERABLE_TYPE, EGL_OPENGL_ES3_BIT, EGL_RED_SIZE, 8,
                          EGL_GREEN_SIZE, 8, EGL_BLUE_SIZE, 8, EGL_ALPHA_SIZE, 8, EGL_DEPTH_SIZE, 24, EGL_NONE)
        cfg, ncfg = _vp(0), _i(0)
        if not egl.eglChooseConfig(dpy, attrs, ctypes.byref(cfg), 1, ctypes.byref(ncfg)) or ncfg.value < 1:
            raise RuntimeError("eglChooseConfig: no RGBA8 + depth24 ES3 pbuffer config")
        sattrs = (_i * 5)(EGL_WIDTH, 16, EGL_HEIGHT, 16, EGL_NONE)   # rendering goes to a framebuffer object
        surf = egl.eglCreatePbufferSurface(dpy, cfg, sattrs)
        cattrs = (_i * 3)(EGL_CONTEXT_CLIENT_VERSION, 3, EGL_NONE)
        ctx = egl.eglCreateContext(dpy, cfg, None, cattrs)
        if not surf or not ctx or not egl.eglMakeCurrent(dpy, surf, surf, ctx):
            raise RuntimeError("could not create / bind the EGL pbuffer context")
        self._keep = (dpy, surf, ctx)

        def get_proc(name):
            addr = egl.eglGetProcAddress(name)
            if not addr:
                try:
                    addr = ctypes.cast(getattr(self._gles, name.decode()), _vp).value
                except AttributeError:
                    addr = None
            return addr

        self.gl = _GLFunctions(get_proc)

    # -- Mesa swrast_dri.so through the DRI software loader interface ---------------------------------------------------------
    def _init_drisw(self):
        so = build_drisw_loader()
        L = self._drisw = ctypes.CDLL(str(so), mode=ctypes.RTLD_GLOBAL)
        L.drisw_open.restype, L.drisw_open.argtypes = _i, [ctypes.c_char_p, _i]
        L.drisw_error.restype = ctypes.c_char_p
        L.drisw_get_proc.restype, L.drisw_get_proc.argtypes = _vp, [ctypes.c_char_p]
        os.environ.setdefault("GALLIUM_DRIVER", "llvmpipe")
        rc = L.drisw_open(b"/usr/lib/x86_64-linux-gnu/dri/swrast_dri.so", 1)
        if rc != 0:
            raise RuntimeError(f"drisw loader: {L.drisw_error().decode()}")
        self.gl = _GLFunctions(L.drisw_get_proc)

    # -- common ------------------------------------------------------------------------------------------------------------
    def _shader(self, kind, text):
        gl = self.gl
        if self.backend == "llvmpipe":   # a desktop core-profile context: the same shaders in GLSL 3.30
            text = text.replace("#version 300 es", "#version 330 core")
        sh = gl.glCreateShader(kind)
        src = ctypes.c_char_p(text.encode())
        gl.glShaderSource(sh, 1, ctypes.byref(src), None)
        gl.glCompileShader(sh)
        ok = _i(0)
        gl.glGetShaderiv(sh, GL_COMPILE_STATUS, ctypes.byref(ok))
        if not ok.value:
            log = ctypes.create_string_buffer(4096)
            gl.glGetShaderInfoLog(sh, 4096, None, log)
            raise RuntimeError(f"shader compile failed: {log.value.decode()}")
        return sh

    def _make_program(self, vertex_shader):
        gl = self.gl
        prog = gl.glCreateProgram()
        gl.glAttachShader(prog, self._shader(GL_VERTEX_SHADER, vertex_shader))
        gl.glAttachShader(prog, self._shader(GL_FRAGMENT_SHADER, _FS))
        gl.glBindAttribLocation(prog, 0, b"pos")
        gl.glBindAttribLocation(prog, 1, b"col")
        gl.glLinkProgram(prog)
        ok = _i(0)
        gl.glGetProgramiv(prog, GL_LINK_STATUS, ctypes.byref(ok))
        if not ok.value:
            log = ctypes.create_string_buffer(4096)
            gl.glGetProgramInfoLog(prog, 4096, None, log)
            raise RuntimeError(f"program link failed: {log.value.decode()}")
        return prog

    def _target(self, w, h, depth_format):
        """A (w, h) RGBA8 + depth framebuffer object, kept between calls of one size."""
        gl = self.gl
        if self._fbo_size == (w, h, depth_format):
            return
        if self._fbo is not None:
            fbo, rbs = self._fbo
            gl.glDeleteFramebuffers(1, ctypes.byref(fbo))
            gl.glDeleteRenderbuffers(2, rbs)
        fbo, rbs = _u(0), (_u * 2)()
        gl.glGenFramebuffers(1, ctypes.byref(fbo))
        gl.glBindFramebuffer(GL_FRAMEBUFFER, fbo.value)
        gl.glGenRenderbuffers(2, rbs)
        gl.glBindRenderbuffer(GL_RENDERBUFFER, rbs[0])
        gl.glRenderbufferStorage(GL_RENDERBUFFER, GL_RGBA8, w, h)
        gl.glFramebufferRenderbuffer(GL_FRAMEBUFFER, GL_COLOR_ATTACHMENT0, GL_RENDERBUFFER, rbs[0])
        gl.glBindRenderbuffer(GL_RENDERBUFFER, rbs[1])
        gl.glRenderbufferStorage(GL_RENDERBUFFER, depth_format, w, h)
        gl.glFramebufferRenderbuffer(GL_FRAMEBUFFER, GL_DEPTH_ATTACHMENT, GL_RENDERBUFFER, rbs[1])
        st = gl.glCheckFramebufferStatus(GL_FRAMEBUFFER)
        if st != GL_FRAMEBUFFER_COMPLETE:
            raise RuntimeError(f"framebuffer incomplete: 0x{st:x}")
        self._fbo, self._fbo_size = (fbo, rbs), (w, h, depth_format)

    def upload_mesh(self, verts, faces):
        """Keeps the (F, 3, 3) float32 corner positions in world space and the base-256 id colours (meshes.py:1783-1798)."""
        verts = np.ascontiguousarray(verts, dtype=np.float32)
        faces = np.ascontiguousarray(faces, dtype=np.int64)
        self._corners = verts[faces]                                   # (F, 3, 3) float32
        F = faces.shape[0]
        ids = np.arange(F, dtype=np.int64)
        col = np.stack([(ids >> (8 * k)) & 255 for k in range(3)] + [np.full(F, 255)], axis=1).astype(np.uint8)
        self._colors = np.ascontiguousarray(np.repeat(col[:, None, :], 3, axis=1))   # every corner: provoking-vertex proof
        self.n_faces = F

    def render_ids(self, cam, h, w, far=None, depth_func=GL_LEQUAL, depth_format=GL_DEPTH_COMPONENT24, transform="camera"):
        """One view -> (h, w) int32 face ids, -1 = background.  cam: the 16-float record of include/geograster.h.

        transform="camera": positions in camera space (R1 in float32, see the module docstring).
        transform="vtk_matrix": WORLD-space float32 positions and ONE float32 4x4 matrix applied by the vertex shader, the
            way VTK's polydata mapper does it (MCDCMatrix = projection . view, composed in float64 from the pyvista
            camera's position / focal point / view-up / vertical view angle, cameras.py:446-477, uploaded as float32)."""
        gl = self.gl
        cam = np.asarray(cam, dtype=np.float32).reshape(16)
        R, t = cam[0:9].reshape(3, 3), cam[9:12]
        f_eff, cxp, cyp, near = (np.float32(cam[12]), float(cam[13]), float(cam[14]), float(cam[15]))
        if abs(cxp - w / 2.0) > 1e-6 or abs(cyp - h / 2.0) > 1e-6:
            raise ValueError("the pyvista camera of the reference has its principal point at the window centre")
        # R1 of DESIGN.md in float32, each operation rounded individually: d = p - t; q_c = (R_0c d_x + R_1c d_y) + R_2c d_z
        d = self._corners - t[None, None, :]
        q = np.empty_like(d)
        for c in range(3):
            q[..., c] = (R[0, c] * d[..., 0] + R[1, c] * d[..., 1]) + R[2, c] * d[..., 2]
        if far is None:
            zmax = float(np.nanmax(q[..., 2]))
            far = max(2.0 * zmax, 10.0 * near)
        A = (far + near) / (far - near)
        B = -2.0 * far * near / (far - near)
        self._target(w, h, depth_format)
        gl.glViewport(0, 0, w, h)
        for cap in (GL_CULL_FACE, GL_DITHER, GL_BLEND):
            gl.glDisable(cap)
        if self.backend == "llvmpipe":
            gl.glDisable(GL_MULTISAMPLE)
        gl.glEnable(GL_DEPTH_TEST)
        gl.glDepthFunc(depth_func)
        gl.glClearColor(1.0, 1.0, 1.0, 1.0)
        gl.glClearDepthf(1.0)
        gl.glClear(GL_COLOR_BUFFER_BIT | GL_DEPTH_BUFFER_BIT)
        if transform == "camera":
            gl.glUseProgram(self._prog)
            gl.glUniform4f(self._uproj, float(2.0 * f_eff / w), float(-2.0 * f_eff / h), float(A), float(B))
            pos = np.ascontiguousarray(q.reshape(-1, 3))
        elif transform == "vtk_matrix":
            R64, t64 = R.astype(np.float64), t.astype(np.float64)
            right, up, fwd = R64[:, 0], -R64[:, 1], R64[:, 2]          # view-up of cameras.py:468 is R . (0, -1, 0)
            view = np.eye(4)
            view[0, :3], view[1, :3], view[2, :3] = right, up, -fwd    # eye space looks down -z
            view[:3, 3] = -view[:3, :3] @ t64
            fe = float(f_eff)
            proj = np.array([[2.0 * fe / w, 0, 0, 0], [0, 2.0 * fe / h, 0, 0], [0, 0, -A, B], [0, 0, -1.0, 0]])
            mcdc = np.ascontiguousarray((proj @ view).T.astype(np.float32))   # column-major for GL
            gl.glUseProgram(self._prog_matrix)
            gl.glUniformMatrix4fv(self._umcdc, 1, 0, mcdc.ctypes.data_as(_vp))
            pos = np.ascontiguousarray(self._corners.reshape(-1, 3))
        else:
            raise ValueError(transform)
        bufs = (_u * 2)()
        gl.glGenBuffers(2, bufs)
        gl.glBindBuffer(GL_ARRAY_BUFFER, bufs[0])
        gl.glBufferData(GL_ARRAY_BUFFER, pos.nbytes, pos.ctypes.data_as(_vp), GL_STATIC_DRAW)
        gl.glEnableVertexAttribArray(0)
        gl.glVertexAttribPointer(0, 3, GL_FLOAT, 0, 0, None)
        gl.glBindBuffer(GL_ARRAY_BUFFER, bufs[1])
        gl.glBufferData(GL_ARRAY_BUFFER, self._colors.nbytes, self._colors.ctypes.data_as(_vp), GL_STATIC_DRAW)
        gl.glEnableVertexAttribArray(1)
        gl.glVertexAttribPointer(1, 4, GL_UNSIGNED_BYTE, 1, 0, None)
        n = 3 * self.n_faces
        step = 3 * (1 << 20)   # draw in id order, a million faces per call
        for first in range(0, n, step):
            gl.glDrawArrays(GL_TRIANGLES, first, min(step, n - first))
        gl.glFinish()
        out = np.empty((h, w, 4), dtype=np.uint8)
        gl.glPixelStorei(GL_PACK_ALIGNMENT, 1)
        gl.glReadPixels(0, 0, w, h, GL_RGBA, GL_UNSIGNED_BYTE, out.ctypes.data_as(_vp))
        err = gl.glGetError()
        gl.glDeleteBuffers(2, bufs)
        if err:
            raise RuntimeError(f"GL error 0x{err:x}")
        out = out[::-1]                                                # GL rows run bottom-up
        ids = out[..., 0].astype(np.int64) + (out[..., 1].astype(np.int64) << 8) + (out[..., 2].astype(np.int64) << 16)
        ids[ids > self.n_faces] = -1                                   # meshes.py:1836 (sic: >, not >=)
        return ids.astype(np.int32)


def build_drisw_loader() -> Path:
    out = _HERE / "_build" / "libdrisw_loader.so"
    src = _HERE / "drisw_loader.c"
    if not out.is_file() or out.stat().st_mtime < src.stat().st_mtime:
        out.parent.mkdir(exist_ok=True)
        subprocess.run(["gcc", "-O2", "-shared", "-fPIC", "-o", str(out), str(src), "-ldl"], check=True)
    return out
