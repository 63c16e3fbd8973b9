/*
 * oracle/oracle_raster.c -- TEST INFRASTRUCTURE ONLY (CPU oracle, never shipped, never on the product path).
 *
 * Plain-C restatement of the face-ID rasterization stage of the reference hot path:
 *   geograypher/meshes/meshes.py:1678-1856   TexturedPhotogrammetryMesh.pix2face (single-camera branch)
 *   geograypher/cameras/cameras.py:446-477   PhotogrammetryCamera.get_pyvista_camera (view parameters)
 *   geograypher/cameras/cameras.py:179-200   get_image_size  (h, w) = (int(H*s), int(W*s))
 *
 * The arithmetic of that stage lives in a third-party dependency that is absent from /root/reference and
 * from this image: vtk==9.2.6 driven through pyvista==0.42.2 (poetry.lock:3361-3362, 2342-2343), i.e. an
 * OpenGL polygon rasterizer (whichever GL implementation the host provides; Mesa's in the reference's Dockerfile).  What is restated here is the *published* algorithm of that stage (OpenGL 4.6
 * core spec 14.6.1 "Basic Polygon Rasterization": point sampling at pixel centres, a consistent
 * shared-edge fill rule, nearest-depth-wins with window-space-linear depth) as the fixed rule-set R0-R7 of
 * DESIGN.md.  The HIP kernels implement the same rule-set independently; tests require bit equality.
 *
 * PARITY STATUS: pinned twice.
 *  (1) To the reference's own known-answer tests for this stage (tests/test_derived_meshes.py:23-76 pixel colours;
 *      tests/test_derived_cameras.py:339-415 shape/dtype/range properties), restated in tests/test_reference_kats.py.
 *  (2) To two REAL third-party OpenGL rasterizers found in the build image -- Mesa 23.2.1 llvmpipe (8 sub-pixel bits; the
 *      software GL family the reference's Dockerfile:6-13 installs) and Google SwiftShader 4.1 (ES 3.0, 4 sub-pixel bits) --
 *      which rendered the id image the way meshes.py:1776-1836 does (tests/golden/make_golden_gl.py -> reference_gl_*.npz;
 *      tests/test_gl_pin.py): identical on EVERY pixel that oracle_envelope.c calls implementation-independent at
 *      delta = 2^-bits + 2e-3 px, identical on 99.996-99.998 % of ALL pixels of full-size C2 / C5 views against llvmpipe
 *      (profiles/r05_gl_pin.log), every differing pixel an edge pixel.
 *  What stays unpinned is VTK ITSELF (its shaders and matrices on top of the GL implementation): vtk==9.2.6 is not in the
 *  image and the reference holds no golden pix2face array.
 *
 * Two entry points compute the same function:
 *   orc_raster_spec  -- literal rule-set, every pixel of the bounding box evaluated from the closed forms.
 *   orc_raster_fast  -- same results, incremental integer edge stepping (exact), used for timing the CPU
 *                       baseline and for full-size parity runs.  tests/ check fast == spec.
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off: no FMA contraction, SSE2 scalar IEEE arithmetic).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_SUB 256          /* sub-pixel units per pixel (8 fractional bits)                  R1 */
#define ORC_HALF 128         /* pixel centre offset                                             R2 */
#define ORC_GUARD 16384.0f   /* |screen coordinate| must be below this many pixels              R1 */

typedef struct {
  int32_t X, Y; /* snapped window coordinates, sub-pixel units */
  float iz;     /* 1 / camera-space depth                      */
  int valid;
} orc_vtx;

/* R1: vertex transform, fp32, every operation individually rounded (no fused multiply-add).
 * cam[0..8]  R = cam_to_world rotation (row-major), cam[9..11] t = camera position,
 * cam[12] f_eff (pixels), cam[13] cxp, cam[14] cyp (principal point in pixels of the rendered image),
 * cam[15] near.   q = R^T (p - t); camera frame +X right, +Y down, +Z forward (cameras.py:446-477). */
/* R1-GL (round 6, opt-in: orc_set_vertex_order(1)): the second half of the vertex stage in the ORDER OF OPERATIONS of an OpenGL
 * pipeline, as Mesa 23.2's llvmpipe executes it behind a vertex shader that multiplies the camera-space point with the
 * projection's diagonal (tests/golden/gl_raster.py; src/gallium/auxiliary/draw/draw_llvm.c generate_viewport;
 * src/gallium/drivers/llvmpipe/lp_setup_tri.c):  clip = (P_x q_x, P_y q_y, ., q_z) with P_x = 2 f / w, P_y = -2 f / h;
 * rw = 1 / q_z;  ndc = clip * rw;  win = fma(ndc, size / 2, size / 2) (llvm.fmuladd: fused on every x86 host with FMA3);
 * fixed = lrintf(256 (win - 0.5)) (round half to even), rows bottom-up.  95 % of the pixels on which rule R1 and llvmpipe
 * disagree are vertices that land on the neighbouring 1/256 px step under this order (profiles/r06_gl_residue.txt).  The
 * principal point must be the window centre (the pyvista camera's, cameras.py:446-477). */
static int orc_vertex_order = 0;
void orc_set_vertex_order(int gl_order) { orc_vertex_order = gl_order; }
static int orc_h = 0, orc_w = 0;   /* image size of the running raster call (the GL order needs the viewport) */

static int orc_snap(float qx, float qy, float qz, const float *cam, int32_t *X, int32_t *Y, float *izp) {
  float iz = 1.0f / qz;
  *izp = iz;
  if (!orc_vertex_order) {
    float fx = cam[12] * qx;
    float fy = cam[12] * qy;
    float sx = cam[13] + fx * iz;
    float sy = cam[14] + fy * iz;
    if (!(fabsf(sx) < ORC_GUARD) || !(fabsf(sy) < ORC_GUARD)) return 0;
    *X = (int32_t)floorf(sx * 256.0f + 0.5f);
    *Y = (int32_t)floorf(sy * 256.0f + 0.5f);
    return 1;
  }
  const float two_f = 2.0f * cam[12];
  const float px = two_f / (float)orc_w, py = -(two_f / (float)orc_h);
  const float hw = 0.5f * (float)orc_w, hh = 0.5f * (float)orc_h;
  float xc = px * qx, yc = py * qy;
  float xn = xc * iz, yn = yc * iz;
  float xw = fmaf(xn, hw, hw), yw = fmaf(yn, hh, hh);
  if (!(fabsf(xw) < ORC_GUARD) || !(fabsf(yw) < ORC_GUARD)) return 0;
  float fx = (xw - 0.5f) * 256.0f, fy = (yw - 0.5f) * 256.0f;
  *X = (int32_t)rintf(fx) + ORC_HALF;                      /* rintf: round half to even (default rounding mode), like lrintf / cvtps2dq */
  *Y = 256 * orc_h - ORC_HALF - (int32_t)rintf(fy);
  return 1;
}

static orc_vtx orc_project(const float *p, const float *cam) {
  orc_vtx v;
  float dx = p[0] - cam[9];
  float dy = p[1] - cam[10];
  float dz = p[2] - cam[11];
  float m0, m1, m2;
  m0 = cam[0] * dx; m1 = cam[3] * dy; m2 = cam[6] * dz;
  float qx = (m0 + m1) + m2;
  m0 = cam[1] * dx; m1 = cam[4] * dy; m2 = cam[7] * dz;
  float qy = (m0 + m1) + m2;
  m0 = cam[2] * dx; m1 = cam[5] * dy; m2 = cam[8] * dz;
  float qz = (m0 + m1) + m2;
  v.valid = (qz > cam[15]) ? 1 : 0; /* false for NaN */
  v.X = 0; v.Y = 0; v.iz = 0.0f;
  if (!v.valid) return v;
  float iz;
  if (!orc_snap(qx, qy, qz, cam, &v.X, &v.Y, &iz)) { v.valid = 0; v.X = 0; v.Y = 0; return v; }
  v.iz = iz;
  return v;
}

typedef struct {
  int32_t X[3], Y[3];
  float iz0, A, B;
  int32_t jmin, jmax, imin, imax; /* inclusive pixel bounding box, clamped to the image */
} orc_tri;

static inline int32_t orc_min3(int32_t a, int32_t b, int32_t c) { int32_t m = a < b ? a : b; return m < c ? m : c; }
static inline int32_t orc_max3(int32_t a, int32_t b, int32_t c) { int32_t m = a > b ? a : b; return m > c ? m : c; }

/* R2 + R4 setup from three snapped vertices. Returns 0 when the triangle is discarded. */
static int orc_setup_snapped(orc_vtx v0, orc_vtx v1, orc_vtx v2, int h, int w, orc_tri *t) {
  int64_t area2 = (int64_t)(v1.X - v0.X) * (int64_t)(v2.Y - v0.Y) - (int64_t)(v2.X - v0.X) * (int64_t)(v1.Y - v0.Y);
  if (area2 == 0) return 0;
  if (area2 < 0) { orc_vtx s = v1; v1 = v2; v2 = s; area2 = -area2; } /* both windings are drawn */
  t->X[0] = v0.X; t->X[1] = v1.X; t->X[2] = v2.X;
  t->Y[0] = v0.Y; t->Y[1] = v1.Y; t->Y[2] = v2.Y;
  int32_t Xmin = orc_min3(v0.X, v1.X, v2.X), Xmax = orc_max3(v0.X, v1.X, v2.X);
  int32_t Ymin = orc_min3(v0.Y, v1.Y, v2.Y), Ymax = orc_max3(v0.Y, v1.Y, v2.Y);
  /* pixel j has its centre at 256*j+128: keep j with Xmin <= 256j+128 <= Xmax (arithmetic shifts = floor) */
  int32_t jmin = (Xmin - ORC_HALF + (ORC_SUB - 1)) >> 8;
  int32_t jmax = (Xmax - ORC_HALF) >> 8;
  int32_t imin = (Ymin - ORC_HALF + (ORC_SUB - 1)) >> 8;
  int32_t imax = (Ymax - ORC_HALF) >> 8;
  if (jmin < 0) jmin = 0;
  if (imin < 0) imin = 0;
  if (jmax > w - 1) jmax = w - 1;
  if (imax > h - 1) imax = h - 1;
  if (jmin > jmax || imin > imax) return 0;
  t->jmin = jmin; t->jmax = jmax; t->imin = imin; t->imax = imax;
  /* R4: plane of 1/z over the window, gradients in double then rounded to float */
  double d1 = (double)v1.iz - (double)v0.iz;
  double d2 = (double)v2.iz - (double)v0.iz;
  double a2 = (double)area2;
  double n1, n2;
  n1 = d1 * (double)(v2.Y - v0.Y); n2 = d2 * (double)(v1.Y - v0.Y);
  double Ad = (n1 - n2) / a2;
  n1 = d2 * (double)(v1.X - v0.X); n2 = d1 * (double)(v2.X - v0.X);
  double Bd = (n1 - n2) / a2;
  t->iz0 = v0.iz; t->A = (float)Ad; t->B = (float)Bd;
  return 1;
}

/* R7: a face that is partly in front of the near plane or partly inside the guard band is CLIPPED, as the OpenGL
 * pipeline clips primitives to the view volume (OpenGL 4.6 core 13.7), instead of being dropped: Sutherland-Hodgman in
 * camera space against the near plane z >= near and the four guard planes |s| <= 16383 px (one pixel inside the
 * validity limit of R1, so that the rounded projections of the new vertices stay valid), in that order, in double
 * precision with every operation individually rounded.  A crossing is always computed from the inside vertex towards
 * the outside one, so two faces that share an edge compute the same new vertex.  The clipped polygon (<= 8 vertices)
 * is rounded to fp32 camera-space points, projected like R1, and drawn as a triangle fan; every triangle of the fan
 * carries the face's id and its own 1/z plane (R4).  Faces entirely behind the near plane, faces with a non-finite
 * vertex and cameras with near <= 0 or f_eff <= 0 keep the old rule: dropped.  Returns the number of triangles. */
#define ORC_CLIP_G 16383.0
typedef struct { double x, y, z; } orc_p3;

static void orc_camspace(const float *p, const float *cam, float q[3]) { /* the first half of R1 */
  float dx = p[0] - cam[9];
  float dy = p[1] - cam[10];
  float dz = p[2] - cam[11];
  float m0, m1, m2;
  m0 = cam[0] * dx; m1 = cam[3] * dy; m2 = cam[6] * dz;
  q[0] = (m0 + m1) + m2;
  m0 = cam[1] * dx; m1 = cam[4] * dy; m2 = cam[7] * dz;
  q[1] = (m0 + m1) + m2;
  m0 = cam[2] * dx; m1 = cam[5] * dy; m2 = cam[8] * dz;
  q[2] = (m0 + m1) + m2;
}

static inline double orc_plane(const double *pl, orc_p3 p) {
  double t1 = pl[0] * p.x, t2 = pl[1] * p.y, t3 = pl[2] * p.z;
  return ((t1 + t2) + t3) + pl[3];
}

static inline orc_p3 orc_cross(orc_p3 in, double din, orc_p3 out, double dout) { /* from the inside vertex */
  double t = din / (din - dout);
  orc_p3 r;
  double ex = out.x - in.x, ey = out.y - in.y, ez = out.z - in.z;
  double px = t * ex, py = t * ey, pz = t * ez;
  r.x = in.x + px; r.y = in.y + py; r.z = in.z + pz;
  return r;
}

static int orc_clip_face(const float *verts, const int32_t *face, const float *cam, orc_vtx out[8]) {
  float q[3][3];
  for (int k = 0; k < 3; ++k) orc_camspace(verts + 3 * (int64_t)face[k], cam, q[k]);
  const float fe = cam[12], cxp = cam[13], cyp = cam[14], nearp = cam[15];
  if (!(nearp > 0.0f) || !(fe > 0.0f) || !isfinite(fe) || !isfinite(cxp) || !isfinite(cyp)) return 0;
  int front = 0;
  for (int k = 0; k < 3; ++k) {
    if (!isfinite(q[k][0]) || !isfinite(q[k][1]) || !isfinite(q[k][2])) return 0;
    if (q[k][2] > nearp) ++front;
  }
  if (front == 0) return 0;
  const double planes[5][4] = {
      {0.0, 0.0, 1.0, -(double)nearp},
      {-(double)fe, 0.0, ORC_CLIP_G - (double)cxp, 0.0},
      {(double)fe, 0.0, ORC_CLIP_G + (double)cxp, 0.0},
      {0.0, -(double)fe, ORC_CLIP_G - (double)cyp, 0.0},
      {0.0, (double)fe, ORC_CLIP_G + (double)cyp, 0.0},
  };
  orc_p3 a[8], b[8];
  int n = 3;
  for (int k = 0; k < 3; ++k) { a[k].x = q[k][0]; a[k].y = q[k][1]; a[k].z = q[k][2]; }
  for (int pl = 0; pl < 5 && n > 0; ++pl) {
    int m = 0;
    for (int i = 0; i < n; ++i) {
      orc_p3 S = a[i], E = a[(i + 1) % n];
      double dS = orc_plane(planes[pl], S), dE = orc_plane(planes[pl], E);
      int inS = dS >= 0.0, inE = dE >= 0.0;
      if (inS && inE) { if (m < 8) b[m] = E; ++m; }
      else if (inS && !inE) { if (m < 8) b[m] = orc_cross(S, dS, E, dE); ++m; }
      else if (!inS && inE) { if (m < 8) b[m] = orc_cross(E, dE, S, dS); ++m; if (m < 8) b[m] = E; ++m; }
    }
    if (m > 8) return 0; /* cannot happen for a convex polygon and five planes */
    n = m;
    for (int i = 0; i < n; ++i) a[i] = b[i];
  }
  if (n < 3) return 0;
  for (int i = 0; i < n; ++i) {
    float qx = (float)a[i].x, qy = (float)a[i].y, qz = (float)a[i].z;
    if (!(qz > 0.0f)) return 0;
    float iz;
    if (!orc_snap(qx, qy, qz, cam, &out[i].X, &out[i].Y, &iz)) return 0;
    out[i].iz = iz;
    out[i].valid = 1;
  }
  return n;
}

/* All triangles a face contributes (R2/R4 set up): one when its three vertices are valid, the fan of its clipped
 * polygon (R7) when some are not, none when it is dropped.  t: room for 6. */
static int orc_face_tris(const float *verts, const int32_t *face, const float *cam, int h, int w, orc_tri *t) {
  orc_vtx v0 = orc_project(verts + 3 * (int64_t)face[0], cam);
  orc_vtx v1 = orc_project(verts + 3 * (int64_t)face[1], cam);
  orc_vtx v2 = orc_project(verts + 3 * (int64_t)face[2], cam);
  if (v0.valid && v1.valid && v2.valid) return orc_setup_snapped(v0, v1, v2, h, w, t);
  orc_vtx poly[8];
  int n = orc_clip_face(verts, face, cam, poly);
  int count = 0;
  for (int k = 1; k + 1 < n; ++k) count += orc_setup_snapped(poly[0], poly[k], poly[k + 1], h, w, t + count);
  return count;
}

/* R3: edge k runs from vertex a=k to b=(k+1)%3; returns 1 when pixel centre (Px,Py) is covered */
static inline int64_t orc_edge(const orc_tri *t, int k, int64_t Px, int64_t Py) {
  int a = k, b = (k + 1) % 3;
  int64_t dx = (int64_t)t->X[b] - t->X[a], dy = (int64_t)t->Y[b] - t->Y[a];
  return dx * (Py - t->Y[a]) - dy * (Px - t->X[a]);
}
static inline int orc_owns(const orc_tri *t, int k) { /* left and BOTTOM edges own their pixels (rows top-down): the tie convention of Mesa's GL, DESIGN.md R3 */
  int a = k, b = (k + 1) % 3;
  int32_t dx = t->X[b] - t->X[a], dy = t->Y[b] - t->Y[a];
  return (dy < 0) || (dy == 0 && dx < 0);
}

/* R4: fragment depth key */
static inline int32_t orc_zbits(const orc_tri *t, int32_t Px, int32_t Py) {
  float fx = (float)(Px - t->X[0]);
  float fy = (float)(Py - t->Y[0]);
  float m0 = t->A * fx;
  float m1 = t->B * fy;
  float s = m0 + m1;
  float z = t->iz0 + s;
  int32_t zb;
  memcpy(&zb, &z, 4);
  if (zb < 1) zb = 1;
  return zb;
}

/* R5: nearest (largest 1/z) wins, equal depth -> lower face id */
static inline void orc_resolve(int32_t *zbuf, int32_t *ids, int64_t p, int32_t zb, int32_t f) {
  if (zb > zbuf[p] || (zb == zbuf[p] && f < ids[p])) { zbuf[p] = zb; ids[p] = f; }
}

static void orc_finish(const int32_t *zbuf, const int32_t *ids, float *depth, int64_t n) {
  if (!depth) return;
  for (int64_t p = 0; p < n; ++p) {
    if (ids[p] < 0) { depth[p] = INFINITY; continue; }
    float iz; memcpy(&iz, &zbuf[p], 4);
    depth[p] = 1.0f / iz;
  }
}

/* Literal rule-set. ids: int32 [h*w], background -1. depth (optional): camera-space Z, +inf background.
 * zbuf: caller scratch int32 [h*w]. */
int orc_raster_spec(const float *verts, const int32_t *faces, int64_t V, int64_t F, const float *cam, int h, int w,
                    int32_t *ids, float *depth, int32_t *zbuf) {
  (void)V;
  orc_h = h; orc_w = w;
  int64_t n = (int64_t)h * w;
  for (int64_t p = 0; p < n; ++p) { ids[p] = -1; zbuf[p] = 0; }
  for (int64_t f = 0; f < F; ++f) {
   orc_tri tris[6];
   const int ntri = orc_face_tris(verts, faces + 3 * f, cam, h, w, tris);
   for (int it = 0; it < ntri; ++it) {
    const orc_tri t = tris[it];
    int own[3] = {orc_owns(&t, 0), orc_owns(&t, 1), orc_owns(&t, 2)};
    for (int32_t i = t.imin; i <= t.imax; ++i) {
      for (int32_t j = t.jmin; j <= t.jmax; ++j) {
        int32_t Px = j * ORC_SUB + ORC_HALF, Py = i * ORC_SUB + ORC_HALF;
        int inside = 1;
        for (int k = 0; k < 3; ++k) {
          int64_t e = orc_edge(&t, k, Px, Py);
          if (!(e > 0 || (e == 0 && own[k]))) inside = 0;
        }
        if (!inside) continue;
        orc_resolve(zbuf, ids, (int64_t)i * w + j, orc_zbits(&t, Px, Py), (int32_t)f);
      }
    }
   }
  }
  orc_finish(zbuf, ids, depth, n);
  return 0;
}

/* Same function, incremental (exact integer) edge stepping along rows. */
int orc_raster_fast(const float *verts, const int32_t *faces, int64_t V, int64_t F, const float *cam, int h, int w,
                    int32_t *ids, float *depth, int32_t *zbuf) {
  (void)V;
  orc_h = h; orc_w = w;
  int64_t n = (int64_t)h * w;
  for (int64_t p = 0; p < n; ++p) { ids[p] = -1; zbuf[p] = 0; }
  for (int64_t f = 0; f < F; ++f) {
   orc_tri tris[6];
   const int ntri = orc_face_tris(verts, faces + 3 * f, cam, h, w, tris);
   for (int it = 0; it < ntri; ++it) {
    const orc_tri t = tris[it];
    int64_t bias[3], stepx[3], e_row[3];
    int32_t Px0 = t.jmin * ORC_SUB + ORC_HALF;
    for (int k = 0; k < 3; ++k) {
      int a = k, b = (k + 1) % 3;
      /* covered  <=>  e > 0 || (e == 0 && owns)  <=>  e + (owns ? 0 : -1) >= 0 */
      bias[k] = orc_owns(&t, k) ? 0 : -1;
      stepx[k] = -(int64_t)(t.Y[b] - t.Y[a]) * ORC_SUB;
      (void)a;
    }
    for (int32_t i = t.imin; i <= t.imax; ++i) {
      int32_t Py = i * ORC_SUB + ORC_HALF;
      for (int k = 0; k < 3; ++k) e_row[k] = orc_edge(&t, k, Px0, Py) + bias[k];
      int64_t e0 = e_row[0], e1 = e_row[1], e2 = e_row[2];
      int64_t p = (int64_t)i * w + t.jmin;
      for (int32_t j = t.jmin; j <= t.jmax; ++j, ++p, e0 += stepx[0], e1 += stepx[1], e2 += stepx[2]) {
        if ((e0 | e1 | e2) < 0) continue;
        orc_resolve(zbuf, ids, p, orc_zbits(&t, j * ORC_SUB + ORC_HALF, Py), (int32_t)f);
      }
    }
   }
  }
  orc_finish(zbuf, ids, depth, n);
  return 0;
}

/* n_views cameras (cams: n_views x 16), ids: n_views x h x w. Views are independent; with OpenMP they run
 * on separate threads (each with its own z-buffer). Returns the number of threads used. */
int orc_raster_views(const float *verts, const int32_t *faces, int64_t V, int64_t F, const float *cams, int n_views,
                     int h, int w, int32_t *ids, int n_threads) {
  int64_t n = (int64_t)h * w;
  int used = 1;
  orc_h = h; orc_w = w;
#ifdef _OPENMP
  if (n_threads < 1) n_threads = 1;
  used = n_threads;
#pragma omp parallel for schedule(dynamic, 1) num_threads(n_threads)
#endif
  for (int v = 0; v < n_views; ++v) {
    int32_t *zbuf = (int32_t *)malloc(sizeof(int32_t) * (size_t)n);
    orc_raster_fast(verts, faces, V, F, cams + 16 * (int64_t)v, h, w, ids + (int64_t)v * n, NULL, zbuf);
    free(zbuf);
  }
  (void)n_threads;
  return used;
}

/* ---- aggregation stage restated in C (meshes.py:1987-2002, 2044-2084) for uint8 index labels ---------------
 * Per view: textured_faces[pix2face.flatten()] = one_hot(label)  => the LAST pixel in row-major order wins per
 * face; a pix2face of -1 indexes the LAST face (reference's own TODO, meshes.py:1998-2001) when compat != 0.
 * Across views: votes[f][c] += one_hot, counts[f] += 1 for every face touched.
 * labels >= C (e.g. 255 = ignore) give an all-zero one-hot row that still counts (predictors/segmentor.py:37-69).
 * winner: caller scratch int64 [F]. */
int orc_project_labels(const int32_t *ids, const uint8_t *labels, int h, int w, int64_t F, int C, int compat,
                       uint32_t *votes, uint32_t *counts, int64_t *winner) {
  int64_t n = (int64_t)h * w;
  for (int64_t f = 0; f < F; ++f) winner[f] = -1;
  for (int64_t p = 0; p < n; ++p) {
    int64_t f = ids[p];
    if (f < 0) { if (!compat) continue; f = F + f; }
    if (f < 0 || f >= F) return -1;
    winner[f] = p; /* row-major sweep: later pixels overwrite earlier ones */
  }
  for (int64_t f = 0; f < F; ++f) {
    if (winner[f] < 0) continue;
    uint8_t l = labels[winner[f]];
    if ((int)l < C) votes[f * C + l] += 1;
    counts[f] += 1;
  }
  return 0;
}
