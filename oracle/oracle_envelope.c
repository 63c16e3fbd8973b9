/*
 * oracle/oracle_envelope.c -- TEST INFRASTRUCTURE ONLY (CPU oracle, never shipped, never on the product path).
 *
 * What can be said about exact per-pixel face ids WITHOUT the reference's VTK stack
 * (geograypher/meshes/meshes.py:1776-1836 renders through vtk==9.2.6, which is not in this image.  Two GL implementations
 * are -- Mesa llvmpipe, GL_SUBPIXEL_BITS = 8, and SwiftShader, GL_SUBPIXEL_BITS = 4 -- and tests/test_gl_pin.py holds this
 * file's classification against renders of both: 0 disagreements on the pixels it calls implementation-independent when
 * `delta` is set for the implementation's own grid, 2^-bits + 2e-3.  With the 8-bit default, SwiftShader's 4-bit grid
 * disagrees on thousands of "independent" pixels: the default claims nothing about implementations coarser than 8 bits.)
 * OpenGL 4.6 (14.6.1, 13.7) fixes what a rasterizer must do only up to three implementation choices: how window
 * coordinates are snapped to the sub-pixel grid (and how fine that grid is: at least 4 bits), which of two triangles owns
 * a sample that lies exactly on their shared edge, and the precision of the depth comparison.  Every conforming
 * implementation WITH AT LEAST THE SUB-PIXEL RESOLUTION `delta` STANDS FOR -- VTK on such a driver, the HIP kernels of
 * this repository, oracle_raster.c -- must therefore produce the SAME face id at a pixel whenever none of the three
 * choices can matter there.  This file provides
 *
 *   orc_envelope      the classification: a pixel is IMPLEMENTATION-INDEPENDENT when, evaluated in float64 without any
 *                     snapping, (1) its centre is farther than `delta` pixels (default 1/256 + 2e-3: one sub-pixel step of
 *                     the coarsest grid OpenGL allows for 8 sub-pixel bits, plus the fp32 vertex-transform error at
 *                     4000-pixel focal lengths) from every edge of every triangle that could reach it, OR the triangles
 *                     it is that close to lie clearly behind the winner; (2) the nearest surely-covering triangle is nearer
 *                     than any other triangle that could cover the pixel by a relative depth gap `gap_rel` (default 1e-5,
 *                     two orders above the fp32 error of 1/z) BEYOND the depth uncertainty of both triangles at that
 *                     pixel.  That uncertainty has two sources an implementation is free about: moving the vertices by
 *                     up to `delta` (snapping) tilts the triangle's depth plane -- by up to (max 1/z - min 1/z) x
 *                     4 delta / (smallest altitude), everything for a needle-shaped sliver --, and evaluating the plane
 *                     as anchor + gradient . offset in 32-bit floats loses (|anchor| + |A dx| + |B dy|) x 2^-21, far more
 *                     than the result's own rounding when the two gradient terms cancel (the same slivers).  Round 2
 *                     ignored both and 1-4 pixels per view of a forest scene (20 000 cone-and-cylinder trees seen
 *                     obliquely: thousands of slivers) were claimed that the rule-set oracle decides differently.
 *                     All other pixels are IMPLEMENTATION-DEFINED: two
 *                     conforming rasterizers may disagree there, and a comparison with VTK cannot ask for equality.
 *   orc_raster_float  a SECOND, independently written rasterizer with a different conforming convention: no sub-pixel
 *                     snapping at all (float64 edge functions of the float64 projections), closed triangles (a sample on
 *                     a shared edge is covered by both neighbours), perspective-correct depth from barycentric
 *                     interpolation of 1/z, nearest wins, exact ties go to the HIGHER face id (the opposite of R5).  It
 *                     shares no arithmetic with oracle_raster.c beyond the camera model of cameras.py:446-477.
 *
 * tests/test_envelope.py: both oracles and the HIP path agree on every implementation-independent pixel of the C1 and C2
 * views; the implementation-defined fraction is reported.
 * Limit: faces that straddle the near plane are not classified -- a view that contains one is reported through the
 * return value and its pixels inside the face's clipped outline must not be claimed (none in the BASELINE scenes).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct { double x, y, iz; int ok; } env_vtx;

/* float64 projection of an fp32 vertex with an fp32 camera record (include/geograster.h: R (9), t (3), f_eff, cxp, cyp, near) */
static env_vtx env_project(const float *p, const float *cam) {
  env_vtx v;
  const double dx = (double)p[0] - (double)cam[9], dy = (double)p[1] - (double)cam[10], dz = (double)p[2] - (double)cam[11];
  const double qx = (double)cam[0] * dx + (double)cam[3] * dy + (double)cam[6] * dz;
  const double qy = (double)cam[1] * dx + (double)cam[4] * dy + (double)cam[7] * dz;
  const double qz = (double)cam[2] * dx + (double)cam[5] * dy + (double)cam[8] * dz;
  v.ok = (qz > (double)cam[15]) && isfinite(qx) && isfinite(qy) && isfinite(qz);
  v.x = v.y = v.iz = 0.0;
  if (!v.ok) return v;
  v.iz = 1.0 / qz;
  v.x = (double)cam[13] + (double)cam[12] * qx * v.iz; /* pixel units, pixel centres at j + 0.5 */
  v.y = (double)cam[14] + (double)cam[12] * qy * v.iz;
  return v;
}

/* pixels j in [0, n) whose centre j + 0.5 lies in [lo, hi]; the clamp comes first (a projection may be astronomically far) */
static void env_pixel_range(double lo, double hi, int n, int *j0, int *j1) {
  lo = fmin(fmax(lo, -1.0), (double)n + 1.0);
  hi = fmin(fmax(hi, -1.0), (double)n + 1.0);
  *j0 = (int)ceil(lo - 0.5);
  *j1 = (int)floor(hi - 0.5);
  if (*j0 < 0) *j0 = 0;
  if (*j1 > n - 1) *j1 = n - 1;
}

/* Returns the number of faces that straddle the near plane (not classified).  cls: 0 independent background,
 * 1 independent face (ids holds it), 2 implementation-defined.  work: caller scratch, 4 doubles + 2 int32 per pixel. */
int orc_envelope(const float *verts, const int32_t *faces, int64_t F, const float *cam, int h, int w, double delta,
                 double gap_rel, uint8_t *cls, int32_t *ids, double *zA, double *zB, int32_t *fA, uint8_t *sureA,
                 double *zAlo, double *zAhi) {
  /* zA: nominal depth of the nearest candidate, [zAlo, zAhi] its uncertainty interval; zB: the smallest LOWER bound of any
   * other candidate's depth */
  const int64_t n = (int64_t)h * w;
  for (int64_t p = 0; p < n; ++p) { zA[p] = INFINITY; zB[p] = INFINITY; zAlo[p] = INFINITY; zAhi[p] = INFINITY; fA[p] = -1; sureA[p] = 0; }
  int straddle = 0;
  for (int64_t f = 0; f < F; ++f) {
    const env_vtx a = env_project(verts + 3 * (int64_t)faces[3 * f], cam);
    const env_vtx b = env_project(verts + 3 * (int64_t)faces[3 * f + 1], cam);
    const env_vtx c = env_project(verts + 3 * (int64_t)faces[3 * f + 2], cam);
    if (!a.ok || !b.ok || !c.ok) { if (a.ok || b.ok || c.ok) ++straddle; continue; }
    double area = (b.x - a.x) * (c.y - a.y) - (c.x - a.x) * (b.y - a.y);
    if (fabs(area) < 1e-300) continue;
    env_vtx v0 = a, v1 = b, v2 = c;
    if (area < 0) { v1 = c; v2 = b; area = -area; }
    const double xmin = fmin(v0.x, fmin(v1.x, v2.x)) - delta, xmax = fmax(v0.x, fmax(v1.x, v2.x)) + delta;
    const double ymin = fmin(v0.y, fmin(v1.y, v2.y)) - delta, ymax = fmax(v0.y, fmax(v1.y, v2.y)) + delta;
    int j0, j1, i0, i1;
    env_pixel_range(xmin, xmax, w, &j0, &j1);
    env_pixel_range(ymin, ymax, h, &i0, &i1);
    if (j0 > j1 || i0 > i1) continue;
    const env_vtx *vs[3] = {&v0, &v1, &v2};
    double ex[3], ey[3], len[3], maxlen = 0.0;
    for (int k = 0; k < 3; ++k) {
      const env_vtx *p0 = vs[k], *p1 = vs[(k + 1) % 3];
      ex[k] = p1->x - p0->x; ey[k] = p1->y - p0->y;
      len[k] = sqrt(ex[k] * ex[k] + ey[k] * ey[k]);
      if (len[k] > maxlen) maxlen = len[k];
    }
    /* depth uncertainty of this triangle: (a) vertices moved by delta tilt the plane */
    const double izmin = fmin(v0.iz, fmin(v1.iz, v2.iz)), izmax = fmax(v0.iz, fmax(v1.iz, v2.iz));
    const double hmin = maxlen > 0 ? area / maxlen : 0.0;                /* smallest altitude, pixels */
    const double tilt = (izmax - izmin) * (hmin > 4.0 * delta ? 4.0 * delta / hmin : 1.0);
    /* (b) plane form in fp32: gradients of 1/z per pixel */
    const double gA = ((v1.iz - v0.iz) * (v2.y - v0.y) - (v2.iz - v0.iz) * (v1.y - v0.y)) / area;
    const double gB = ((v2.iz - v0.iz) * (v1.x - v0.x) - (v1.iz - v0.iz) * (v2.x - v0.x)) / area;
    for (int i = i0; i <= i1; ++i) {
      for (int j = j0; j <= j1; ++j) {
        const double px = j + 0.5, py = i + 0.5;
        double dmin = INFINITY, bary[3];
        for (int k = 0; k < 3; ++k) {
          const env_vtx *p0 = vs[k];
          const double e = ex[k] * (py - p0->y) - ey[k] * (px - p0->x); /* > 0 inside (area > 0, y down) */
          bary[(k + 2) % 3] = e / area;
          const double d = len[k] > 0 ? e / len[k] : -INFINITY;        /* signed distance to the edge line, pixels */
          if (d < dmin) dmin = d;
        }
        if (dmin < -delta) continue;                                    /* cannot cover under any convention */
        const int sure = dmin > delta;
        const double iz = bary[0] * v0.iz + bary[1] * v1.iz + bary[2] * v2.iz; /* 1/z is affine in window space */
        const double z = iz > 0 ? 1.0 / iz : INFINITY;
        double mag = 0.0;                                               /* largest anchor + |A dx| + |B dy| over the anchors */
        for (int k = 0; k < 3; ++k) {
          const double m = fabs(vs[k]->iz) + fabs(gA * (px - vs[k]->x)) + fabs(gB * (py - vs[k]->y));
          if (m > mag) mag = m;
        }
        const double err = tilt + mag * 0x1p-21;
        const double zlo = 1.0 / (iz + err), zhi = iz - err > 0 ? 1.0 / (iz - err) : INFINITY;
        const int64_t p = (int64_t)i * w + j;
        if (z < zA[p]) {
          if (zAlo[p] < zB[p]) zB[p] = zAlo[p];
          zA[p] = z; zAlo[p] = zlo; zAhi[p] = zhi; fA[p] = (int32_t)f; sureA[p] = (uint8_t)sure;
        } else if (zlo < zB[p]) zB[p] = zlo;
      }
    }
  }
  for (int64_t p = 0; p < n; ++p) {
    if (fA[p] < 0) { cls[p] = 0; ids[p] = -1; continue; }
    const int clear = !(zB[p] < zAhi[p] * (1.0 + gap_rel));
    if (sureA[p] && clear) { cls[p] = 1; ids[p] = fA[p]; }
    else { cls[p] = 2; ids[p] = fA[p]; }
  }
  return straddle;
}

/* Second rasterizer: un-snapped float64, closed triangles, barycentric 1/z, nearest wins, ties -> higher id.
 * Faces with a vertex at or behind the near plane are skipped (reported through the return value). */
int orc_raster_float(const float *verts, const int32_t *faces, int64_t F, const float *cam, int h, int w, int32_t *ids,
                     double *zbuf) {
  const int64_t n = (int64_t)h * w;
  for (int64_t p = 0; p < n; ++p) { ids[p] = -1; zbuf[p] = INFINITY; }
  int skipped = 0;
  for (int64_t f = 0; f < F; ++f) {
    env_vtx v[3];
    int ok = 1, any = 0;
    for (int k = 0; k < 3; ++k) { v[k] = env_project(verts + 3 * (int64_t)faces[3 * f + k], cam); ok &= v[k].ok; any |= v[k].ok; }
    if (!ok) { skipped += any; continue; }
    /* twice the signed area; orientation is normalised by dividing the barycentric numerators by it */
    const double area = (v[1].x - v[0].x) * (v[2].y - v[0].y) - (v[2].x - v[0].x) * (v[1].y - v[0].y);
    if (area == 0.0) continue;
    const double xmin = fmin(v[0].x, fmin(v[1].x, v[2].x)), xmax = fmax(v[0].x, fmax(v[1].x, v[2].x));
    const double ymin = fmin(v[0].y, fmin(v[1].y, v[2].y)), ymax = fmax(v[0].y, fmax(v[1].y, v[2].y));
    int j0, j1, i0, i1;
    env_pixel_range(xmin, xmax, w, &j0, &j1);
    env_pixel_range(ymin, ymax, h, &i0, &i1);
    for (int i = i0; i <= i1; ++i) {
      for (int j = j0; j <= j1; ++j) {
        const double px = j + 0.5, py = i + 0.5;
        /* barycentric coordinates of the sample: b_k = area(sample, v_{k+1}, v_{k+2}) / area(v0, v1, v2) */
        const double b0 = ((v[1].x - px) * (v[2].y - py) - (v[2].x - px) * (v[1].y - py)) / area;
        const double b1 = ((v[2].x - px) * (v[0].y - py) - (v[0].x - px) * (v[2].y - py)) / area;
        const double b2 = 1.0 - b0 - b1;
        if (b0 < 0.0 || b1 < 0.0 || b2 < 0.0) continue;                /* closed triangle: edges included */
        const double iz = b0 * v[0].iz + b1 * v[1].iz + b2 * v[2].iz;
        if (!(iz > 0.0)) continue;
        const double z = 1.0 / iz;
        const int64_t p = (int64_t)i * w + j;
        if (z < zbuf[p] || (z == zbuf[p] && (int32_t)f > ids[p])) { zbuf[p] = z; ids[p] = (int32_t)f; }
      }
    }
  }
  return skipped;
}
