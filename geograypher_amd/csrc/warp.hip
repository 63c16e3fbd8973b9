// geograypher_amd/csrc/warp.hip -- row f1: distortion warp through a cached sampling map, dense Newton inverse of the Metashape
// lens model.
#include "gr_internal.hpp"

using namespace grimpl;

namespace {

// K8  distortion warp (row f1): out[i][j] = in[nearest(map_r[i][j]), nearest(map_c[i][j])] or fill.
//     Replaces skimage.transform.warp(order=0, mode="constant") driven by utils/image.py:72-126 on the face-id image
//     (meshes.py:1842-1854).  Nearest = floor(x + 0.5) (scipy.ndimage.map_coordinates, order 0); a sample outside the
//     input reads `fill`.  roundtrip != 0 reproduces the reference's float rescale + truncation (image.py:102, 123)
//     bit for bit: v -> trunc(((v - lo) / range) * range + lo) in double precision.
__global__ __launch_bounds__(256) void k_warp_nearest_i32(const int32_t *__restrict__ in, int h_in, int w_in,
                                                          const double *__restrict__ map_r,
                                                          const double *__restrict__ map_c, int64_t n_out, int32_t fill,
                                                          int roundtrip, double lo, double range,
                                                          int32_t *__restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n_out) return;
  const double r = floor(map_r[i] + 0.5), c = floor(map_c[i] + 0.5);
  int32_t v = fill;
  if (r >= 0.0 && r < (double)h_in && c >= 0.0 && c < (double)w_in) v = in[(int64_t)r * w_in + (int64_t)c];
  if (roundtrip) {
    const double t = ((double)v - lo) / range;
    const double o = t * range + lo;
    v = (int32_t)o;  // C truncation, as numpy's astype
  }
  out[i] = v;
}

// float64 images, C channels: order 0 (nearest) or 1 (bilinear, samples outside the input read `fill`: scipy's
// "grid-constant" boundary as used by the skimage version the reference pins).
__global__ __launch_bounds__(256) void k_warp_f64(const double *__restrict__ in, int h_in, int w_in, int C,
                                                  const double *__restrict__ map_r, const double *__restrict__ map_c,
                                                  int64_t n_out, int order, double fill, double *__restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n_out * C) return;
  const int64_t p = i / C;
  const int ch = (int)(i - p * C);
  const double mr = map_r[p], mc = map_c[p];
  auto at = [&](double rr, double cc) -> double {
    if (rr >= 0.0 && rr < (double)h_in && cc >= 0.0 && cc < (double)w_in)
      return in[((int64_t)rr * w_in + (int64_t)cc) * C + ch];
    return fill;
  };
  double v;
  if (order == 0) {
    v = at(floor(mr + 0.5), floor(mc + 0.5));
  } else {
    const double r0 = floor(mr), c0 = floor(mc);
    const double tr = mr - r0, tc = mc - c0;
    const double top = at(r0, c0) * (1.0 - tc) + at(r0, c0 + 1.0) * tc;
    const double bot = at(r0 + 1.0, c0) * (1.0 - tc) + at(r0 + 1.0, c0 + 1.0) * tc;
    v = top * (1.0 - tr) + bot * tr;
    if (!(mr == mr) || !(mc == mc)) v = fill;  // NaN coordinates
  }
  out[i] = v;
}

// K8b  inverse of the Metashape frame-camera model (row f1).  The reference inverts the lens model numerically ONCE per
//      distortion key on the host: scipy griddata over every 8th pixel of the forward map (cameras.py:1045-1062,
//      utils/indexing.py:87-150) -- minutes at 5280 x 3956, and 0.02 px off the true inverse (the piecewise-linear
//      interpolation error of its 8-pixel triangles).  Here every pixel (i, j) of the warped image solves
//      forward(row, col) = (i, j) by Newton's method in float64 with the analytic Jacobian of
//      derived_cameras.py:163-208, from the identity guess: quadratic convergence, 1e-13 px after at most 8 steps for
//      the distortions photogrammetry lenses have.  `fill` where the solution lies outside the ideal image or the
//      iteration has not converged.  par: f, cx, cy, image_width, image_height, k1..k4, p1, p2, b1, b2.
struct LensModel { double f, cx, cy, W, H, k1, k2, k3, k4, p1, p2, b1, b2; };

// model and its Jacobian at the ORIGINAL-resolution ideal pixel (xp, yp): (u, v) = distorted pixel
__device__ __forceinline__ void lens_forward(const LensModel &m, double xp, double yp, double &u, double &v, double &ux,
                                             double &uy, double &vx, double &vy) {
  const double x = (xp - m.W * 0.5) / m.f, y = (yp - m.H * 0.5) / m.f;
  const double r2 = x * x + y * y;
  const double R = 1.0 + r2 * (m.k1 + r2 * (m.k2 + r2 * (m.k3 + r2 * m.k4)));
  const double Rp = 2.0 * (m.k1 + r2 * (2.0 * m.k2 + r2 * (3.0 * m.k3 + r2 * 4.0 * m.k4)));  // dR/dx = Rp x, dR/dy = Rp y
  const double xd = x * R + (m.p1 * (r2 + 2.0 * x * x) + 2.0 * m.p2 * x * y);
  const double yd = y * R + (m.p2 * (r2 + 2.0 * y * y) + 2.0 * m.p1 * x * y);
  const double xdx = R + x * x * Rp + 6.0 * m.p1 * x + 2.0 * m.p2 * y, xdy = x * y * Rp + 2.0 * m.p1 * y + 2.0 * m.p2 * x;
  const double ydx = x * y * Rp + 2.0 * m.p2 * x + 2.0 * m.p1 * y, ydy = R + y * y * Rp + 6.0 * m.p2 * y + 2.0 * m.p1 * x;
  u = m.W * 0.5 + m.cx + xd * m.f + xd * m.b1 + yd * m.b2;
  v = m.H * 0.5 + m.cy + yd * m.f;
  const double inv_f = 1.0 / m.f;  // d x / d xp
  ux = ((m.f + m.b1) * xdx + m.b2 * ydx) * inv_f; uy = ((m.f + m.b1) * xdy + m.b2 * ydy) * inv_f;
  vx = m.f * ydx * inv_f; vy = m.f * ydy * inv_f;
}

__global__ __launch_bounds__(256) void k_invert_distortion(LensModel m, int h, int w, double scale, int unit_scale,
                                                           int iters, double fill, double *__restrict__ map_r,
                                                           double *__restrict__ map_c) {
  const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (p >= (int64_t)h * w) return;
  const int i = (int)(p / w), j = (int)(p - (int64_t)i * w);
  // the forward map of cameras.py:1012-1043: at scale 1 the model is evaluated at the pixel index itself, otherwise at the
  // original-resolution position (index + 0.5) / scale of the scaled pixel's centre, and its result is scaled back
  const double off = unit_scale ? 0.0 : 0.5, s = unit_scale ? 1.0 : scale, inv_s = 1.0 / s;
  double r = (double)i, c = (double)j;
  double er = 0.0, ec = 0.0;
  for (int it = 0; it <= iters; ++it) {
    double u, v, ux, uy, vx, vy;
    lens_forward(m, (c + off) * inv_s, (r + off) * inv_s, u, v, ux, uy, vx, vy);
    er = v * s - (double)i; ec = u * s - (double)j;  // residual in pixels of the scaled warped image
    if (it == iters) break;
    // d(row', col') / d(row, col): the scale factors cancel
    const double det = vy * ux - vx * uy;
    if (!(fabs(det) > 1e-300)) break;
    double dr = (ux * er - vx * ec) / det, dc = (vy * ec - uy * er) / det;
    dr = fmin(fmax(dr, -(double)h), (double)h); dc = fmin(fmax(dc, -(double)w), (double)w);
    r -= dr; c -= dc;
  }
  const double tol = 1e-9 * (double)max(h, w);
  const bool ok = fabs(er) < tol && fabs(ec) < tol && r >= 0.0 && r <= (double)(h - 1) && c >= 0.0 && c <= (double)(w - 1);
  map_r[p] = ok ? r : fill;
  map_c[p] = ok ? c : fill;
}

}  // namespace

extern "C" {

int gr_warp_nearest_i32(gr_ctx *c, const int32_t *in, int h_in, int w_in, const double *map_rows, const double *map_cols,
                        int h_out, int w_out, int32_t fill, int reference_float_roundtrip, double value_min,
                        double value_range, int32_t *out, void *stream) {
  if (!c) return GR_EINVAL;
  if (!in || !map_rows || !map_cols || !out || h_in <= 0 || w_in <= 0 || h_out <= 0 || w_out <= 0)
    return fail(c, GR_EINVAL, "bad warp args");
  if (reference_float_roundtrip && !(value_range > 0.0)) return fail(c, GR_EINVAL, "value_range must be positive");
  hipStream_t s = (hipStream_t)stream;
  GR_HIP(c, hipSetDevice(c->device));
  const int64_t n = (int64_t)h_out * w_out;
  hipLaunchKernelGGL(k_warp_nearest_i32, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, s, in, h_in, w_in, map_rows,
                     map_cols, n, fill, reference_float_roundtrip, value_min, value_range, out);
  GR_HIP(c, hipGetLastError());
  return GR_OK;
}

int gr_warp_f64(gr_ctx *c, const double *in, int h_in, int w_in, int C, const double *map_rows, const double *map_cols,
                int h_out, int w_out, int order, double fill, double *out, void *stream) {
  if (!c) return GR_EINVAL;
  if (!in || !map_rows || !map_cols || !out || h_in <= 0 || w_in <= 0 || h_out <= 0 || w_out <= 0 || C <= 0)
    return fail(c, GR_EINVAL, "bad warp args");
  if (order != 0 && order != 1) return fail(c, GR_EINVAL, "interpolation order %d not supported (0 or 1)", order);
  hipStream_t s = (hipStream_t)stream;
  GR_HIP(c, hipSetDevice(c->device));
  const int64_t n = (int64_t)h_out * w_out;
  hipLaunchKernelGGL(k_warp_f64, dim3((unsigned)ceil_div(n * C, 256)), dim3(256), 0, s, in, h_in, w_in, C, map_rows,
                     map_cols, n, order, fill, out);
  GR_HIP(c, hipGetLastError());
  return GR_OK;
}

int gr_invert_distortion_f64(gr_ctx *c, const double *par_h, int h, int w, double image_scale, int max_iters, double fill,
                              double *map_rows, double *map_cols, void *stream) {
  if (!c) return GR_EINVAL;
  if (!par_h || !map_rows || !map_cols || h <= 0 || w <= 0 || !(image_scale > 0.0) || max_iters < 1 || max_iters > 64)
    return fail(c, GR_EINVAL, "bad lens-inversion args");
  if (!(par_h[0] > 0.0) || !(par_h[3] > 0.0) || !(par_h[4] > 0.0)) return fail(c, GR_EINVAL, "focal length and image size must be positive");
  hipStream_t s = (hipStream_t)stream;
  GR_HIP(c, hipSetDevice(c->device));
  LensModel m = {par_h[0], par_h[1], par_h[2], par_h[3], par_h[4], par_h[5], par_h[6], par_h[7], par_h[8], par_h[9], par_h[10],
                 par_h[11], par_h[12]};
  const int64_t n = (int64_t)h * w;
  const int unit = fabs(image_scale - 1.0) <= 1e-8 + 1e-5 * 1.0 ? 1 : 0;  // numpy.isclose(image_scale, 1.0), cameras.py:1012
  hipLaunchKernelGGL(k_invert_distortion, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, s, m, h, w, image_scale, unit,
                     max_iters, fill, map_rows, map_cols);
  GR_HIP(c, hipGetLastError());
  return GR_OK;
}

}  // extern "C"
