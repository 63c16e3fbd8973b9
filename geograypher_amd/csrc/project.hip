// geograypher_amd/csrc/project.hip -- everything behind the rasterizer: last-writer-wins winners from id images, per-face
// votes / nansums, per-view textures, texture gathers, sparse (face, class) pairs, finalize, argmax -- kernels and entry points.
#include <hipcub/hipcub.hpp>

#include "gr_internal.hpp"

using namespace grimpl;

namespace {

// last-writer-wins candidate of the unfused pass (K5): issue the global atomicMax only when neither the right nor the
// lower neighbour shows the same face.  key = pixel + 1 (the label is looked up by the vote kernel).
__device__ __forceinline__ void winner_pixel(uint32_t *__restrict__ winner, int f, int fr, int fb, int64_t p, int64_t F,
                                              int compat) {
  if (compat) {  // meshes.py:1998-2001: index -1 aliases the last face
    const int last = (int)F - 1;
    if (f == -1) f = last;
    if (fr == -1) fr = last;
    if (fb == -1) fb = last;
  }
  if (f < 0 || f >= F) return;
  if (fr == f || fb == f) return;  // a later pixel of the same face exists
  atomicMax(&winner[f], (uint32_t)(p + 1));
}

// ------------------------------------------------------------------------------------------------------------------
// K5  last-writer-wins winners from id images already in memory (the unfused path).  Four pixels per thread.  A pixel
//     can only be its face's LAST pixel in row-major order if neither its right nor its lower neighbour shows the same
//     face, so only those candidates issue the global atomicMax (~1-3 per visible face instead of ~80).  key = pixel + 1.
// ------------------------------------------------------------------------------------------------------------------
// grid (ceil(w/1024), ceil(h/WIN_ROWS), views): a thread owns 4 consecutive columns and walks WIN_ROWS rows downwards;
// the row below is loaded once and becomes the current row of the next step (16-byte id loads).
#define WIN_ROWS 16
__global__ __launch_bounds__(256) void k_winner(const int32_t *__restrict__ ids, uint32_t *__restrict__ winner, int64_t F,
                                                int h, int w, int compat) {
  const int slot = blockIdx.z;
  const int y0 = blockIdx.y * WIN_ROWS;
  const int x0 = (blockIdx.x * 256 + threadIdx.x) * 4;
  if (x0 >= w) return;
  const int64_t P = (int64_t)h * w;
  const int32_t *img = ids + slot * P;
  uint32_t *win = winner + slot * F;
  const bool vec = ((w & 3) == 0) && ((reinterpret_cast<uintptr_t>(img) & 15) == 0);
  auto load_row = [&](int y, int (&f)[5]) {
    const int32_t *row = img + (int64_t)y * w;
    if (vec) {
      const int4 c = *reinterpret_cast<const int4 *>(row + x0);
      f[0] = c.x; f[1] = c.y; f[2] = c.z; f[3] = c.w;
      f[4] = (x0 + 4 < w) ? row[x0 + 4] : -2;
    } else {
#pragma unroll
      for (int k = 0; k < 5; ++k) f[k] = (x0 + k < w) ? row[x0 + k] : -2;
    }
  };
  int cur[5], nxt[5];
  load_row(y0, cur);
  const int y1 = min(y0 + WIN_ROWS, h);
  for (int y = y0; y < y1; ++y) {
    const bool has_below = (y + 1 < h);
    if (has_below) load_row(y + 1, nxt);
    else { nxt[0] = nxt[1] = nxt[2] = nxt[3] = nxt[4] = -2; }
    const int64_t p0 = (int64_t)y * w + x0;
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (x0 + k < w) winner_pixel(win, cur[k], cur[k + 1], nxt[k], p0 + k, F, compat);
#pragma unroll
    for (int k = 0; k < 5; ++k) cur[k] = nxt[k];
  }
}

// K6  per-face vote: one thread per face walks the views of the launch group IN ORDER (deterministic, no atomics needed:
//     a face belongs to exactly one thread).  The label of the winning pixel is looked up here (one byte per visible
//     face and view; neighbouring faces win neighbouring pixels): votes[f][label] += 1, counts[f] += 1; a label >= C
//     (255 = ignore) is an all-zero one-hot row that still counts (predictors/segmentor.py:37-69).  Winners are
//     cleared for reuse (only the faces a view shows were written: a tenth of the array).
//     Fused aggregation: a wave reads the winners of its 64 faces only for the views in which the tile kernel marked the group
//     (round 5: a bit per 256-face chunk from the cull pass's block lists -- half of the winners it made the pass read were empty).
__global__ __launch_bounds__(256) void k_vote_labels(uint32_t *__restrict__ winner, const uint8_t *__restrict__ labels,
                                                     int n_views, int64_t F, int64_t P, int C,
                                                     uint32_t *__restrict__ votes, uint32_t *__restrict__ counts,
                                                     const unsigned long long *__restrict__ stats, int group,
                                                     const uint32_t *__restrict__ touched, int tw, uint32_t *__restrict__ visits) {
  const int64_t f = (int64_t)blockIdx.x * 256 + threadIdx.x;
  // which views of the group hold a winner for this WAVE's 64 faces: the byte the tile kernel's epilogue set for the group
  // beside its winner atomic (exact: a group no pixel of the view voted into is not read at all; background pixels that alias
  // the last face mark its group like any other winner); without the map (ids given by the caller) every view can.  Lane v
  // looks at view v.
  unsigned long long dirty = ~0ull;
  if (touched) {
    const int v = threadIdx.x & 63;
    const int64_t g = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    bool d = false;
    if (v < n_views && g * 64 < F) d = reinterpret_cast<const uint8_t *>(touched + (int64_t)v * tw)[g] != 0;
    dirty = __ballot(d);
    // (view, group) pairs visited, for gr_raster_stats.chunk_visits: a slot per group, no contention
    if (visits && v == 0 && dirty) visits[g] += (uint32_t)__popcll(dirty);
  }
  if (f >= F) return;
  // a launch group whose binning overflowed (and every group after it) must not vote: its winners are incomplete.  The
  // caller learns how many views were folded in (gr_raster_status: views_done) and repeats the call for the rest.
  const bool skip = stats != nullptr && stats[4] <= (unsigned long long)group;
  uint32_t c = 0;
  // up to sixteen classes: the face's votes of the whole launch group are collected in TWO registers, a byte per class (a group has
  // at most 64 views), and folded into votes[] once at the end -- loads first, then stores.  (`votes[f][label] += 1` view by view
  // is a chain of dependent read-modify-writes on one array: the compiler must finish each before the next may start, eight
  // memory round trips per batch of eight views, and the kernel is nothing but latency: round 5, 1.9 us per C2 view.)
  const bool packed = C <= 16;
  unsigned long long acc0 = 0ull, acc1 = 0ull;
  // eight views' winners are requested together (the kernel is a stream over winner[views][F]: memory-level
  // parallelism, not arithmetic, sets its speed), then their labels, then the votes in view order
  for (int v0 = 0; v0 < n_views; v0 += 8) {
    const uint32_t d8 = (uint32_t)(dirty >> v0) & 0xFFu;
    if (d8 == 0u) continue;
    uint32_t key[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) key[k] = (v0 + k < n_views && ((d8 >> k) & 1u)) ? winner[(int64_t)(v0 + k) * F + f] : 0u;
    uint32_t lab[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) lab[k] = (key[k] && !skip) ? (uint32_t)labels[(int64_t)(v0 + k) * P + (key[k] - 1)] : 0u;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      if (key[k] == 0) continue;
      winner[(int64_t)(v0 + k) * F + f] = 0;
      if (skip) continue;
      if ((int)lab[k] < C) {
        if (!packed) votes[f * C + lab[k]] += 1u;
        else if (lab[k] < 8u) acc0 += 1ull << (8u * lab[k]);
        else acc1 += 1ull << (8u * (lab[k] - 8u));
      }
      ++c;
    }
  }
  if (acc0 | acc1) {
    uint32_t cur[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) cur[k] = k < C ? votes[f * C + k] : 0u;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const uint32_t add = (uint32_t)((k < 8 ? acc0 : acc1) >> (8 * (k & 7))) & 0xFFu;
      if (k < C && add) votes[f * C + k] = cur[k] + add;
    }
  }
  if (c) counts[f] += c;
}

__global__ __launch_bounds__(256) void k_vote_values(uint32_t *__restrict__ winner, const double *__restrict__ img,
                                                     int n_views, int64_t F, int64_t P, int C,
                                                     double *__restrict__ sums, uint32_t *__restrict__ counts) {
  const int64_t f = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (f >= F) return;
  // meshes.py:2060-2062 to the letter: summed = np.nansum([summed, projection], axis=0) drops a NaN of the RUNNING sum as
  // well as one of the projection -- a sum that went NaN (+inf of one view met -inf of another) starts again from zero at the
  // next view, whether that view shows the face or not.  (Found by tools/fuzz_stages.py; rounds 1-3 kept the NaN.)
  double *const acc = sums + f * C;
  for (int ch = 0; ch < C; ++ch)
    if (isnan(acc[ch])) acc[ch] = 0.0;  // left by an earlier launch: this launch holds a next view
  uint32_t c = 0;
  int last_seen = -1;
  for (int v0 = 0; v0 < n_views; v0 += 8) {  // eight views' winners are requested together, then consumed in view order
    uint32_t keyv[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) keyv[k] = (v0 + k < n_views) ? winner[(int64_t)(v0 + k) * F + f] : 0u;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const uint32_t key = keyv[k];
      if (key == 0) continue;
      const int v = v0 + k;
      winner[(int64_t)v * F + f] = 0;
      const double *row = img + ((int64_t)v * P + (key - 1)) * C;
      bool any_finite = false;
      for (int ch = 0; ch < C; ++ch) {
        const double x = row[ch];
        if (isfinite(x)) any_finite = true;
        double a = acc[ch];
        if (isnan(a)) a = 0.0;          // the running sum's NaN counts as 0 ...
        if (!isnan(x)) a += x;          // ... like the projection's
        acc[ch] = a;
      }
      if (any_finite) ++c;
      last_seen = v;
    }
  }
  if (last_seen >= 0 && last_seen < n_views - 1)  // a NaN made at the face's last view here is dropped by the view behind it
    for (int ch = 0; ch < C; ++ch)
      if (isnan(acc[ch])) acc[ch] = 0.0;
  if (c) counts[f] += c;
}

__global__ __launch_bounds__(256) void k_project_view(uint32_t *__restrict__ winner, const double *__restrict__ img,
                                                      int64_t F, int C, double *__restrict__ tex) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= F * C) return;
  const int64_t f = i / C;
  const int ch = (int)(i - f * C);
  const uint32_t key = winner[f];
  tex[i] = key ? img[(int64_t)(key - 1) * C + ch] : __longlong_as_double(0x7FF8000000000000ll);
}

__global__ __launch_bounds__(256) void k_clear_u32(uint32_t *p, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) p[i] = 0;
}

// K7  render_flat gather: out[p][c] = tex[ids[p]][c] or NaN
__global__ __launch_bounds__(256) void k_gather_texture(const int32_t *__restrict__ ids, int64_t n_pix,
                                                        const double *__restrict__ tex, int64_t F, int C,
                                                        double *__restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n_pix * C) return;
  const int64_t p = i / C;
  const int ch = (int)(i - p * C);
  const int f = ids[p];
  out[i] = (f >= 0 && f < F) ? tex[(int64_t)f * C + ch] : __longlong_as_double(0x7FF8000000000000ll);
}

// K9  save_renders epilogue (row f2): gather the face texture and cast it the way meshes.py:2325-2337 does --
//     values < 0, > 255 or non-finite (and pixels without a face) become `null_value`, the rest is truncated to uint8.
__global__ __launch_bounds__(256) void k_gather_texture_u8(const int32_t *__restrict__ ids, int64_t n_pix,
                                                           const double *__restrict__ tex, int64_t F, int C,
                                                           uint8_t null_value, uint8_t *__restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n_pix * C) return;
  const int64_t p = i / C;
  const int ch = (int)(i - p * C);
  const int f = ids[p];
  uint8_t v = null_value;
  if (f >= 0 && f < F) {
    const double x = tex[(int64_t)f * C + ch];
    if (x >= 0.0 && x <= 255.0) v = (uint8_t)x;  // false for NaN; truncation like numpy's astype(uint8)
  }
  out[i] = v;
}

// K10 sparse index aggregation (row f3, derived_meshes.py:470-520): one thread per face walks the views of the batch;
//     a finite winner value v is one observation of class int(v): counts[f] += 1 and the pair key f * n_classes + class
//     is appended to `keys` (wave ballot + one atomic per wave).  The pairs are counted later by sort + run-length.
__global__ __launch_bounds__(256) void k_emit_index_pairs(uint32_t *__restrict__ winner, const double *__restrict__ img,
                                                          int n_views, int64_t F, int64_t P, long long n_classes,
                                                          uint32_t *__restrict__ counts,
                                                          unsigned long long *__restrict__ keys, long long key_cap,
                                                          unsigned long long *__restrict__ key_count,
                                                          int *__restrict__ bad) {
  const int64_t f = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int lane = threadIdx.x & 63;
  uint32_t c = 0;
  for (int v = 0; v < n_views; ++v) {
    bool emit = false;
    unsigned long long key = 0;
    if (f < F) {
      const uint32_t w = winner[v * F + f];
      if (w != 0) {
        winner[v * F + f] = 0;
        const double x = img[(int64_t)v * P + (w - 1)];
        if (isfinite(x)) {
          ++c;
          const long long cls = (long long)x;  // astype(int): truncation
          if (cls < 0 || cls >= n_classes) atomicOr(bad, 1);
          else { emit = true; key = (unsigned long long)f * (unsigned long long)n_classes + (unsigned long long)cls; }
        }
      }
    }
    const unsigned long long m = __ballot(emit);
    if (m) {
      const int leader = __ffsll((long long)m) - 1;
      unsigned long long base = 0;
      if (lane == leader) base = atomicAdd(key_count, (unsigned long long)__popcll(m));
      base = __shfl(base, leader);
      if (emit) {
        const unsigned long long idx = base + __popcll(m & ((1ull << lane) - 1ull));
        if ((long long)idx < key_cap) keys[idx] = key;
      }
    }
  }
  if (f < F && c) counts[f] += c;
}

__global__ __launch_bounds__(256) void k_finalize_votes(const uint32_t *__restrict__ votes,
                                                        const uint32_t *__restrict__ counts, int64_t F, int C,
                                                        double *__restrict__ average, double *__restrict__ summed,
                                                        double *__restrict__ counts_f64) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= F * C) return;
  const int64_t f = i / C;
  const uint32_t c = counts[f];
  const double nan = __longlong_as_double(0x7FF8000000000000ll);
  const double s = c ? (double)votes[i] : nan;
  summed[i] = s;
  average[i] = c ? s / (double)c : nan;  // numpy: nan / 0 = nan
  if (i == f * C) counts_f64[f] = (double)c;
}

__global__ __launch_bounds__(256) void k_finalize_sums(double *__restrict__ sums, const uint32_t *__restrict__ counts,
                                                       int64_t F, int C, double *__restrict__ average,
                                                       double *__restrict__ counts_f64) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= F * C) return;
  const int64_t f = i / C;
  const uint32_t c = counts[f];
  const double nan = __longlong_as_double(0x7FF8000000000000ll);
  const double s = c ? sums[i] : nan;
  sums[i] = s;
  average[i] = c ? s / (double)c : nan;
  if (i == f * C) counts_f64[f] = (double)c;
}

// utils/indexing.py:9-32
__global__ __launch_bounds__(256) void k_argmax_nonzero(const double *__restrict__ arr, int64_t F, int C,
                                                        double *__restrict__ out) {
  const int64_t f = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (f >= F) return;
  const double *row = arr + f * C;
  double best = row[0], sum = 0.0;
  int arg = 0;
  bool bad = false;
  // np.argmax: first maximum; a NaN is "maximal" and the first NaN wins
  bool best_nan = isnan(best);
  for (int c = 0; c < C; ++c) {
    const double x = row[c];
    if (!isfinite(x)) bad = true;
    sum += x;
    if (c > 0 && !best_nan) {
      if (isnan(x)) { best_nan = true; arg = c; }
      else if (x > best) { best = x; arg = c; }
    }
  }
  out[f] = (bad || sum == 0.0) ? __longlong_as_double(0x7FF8000000000000ll) : (double)arg;
}

}  // namespace

namespace grimpl {

// unfused label projection for id images already in memory: winner pass + vote pass per launch group
int project_labels(gr_ctx *c, const int32_t *ids, const uint8_t *labels, int n_views, int h, int w, int C, uint32_t *votes,
                   uint32_t *counts, int flags, hipStream_t s) {
  const int64_t P = (int64_t)h * w, F = c->F;
  const int B = n_views < GR_MAX_BATCH ? n_views : GR_MAX_BATCH;
  int rc = ensure_winner(c, sizeof(uint32_t) * (size_t)F * B, s);
  if (rc) return rc;
  note_stream(c, s);
  uint32_t *win = (uint32_t *)c->winner;
  for (int v0 = 0; v0 < n_views; v0 += B) {
    const int nb = (n_views - v0) < B ? (n_views - v0) : B;
    {
      Timed t(c, s, ST_PROJECT);
      hipLaunchKernelGGL(k_winner, dim3((unsigned)ceil_div(ceil_div(w, 4), 256), (unsigned)ceil_div(h, WIN_ROWS), nb), dim3(256), 0,
                         s, ids + v0 * P, win, F, h, w, (flags & GR_FLAG_NEG1_IS_LAST_FACE) ? 1 : 0);
    }
    {
      Timed t(c, s, ST_VOTE);
      hipLaunchKernelGGL(k_vote_labels, dim3((unsigned)ceil_div(F, 256)), dim3(256), 0, s, win, labels + v0 * P, nb, F, P, C,
                         votes, counts, (const unsigned long long *)nullptr, 0, (const uint32_t *)nullptr, 0, (uint32_t *)nullptr);
    }
  }
  GR_HIP(c, hipGetLastError());
  return GR_OK;
}

// the fused aggregation's vote pass of one launch group (raster_views, geograster.hip)
void launch_vote_labels(gr_ctx *c, hipStream_t vs, uint32_t *win, const uint8_t *labels, int nb, int64_t F, int64_t P, int C,
                        uint32_t *votes, uint32_t *counts, int group, const uint32_t *touched, int tw, int flags) {
  Timed t(c, vs, ST_VOTE);
  hipLaunchKernelGGL(k_vote_labels, dim3((unsigned)ceil_div(F, 256)), dim3(256), 0, vs, win, labels, nb, F, P, C, votes, counts,
                     (const unsigned long long *)c->stats, group, touched, tw, touched ? c->visits : (uint32_t *)nullptr);
  (void)flags;
}

}  // namespace grimpl

extern "C" {

int gr_gather_texture_f64(gr_ctx *c, const int32_t *ids, int64_t n_pix, const double *face_tex, int64_t F, int C,
                          double *out, void *stream) {
  if (!c || !ids || !face_tex || !out || n_pix < 0 || F <= 0 || C <= 0) return fail(c, GR_EINVAL, "bad gather args");
  if (n_pix == 0) return GR_OK;
  hipStream_t s = (hipStream_t)stream;
  GR_HIP(c, hipSetDevice(c->device));
  Timed t(c, s, ST_GATHER);
  hipLaunchKernelGGL(k_gather_texture, dim3((unsigned)ceil_div(n_pix * C, 256)), dim3(256), 0, s, ids, n_pix, face_tex,
                     F, C, out);
  GR_HIP(c, hipGetLastError());
  return GR_OK;
}

int gr_project_labels_u8(gr_ctx *c, const int32_t *ids, const uint8_t *labels, int n_views, int h, int w, int C,
                         uint32_t *votes, uint32_t *counts, int flags, void *stream) {
  int rc = check_common(c, n_views, h, w);
  if (rc) return rc;
  if (c->F <= 0) return fail(c, GR_ENOMESH, "gr_mesh_upload has not been called");
  if (!ids || !labels || !votes || !counts || C <= 0 || C > 255) return fail(c, GR_EINVAL, "bad project args C=%d", C);
  if (n_views == 0) return GR_OK;
  hipStream_t s = (hipStream_t)stream;
  GR_HIP(c, hipSetDevice(c->device));
  return project_labels(c, ids, labels, n_views, h, w, C, votes, counts, flags, s);
}

int gr_project_values_f64(gr_ctx *c, const int32_t *ids, const double *img, int n_views, int h, int w, int C,
                          double *sums, uint32_t *counts, int flags, void *stream) {
  int rc = check_common(c, n_views, h, w);
  if (rc) return rc;
  if (c->F <= 0) return fail(c, GR_ENOMESH, "gr_mesh_upload has not been called");
  if (!ids || !img || !sums || !counts || C <= 0) return fail(c, GR_EINVAL, "bad project args");
  if (n_views == 0) return GR_OK;
  hipStream_t s = (hipStream_t)stream;
  GR_HIP(c, hipSetDevice(c->device));
  const int64_t P = (int64_t)h * w, F = c->F;
  const int B = n_views < GR_MAX_BATCH ? n_views : GR_MAX_BATCH;
  rc = ensure_winner(c, sizeof(uint32_t) * (size_t)F * B, s);
  if (rc) return rc;
  note_stream(c, s);
  uint32_t *win = (uint32_t *)c->winner;
  for (int v0 = 0; v0 < n_views; v0 += B) {
    const int nb = (n_views - v0) < B ? (n_views - v0) : B;
    {
      Timed t(c, s, ST_PROJECT);
      hipLaunchKernelGGL(k_winner, dim3((unsigned)ceil_div(ceil_div(w, 4), 256), (unsigned)ceil_div(h, WIN_ROWS), nb), dim3(256), 0, s, ids + v0 * P, win, F, h, w,
                         (flags & GR_FLAG_NEG1_IS_LAST_FACE) ? 1 : 0);
    }
    {
      Timed t(c, s, ST_VOTE);
      hipLaunchKernelGGL(k_vote_values, dim3((unsigned)ceil_div(F, 256)), dim3(256), 0, s, win, img + v0 * P * C, nb, F,
                         P, C, sums, counts);
    }
  }
  GR_HIP(c, hipGetLastError());
  return GR_OK;
}

int gr_project_view_f64(gr_ctx *c, const int32_t *ids, const double *img, int h, int w, int C, double *tex, int flags,
                        void *stream) {
  int rc = check_common(c, 1, h, w);
  if (rc) return rc;
  if (c->F <= 0) return fail(c, GR_ENOMESH, "gr_mesh_upload has not been called");
  if (!ids || !img || !tex || C <= 0) return fail(c, GR_EINVAL, "bad project args");
  hipStream_t s = (hipStream_t)stream;
  GR_HIP(c, hipSetDevice(c->device));
  const int64_t F = c->F;
  rc = ensure_winner(c, sizeof(uint32_t) * (size_t)F, s);
  if (rc) return rc;
  note_stream(c, s);
  uint32_t *win = (uint32_t *)c->winner;
  {
    Timed t(c, s, ST_PROJECT);
    hipLaunchKernelGGL(k_winner, dim3((unsigned)ceil_div(ceil_div(w, 4), 256), (unsigned)ceil_div(h, WIN_ROWS), 1), dim3(256), 0, s, ids, win, F, h, w,
                         (flags & GR_FLAG_NEG1_IS_LAST_FACE) ? 1 : 0);
  }
  {
    Timed t(c, s, ST_VOTE);
    hipLaunchKernelGGL(k_project_view, dim3((unsigned)ceil_div(F * C, 256)), dim3(256), 0, s, win, img, F, C, tex);
    hipLaunchKernelGGL(k_clear_u32, dim3((unsigned)ceil_div(F, 256)), dim3(256), 0, s, win, F);
  }
  GR_HIP(c, hipGetLastError());
  return GR_OK;
}

int gr_gather_texture_u8(gr_ctx *c, const int32_t *ids, int64_t n_pix, const double *face_tex, int64_t F, int C,
                         int null_value, uint8_t *out, void *stream) {
  if (!c) return GR_EINVAL;
  if (!ids || !face_tex || !out || n_pix < 0 || F <= 0 || C <= 0 || null_value < 0 || null_value > 255)
    return fail(c, GR_EINVAL, "bad gather args");
  if (n_pix == 0) return GR_OK;
  hipStream_t s = (hipStream_t)stream;
  GR_HIP(c, hipSetDevice(c->device));
  Timed t(c, s, ST_GATHER);
  hipLaunchKernelGGL(k_gather_texture_u8, dim3((unsigned)ceil_div(n_pix * C, 256)), dim3(256), 0, s, ids, n_pix, face_tex,
                     F, C, (uint8_t)null_value, out);
  GR_HIP(c, hipGetLastError());
  return GR_OK;
}

int gr_project_index_pairs(gr_ctx *c, const int32_t *ids, const double *img, int n_views, int h, int w, int64_t n_classes,
                           uint32_t *counts, uint64_t *keys, int64_t key_cap, uint64_t *key_count, int flags,
                           void *stream) {
  int rc = check_common(c, n_views, h, w);
  if (rc) return rc;
  if (c->F <= 0) return fail(c, GR_ENOMESH, "gr_mesh_upload has not been called");
  if (!ids || !img || !counts || !keys || !key_count || n_classes <= 0 || key_cap < 0)
    return fail(c, GR_EINVAL, "bad sparse projection args");
  if (n_views == 0) return GR_OK;
  hipStream_t s = (hipStream_t)stream;
  GR_HIP(c, hipSetDevice(c->device));
  const int64_t P = (int64_t)h * w, F = c->F;
  const int B = n_views < GR_MAX_BATCH ? n_views : GR_MAX_BATCH;
  rc = ensure_winner(c, sizeof(uint32_t) * (size_t)F * B, s);
  if (rc) return rc;
  note_stream(c, s);
  uint32_t *win = (uint32_t *)c->winner;
  // the "a value is no class index" flag: the context's flag word, read back below -- or, deferred, the caller's SECOND
  // 64-bit word behind the pair counter (a word of its own: the counter takes 64-bit atomics, the flag a 32-bit one)
  const bool defer = (flags & GR_FLAG_DEFER_CHECK) != 0;
  int *bad_flag = defer ? reinterpret_cast<int *>(key_count + 1) : c->flag;
  if (!defer) GR_HIP(c, hipMemsetAsync(c->flag, 0, sizeof(int), s));
  for (int v0 = 0; v0 < n_views; v0 += B) {
    const int nb = (n_views - v0) < B ? (n_views - v0) : B;
    {
      Timed t(c, s, ST_PROJECT);
      hipLaunchKernelGGL(k_winner, dim3((unsigned)ceil_div(ceil_div(w, 4), 256), (unsigned)ceil_div(h, WIN_ROWS), nb), dim3(256), 0, s, ids + v0 * P, win, F, h, w,
                         (flags & GR_FLAG_NEG1_IS_LAST_FACE) ? 1 : 0);
    }
    {
      Timed t(c, s, ST_VOTE);
      hipLaunchKernelGGL(k_emit_index_pairs, dim3((unsigned)ceil_div(F, 256)), dim3(256), 0, s, win, img + v0 * P, nb, F, P,
                         (long long)n_classes, counts, (unsigned long long *)keys, (long long)key_cap,
                         (unsigned long long *)key_count, bad_flag);
    }
  }
  GR_HIP(c, hipGetLastError());
  if (defer) return GR_OK;
  int bad = 0;
  GR_HIP(c, hipMemcpyAsync(&bad, c->flag, sizeof(int), hipMemcpyDeviceToHost, s));
  GR_HIP(c, hipStreamSynchronize(s));
  if (bad) return fail(c, GR_EINDEX, "an image value is not a class index in [0, %lld)", (long long)n_classes);
  return GR_OK;
}

int gr_count_pairs(gr_ctx *c, uint64_t *keys, int64_t n, uint64_t *unique_keys, uint32_t *pair_counts, int64_t *n_unique_h,
                   void *stream) {
  if (!c) return GR_EINVAL;
  if (!keys || !unique_keys || !pair_counts || !n_unique_h || n < 0 || n > 0x7FFFFFFFll)
    return fail(c, GR_EINVAL, "bad pair-count args");
  *n_unique_h = 0;
  if (n == 0) return GR_OK;
  hipStream_t s = (hipStream_t)stream;
  GR_HIP(c, hipSetDevice(c->device));
  // radix sort (keys -> sorted copy in context scratch) + run-length encode, both rocPRIM through hipcub
  size_t sort_bytes = 0, rle_bytes = 0;
  unsigned long long *kin = (unsigned long long *)keys, *uo = (unsigned long long *)unique_keys;
  GR_HIP(c, hipcub::DeviceRadixSort::SortKeys(nullptr, sort_bytes, kin, kin, (int)n, 0, 64, s));
  int *d_runs = nullptr;
  GR_HIP(c, hipcub::DeviceRunLengthEncode::Encode(nullptr, rle_bytes, kin, uo, pair_counts, d_runs, (int)n, s));
  const size_t tmp_bytes = (sort_bytes > rle_bytes ? sort_bytes : rle_bytes) + 256;
  const size_t need = tmp_bytes + sizeof(unsigned long long) * (size_t)n + 256;
  if (c->sort_bytes < need) {
    quiesce(c);
    GR_HIP(c, hipStreamSynchronize(s));
    if (c->sort_tmp) (void)hipFree(c->sort_tmp);
    c->sort_tmp = nullptr; c->sort_bytes = 0;
    if (hipMalloc(&c->sort_tmp, need) != hipSuccess) return fail(c, GR_ENOMEM, "sort scratch allocation failed");
    c->sort_bytes = need;
  }
  note_stream(c, s);
  char *base = (char *)c->sort_tmp;
  unsigned long long *sorted = (unsigned long long *)base;
  void *tmp = base + ((sizeof(unsigned long long) * (size_t)n + 255) / 256) * 256;
  size_t tb = sort_bytes;
  GR_HIP(c, hipcub::DeviceRadixSort::SortKeys(tmp, tb, kin, sorted, (int)n, 0, 64, s));
  tb = rle_bytes;
  GR_HIP(c, hipcub::DeviceRunLengthEncode::Encode(tmp, tb, sorted, uo, pair_counts, (int *)c->flag, (int)n, s));
  int runs = 0;
  GR_HIP(c, hipMemcpyAsync(&runs, c->flag, sizeof(int), hipMemcpyDeviceToHost, s));
  GR_HIP(c, hipStreamSynchronize(s));
  *n_unique_h = runs;
  return GR_OK;
}

int gr_finalize_votes(gr_ctx *c, const uint32_t *votes, const uint32_t *counts, int64_t F, int C, double *average,
                      double *summed, double *counts_f64, void *stream) {
  if (!c || !votes || !counts || !average || !summed || !counts_f64 || F <= 0 || C <= 0)
    return fail(c, GR_EINVAL, "bad finalize args");
  hipStream_t s = (hipStream_t)stream;
  GR_HIP(c, hipSetDevice(c->device));
  hipLaunchKernelGGL(k_finalize_votes, dim3((unsigned)ceil_div(F * C, 256)), dim3(256), 0, s, votes, counts, F, C,
                     average, summed, counts_f64);
  GR_HIP(c, hipGetLastError());
  return GR_OK;
}

int gr_finalize_sums_f64(gr_ctx *c, double *sums, const uint32_t *counts, int64_t F, int C, double *average,
                         double *counts_f64, void *stream) {
  if (!c || !sums || !counts || !average || !counts_f64 || F <= 0 || C <= 0)
    return fail(c, GR_EINVAL, "bad finalize args");
  hipStream_t s = (hipStream_t)stream;
  GR_HIP(c, hipSetDevice(c->device));
  hipLaunchKernelGGL(k_finalize_sums, dim3((unsigned)ceil_div(F * C, 256)), dim3(256), 0, s, sums, counts, F, C,
                     average, counts_f64);
  GR_HIP(c, hipGetLastError());
  return GR_OK;
}

int gr_argmax_nonzero_f64(gr_ctx *c, const double *array, int64_t F, int C, double *out, void *stream) {
  if (!c || !array || !out || F <= 0 || C <= 0) return fail(c, GR_EINVAL, "bad argmax args");
  hipStream_t s = (hipStream_t)stream;
  GR_HIP(c, hipSetDevice(c->device));
  hipLaunchKernelGGL(k_argmax_nonzero, dim3((unsigned)ceil_div(F, 256)), dim3(256), 0, s, array, F, C, out);
  GR_HIP(c, hipGetLastError());
  return GR_OK;
}

}  // extern "C"
