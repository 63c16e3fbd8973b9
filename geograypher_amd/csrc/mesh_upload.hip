// geograypher_amd/csrc/mesh_upload.hip -- gr_mesh_upload: once per mesh, the faces are ordered along a Morton curve of their
// centroids and de-indexed (soup), with a bounding sphere and a chunk list per block of 64 faces.
#include <hipcub/hipcub.hpp>

#include "gr_internal.hpp"

using namespace grimpl;

namespace {

// ------------------------------------------------------------------------------------------------------------------
// K0  (once per mesh upload) the mesh is re-ordered along a Morton curve of the face centroids and de-indexed:
//     soup[n] = the 9 vertex coordinates of face orig[n].  A block of GR_BLOCK = 64 consecutive soup faces is then a
//     compact patch whatever the caller's face order: one bounding sphere per block rejects most of a survey mesh with
//     one test per view, and the 64 faces of a wave fall into one to four tiles (few, long runs for the tile counters).
//     Rasterization does not depend on the order in which faces are processed (ds_max_u64 resolve), ids are the caller's.
// ------------------------------------------------------------------------------------------------------------------
// order-preserving float -> uint32 (for atomicMin / atomicMax on floats)
__device__ __forceinline__ uint32_t float_ordered(float f) {
  const uint32_t b = (uint32_t)__float_as_int(f);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__host__ __device__ __forceinline__ float ordered_float(uint32_t u) {
  const uint32_t b = (u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u;
  union { uint32_t i; float f; } c; c.i = b; return c.f;
}

// bounds[0..2] = min, bounds[3..5] = max of the finite vertex coordinates (ordered-uint encoding)
__global__ __launch_bounds__(256) void k_mesh_bounds(const float *__restrict__ verts, int64_t V, uint32_t *__restrict__ bounds) {
  float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
  for (int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x; v < V; v += (int64_t)gridDim.x * 256)
    for (int d = 0; d < 3; ++d) {
      const float x = verts[3 * v + d];
      if (isfinite(x)) { lo[d] = fminf(lo[d], x); hi[d] = fmaxf(hi[d], x); }
    }
  for (int d = 0; d < 3; ++d) {
    for (int o = 32; o > 0; o >>= 1) {
      lo[d] = fminf(lo[d], __shfl_xor(lo[d], o));
      hi[d] = fmaxf(hi[d], __shfl_xor(hi[d], o));
    }
    if ((threadIdx.x & 63) == 0) {
      atomicMin(&bounds[d], float_ordered(lo[d]));
      atomicMax(&bounds[3 + d], float_ordered(hi[d]));
    }
  }
}

__device__ __forceinline__ uint32_t spread16(uint32_t x) {  // abcd -> 0a0b0c0d
  x &= 0xFFFFu;
  x = (x | (x << 8)) & 0x00FF00FFu;
  x = (x | (x << 4)) & 0x0F0F0F0Fu;
  x = (x | (x << 2)) & 0x33333333u;
  x = (x | (x << 1)) & 0x55555555u;
  return x;
}

// 32-bit Morton code of the face centroid on the two axes of largest extent (16 bits each); code[f], idx[f] = f
__global__ __launch_bounds__(256) void k_face_codes(const float *__restrict__ verts, const int32_t *__restrict__ faces, int64_t F,
                                                    int ax0, int ax1, float lo0, float inv0, float lo1, float inv1,
                                                    uint32_t *__restrict__ code, int32_t *__restrict__ idx) {
  const int64_t f = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (f >= F) return;
  const float *p0 = verts + 3 * (int64_t)faces[3 * f], *p1 = verts + 3 * (int64_t)faces[3 * f + 1],
              *p2 = verts + 3 * (int64_t)faces[3 * f + 2];
  const float c0 = (p0[ax0] + p1[ax0] + p2[ax0]) * (1.0f / 3.0f), c1 = (p0[ax1] + p1[ax1] + p2[ax1]) * (1.0f / 3.0f);
  const float q0 = (c0 - lo0) * inv0, q1 = (c1 - lo1) * inv1;  // NaN -> 0 below
  const uint32_t u0 = (uint32_t)fminf(fmaxf(q0, 0.0f), 65535.0f), u1 = (uint32_t)fminf(fmaxf(q1, 0.0f), 65535.0f);
  code[f] = spread16(u0) | (spread16(u1) << 1);
  idx[f] = (int32_t)f;
}

// soup[n] = the 9 vertex coordinates of face orig[n]   (one thread per (face, corner))
__global__ __launch_bounds__(256) void k_build_soup(const float *__restrict__ verts, const int32_t *__restrict__ faces,
                                                    const int32_t *__restrict__ orig, int64_t F, float *__restrict__ soup) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= 3 * F) return;
  const int64_t n = i / 3;
  const int k = (int)(i - 3 * n);
  const float *p = verts + 3 * (int64_t)faces[3 * (int64_t)orig[n] + k];
  soup[3 * i + 0] = p[0]; soup[3 * i + 1] = p[1]; soup[3 * i + 2] = p[2];
}

// bounding sphere of every block of 64 consecutive soup faces (one wave per block)
__global__ __launch_bounds__(256) void k_block_bounds(const float *__restrict__ soup, int64_t F, float4 *__restrict__ blk) {
  const int64_t f = (int64_t)blockIdx.x * 256 + threadIdx.x;
  float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
  if (f < F) {
    for (int k = 0; k < 3; ++k) {
      const float *p = soup + 9 * f + 3 * k;
      for (int d = 0; d < 3; ++d) { lo[d] = fminf(lo[d], p[d]); hi[d] = fmaxf(hi[d], p[d]); }
    }
  }
  for (int d = 0; d < 3; ++d) {
    for (int o = 32; o > 0; o >>= 1) {
      lo[d] = fminf(lo[d], __shfl_xor(lo[d], o));
      hi[d] = fmaxf(hi[d], __shfl_xor(hi[d], o));
    }
  }
  const int64_t b = f >> 6;  // wave-uniform
  if ((threadIdx.x & 63) == 0 && b * GR_BLOCK < F) {
    float c[3], r2 = 0.f;
    for (int d = 0; d < 3; ++d) {
      c[d] = 0.5f * (lo[d] + hi[d]);
      const float e = 0.5f * (hi[d] - lo[d]);
      r2 += e * e;
    }
    // NaN / inf vertices give a NaN radius: the cull test below is written so that NaN never culls
    blk[b] = make_float4(c[0], c[1], c[2], sqrtf(r2) * 1.0001f + 1e-6f);
  }
}

// K0c  (once per upload) the DISTINCT vertices of every block of 64 soup faces.  The set-up kernel transforms a vertex once
//      per block and view instead of once per face corner: a patch of a manifold mesh has about 48 distinct vertices for its
//      192 corners (a face soup has 192: no worse than before).  One wave per block: the 192 corner indices go to LDS, every
//      corner finds the first corner with the same mesh index (192 broadcast reads), first corners are numbered in order
//      (prefix sum over the wave), and
//        bvert[block][n]  = coordinates of the block's n-th distinct vertex
//        bidx[face]       = n(corner 0) | n(corner 1) << 8 | n(corner 2) << 16 | (distinct vertices of the block - 1) << 24
__global__ __launch_bounds__(256) void k_block_vertices(const float *__restrict__ verts, const int32_t *__restrict__ faces,
                                                        const int32_t *__restrict__ orig, int64_t F, float *__restrict__ bvert,
                                                        uint32_t *__restrict__ bidx) {
  __shared__ int gi_s[4][GR_BLOCK_VERTS];
  __shared__ int lid_s[4][GR_BLOCK_VERTS];
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  int *gi = gi_s[wv], *lid = lid_s[wv];
  const int64_t b = (int64_t)blockIdx.x * 4 + wv;
  if (b * GR_BLOCK >= F) return;   // wave-uniform
  const int64_t f = b * GR_BLOCK + lane;
  const bool live = f < F;
  int g[3] = {-1, -1, -1};
  if (live) {
    const int64_t o = orig[f];
    g[0] = faces[3 * o]; g[1] = faces[3 * o + 1]; g[2] = faces[3 * o + 2];
  }
  for (int k = 0; k < 3; ++k) gi[3 * lane + k] = g[k];
  __builtin_amdgcn_wave_barrier();
  int first[3] = {3 * lane, 3 * lane + 1, 3 * lane + 2};
  for (int s = GR_BLOCK_VERTS - 1; s >= 0; --s) {  // descending: the smallest matching corner is kept
    const int v = gi[s];
#pragma unroll
    for (int k = 0; k < 3; ++k)
      if (v == g[k] && s < first[k]) first[k] = s;
  }
  int mine = 0;
#pragma unroll
  for (int k = 0; k < 3; ++k) mine += (live && first[k] == 3 * lane + k) ? 1 : 0;
  int incl = mine;
  for (int d = 1; d < 64; d <<= 1) {
    const int o = __shfl_up(incl, d);
    if (lane >= d) incl += o;
  }
  const int total = __shfl(incl, 63);
  int n = incl - mine;
#pragma unroll
  for (int k = 0; k < 3; ++k)
    if (live && first[k] == 3 * lane + k) {
      lid[3 * lane + k] = n;
      const float *p = verts + 3 * (int64_t)g[k];
      float *d = bvert + ((int64_t)b * GR_BLOCK_VERTS + n) * 3;
      d[0] = p[0]; d[1] = p[1]; d[2] = p[2];
      ++n;
    }
  __builtin_amdgcn_wave_barrier();
  if (live)
    bidx[f] = (uint32_t)lid[first[0]] | ((uint32_t)lid[first[1]] << 8) | ((uint32_t)lid[first[2]] << 16) | ((uint32_t)(total - 1) << 24);
}

__global__ __launch_bounds__(256) void k_validate_faces(const int32_t *__restrict__ faces, int64_t n, int64_t V,
                                                        int *__restrict__ bad) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int v = faces[i];
  if (v < 0 || v >= V) atomicOr(bad, 1);
}

}  // namespace

extern "C" {

int gr_mesh_upload(gr_ctx *c, const float *verts, const int32_t *faces, int64_t V, int64_t F, void *stream) {
  if (!c) return GR_EINVAL;
  if (!verts || !faces || V <= 0 || F <= 0 || V > 0x7FFFFFFFll || F > 0x7FFFFFF0ll)
    return fail(c, GR_EINVAL, "bad mesh V=%lld F=%lld", (long long)V, (long long)F);
  hipStream_t s = (hipStream_t)stream;
  GR_HIP(c, hipSetDevice(c->device));
  // scratch of the upload: [0] bad-index flag, [1..6] vertex bounds (ordered-uint min x3, max x3)
  uint32_t init[8] = {0, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0, 0, 0, 0};
  uint32_t *up = reinterpret_cast<uint32_t *>(c->flag);
  GR_HIP(c, hipMemcpyAsync(up, init, sizeof(init), hipMemcpyHostToDevice, s));
  hipLaunchKernelGGL(k_validate_faces, dim3((unsigned)ceil_div(3 * F, 256)), dim3(256), 0, s, faces, 3 * F, V, c->flag);
  // few blocks: every wave ends with six atomics on the same six words (2048 blocks spent 0.56 ms queueing on them)
  hipLaunchKernelGGL(k_mesh_bounds, dim3((unsigned)std::min<int64_t>(ceil_div(V, 256), 128)), dim3(256), 0, s, verts, V,
                     up + 1);
  uint32_t got[8];
  GR_HIP(c, hipMemcpyAsync(got, up, sizeof(got), hipMemcpyDeviceToHost, s));
  GR_HIP(c, hipStreamSynchronize(s));
  if (got[0]) return fail(c, GR_EINDEX, "face index outside [0, %lld)", (long long)V);
  // the two axes of largest extent carry the Morton code (16 bits each)
  float lo[3], ext[3];
  for (int d = 0; d < 3; ++d) {
    lo[d] = ordered_float(got[1 + d]);
    const float hi = ordered_float(got[4 + d]);
    ext[d] = (hi >= lo[d]) ? hi - lo[d] : 0.0f;  // no finite vertex on this axis: extent 0
    if (!(ext[d] >= 0.0f) || std::isinf(ext[d])) ext[d] = 0.0f;
  }
  {  // signature of the mesh for the learned-binning table: counts and vertex bounds, mixed (splitmix64 steps)
    uint64_t h = 0x9E3779B97F4A7C15ull ^ (uint64_t)F;
    auto mix = [&h](uint64_t v) { h += v + 0x9E3779B97F4A7C15ull; h = (h ^ (h >> 30)) * 0xBF58476D1CE4E5B9ull; h = (h ^ (h >> 27)) * 0x94D049BB133111EBull; h ^= h >> 31; };
    mix((uint64_t)V);
    for (int d = 1; d <= 6; ++d) mix(got[d]);
    c->mesh_sig = h ? h : 1;
  }
  int ax0 = 0, ax1 = 1, axs = 2;  // ax0, ax1: largest extents
  if (ext[axs] > ext[ax0]) std::swap(axs, ax0);
  if (ext[axs] > ext[ax1]) std::swap(axs, ax1);
  const float inv0 = ext[ax0] > 0.0f ? 65535.0f / ext[ax0] : 0.0f, inv1 = ext[ax1] > 0.0f ? 65535.0f / ext[ax1] : 0.0f;

  const int64_t nblk = ceil_div(F, GR_BLOCK);
  if (c->blk_cap < nblk || c->soup_cap < F) quiesce(c);  // nothing may still read the buffers that are replaced below
  note_stream(c, s);
  if (c->blk_cap < nblk) {
    if (c->blk) (void)hipFree(c->blk);
    c->blk = nullptr; c->blk_cap = 0;
    if (hipMalloc(&c->blk, sizeof(float4) * nblk) != hipSuccess) return fail(c, GR_ENOMEM, "block bounds allocation failed");
    c->blk_cap = nblk;
  }
  if (c->soup_cap < F) {
    if (c->soup) (void)hipFree(c->soup);
    if (c->orig) (void)hipFree(c->orig);
    c->soup = nullptr; c->orig = nullptr; c->soup_cap = 0;
    if (c->bvert) (void)hipFree(c->bvert);
    if (c->bidx) (void)hipFree(c->bidx);
    c->bvert = nullptr; c->bidx = nullptr;
    if (hipMalloc(&c->soup, sizeof(float) * 9 * F) != hipSuccess) return fail(c, GR_ENOMEM, "face soup allocation failed");
    if (hipMalloc(&c->orig, sizeof(int32_t) * F) != hipSuccess) return fail(c, GR_ENOMEM, "face order allocation failed");
    if (hipMalloc(&c->bvert, sizeof(float) * 3 * GR_BLOCK_VERTS * ceil_div(F, GR_BLOCK)) != hipSuccess)
      return fail(c, GR_ENOMEM, "block vertex allocation failed");
    if (hipMalloc(&c->bidx, sizeof(uint32_t) * F) != hipSuccess) return fail(c, GR_ENOMEM, "block index allocation failed");
    c->soup_cap = F;
  }
  // Morton codes -> stable radix sort of (code, face) pairs (rocPRIM through hipcub) -> orig[]
  size_t sort_bytes = 0;
  GR_HIP(c, hipcub::DeviceRadixSort::SortPairs(nullptr, sort_bytes, (uint32_t *)nullptr, (uint32_t *)nullptr,
                                               (int32_t *)nullptr, (int32_t *)nullptr, (int)F, 0, 32, s));
  const size_t arr = ((size_t)F * 4 + 255) / 256 * 256;
  const size_t need = 3 * arr + sort_bytes + 256;
  if (c->sort_bytes < need) {
    GR_HIP(c, hipStreamSynchronize(s));
    if (c->sort_tmp) (void)hipFree(c->sort_tmp);
    c->sort_tmp = nullptr; c->sort_bytes = 0;
    if (hipMalloc(&c->sort_tmp, need) != hipSuccess) return fail(c, GR_ENOMEM, "sort scratch allocation failed");
    c->sort_bytes = need;
  }
  char *base = static_cast<char *>(c->sort_tmp);
  uint32_t *code_in = reinterpret_cast<uint32_t *>(base), *code_out = reinterpret_cast<uint32_t *>(base + arr);
  int32_t *idx_in = reinterpret_cast<int32_t *>(base + 2 * arr);
  void *tmp = base + 3 * arr;
  hipLaunchKernelGGL(k_face_codes, dim3((unsigned)ceil_div(F, 256)), dim3(256), 0, s, verts, faces, F, ax0, ax1, lo[ax0], inv0,
                     lo[ax1], inv1, code_in, idx_in);
  size_t tb = sort_bytes;
  GR_HIP(c, hipcub::DeviceRadixSort::SortPairs(tmp, tb, code_in, code_out, idx_in, c->orig, (int)F, 0, 32, s));
  hipLaunchKernelGGL(k_build_soup, dim3((unsigned)ceil_div(3 * F, 256)), dim3(256), 0, s, verts, faces, c->orig, F, c->soup);
  hipLaunchKernelGGL(k_block_bounds, dim3((unsigned)ceil_div(F, 256)), dim3(256), 0, s, c->soup, F, c->blk);
  hipLaunchKernelGGL(k_block_vertices, dim3((unsigned)ceil_div(nblk, 4)), dim3(256), 0, s, verts, faces, c->orig, F, c->bvert, c->bidx);
  GR_HIP(c, hipGetLastError());
  c->verts = verts; c->faces = faces; c->V = V; c->F = F;
  c->stats_deferred = false;   // (view totals a raster call on the old mesh left for the status call: void)
  c->visits_pending = false;   // (... and the visit counters of its vote passes: sized for the old mesh)
  return GR_OK;
}

}  // extern "C"
