// geograypher_amd/csrc/geograster.hip -- hand-written CDNA4 (gfx950, wave64) kernels + the C ABI of include/geograster.h.
//
// Hot path of geograypher re-designed for MI355X (reference lines in include/geograster.h and DESIGN.md):
//   pix2face            k_cull_blocks -> k_setup_cull (+ k_clip_faces) -> k_bin_stats -> k_raster_tile
//                       (single-pass binning; exact fallback: k_setup_cull<false> -> k_scan_tiles -> k_fill_compile)
//                                                                                          (meshes.py:1776-1836)
//   project/aggregate   k_raster_tile<FUSE> | k_winner  ->  k_vote_labels / k_vote_values (meshes.py:1987-2002, 2057-2067)
//   render_flat gather  k_gather_texture                                                  (meshes.py:1921-1937)
//   distortion (row f1) k_invert_distortion (once per lens), k_warp_nearest_i32 / k_warp_f64 (cameras.py:995-1156)
// No MFMA anywhere: there is no dense contraction on this path.  The work is integer edge functions, an
// LDS-resident depth|id tile per workgroup, wave ballot/popcount compaction of surviving faces and global
// atomicMax/atomicAdd for bins and per-face winners.  The tile kernel is bound by VALU issue, then by the LDS pipe
// (DESIGN.md section 5): its inner pieces are written for instruction count.
//
// Rule-set R0-R7 is implemented here independently of oracle/oracle_raster.c; tests demand equality.
// Compile with -ffp-contract=off: every floating-point operation below is individually rounded on purpose.

#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <algorithm>

#include <cmath>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <new>
#include <vector>

#include "geograster.h"

// ------------------------------------------------------------------------------------------------------------------
// constants
// ------------------------------------------------------------------------------------------------------------------
#define GR_TILE 64          // tile width in pixels (a workgroup rasterizes 64x32 or 64x64 tiles out of LDS)
#define GR_TILE_LOG2 6
#define GR_MAX_BATCH 64     // views per launch group (amortises kernel boundaries and per-launch tails)
#define GR_ENT_Q 3          // int4 per compiled (face, tile) entry: 48 bytes, 12 words
#define GR_CTRL_HDR 8       // ctrl words before the tile arrays: rec_count, total_entries, overflow, work_count, clip_count, big_count
#define GR_MAX_DIM 16384    // h, w limit (guard band and 16-bit bbox packing)

namespace {

struct BinArgs {
  uint32_t *ctrl;        // [slot][GR_CTRL_HDR + 4*Tcap]  rec_count,total,overflow,work_count,clip_count,- | cntS[T] | cntB[T] | offset[T] | curB[T]
                         //   cntS: entries whose list position was handed out in k_setup_cull (faces touching <= 2x2 tiles)
                         //   cntB: entries of larger faces, placed by k_fill_compile behind the cntS block of their tile (exact path)
  int4 *rec;             // [slot][4][F]  plane0 {X0,Y0,X1,Y1} plane1 {X2,Y2,iz0,face} plane2 {A,B,jmin|jmax<<16,imin|imax<<16}
                         //               plane3 {list position in up to 4 tiles}
  const float *soup;     // [F][9] the three vertex positions of every face, in Morton order (built once per upload)
  const int32_t *orig;   // [F] soup position -> face id of the caller's mesh
  const float4 *blk;     // [ceil(F/64)] bounding sphere (centre, radius) of each block of GR_BLOCK faces, local frame
  const uint32_t *blk_chunks;  // [ceil(F/64)][17] count (or ~0: more than 16) + the 256-face chunks of CALLER ids the block's faces lie in
  uint32_t *touched;     // [slot][tw] bit per 256-face chunk of caller ids that a surviving block reaches (+ last word: all), or null
  int tw;                // words per slot of `touched`
  uint32_t *work;        // [slot][work_stride] blocks of this view that passed the frustum test (ctrl[3] = count)
  uint32_t *clip;        // [slot][F] from the front: soup faces that straddle the near plane / guard band (R7; ctrl[4] = count);
                         //           from the back: faces over more than 2 x 2 tiles (single-pass binning; ctrl[5] = count)
  int64_t work_stride;
  int4 *comp;            // [slot][ent_cap][GR_ENT_Q]  compiled (face, tile) entries grouped by tile, 48 bytes each (ent40: 40 bytes
                         //                            each at the front of the same slot memory)
  uint8_t *nrow8;        // [slot][ent_cap] rows of each entry inside its tile (the tile kernel's scan input: a compact stream)
  unsigned long long *stats;  // [6] records, entries, max_entries, overflow, first overflowed launch group (over the call),
                              //     short-form miss (a face the 40-byte entry cannot hold: the caller repeats with 48 bytes)
  int group;             // index of this launch group inside the call
  int64_t ctrl_stride;   // words per slot
  int64_t rec_stride;    // int4 per slot (= 3*F)
  int64_t ent_cap;       // entries per slot
  int64_t F;
  int T, TX, TY, Tcap;
  int h, w;
  int twl, thl;          // log2 of the tile width / height in pixels
  int cap_tile;          // > 0: single-pass binning, every tile owns cap_tile entry slots (list base = tile * cap_tile)
  int ent40;             // 1: entries are written in the SHORT form (40 bytes, store_entry below); single-pass binning only
  int var;               // variant bits (GR_OPT_VARIANT): 1 = one tile per workgroup instead of four, 4 = votes on the caller's stream, 8 = no speculative first chunk, 16 = chains of four whatever the launch looks like, 32 = votes without chunk bitmaps
  int dbg;               // timing-only ablation mask (GR_OPT_DEBUG): 1 skip scanline loop, 2 skip id stores, 4 skip triangles,
                         // fused epilogue: 8 skip winner atomics, 16 skip label loads; set-up: 32 no entry compilation, 64 no
                         // second-to-fourth tiles of small faces, 256 no depth gradients
};

__device__ __forceinline__ int imin3(int a, int b, int c) { return min(a, min(b, c)); }
__device__ __forceinline__ int imax3(int a, int b, int c) { return max(a, max(b, c)); }

struct Vtx {
  int X, Y;
  float iz;
  bool valid;
  bool front, finite;  // q_z > near; camera-space point finite (R7: which invalid faces are clipped instead of dropped)
};

// R1 -- vertex transform, fp32, each operation individually rounded
__device__ __forceinline__ Vtx project_vertex(const float *__restrict__ p, const float *__restrict__ cam) {
  Vtx v;
  const float dx = p[0] - cam[9];
  const float dy = p[1] - cam[10];
  const float dz = p[2] - cam[11];
  float m0, m1, m2;
  m0 = cam[0] * dx; m1 = cam[3] * dy; m2 = cam[6] * dz;
  const float qx = (m0 + m1) + m2;
  m0 = cam[1] * dx; m1 = cam[4] * dy; m2 = cam[7] * dz;
  const float qy = (m0 + m1) + m2;
  m0 = cam[2] * dx; m1 = cam[5] * dy; m2 = cam[8] * dz;
  const float qz = (m0 + m1) + m2;
  v.valid = qz > cam[15];
  v.front = v.valid;
  v.finite = isfinite(qx) && isfinite(qy) && isfinite(qz);
  const float iz = 1.0f / qz;  // correctly rounded (-fhip-fp32-correctly-rounded-divide-sqrt)
  const float fx = cam[12] * qx;
  const float fy = cam[12] * qy;
  const float sx = cam[13] + fx * iz;
  const float sy = cam[14] + fy * iz;
  v.valid = v.valid && (fabsf(sx) < 16384.0f) && (fabsf(sy) < 16384.0f);
  v.X = (int)floorf(sx * 256.0f + 0.5f);
  v.Y = (int)floorf(sy * 256.0f + 0.5f);
  v.iz = iz;
  return v;
}

// ------------------------------------------------------------------------------------------------------------------
// K0  (once per mesh upload) the mesh is re-ordered along a Morton curve of the face centroids and de-indexed:
//     soup[n] = the 9 vertex coordinates of face orig[n].  A block of GR_BLOCK = 64 consecutive soup faces is then a
//     compact patch whatever the caller's face order: one bounding sphere per block rejects most of a survey mesh with
//     one test per view, and the 64 faces of a wave fall into one to four tiles (few, long runs for the tile counters).
//     Rasterization does not depend on the order in which faces are processed (ds_max_u64 resolve), ids are the caller's.
// ------------------------------------------------------------------------------------------------------------------
#define GR_BLOCK 64
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

// ------------------------------------------------------------------------------------------------------------------
// K1  transform + cull + per-tile counts (+ compiled entries in single-pass mode).   grid (<= 1024, views)
//     (a) work list: the 64-face blocks whose bounding sphere passed k_cull_blocks (~87 % of a survey mesh is rejected
//         per view before a single face is read); every wave takes its own blocks;
//     (b) the face's three vertices are read from the de-indexed soup (36 coalesced bytes per lane);
//     (c) exact path: survivors compacted with wave ballot + popcount, ONE atomicAdd per wave; record planes written
//         as consecutive 16-byte slots (full-rate coalesced stores);
//     (d) tile counting is aggregated per wave as well: neighbouring lanes that hit the same tile share one returning
//         atomicAdd and receive consecutive list positions; single-pass mode compiles and stores the entries at once,
//         the exact path leaves that to k_fill_compile (no atomics there for faces over at most 2x2 tiles).
// ------------------------------------------------------------------------------------------------------------------
// Lanes of a wave that hit the same tile form a group: leader lane, rank inside the group, group size -- found with
// ballots and shuffles only (no memory traffic), so that the leaders' atomics can all be issued back to back.  The 64
// faces of a wave are a compact patch of the mesh (Morton order): a handful of distinct tiles, hence few iterations.
__device__ __forceinline__ void wave_group(int t, int lane, int &leader, int &rank, int &size) {
  leader = lane; rank = 0; size = 0;
  unsigned long long rem = __ballot(t >= 0);
  while (rem) {
    const int l = __ffsll((long long)rem) - 1;
    const int tl = __builtin_amdgcn_readlane(t, l);
    const unsigned long long m = __ballot(t == tl);
    if (t == tl) {
      leader = l;
      rank = __popcll(m & ((1ull << lane) - 1ull));
      size = __popcll(m);
    }
    rem &= ~m;
  }
}

// The same with at most `max_groups` groups looked for: lanes that are left over stand alone (leader = itself, size 1).
// For the (face, tile) pairs of big faces, where a step of 64 pairs can name 64 different tiles.
__device__ __forceinline__ void wave_group_capped(int t, int lane, int &leader, int &rank, int &size, int max_groups) {
  leader = lane; rank = 0; size = 1;
  unsigned long long rem = __ballot(t >= 0);
  for (int g = 0; rem && g < max_groups; ++g) {
    const int l = __ffsll((long long)rem) - 1;
    const int tl = __builtin_amdgcn_readlane(t, l);
    const unsigned long long m = __ballot(t == tl);
    if (t == tl) {
      leader = l;
      rank = __popcll(m & ((1ull << lane) - 1ull));
      size = __popcll(m);
    }
    rem &= ~m;
  }
}

// K0a  (once per upload) for every block of 64 soup faces: the 256-face chunks of the CALLER's face ids its faces lie in
//      (at most 16 listed; a block whose faces are scattered over more says so).  The fused aggregation marks, per view,
//      the chunks that surviving blocks reach, and its vote kernel -- one workgroup per chunk -- reads the winners of the
//      views that can have any.  One wave per block.
#define GR_CHUNK_LIST 16
__global__ __launch_bounds__(256) void k_block_chunks(const int32_t *__restrict__ orig, int64_t F, uint32_t *__restrict__ out) {
  const int64_t b = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const int64_t f = b * GR_BLOCK + lane;
  if (b * GR_BLOCK >= F) return;
  const int ch = f < F ? (orig[f] >> 8) : -1;
  unsigned long long rem = __ballot(ch >= 0);
  int n = 0;
  while (rem && n < GR_CHUNK_LIST) {
    const int cl = __builtin_amdgcn_readlane(ch, __ffsll((long long)rem) - 1);
    if (lane == 0) out[b * (GR_CHUNK_LIST + 1) + 1 + n] = (uint32_t)cl;
    rem &= ~__ballot(ch == cl);
    ++n;
  }
  if (lane == 0) out[b * (GR_CHUNK_LIST + 1)] = rem ? 0xFFFFFFFFu : (uint32_t)n;
}

// K0b  per view: sphere-vs-frustum test of every 64-face block (one thread per block); survivors are appended to the
//      view's work list with one wave-aggregated atomic.  grid (ceil(nblk/256), views)
__global__ __launch_bounds__(256) void k_cull_blocks(const float *__restrict__ cams, BinArgs a, int nblk) {
  const int slot = blockIdx.y;
  const int b = blockIdx.x * 256 + threadIdx.x;
  const float *cam = cams + (int64_t)slot * GR_CAM_FLOATS;
  uint32_t *ctrl = a.ctrl + slot * a.ctrl_stride;
  bool keep = false;
  if (b < nblk) {
    // camera space; planes carry a 2-pixel margin; any NaN keeps the block
    const float4 sp = a.blk[b];
    const float dx = sp.x - cam[9], dy = sp.y - cam[10], dz = sp.z - cam[11];
    const float qx = cam[0] * dx + cam[3] * dy + cam[6] * dz;
    const float qy = cam[1] * dx + cam[4] * dy + cam[7] * dz;
    const float qz = cam[2] * dx + cam[5] * dy + cam[8] * dz;
    const float fe = fabsf(cam[12]), r = sp.w * 1.001f;
    const float mxl = cam[13] + 2.0f, mxr = (float)a.w - cam[13] + 2.0f;
    const float myt = cam[14] + 2.0f, myb = (float)a.h - cam[14] + 2.0f;
    bool out = (qz + r < cam[15]);
    out = out || (cam[12] * qx + mxl * qz < -r * (fe + fabsf(mxl)));
    out = out || (-cam[12] * qx + mxr * qz < -r * (fe + fabsf(mxr)));
    out = out || (cam[12] * qy + myt * qz < -r * (fe + fabsf(myt)));
    out = out || (-cam[12] * qy + myb * qz < -r * (fe + fabsf(myb)));
    keep = !out;
  }
  const unsigned long long m = __ballot(keep);
  if (m != 0ull) {
    const int lane = threadIdx.x & 63;
    const int leader = __ffsll((long long)m) - 1;
    uint32_t base = 0;
    if (lane == leader) base = atomicAdd(&ctrl[3], (uint32_t)__popcll(m));
    base = __shfl(base, leader);
    if (keep) a.work[(int64_t)slot * a.work_stride + base + __popcll(m & ((1ull << lane) - 1ull))] = (uint32_t)b;
  }
  // fused aggregation: which 256-face chunks of caller ids can receive winners in this view.  The workgroup's 256 blocks
  // are neighbours on the Morton curve and share most of their chunks: the bits are collected in LDS and every non-zero
  // word leaves the workgroup as one atomicOr.
  if (a.touched) {
    extern __shared__ uint32_t bits[];
    for (int i = threadIdx.x; i < a.tw; i += 256) bits[i] = 0u;
    __syncthreads();
    if (keep) {
      const uint32_t *cl = a.blk_chunks + (int64_t)b * (GR_CHUNK_LIST + 1);
      const uint32_t n = cl[0];
      if (n == 0xFFFFFFFFu) atomicOr(&bits[a.tw - 1], 1u);
      else
        for (uint32_t i = 0; i < n; ++i) {
          const uint32_t ch = cl[1 + i];
          atomicOr(&bits[ch >> 5], 1u << (ch & 31u));
        }
    }
    __syncthreads();
    uint32_t *dst = a.touched + (int64_t)slot * a.tw;
    for (int i = threadIdx.x; i < a.tw; i += 256)
      if (bits[i]) atomicOr(&dst[i], bits[i]);
  }
}

__device__ __forceinline__ bool compile_entry(const BinArgs &a, uint32_t *__restrict__ ctrl, int4 *__restrict__ comp,
                                              uint8_t *__restrict__ nr8, int64_t idx, const int4 p0, const int4 p1,
                                              const int4 p2, int px0, int py0, int TW, int TH);
__device__ __forceinline__ bool build_entry(const int4 p0, const int4 p1, const int4 p2, int px0, int py0, int TW, int TH,
                                            int4 &e0, int4 &e1, int4 &e2, int &rows);
__device__ __forceinline__ void store_entry(const BinArgs &a, uint32_t *__restrict__ ctrl, int4 *__restrict__ comp,
                                            uint8_t *__restrict__ nr8, int64_t idx, const int4 e0, const int4 e1, const int4 e2,
                                            int rows);
__device__ __forceinline__ int wave_incl_scan(int x);

// R1 / R2 / R4 for one face of the soup: the record (three int4) that compile_entry turns into per-tile entries, and the
// range of tiles its pixel bounding box touches.  Returns false for faces that draw nothing in this view; clip_me: the face
// straddles the near plane or the guard band (R7).  Used by K1 and, for faces over more than 2 x 2 tiles, by k_bin_big:
// same code, same bits.
__device__ __forceinline__ bool face_setup(const BinArgs &a, const float *__restrict__ cam, int64_t f, int4 &r0, int4 &r1,
                                           int4 &r2, int &tx0, int &tx1, int &ty0, int &ty1, bool &clip_me) {
  // the face's three vertices sit side by side in the soup: one coalesced 36-byte read per lane instead of an index
  // load followed by three dependent 12-byte gathers (one dependent memory round trip less per wave)
  const float *sp = a.soup + 9 * f;
  Vtx v0 = project_vertex(sp, cam);
  Vtx v1 = project_vertex(sp + 3, cam);
  Vtx v2 = project_vertex(sp + 6, cam);
  clip_me = !(v0.valid && v1.valid && v2.valid) && (v0.front || v1.front || v2.front) && v0.finite && v1.finite && v2.finite;
  if (!(v0.valid && v1.valid && v2.valid)) return false;
  long long area2 = (long long)(v1.X - v0.X) * (long long)(v2.Y - v0.Y) - (long long)(v2.X - v0.X) * (long long)(v1.Y - v0.Y);
  if (area2 == 0) return false;
  if (area2 < 0) {  // both windings are drawn: normalise to positive area
    Vtx s = v1; v1 = v2; v2 = s;
    area2 = -area2;
  }
  const int Xmin = imin3(v0.X, v1.X, v2.X), Xmax = imax3(v0.X, v1.X, v2.X);
  const int Ymin = imin3(v0.Y, v1.Y, v2.Y), Ymax = imax3(v0.Y, v1.Y, v2.Y);
  int jmin = (Xmin - 128 + 255) >> 8, jmax = (Xmax - 128) >> 8;  // R2: pixel centres inside the bbox
  int imin = (Ymin - 128 + 255) >> 8, imax = (Ymax - 128) >> 8;
  jmin = max(jmin, 0); imin = max(imin, 0);
  jmax = min(jmax, a.w - 1); imax = min(imax, a.h - 1);
  if (jmin > jmax || imin > imax) return false;
  // R4: gradients of 1/z in double, rounded once to float
  float A = 0.f, B = 0.f;
  if (!(a.dbg & 256)) {
    const double d1 = (double)v1.iz - (double)v0.iz;
    const double d2 = (double)v2.iz - (double)v0.iz;
    const double a2 = (double)area2;
    double n1, n2;
    n1 = d1 * (double)(v2.Y - v0.Y); n2 = d2 * (double)(v1.Y - v0.Y);
    A = (float)((n1 - n2) / a2);
    n1 = d2 * (double)(v1.X - v0.X); n2 = d1 * (double)(v2.X - v0.X);
    B = (float)((n1 - n2) / a2);
  }
  r0 = make_int4(v0.X, v0.Y, v1.X, v1.Y);
  r1 = make_int4(v2.X, v2.Y, __float_as_int(v0.iz), a.orig[f]);
  r2 = make_int4(__float_as_int(A), __float_as_int(B), jmin | (jmax << 16), imin | (imax << 16));
  tx0 = jmin >> a.twl; tx1 = jmax >> a.twl;
  ty0 = imin >> a.thl; ty1 = imax >> a.thl;
  return true;
}

// Single-pass binning of the wave's faces that reach over more than 2 x 2 tiles (`big`: this lane holds one, records r0 .. r2,
// tile rectangle tx0 .. ty1).  A per-lane walk over the tiles would leave 63 lanes waiting for the largest face -- 112 us per
// view on a scene with 20 000 trees seen obliquely (canopy and trunk faces of 300 x 40 pixels), where the terrain alone
// takes 7.  Instead the wave prefix-sums the tile counts of its faces and EXPANDS: the (face, tile) pairs are taken 64 at a
// time, a pair finds its face by a 6-step search over the prefix sums and pulls the record out of the owning lane's
// registers (ds_bpermute).  A tile the triangle does not touch takes no list slot.  The pairs of a step that name the same
// tile (neighbouring faces of one tree do) share ONE returning counter atomic (wave_group_capped: at most 16 groups are
// looked for, left-over pairs stand alone); all atomics of a step are in flight together.
__device__ __forceinline__ void bin_big_pairs(const BinArgs &a, uint32_t *__restrict__ ctrl, const int slot, const int lane,
                                              const bool big, const int4 r0, const int4 r1, const int4 r2, const int tx0,
                                              const int tx1, const int ty0, const int ty1) {
  uint32_t *cntS = ctrl + GR_CTRL_HDR;
  int4 *comp = a.comp + slot * a.ent_cap * GR_ENT_Q;
  uint8_t *nr8 = a.nrow8 + slot * a.ent_cap;
  const int TW = 1 << a.twl, TH = 1 << a.thl;
  const int ntx = tx1 - tx0 + 1;
  const int nt = big ? ntx * (ty1 - ty0 + 1) : 0;
  const int incl = wave_incl_scan(nt);
  const int total = __builtin_amdgcn_readlane(incl, 63);
  const int geo = tx0 | (ty0 << 12) | ((ntx - 1) << 24);  // at most 256 x 512 tiles per image (GR_MAX_DIM)
  for (int k0 = 0; k0 < total; k0 += 64) {
    const int q = k0 + lane;
    int t = 0;  // the face of pair q: the first lane whose inclusive sum exceeds q
#pragma unroll
    for (int step = 32; step >= 1; step >>= 1) t += (__shfl(incl, t + step - 1) <= q) ? step : 0;
    t = min(t, 63);
    const int ex = __shfl(incl, t) - __shfl(nt, t);
    const int g = __shfl(geo, t);
    const int4 p0 = make_int4(__shfl(r0.x, t), __shfl(r0.y, t), __shfl(r0.z, t), __shfl(r0.w, t));
    const int4 p1 = make_int4(__shfl(r1.x, t), __shfl(r1.y, t), __shfl(r1.z, t), __shfl(r1.w, t));
    const int4 p2 = make_int4(__shfl(r2.x, t), __shfl(r2.y, t), __shfl(r2.z, t), __shfl(r2.w, t));
    int tile = -1, rows = 0;
    int4 e0 = {0, 0, 0, 0}, e1 = {0, 0, 0, 0}, e2 = {0, 0, 0, 0};
    if (q < total) {
      const int k = q - ex, gtx = g & 0xFFF, gty = (g >> 12) & 0xFFF, gn = (int)((uint32_t)g >> 24) + 1;
      const int tx = gtx + k % gn, ty = gty + k / gn;
      if (build_entry(p0, p1, p2, tx << a.twl, ty << a.thl, TW, TH, e0, e1, e2, rows)) tile = ty * a.TX + tx;
    }
    int ld, rk, sz;
    wave_group_capped(tile, lane, ld, rk, sz, 16);
    uint32_t base = 0;
    if (tile >= 0 && lane == ld) base = atomicAdd(&cntS[tile], (uint32_t)sz);
    const uint32_t pos = __shfl(base, ld) + (uint32_t)rk;
    if (tile >= 0) {
      if (pos < (uint32_t)a.cap_tile) {
        store_entry(a, ctrl, comp, nr8, (int64_t)tile * a.cap_tile + pos, e0, e1, e2, rows);
      } else atomicOr(&ctrl[2], 1u);
    }
  }
}

// DIRECT = true: single-pass binning.  Every tile owns a fixed segment of a.cap_tile entries; the list position
// returned by the (wave-aggregated) tile counter is final, so the compiled entry is written straight from here and
// the record planes, k_scan_tiles and k_fill_compile are skipped.  A tile that receives more than cap_tile entries
// raises the view's overflow word; the caller then repeats the call with the exact two-pass path (DIRECT = false).
template <bool DIRECT>
__global__ __launch_bounds__(256) void k_setup_cull(const float *__restrict__ cams, BinArgs a) {
  const int slot = blockIdx.y;
  const float *cam = cams + (int64_t)slot * GR_CAM_FLOATS;
  uint32_t *ctrl = a.ctrl + slot * a.ctrl_stride;
  const uint32_t *work = a.work + (int64_t)slot * a.work_stride;
  // every wave takes its own 64-face block from the view's work list (wave-uniform control flow, no workgroup barrier)
  const int lane = threadIdx.x & 63;
  const uint32_t wave0 = blockIdx.x * 4 + (threadIdx.x >> 6), wstep = gridDim.x * 4;
  uint32_t blk_next = work[wave0];       // read alongside the count (any slot of the list is valid memory)
  const uint32_t n_work = ctrl[3];       // (a) blocks that passed k_cull_blocks for this view
  uint32_t n_rec = 0;                    // single-pass binning: the wave's record count (a statistic), added once at the end
  for (uint32_t wi = wave0; wi < n_work; wi += wstep) {
  const int64_t f = (int64_t)blk_next * GR_BLOCK + lane;
  if (wi + wstep < n_work) blk_next = work[wi + wstep];

  bool keep = false, clip_me = false;
  int4 r0 = {0, 0, 0, 0}, r1 = {0, 0, 0, 0}, r2 = {0, 0, 0, 0};
  int tx0 = 0, tx1 = -1, ty0 = 0, ty1 = -1;
  if (f < a.F) keep = face_setup(a, cam, f, r0, r1, r2, tx0, tx1, ty0, ty1, clip_me);
  // R7: faces that straddle the near plane or the guard band go to the view's clip list (k_clip_faces)
  const unsigned long long mc = __ballot(clip_me);
  if (mc) {
    const int lead = __ffsll((long long)mc) - 1;
    uint32_t cb = 0;
    if (lane == lead) cb = atomicAdd(&ctrl[4], (uint32_t)__popcll(mc));
    cb = __shfl(cb, lead);
    if (clip_me) a.clip[(int64_t)slot * a.F + cb + __popcll(mc & ((1ull << lane) - 1ull))] = (uint32_t)f;
  }
  // wave-level compaction of survivors
  const unsigned long long m = __ballot(keep);
  if (m == 0ull) continue;
  const int n = __popcll(m);
  const int prefix = __popcll(m & ((1ull << lane) - 1ull));
  const int leader = __ffsll((long long)m) - 1;
  // (d) tile counts.  Faces touching at most 2x2 tiles get their list positions here (wave-aggregated atomics);
  //     larger faces are only counted (cntB) and placed by k_fill_compile.  Groups are found first (registers only),
  //     then ALL atomics of the wave -- record slot + up to four tile counters -- are issued before any is consumed.
  const bool small_fp = keep && (tx1 - tx0 <= 1) && (ty1 - ty0 <= 1);
  uint32_t *cntS = ctrl + GR_CTRL_HDR;
  uint32_t *cntB = cntS + a.Tcap;
  const int t00 = small_fp ? ty0 * a.TX + tx0 : -1;
  const int t01 = (small_fp && tx1 > tx0) ? ty0 * a.TX + tx1 : -1;
  const int t10 = (small_fp && ty1 > ty0) ? ty1 * a.TX + tx0 : -1;
  const int t11 = (small_fp && tx1 > tx0 && ty1 > ty0) ? ty1 * a.TX + tx1 : -1;
  int l0, k0, n0, l1 = lane, k1 = 0, n1 = 0, l2 = lane, k2 = 0, n2 = 0, l3 = lane, k3 = 0, n3 = 0;
  wave_group(t00, lane, l0, k0, n0);
  if (__ballot(t01 >= 0)) wave_group(t01, lane, l1, k1, n1);
  if (__ballot(t10 >= 0)) wave_group(t10, lane, l2, k2, n2);
  if (__ballot(t11 >= 0)) wave_group(t11, lane, l3, k3, n3);
  uint32_t base = 0, b0 = 0, b1 = 0, b2 = 0, b3 = 0;
  // record count: a list position for the exact path; a statistic otherwise, kept in a register until the wave is done (one
  // atomic per block on the view's one address made every wave of the view queue there: same-address atomics are served
  // one after the other, tools/ubench/atomic_rate.hip)
  if (DIRECT) n_rec += (uint32_t)n;
  else if (lane == leader) base = atomicAdd(&ctrl[0], (uint32_t)n);
  if (t00 >= 0 && lane == l0) b0 = atomicAdd(&cntS[t00], (uint32_t)n0);
  if (t01 >= 0 && lane == l1) b1 = atomicAdd(&cntS[t01], (uint32_t)n1);
  if (t10 >= 0 && lane == l2) b2 = atomicAdd(&cntS[t10], (uint32_t)n2);
  if (t11 >= 0 && lane == l3) b3 = atomicAdd(&cntS[t11], (uint32_t)n3);
  if (!DIRECT) base = __shfl(base, leader);
  int4 r3;
  r3.x = (int)(__shfl(b0, l0) + (uint32_t)k0);
  r3.y = (int)(__shfl(b1, l1) + (uint32_t)k1);
  r3.z = (int)(__shfl(b2, l2) + (uint32_t)k2);
  r3.w = (int)(__shfl(b3, l3) + (uint32_t)k3);
  if (DIRECT) {
    // faces over at most 2x2 tiles: positions came from the wave-aggregated counters; the lanes of a group hold
    // consecutive positions of the same tile segment, so their 48-byte entries are written side by side.  Every such face has
    // a FIRST tile: one dense round of entry compilation.  Second to fourth tiles are the exception (0.5 per face): instead of
    // three more rounds in which most lanes wait (the set-up kernel of a forest scene is VALU-bound: SQ counters in
    // profiles/), those (face, tile) pairs are dealt to the lanes -- prefix sum of the extra tiles per face, 6-step search for
    // the owning lane, records pulled from its registers (ds_bpermute) -- and take one round together.
    int4 *comp = a.comp + slot * a.ent_cap * GR_ENT_Q;
    uint8_t *nr8 = a.nrow8 + slot * a.ent_cap;
    const int TW = 1 << a.twl, TH = 1 << a.thl;
    if (small_fp && !(a.dbg & 32)) {
      if ((uint32_t)r3.x < (uint32_t)a.cap_tile) {
        const int64_t idx = (int64_t)t00 * a.cap_tile + (uint32_t)r3.x;
        compile_entry(a, ctrl, comp, nr8, idx, r0, r1, r2, tx0 << a.twl, ty0 << a.thl, TW, TH);
      } else atomicOr(&ctrl[2], 1u);
    }
    const int shape = small_fp ? ((tx1 > tx0 ? 1 : 0) | (ty1 > ty0 ? 2 : 0)) : 0;  // which neighbours exist: 1 right, 2 below
    const int ne = shape == 3 ? 3 : (shape ? 1 : 0);
    const int incl_e = wave_incl_scan(ne);
    const int total_e = (a.dbg & (32 | 64)) ? 0 : __builtin_amdgcn_readlane(incl_e, 63);
    const int geo = tx0 | (ty0 << 12) | (shape << 24);
    for (int k0 = 0; k0 < total_e; k0 += 64) {
      const int q = k0 + lane;
      int t = 0;  // the face of pair q: the first lane whose inclusive sum exceeds q
#pragma unroll
      for (int step = 32; step >= 1; step >>= 1) t += (__shfl(incl_e, t + step - 1) <= q) ? step : 0;
      t = min(t, 63);
      const int g = __shfl(geo, t);
      const int sh = (g >> 24) & 3;
      const int which = q - (__shfl(incl_e, t) - (sh == 3 ? 3 : 1));  // 0 .. 2: the face's extra tile
      const int k = sh == 3 ? which + 1 : sh;                          // tile slot 1 (right), 2 (below), 3 (below right)
      const int4 p0 = make_int4(__shfl(r0.x, t), __shfl(r0.y, t), __shfl(r0.z, t), __shfl(r0.w, t));
      const int4 p1 = make_int4(__shfl(r1.x, t), __shfl(r1.y, t), __shfl(r1.z, t), __shfl(r1.w, t));
      const int4 p2 = make_int4(__shfl(r2.x, t), __shfl(r2.y, t), __shfl(r2.z, t), __shfl(r2.w, t));
      const int py = __shfl(r3.y, t), pz = __shfl(r3.z, t), pw = __shfl(r3.w, t);
      if (q < total_e) {
        const uint32_t pos = (uint32_t)(k == 1 ? py : k == 2 ? pz : pw);
        const int tx = (g & 0xFFF) + (k & 1), ty = ((g >> 12) & 0xFFF) + (k >> 1);
        if (pos < (uint32_t)a.cap_tile) {
          const int64_t idx = (int64_t)(ty * a.TX + tx) * a.cap_tile + pos;
          compile_entry(a, ctrl, comp, nr8, idx, p0, p1, p2, tx << a.twl, ty << a.thl, TW, TH);
        } else atomicOr(&ctrl[2], 1u);
      }
    }
  }
  if (DIRECT) {
    // faces over more than 2 x 2 tiles: the wave expands their (face, tile) pairs right here, from the records it holds
    // (bin_big_pairs).  Variant bit 64: they go to the view's big list instead (the back of the clip buffer, ctrl[5] = count)
    // and k_bin_big sets them up again, 64 per wave -- one returning atomic per block on ONE address per view.
    const bool big_fp = keep && !small_fp;
    const unsigned long long mb = __ballot(big_fp);
    if (mb) {
      if (!(a.var & 64)) {
        bin_big_pairs(a, ctrl, slot, lane, big_fp, r0, r1, r2, tx0, tx1, ty0, ty1);
      } else {
        const int lead = __ffsll((long long)mb) - 1;
        uint32_t bb = 0;
        if (lane == lead) bb = atomicAdd(&ctrl[5], (uint32_t)__popcll(mb));
        bb = __shfl(bb, lead);
        if (big_fp) a.clip[(int64_t)slot * a.F + (a.F - 1 - (int64_t)(bb + __popcll(mb & ((1ull << lane) - 1ull))))] = (uint32_t)f;
      }
    }
  }
  if (keep && !DIRECT) {
    int4 *rec = a.rec + slot * a.rec_stride;
    const int64_t s = (int64_t)base + prefix;
    rec[s] = r0;
    rec[a.F + s] = r1;
    rec[2 * a.F + s] = r2;
    rec[3 * a.F + s] = r3;
    if (!small_fp)
      for (int ty = ty0; ty <= ty1; ++ty)
        for (int tx = tx0; tx <= tx1; ++tx) atomicAdd(&cntB[ty * a.TX + tx], 1u);
  }
  }  // work list loop
  if (DIRECT && lane == 0 && n_rec) atomicAdd(&ctrl[0], n_rec);
}

// K2d  (single-pass binning) per view: totals of the per-tile counters for gr_raster_status.  grid (views), 1024 threads
__global__ __launch_bounds__(1024) void k_bin_stats(BinArgs a) {
  __shared__ unsigned long long part[16];
  __shared__ uint32_t pmax[16];
  const int slot = blockIdx.x;
  uint32_t *ctrl = a.ctrl + slot * a.ctrl_stride;
  const uint32_t *cnt = ctrl + GR_CTRL_HDR;
  unsigned long long sum = 0;
  uint32_t mx = 0;
  for (int t = threadIdx.x; t < a.T; t += 1024) { const uint32_t c = cnt[t]; sum += c; mx = max(mx, c); }
  for (int o = 32; o > 0; o >>= 1) { sum += __shfl_xor(sum, o); mx = max(mx, (uint32_t)__shfl_xor((int)mx, o)); }
  if ((threadIdx.x & 63) == 0) { part[threadIdx.x >> 6] = sum; pmax[threadIdx.x >> 6] = mx; }
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long total = 0; uint32_t m = 0;
    for (int k = 0; k < 16; ++k) { total += part[k]; m = max(m, pmax[k]); }
    ctrl[1] = (uint32_t)total;
    const bool ovf = m > (uint32_t)a.cap_tile || ctrl[2] != 0;
    atomicAdd(&a.stats[0], (unsigned long long)ctrl[0]);
    atomicAdd(&a.stats[1], total);
    atomicMax(&a.stats[2], (unsigned long long)m);  // direct mode: the largest per-tile count
    if (ovf) { atomicMax(&a.stats[3], 1ull); atomicMin(&a.stats[4], (unsigned long long)a.group); }
    if (ctrl[2] & 2u) atomicMax(&a.stats[5], 1ull);  // a face the 40-byte entry form cannot hold
  }
}

// ------------------------------------------------------------------------------------------------------------------
// K2  exclusive scan of the per-tile counts (cntS + cntB) of one view.  grid (views), 1024 threads
// ------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_scan_tiles(BinArgs a) {
  __shared__ uint32_t wave_tot[16];
  __shared__ uint32_t carry_s;
  const int slot = blockIdx.x;
  uint32_t *ctrl = a.ctrl + slot * a.ctrl_stride;
  const uint32_t *cntS = ctrl + GR_CTRL_HDR;
  const uint32_t *cntB = cntS + a.Tcap;
  uint32_t *off = ctrl + GR_CTRL_HDR + 2 * a.Tcap;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  if (tid == 0) carry_s = 0;
  __syncthreads();
  for (int base = 0; base < a.T; base += 1024) {
    const int t = base + tid;
    const uint32_t c = (t < a.T) ? cntS[t] + cntB[t] : 0u;
    uint32_t incl = c;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const uint32_t o = __shfl_up(incl, d);
      if (lane >= d) incl += o;
    }
    if (lane == 63) wave_tot[wv] = incl;
    __syncthreads();
    uint32_t wbase = 0;
    for (int k = 0; k < wv; ++k) wbase += wave_tot[k];
    const uint32_t carry = carry_s;
    if (t < a.T) off[t] = carry + wbase + incl - c;
    __syncthreads();
    if (tid == 1023) carry_s = carry + wbase + incl;
    __syncthreads();
  }
  if (tid == 0) {
    const uint32_t total = carry_s;
    ctrl[1] = total;
    const bool ovf = (int64_t)total > a.ent_cap;
    ctrl[2] = ovf ? 1u : 0u;
    atomicAdd(&a.stats[0], (unsigned long long)ctrl[0]);
    atomicAdd(&a.stats[1], (unsigned long long)total);
    atomicMax(&a.stats[2], (unsigned long long)total);
    if (ovf) { atomicMax(&a.stats[3], 1ull); atomicMin(&a.stats[4], (unsigned long long)a.group); }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// K3  per (face, tile) entry: the face's edge functions re-based to the CENTRE of the tile and stored as a 48-byte
//     "compiled" entry at its place in the tile's list (build_entry; layout in DESIGN.md section 5):
//       E'_k(x_c, y_c) = C'_k + a_k x_c + b_k y_c   in units of one pixel, covered <=> all E'_k >= 0 (fill rule folded into C'_k)
//     One int32 form for every face: all 64-bit set-up arithmetic happens here, once per entry; the tile rasterizer only
//     streams entries (no index indirection, no 64-bit arithmetic).  12 words:
//       word  0..3   C'_first C'_middle C'_last | slopes (a_first, a_middle: 16 + 16 bits, or the start of 4 x 24 bits)
//       word  4..7   slopes (b_first, b_middle) | slopes (24-bit form only) | iz0 | A
//       word  8..11  B | Xw = X0rel (24 bit) + rows in tile << 24 | ~face | Yw = Y0rel (24 bit) + first row << 24 + flags
//     Single-pass binning calls it from K1 (compile_entry at the position the tile counter returned); the exact path from
//     k_fill_compile below (positions of <= 2x2-tile faces come from K1, larger faces take one cursor atomic per tile).
// ------------------------------------------------------------------------------------------------------------------
// Faces whose snapped bounding box is smaller than GR_FAST_EXT sub-pixels (93 px) take a short form of the set-up: the
// face overlaps the tile, so every pixel the tile rasterizer can probe (x in [-2, TW+2], y in [0, TH]) lies within
// reach = (64 + 3) * 256 + ext < 41152 sub-pixels of every vertex, |dx|, |dy| <= ext, hence
//   |E| <= (|dx| + |dy|) * reach + 1 < 48000 * 41152 < 2^31   and   |A|, |B| = 256 * |d| < 2^23:
// every product has 24-bit factors and every value fits int32 -- no 64-bit arithmetic, no per-tile range test.
// Larger faces take the general form below (identical coverage: both forms are exact).
#define GR_FAST_EXT 24000
#define GR_FLOOR_NOCORR_MAX 16000  // largest slope magnitude for which edge_floor<false> is exact (see there)
__device__ __forceinline__ int pack16(int lo, int hi) { return (lo & 0xFFFF) | (hi << 16); }
__device__ __forceinline__ bool build_entry(const int4 p0, const int4 p1, const int4 p2, int px0, int py0, int TW, int TH,
                                            int4 &e0, int4 &e1, int4 &e2, int &rows) {
  const int X0 = p0.x, Y0 = p0.y, X1 = p0.z, Y1 = p0.w, X2 = p1.x, Y2 = p1.y;
  const int Pxo = px0 * 256 + 128, Pyo = py0 * 256 + 128;  // centre of the tile's first pixel
  const int jlo = max((p2.z & 0xFFFF) - px0, 0), jhi = min((int)((uint32_t)p2.z >> 16) - px0, TW - 1);
  const int ilo = max((p2.w & 0xFFFF) - py0, 0), ihi = min((int)((uint32_t)p2.w >> 16) - py0, TH - 1);
  const int dx0 = X1 - X0, dy0 = Y1 - Y0, dx1 = X2 - X1, dy1 = Y2 - Y1, dx2 = X0 - X2, dy2 = Y0 - Y2;
  const int t0 = ((dy0 < 0) || (dy0 == 0 && dx0 > 0)) ? 0 : -1;  // R3 top-left rule as a bias
  const int t1 = ((dy1 < 0) || (dy1 == 0 && dx1 > 0)) ? 0 : -1;
  const int t2 = ((dy2 < 0) || (dy2 == 0 && dx2 > 0)) ? 0 : -1;
  const int ext = max(imax3(X0, X1, X2) - imin3(X0, X1, X2), imax3(Y0, Y1, Y2) - imin3(Y0, Y1, Y2));
  // row word, CENTRED like everything else the tile kernel reads: float(P_y - Y0) of centred row y_c = y - TH/2 is
  // float(256 y_c + Yw); the entry's first row as y_c (6 bits, signed).  |Pyo - Y0| + 8192 < 2^23 inside the guard band
  const int yw = ((Pyo - Y0 + (TH / 2) * 256) & 0xFFFFFF) | (((ilo - TH / 2) & 0x3F) << 24);
  int nr = (jhi >= jlo) ? max(ihi - ilo + 1, 0) : 0;         // rows of the entry in this tile (<= 64)
  // ONE form for every face, however large: the three edge functions in units of 256 around the tile CENTRE,
  //   E'_k(x_c, y_c) = C'_k + a_k x_c + b_k y_c,   x_c = x - TW/2, y_c = y - TH/2,   a_k = -dy_k, b_k = dx_k (|.| < 2^23),
  //   C'_k = floor(C_k / 256) with C_k the exact edge value (fill-rule bias included) at the centre pixel.
  // Exact because A_k = 256 a_k and B_k = 256 b_k are multiples of 256: E_k >= 0 <=> floor(E_k / 256) >= 0 <=> E'_k >= 0.
  // C'_k can be as large as 2^39 for a face that spans the guard band, but inside the tile (|x_c| <= TW/2 + 2 with the
  // solver's reach, |y_c| <= TH/2) the sum a x_c + b y_c stays within M_k = (TW/2 + 2)|a_k| + (TH/2)|b_k|: a C'_k beyond
  // +-M_k cannot change sign in the tile, so it is CLAMPED to +-(M_k + 1) -- same coverage, and every value the tile
  // kernel forms fits int32 (M_k < 2^29.1).  The edges are stored in an order the tile kernel relies on: FIRST an edge
  // with a > 0 (it bounds the span from the left), LAST one with a < 0 (from the right), the remaining one in between
  // -- a triangle of non-zero area has both kinds (the a_k sum to zero; so do the b_k: the last edge's slopes are not
  // stored).  The plane of 1/z refers to vertex 0 whatever the edge order.
  const int Pxc = Pxo + (TW / 2) * 256, Pyc = Pyo + (TH / 2) * 256;  // centre of the tile's centre pixel
  int c0, c1, c2;
  if (ext < GR_FAST_EXT) {  // every product has 24-bit factors and every value fits int32: no 64-bit arithmetic, no clamp
    c0 = (__mul24(dx0, Pyc - Y0) - __mul24(dy0, Pxc - X0) + t0) >> 8;
    c1 = (__mul24(dx1, Pyc - Y1) - __mul24(dy1, Pxc - X1) + t1) >> 8;
    c2 = (__mul24(dx2, Pyc - Y2) - __mul24(dy2, Pxc - X2) + t2) >> 8;
  } else {
    const long long C0 = ((long long)dx0 * (Pyc - Y0) - (long long)dy0 * (Pxc - X0) + (long long)t0) >> 8;
    const long long C1 = ((long long)dx1 * (Pyc - Y1) - (long long)dy1 * (Pxc - X1) + (long long)t1) >> 8;
    const long long C2 = ((long long)dx2 * (Pyc - Y2) - (long long)dy2 * (Pxc - X2) + (long long)t2) >> 8;
    const long long hx = TW / 2 + 2, hy = TH / 2;
    const long long M0 = hx * abs(dy0) + hy * abs(dx0) + 1, M1 = hx * abs(dy1) + hy * abs(dx1) + 1,
                    M2 = hx * abs(dy2) + hy * abs(dx2) + 1;
    c0 = (int)min(max(C0, -M0), M0);
    c1 = (int)min(max(C1, -M1), M1);
    c2 = (int)min(max(C2, -M2), M2);
  }
  const int a0 = -dy0, a1 = -dy1, a2 = -dy2;
  // The bounding box reaches this tile; the triangle itself may not (the far corner of a diagonal face).  An edge whose
  // value is negative even at the tile corner most in its favour, C' + (TW/2)|a| + (TH/2)|b| < 0, excludes every pixel of
  // the tile: the entry is DEAD (0 rows: the tile kernel never looks at it); k_bin_big asks before it takes a list slot.
  const bool touches = nr > 0 && c0 + (TW / 2) * abs(a0) + (TH / 2) * abs(dx0) >= 0 &&
                       c1 + (TW / 2) * abs(a1) + (TH / 2) * abs(dx1) >= 0 && c2 + (TW / 2) * abs(a2) + (TH / 2) * abs(dx2) >= 0;
  if (!touches) nr = 0;
  rows = nr;
  // float(P_x - X0) of the pixel with CENTRED column x_c = x - TW/2 is float(256 x_c + Xw)
  const int xw = ((Pxo - X0 + (TW / 2) * 256) & 0xFFFFFF) | (nr << 24);
  const int kf = a0 > 0 ? 0 : (a1 > 0 ? 1 : 2);   // first: a > 0
  const int kl = a0 < 0 ? 0 : (a1 < 0 ? 1 : 2);   // last: a < 0
  const int km = 3 - kf - kl;
  auto pick = [](int k, int v0, int v1, int v2) { return k == 0 ? v0 : (k == 1 ? v1 : v2); };
  const int cf = pick(kf, c0, c1, c2), cm = pick(km, c0, c1, c2), cl = pick(kl, c0, c1, c2);
  const int af = pick(kf, a0, a1, a2), am = pick(km, a0, a1, a2);
  const int bf = pick(kf, dx0, dx1, dx2), bm = pick(km, dx0, dx1, dx2);
  // slopes: four values (the last edge's are -(first + middle)).  Two packings: 16 bits each when every slope of the face
  // fits (faces below 128 pixels: nearly all of them), else 24 bits each, flagged in bit 31 of the Yw word
  const bool narrow = max(max(abs(a0), abs(a1)), max(abs(a2), max(abs(dx0), max(abs(dx1), abs(dx2))))) <= 32767;
  int w3, w4, w5;
  if (narrow) {
    w3 = pack16(af, am); w4 = pack16(bf, bm);
    w5 = ext < GR_FAST_EXT ? 0 : 1;  // never read for 16-bit slopes; non-zero tells store_entry that the short form does not fit
  } else {
    w3 = (af & 0xFFFFFF) | (am << 24);
    w4 = ((am >> 8) & 0xFFFF) | (bf << 16);
    w5 = ((bf >> 16) & 0xFF) | (bm << 8);
  }
  e0 = make_int4(cf, cm, cl, w3);
  e1 = make_int4(w4, w5, p1.z, p2.x);
  // bit 31: 24-bit slopes; bit 30: some slope magnitude beyond GR_FLOOR_NOCORR_MAX (the span solver must correct its floor)
  const bool corr = !narrow || max(abs(a0), max(abs(a1), abs(a2))) > GR_FLOOR_NOCORR_MAX;
  // ~face sits in an EVEN word: the tile kernel forms the 64-bit key (depth << 32 | ~face) in the register pair the entry
  // was read into, without a move
  e2 = make_int4(p2.y, xw, (int)~(uint32_t)p1.w, yw | (narrow ? 0 : (int)0x80000000) | (corr ? 0x40000000 : 0));
  return touches;
}

// The SHORT form of an entry, 40 bytes (single-pass binning, a.ent40): what a face whose snapped bounding box stays below
// GR_FAST_EXT sub-pixels (93 px: every face of a survey mesh) needs -- the three edge constants are below 2^23 in magnitude
// there (|E| < 24000 * 64640 before the shift by 8: build_entry), the offsets of vertex 0 from the tile's centre pixel below
// 2^15 (half a tile + the face's extent), the slopes fit 16 bits:
//   s0 = c_first      s1 = c_mid      s2 = c_last[0:24] | first row (6 bits, centred) << 24 | corr << 31
//   s3 = X0rel (16) | Y0rel << 16               s4, s5 = the slope words w3, w4        s6, s7 = iz0, A
//   s8 = ~face (an EVEN word: the key pair)     s9 = B
// (the tile kernel unpacks it with as many instructions as the 48-byte form; a denser packing of the constants cost it five more)
// 17 % fewer bytes written here and read by the tile kernel than the 48-byte form (the binning tax of DESIGN.md section 10).
// A face the short form cannot hold raises bit 1 of the view's overflow word: gr_raster_status reports GR_EOVERFLOW like
// for a tile that outgrew its segment, remembers that this (mesh, image) needs 48-byte entries, and the caller repeats.
__device__ __forceinline__ void store_entry(const BinArgs &a, uint32_t *__restrict__ ctrl, int4 *__restrict__ comp,
                                            uint8_t *__restrict__ nr8, int64_t idx, const int4 e0, const int4 e1, const int4 e2,
                                            int rows) {
  if (a.ent40) {
    // e1.y (the third slope word) is zero for 16-bit slopes; bit 0 of it is build_entry's "too large for the short form"
    if (e1.y != 0 || e2.w < 0) { atomicOr(&ctrl[2], 2u); return; }
    // a chunk of 64 entries (2560 bytes) holds the 64 x {s0 .. s7} first, then the 64 x {s8, s9}: the tile kernel copies the
    // chunk to LDS as it is and reads an entry with two 16-byte reads and one 8-byte read, all aligned
    char *chunk = reinterpret_cast<char *>(comp) + (idx >> 6) * 2560;
    const int t = (int)(idx & 63);
    int4 *d4 = reinterpret_cast<int4 *>(chunk) + t * 2;
    d4[0] = make_int4(e0.x, e0.y, (int)(((uint32_t)e0.z & 0xFFFFFFu) | ((uint32_t)e2.w & 0x3F000000u) | (((uint32_t)e2.w << 1) & 0x80000000u)),
                      (int)(((uint32_t)e2.y & 0xFFFFu) | ((uint32_t)e2.w << 16)));
    d4[1] = make_int4(e0.w, e1.x, e1.z, e1.w);
    reinterpret_cast<uint2 *>(chunk + 2048)[t] = make_uint2((uint32_t)e2.z, (uint32_t)e2.x);
  } else {
    int4 *dst = comp + idx * GR_ENT_Q;
    dst[0] = e0; dst[1] = e1; dst[2] = e2;
  }
  nr8[idx] = (uint8_t)rows;
}

__device__ __forceinline__ bool compile_entry(const BinArgs &a, uint32_t *__restrict__ ctrl, int4 *__restrict__ comp,
                                              uint8_t *__restrict__ nr8, int64_t idx, const int4 p0, const int4 p1,
                                              const int4 p2, int px0, int py0, int TW, int TH) {
  int4 e0, e1, e2;
  int rows;
  const bool touches = build_entry(p0, p1, p2, px0, py0, TW, TH, e0, e1, e2, rows);
  store_entry(a, ctrl, comp, nr8, idx, e0, e1, e2, rows);
  return touches;
}

__global__ __launch_bounds__(256) void k_fill_compile(BinArgs a) {
  const int slot = blockIdx.y;
  uint32_t *ctrl = a.ctrl + slot * a.ctrl_stride;
  const uint32_t n_rec = ctrl[0];
  const uint32_t *cntS = ctrl + GR_CTRL_HDR;
  const uint32_t *off = ctrl + GR_CTRL_HDR + 2 * a.Tcap;
  uint32_t *cur = ctrl + GR_CTRL_HDR + 3 * a.Tcap;
  const int4 *rec0 = a.rec + slot * a.rec_stride;
  int4 *comp = a.comp + slot * a.ent_cap * GR_ENT_Q;
  uint8_t *nr8 = a.nrow8 + slot * a.ent_cap;
  const int TW = 1 << a.twl, TH = 1 << a.thl;
  for (uint32_t r = blockIdx.x * 256 + threadIdx.x; r < n_rec; r += gridDim.x * 256) {
    const int4 p0 = rec0[r], p1 = rec0[a.F + r], p2 = rec0[2 * a.F + r];
    const int tx0 = (p2.z & 0xFFFF) >> a.twl, tx1 = (int)((uint32_t)p2.z >> 16) >> a.twl;
    const int ty0 = (p2.w & 0xFFFF) >> a.thl, ty1 = (int)((uint32_t)p2.w >> 16) >> a.thl;
    const bool small_fp = (tx1 - tx0 <= 1) && (ty1 - ty0 <= 1);
    int4 pos = {0, 0, 0, 0};
    if (small_fp) pos = rec0[3 * a.F + r];
#pragma unroll 1
    for (int ty = ty0; ty <= ty1; ++ty) {
#pragma unroll 1
      for (int tx = tx0; tx <= tx1; ++tx) {
        const int t = ty * a.TX + tx;
        const int k = ((ty - ty0) << 1) | (tx - tx0);
        const uint32_t pk = (uint32_t)(k == 0 ? pos.x : k == 1 ? pos.y : k == 2 ? pos.z : pos.w);
        const int64_t idx = small_fp ? (int64_t)off[t] + pk : (int64_t)off[t] + cntS[t] + atomicAdd(&cur[t], 1u);
        if (idx < a.ent_cap) compile_entry(a, ctrl, comp, nr8, idx, p0, p1, p2, tx << a.twl, ty << a.thl, TW, TH);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// K3c  R7: faces that straddle the near plane or the guard band (the view's clip list, filled by K1) are clipped in
//      camera space -- Sutherland-Hodgman against z >= near and |s| <= 16383 px, double precision, every operation
//      individually rounded, crossings always computed from the inside vertex (two faces sharing an edge get the same
//      new vertex) -- and the fan of the clipped polygon is binned like any other triangle, with the face's id.  Rare
//      (a camera inside the scene, faces larger than the guard band): one thread per face, plain atomics, local arrays.
//      The oracle's orc_clip_face is the same code in C.
// ------------------------------------------------------------------------------------------------------------------
struct P3 { double x, y, z; };

__device__ __forceinline__ double clip_plane(const double *pl, P3 p) {
  const double t1 = pl[0] * p.x, t2 = pl[1] * p.y, t3 = pl[2] * p.z;
  return ((t1 + t2) + t3) + pl[3];
}

__device__ __forceinline__ P3 clip_cross(P3 in, double din, P3 out, double dout) {
  const double t = din / (din - dout);
  const double ex = out.x - in.x, ey = out.y - in.y, ez = out.z - in.z;
  const double px = t * ex, py = t * ey, pz = t * ez;
  P3 r;
  r.x = in.x + px; r.y = in.y + py; r.z = in.z + pz;
  return r;
}

// one triangle of a clipped face: R2 / R4 set-up from three snapped vertices, then binning (no wave aggregation)
template <bool DIRECT>
__device__ void emit_triangle(const BinArgs &a, int slot, uint32_t *ctrl, Vtx v0, Vtx v1, Vtx v2, int face) {
  long long area2 = (long long)(v1.X - v0.X) * (long long)(v2.Y - v0.Y) - (long long)(v2.X - v0.X) * (long long)(v1.Y - v0.Y);
  if (area2 == 0) return;
  if (area2 < 0) { Vtx t = v1; v1 = v2; v2 = t; area2 = -area2; }
  const int Xmin = imin3(v0.X, v1.X, v2.X), Xmax = imax3(v0.X, v1.X, v2.X);
  const int Ymin = imin3(v0.Y, v1.Y, v2.Y), Ymax = imax3(v0.Y, v1.Y, v2.Y);
  int jmin = (Xmin - 128 + 255) >> 8, jmax = (Xmax - 128) >> 8;
  int imin = (Ymin - 128 + 255) >> 8, imax = (Ymax - 128) >> 8;
  jmin = max(jmin, 0); imin = max(imin, 0);
  jmax = min(jmax, a.w - 1); imax = min(imax, a.h - 1);
  if (jmin > jmax || imin > imax) return;
  const double d1 = (double)v1.iz - (double)v0.iz;
  const double d2 = (double)v2.iz - (double)v0.iz;
  const double a2 = (double)area2;
  double n1, n2;
  n1 = d1 * (double)(v2.Y - v0.Y); n2 = d2 * (double)(v1.Y - v0.Y);
  const float A = (float)((n1 - n2) / a2);
  n1 = d2 * (double)(v1.X - v0.X); n2 = d1 * (double)(v2.X - v0.X);
  const float B = (float)((n1 - n2) / a2);
  const int4 r0 = make_int4(v0.X, v0.Y, v1.X, v1.Y);
  const int4 r1 = make_int4(v2.X, v2.Y, __float_as_int(v0.iz), face);
  const int4 r2 = make_int4(__float_as_int(A), __float_as_int(B), jmin | (jmax << 16), imin | (imax << 16));
  const int tx0 = jmin >> a.twl, tx1 = jmax >> a.twl, ty0 = imin >> a.thl, ty1 = imax >> a.thl;
  uint32_t *cntS = ctrl + GR_CTRL_HDR;
  uint32_t *cntB = cntS + a.Tcap;
  if (DIRECT) {
    int4 *comp = a.comp + slot * a.ent_cap * GR_ENT_Q;
    uint8_t *nr8 = a.nrow8 + slot * a.ent_cap;
    for (int ty = ty0; ty <= ty1; ++ty)
      for (int tx = tx0; tx <= tx1; ++tx) {
        const int t = ty * a.TX + tx;
        const uint32_t pos = atomicAdd(&cntS[t], 1u);
        if (pos < (uint32_t)a.cap_tile) {
          const int64_t idx = (int64_t)t * a.cap_tile + pos;
          compile_entry(a, ctrl, comp, nr8, idx, r0, r1, r2, tx << a.twl, ty << a.thl, 1 << a.twl, 1 << a.thl);
        } else atomicOr(&ctrl[2], 1u);
      }
  } else {
    const uint32_t s = atomicAdd(&ctrl[0], 1u);
    if ((int64_t)s >= a.F) { atomicMax(&a.stats[3], 1ull); atomicMin(&a.stats[4], (unsigned long long)a.group); return; }  // more records than faces: the call is rejected
    const bool small_fp = (tx1 - tx0 <= 1) && (ty1 - ty0 <= 1);
    int4 r3 = {0, 0, 0, 0};
    if (small_fp) {
      r3.x = (int)atomicAdd(&cntS[ty0 * a.TX + tx0], 1u);
      if (tx1 > tx0) r3.y = (int)atomicAdd(&cntS[ty0 * a.TX + tx1], 1u);
      if (ty1 > ty0) r3.z = (int)atomicAdd(&cntS[ty1 * a.TX + tx0], 1u);
      if (tx1 > tx0 && ty1 > ty0) r3.w = (int)atomicAdd(&cntS[ty1 * a.TX + tx1], 1u);
    } else {
      for (int ty = ty0; ty <= ty1; ++ty)
        for (int tx = tx0; tx <= tx1; ++tx) atomicAdd(&cntB[ty * a.TX + tx], 1u);
    }
    int4 *rec = a.rec + slot * a.rec_stride;
    rec[s] = r0; rec[a.F + s] = r1; rec[2 * a.F + s] = r2; rec[3 * a.F + s] = r3;
  }
}

template <bool DIRECT>
__global__ __launch_bounds__(64) void k_clip_faces(const float *__restrict__ cams, BinArgs a) {
  // polygon buffers in LDS, one column per thread (dynamically indexed local arrays would put the kernel on scratch
  // memory, which costs every launch ~10 us even when the clip lists are empty)
  __shared__ double px[2][8][64], py[2][8][64], pz[2][8][64];
  __shared__ int sX[8][64], sY[8][64];
  __shared__ float sZ[8][64];
  const int slot = blockIdx.y, tid = threadIdx.x;
  const float *cam = cams + (int64_t)slot * GR_CAM_FLOATS;
  uint32_t *ctrl = a.ctrl + slot * a.ctrl_stride;
  const int64_t n_clip = min((int64_t)ctrl[4], a.F);
  if ((int64_t)blockIdx.x * 64 >= n_clip) return;  // the usual case: nothing to clip in this view
  const float fe = cam[12], cxp = cam[13], cyp = cam[14], nearp = cam[15];
  if (!(nearp > 0.0f) || !(fe > 0.0f) || !isfinite(fe) || !isfinite(cxp) || !isfinite(cyp)) return;
  constexpr double G = 16383.0;
  for (int64_t i = (int64_t)blockIdx.x * 64 + tid; i < n_clip; i += (int64_t)gridDim.x * 64) {
    const int64_t f = a.clip[(int64_t)slot * a.F + i];
    const float *sp = a.soup + 9 * f;
#pragma unroll
    for (int k = 0; k < 3; ++k) {  // the first half of R1
      const float *p = sp + 3 * k;
      const float dx = p[0] - cam[9], dy = p[1] - cam[10], dz = p[2] - cam[11];
      float m0, m1, m2;
      m0 = cam[0] * dx; m1 = cam[3] * dy; m2 = cam[6] * dz;
      px[0][k][tid] = (double)((m0 + m1) + m2);
      m0 = cam[1] * dx; m1 = cam[4] * dy; m2 = cam[7] * dz;
      py[0][k][tid] = (double)((m0 + m1) + m2);
      m0 = cam[2] * dx; m1 = cam[5] * dy; m2 = cam[8] * dz;
      pz[0][k][tid] = (double)((m0 + m1) + m2);
    }
    int n = 3, cur = 0;
    bool bad = false;
#pragma unroll
    for (int pl = 0; pl < 5; ++pl) {
      // plane pl:  a x + b y + c z + d >= 0   (near plane, then sx <= G, sx >= -G, sy <= G, sy >= -G)
      const double pa = pl == 1 ? -(double)fe : pl == 2 ? (double)fe : 0.0;
      const double pb = pl == 3 ? -(double)fe : pl == 4 ? (double)fe : 0.0;
      const double pc = pl == 0 ? 1.0 : pl == 1 ? G - (double)cxp : pl == 2 ? G + (double)cxp : pl == 3 ? G - (double)cyp
                                                                                                        : G + (double)cyp;
      const double pd = pl == 0 ? -(double)nearp : 0.0;
      const double plane[4] = {pa, pb, pc, pd};
      if (n == 0 || bad) break;
      int m = 0;
      for (int e = 0; e < n; ++e) {
        const int e1 = (e + 1) % n;
        const P3 S = {px[cur][e][tid], py[cur][e][tid], pz[cur][e][tid]};
        const P3 E = {px[cur][e1][tid], py[cur][e1][tid], pz[cur][e1][tid]};
        const double dS = clip_plane(plane, S), dE = clip_plane(plane, E);
        const bool inS = dS >= 0.0, inE = dE >= 0.0;
        P3 o0 = E, o1 = E;
        int cnt = 0;
        if (inS && inE) { cnt = 1; }
        else if (inS && !inE) { o0 = clip_cross(S, dS, E, dE); cnt = 1; }
        else if (!inS && inE) { o0 = clip_cross(E, dE, S, dS); cnt = 2; }
        if (cnt >= 1) { if (m < 8) { px[cur ^ 1][m][tid] = o0.x; py[cur ^ 1][m][tid] = o0.y; pz[cur ^ 1][m][tid] = o0.z; } ++m; }
        if (cnt == 2) { if (m < 8) { px[cur ^ 1][m][tid] = o1.x; py[cur ^ 1][m][tid] = o1.y; pz[cur ^ 1][m][tid] = o1.z; } ++m; }
      }
      if (m > 8) bad = true;
      n = m;
      cur ^= 1;
    }
    if (bad || n < 3) continue;
    for (int e = 0; e < n; ++e) {
      const float qx = (float)px[cur][e][tid], qy = (float)py[cur][e][tid], qz = (float)pz[cur][e][tid];
      if (!(qz > 0.0f)) { bad = true; break; }
      const float iz = 1.0f / qz;
      const float fx = fe * qx;
      const float fy = fe * qy;
      const float sx = cxp + fx * iz;
      const float sy = cyp + fy * iz;
      if (!(fabsf(sx) < 16384.0f) || !(fabsf(sy) < 16384.0f)) { bad = true; break; }
      sX[e][tid] = (int)floorf(sx * 256.0f + 0.5f);
      sY[e][tid] = (int)floorf(sy * 256.0f + 0.5f);
      sZ[e][tid] = iz;
    }
    if (bad) continue;
    const int face = a.orig[f];
    Vtx v0;
    v0.X = sX[0][tid]; v0.Y = sY[0][tid]; v0.iz = sZ[0][tid]; v0.valid = v0.front = v0.finite = true;
    for (int k = 1; k + 1 < n; ++k) {
      Vtx v1 = v0, v2 = v0;
      v1.X = sX[k][tid]; v1.Y = sY[k][tid]; v1.iz = sZ[k][tid];
      v2.X = sX[k + 1][tid]; v2.Y = sY[k + 1][tid]; v2.iz = sZ[k + 1][tid];
      emit_triangle<DIRECT>(a, slot, ctrl, v0, v1, v2, face);
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// K4  tile rasterizer (the dominant kernel).  grid (T, views), 256 threads = 4 waves, one 64 x TH tile per workgroup.
//     depth|id keys (u64: 1/z bits << 32 | ~face) live in LDS; visibility is resolved with ds_max_u64, so the result
//     does not depend on list order.
//       phase 1  the tile's list is taken in CHUNKS of 64 entries.  A chunk is copied once into a 3 KiB LDS buffer
//                shared by the workgroup (one 16-byte load + one ds_write_b128 per lane, 48 lanes per wave); in
//                single-pass mode the first chunk and its 64 row counts (the nrow8 stream) are requested before the
//                tile's count is known (the segment address is static: one memory round trip instead of two);
//       phase 2  every wave prefix-sums the same 64 row counts with DPP moves: the chunk's work is total_rows
//                (entry, row) items, taken 64 at a time; the 64-item batches are dealt to the waves round-robin (an
//                average C2 tile has 6.5 batches: 7 are issued, where a per-wave split of the ENTRIES issued 8); an
//                item finds its entry through the wave's LDS mailboxes (starts post, items read, a DPP prefix
//                maximum carries the latest start forward) and reads the entry's 12 words with three ds_read_b128
//                (12 LDS cycles per batch; the register-resident entries of round 1 cost twelve ds_bpermute = 48);
//       phase 3  ONE SCANLINE OF ONE TRIANGLE PER LANE: the exact covered span [xs, xe] comes from the three edge
//                inequalities (span_solve: probe-free float floor division, exact by construction -- edge_floor), then
//                the lane walks the span two pixels at a time and issues one ds_max_u64 per covered pixel.
//     What bounds it: VALU issue (74-80 % of the SIMD cycles), then the LDS pipe (62-66 %); 7 workgroups
//     fit a CU (21.25 KiB of LDS each).
//     Epilogues: ids -> 16-byte stores (4 pixels per lane); fused projection -> per-face winners (see fused_winners).
// ------------------------------------------------------------------------------------------------------------------
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

struct RasterOut {
  int32_t *ids;      // [slot][h][w] or null
  float *depth;      // [slot][h][w] or null
  uint32_t *winner;  // fused projection: [slot][F] keys = (last pixel of the face in the view) + 1, or null
  int64_t F;
  int compat;        // GR_FLAG_NEG1_IS_LAST_FACE
};

// LDS image of a tile: rows of TW keys padded by GR_LDS_PAD keys (stride 69 keys = 552 B).  The rows of one triangle
// walk their spans in step; with a row offset of 5 key-banks a pile-up on one bank needs a left edge that recedes
// 5 px per row (a pad of 1 piled up every 45-degree edge: 530 of 1820 LDS cycles per tile were bank conflicts), and a
// pixel's address advances by a plain +8 bytes along the scanline (no wrap arithmetic in the inner loop).
#ifndef GR_LDS_PAD
#define GR_LDS_PAD 5
#endif
template <int TWL, int PAD>
__device__ __forceinline__ int lds_idx(int row, int col) {
  return __mul24(row, (1 << TWL) + PAD) + col;
}

// floor(E / m) for an integer edge value E (|E| < 2^23 wherever the result matters) and an edge slope magnitude
// 0 <= m < 2^15, clamped to [-66, 65] (-67 / 66 with the correction): the scanline solver of the tile kernel.
//   g = (E + 0.5) * rcp(m) in fp32.  (E + 0.5) / m is never an integer and at least 0.5 / m away from one; the fp32 error
//   of g (v_rcp_f32: 1 ulp, one rounded multiply) is below 66 * 1.8e-7 = 1.2e-5 wherever |g| <= 66.  For m <= 16000
//   the gap is 3.1e-5: floor(g) IS floor(E / m) -- checked exhaustively on the CPU against integer division with the
//   reciprocal perturbed by up to 3.5 ulp (tests/test_span_floor.py) -- so no probe of the edge function is needed.
//   CORR (m up to 32767): one exact remainder puts a proposal that is off by one right.
//   m == 0 (an edge parallel to the scanline): g = +-inf, clamped to "no constraint" / "empty" by the sign of E.
template <bool CORR>
__device__ __forceinline__ int edge_floor(int E, int m, float mf) {
  float g = ((float)E + 0.5f) * __builtin_amdgcn_rcpf(mf);
  g = __builtin_amdgcn_fmed3f(g, -34.0f, 33.0f);  // centred columns -32 .. 31, plus the solver's reach
  int fl;
  asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(fl) : "v"(g));  // (int)floorf(g) in one instruction
  if (CORR) {
    const int rem = E - __mul24(fl, m);
    fl += (rem >= m ? 1 : 0) - (rem < 0 ? 1 : 0);
  }
  return fl;
}

// exact covered span [xs, xe] of one scanline in CENTRED tile coordinates (x_c = x - TW/2 in [-TW/2, TW/2 - 1],
// y_c = y - TH/2): the first edge (a > 0) bounds it from the left, x_c >= ceil(-E'/a) = -floor(E'/a); the last (a < 0)
// from the right, x_c <= floor(E'/|a|); the middle one does either (a == 0 works as either).  WIDE = false: every lane's
// slopes are packed in 16 bits and at most GR_FLOOR_NOCORR_MAX; WIDE = true: any packing, exact correction.
template <int TW, bool WIDE>
__device__ __forceinline__ void span_solve(int C0, int C1, int C2, int w3, int w4, int w5, bool wide24, int yc, int &xs, int &xe) {
  int a0 = (int)((uint32_t)w3 << 16) >> 16, a1 = w3 >> 16, b0 = (int)((uint32_t)w4 << 16) >> 16, b1 = w4 >> 16;
  if (WIDE) {
    const int A0 = (int)((uint32_t)w3 << 8) >> 8, A1 = (int)((((uint32_t)w3 >> 24) | ((uint32_t)w4 << 8)) << 8) >> 8;
    const int B0 = (int)((((uint32_t)w4 >> 16) | ((uint32_t)w5 << 16)) << 8) >> 8, B1 = w5 >> 8;
    a0 = wide24 ? A0 : a0; a1 = wide24 ? A1 : a1; b0 = wide24 ? B0 : b0; b1 = wide24 ? B1 : b1;
  }
  const int m2 = a0 + a1, b2 = -(b0 + b1);   // the last edge: a2 = -(a0 + a1) < 0, stored nowhere
  const int m1 = a1 < 0 ? -a1 : a1;
  const int f0 = edge_floor<WIDE>(C0 + __mul24(b0, yc), a0, (float)a0);
  const int f1 = edge_floor<WIDE>(C1 + __mul24(b1, yc), m1, (float)m1);
  const int f2 = edge_floor<WIDE>(C2 + __mul24(b2, yc), m2, (float)m2);
  xs = max(-(TW / 2), -f0);
  xe = min(TW / 2 - 1, f2);
  const int lo = max(xs, -f1), hi = min(xe, f1);
  xs = a1 > 0 ? lo : xs;
  xe = a1 > 0 ? xe : hi;
}

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef int v4i __attribute__((ext_vector_type(4)));

// Inclusive prefix sum over the 64 lanes with DPP moves only: the LDS pipe (ds_bpermute shuffles included) is the tile
// kernel's scarcest resource, VALU issue is not (one extra ds_bpermute per 64-item batch costs 0.34 us per C2 view, 48
// extra VALU instructions 0.9).  Sources outside a row / masked rows contribute the `old` operand, 0.
__device__ __forceinline__ int wave_incl_scan(int x) {
  x += __builtin_amdgcn_update_dpp(0, x, 0x111 /* row_shr:1 */, 0xf, 0xf, false);
  x += __builtin_amdgcn_update_dpp(0, x, 0x112 /* row_shr:2 */, 0xf, 0xf, false);
  x += __builtin_amdgcn_update_dpp(0, x, 0x114 /* row_shr:4 */, 0xf, 0xf, false);
  x += __builtin_amdgcn_update_dpp(0, x, 0x118 /* row_shr:8 */, 0xf, 0xf, false);
  x += __builtin_amdgcn_update_dpp(0, x, 0x142 /* row_bcast:15 */, 0xa, 0xf, false);  // rows 1, 3 += total of rows 0, 2
  x += __builtin_amdgcn_update_dpp(0, x, 0x143 /* row_bcast:31 */, 0xc, 0xf, false);  // rows 2, 3 += total of rows 0-1
  return x;
}

// Inclusive prefix maximum (unsigned), same DPP pattern.
__device__ __forceinline__ uint32_t wave_incl_max(uint32_t x) {
  x = max(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111 /* row_shr:1 */, 0xf, 0xf, false));
  x = max(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112 /* row_shr:2 */, 0xf, 0xf, false));
  x = max(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114 /* row_shr:4 */, 0xf, 0xf, false));
  x = max(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118 /* row_shr:8 */, 0xf, 0xf, false));
  // the two row-broadcast steps as single instructions (the compiler makes three of each): rows 1, 3 take the maximum
  // with lane 15 of the row before, then rows 2, 3 with lane 31; masked-out rows keep their value
  asm("s_nop 1\n\tv_max_u32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\ts_nop 1\n\t"
      "v_max_u32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf" : "+v"(x));
  return x;
}

// Phase 3 for ONE work item: scanline `q - et` of the entry whose 12 words (e0, e1, e2) the lane holds.
// The tile kernel is VALU-issue bound (SQ_ACTIVE_INST_VALU: 85 % of the SIMD cycles), so this function is written for
// instruction count: packed fp32 operands are broadcast by op_sel instead of being copied into register pairs, the
// 64-bit key is formed in the pair the entry word ~face was read into, an odd span is extended to the LEFT (only the
// first step has a spare slot, steered to the row's padding key), and the row addresses come from one multiply-add.
// The fields of an entry the scanline code works with, from either form (store_entry).
struct EntryView {
  int c0, c1, c2, w3, w4, w5;   // edge constants, slope words
  int X0rel, Y0rel, y_first;    // float(P - vertex 0) offsets of the centred pixel (0, 0); the entry's first row, centred
  bool wide24, corr;            // 24-bit slope packing; a slope beyond GR_FLOOR_NOCORR_MAX
  f32x2 izA;                    // {iz0, A}
  float B;
  uint32_t key;                 // ~face
};

__device__ __forceinline__ EntryView entry_view(const int4 e0, const int4 e1, const int4 e2) {
  EntryView v;
  v.c0 = e0.x; v.c1 = e0.y; v.c2 = e0.z; v.w3 = e0.w; v.w4 = e1.x; v.w5 = e1.y;
  const int xw = e2.y, yw = e2.w;
  v.X0rel = (xw << 8) >> 8;     // biased by TW/2 columns: float(P_x - X0) = float(256 x_c + X0rel)
  v.Y0rel = (yw << 8) >> 8;     // biased by TH/2 rows
  v.y_first = (yw << 2) >> 26;
  v.wide24 = yw < 0;
  v.corr = (uint32_t)yw >= 0x40000000u;  // compile_entry's flags: 24-bit slopes or a slope beyond 16000
  v.izA.x = __int_as_float(e1.z); v.izA.y = __int_as_float(e1.w);  // the two words as the entry holds them
  v.B = __int_as_float(e2.x);
  v.key = (uint32_t)e2.z;
  return v;
}

// the 40-byte form: five 8-byte words (store_entry)
__device__ __forceinline__ EntryView entry_view(const uint2 s01, const uint2 s23, const uint2 s45, const uint2 s67, const uint2 s89) {
  EntryView v;
  v.c0 = (int)s01.x; v.c1 = (int)s01.y;
  v.c2 = __builtin_amdgcn_sbfe(s23.x, 0, 24);
  v.w3 = (int)s45.x; v.w4 = (int)s45.y; v.w5 = 0;
  v.X0rel = __builtin_amdgcn_sbfe(s23.y, 0, 16);
  v.Y0rel = (int)s23.y >> 16;
  v.y_first = __builtin_amdgcn_sbfe(s23.x, 24, 6);
  v.wide24 = false;
  v.corr = (int)s23.x < 0;
  v.izA.x = __uint_as_float(s67.x); v.izA.y = __uint_as_float(s67.y);
  v.B = __uint_as_float(s89.y);
  v.key = s89.x;
  return v;
}

template <int TWL, int TH, int PAD>
__device__ __forceinline__ void raster_item(unsigned long long *keys, const EntryView &e, const int r, const bool live) {
  constexpr int TW = 1 << TWL;
  const int X0rel = e.X0rel, Y0rel = e.Y0rel;
  const int yc = e.y_first + r;  // centred row of the item: the entry's first row + the item's row within the entry
  // faces with a slope beyond GR_FLOOR_NOCORR_MAX (edges longer than 62 pixels) or 24-bit slopes take the span solver
  // with the exact correction; the choice is made per wave so that the usual case carries no extra instructions
  const bool wide24 = e.wide24;
  const bool wide = live && e.corr;
  int xs = 0, xe = -1;
  if (__ballot(wide) != 0ull) {
    if (live) span_solve<TW, true>(e.c0, e.c1, e.c2, e.w3, e.w4, e.w5, wide24, yc, xs, xe);
  } else {
    if (live) span_solve<TW, false>(e.c0, e.c1, e.c2, e.w3, e.w4, e.w5, false, yc, xs, xe);
  }
  // two pixels per step with packed fp32 math (v_pk_mul_f32 / v_pk_add_f32: same IEEE results as the scalar forms,
  // R4 op for op: z = iz0 + (A * float(P_x - X0) + B * float(P_y - Y0))).  float(P_x - X0) advances by exact float adds
  // (integers below 2^24).
  if (live && xs <= xe) {
    const float m1 = e.B * (float)(yc * 256 + Y0rel);
    const bool even = ((xe - xs) & 1) != 0;       // an even number of pixels xs .. xe
    const int x0 = even ? xs : xs - 1;            // x0 .. xe is always an even number; pixel xs - 1 is computed, not stored
    const float fx0 = (float)(x0 * 256 + X0rel);
    f32x2 fx = {fx0, fx0 + 256.0f};
    const f32x2 step = {512.0f, 512.0f};
    const f32x2 izA = e.izA;                      // {iz0, A}: the two words as the entry holds them
    f32x2 mp;
    mp.x = m1;                                    // the high half is never selected (op_sel_hi)
    // byte offset of the row's centred column 0: the key rows are (TW + PAD) * 8 bytes apart
    const int row = __mul24(yc, (TW + PAD) * 8) + ((TH / 2) * (TW + PAD) + TW / 2) * 8;
    int kp = row + x0 * 8;
    const int kend = row + xe * 8;
    const uint32_t key_a = e.key;
    uint32_t key_b = key_a;                       // a second copy: each pixel of a step forms its key in its own pair
    asm("v_mov_b32 %0, %1" : "=v"(key_b) : "v"(key_a));
    auto pixel_pair = [&](bool first_too) {
      f32x2 t, z;
      asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1]" : "=v"(t) : "v"(izA), "v"(fx));   // A * fx
      asm("v_pk_add_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(t) : "v"(mp), "v"(t));                   // + m1
      asm("v_pk_add_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(z) : "v"(izA), "v"(t));                  // iz0 +
      const int zb0 = max(__float_as_int(z.x), 1), zb1 = max(__float_as_int(z.y), 1);
      unsigned long long *const k = reinterpret_cast<unsigned long long *>(reinterpret_cast<char *>(keys) + kp);
      if (first_too) atomicMax(k, ((unsigned long long)(uint32_t)zb0 << 32) | key_a);
      atomicMax(k + 1, ((unsigned long long)(uint32_t)zb1 << 32) | key_b);
      kp += 16;
      fx += step;
    };
    pixel_pair(even);
    while (kp < kend) pixel_pair(true);
  }
}

// Phases 2-3 for one CHUNK of up to 64 entries staged in LDS (`ent`, 48 bytes each).  Every wave of the workgroup scans
// the same 64 row counts; batch b of the chunk belongs to wave (b + rot) % NW.  tab: the wave's 64 mailbox words in LDS,
// gen: the wave's batch counter (mailbox generation).  Returns the number of batches of the chunk.
// item -> entry: an entry that starts inside the batch posts gen | lane | slot into the mailbox of its start slot; the
// words of the current batch are larger than any stale one (gen grows), and among them the latest start is the largest,
// so an unsigned prefix maximum over the RAW words carries the right entry to every item lane.
template <int TWL, int TH, int NW, int PAD, bool SHORT>
__device__ __forceinline__ int raster_chunk_gather(unsigned long long *keys, const int tab_base, const int tab_self, uint32_t &gen,
                                                   const int4 *ent, const int nrows, const int lane,
                                                   const int first_b, const int dbg) {
  char *const lds = reinterpret_cast<char *>(keys);
  const int incl = wave_incl_scan(nrows);
  int total = __builtin_amdgcn_readlane(incl, 63);
  const int excl = incl - nrows;
  if (dbg & 1) total = 0;
  for (int k0 = first_b * 64; k0 < total; k0 += 64 * NW) {
    const int q = k0 + lane;
    gen += 1u << 12;
    const int slot = excl - k0;
    if (nrows > 0 && slot >= 0 && slot < 64)
      *reinterpret_cast<uint32_t *>(lds + tab_base + slot * 4) = gen | (uint32_t)(lane << 6) | (uint32_t)slot;
    const int carry_t = __popcll(__ballot(incl <= k0));  // the entry that holds item k0: it exists (k0 < total), <= 63
    const int carry_r = k0 - __builtin_amdgcn_readlane(excl, carry_t);  // row of item k0 within that entry
    const uint32_t m = wave_incl_max(*reinterpret_cast<const uint32_t *>(lds + tab_self));
    const bool started = m >= gen;                       // some entry starts at or before this lane's item in the batch
    const int t = started ? (int)((m >> 6) & 63u) : carry_t;  // always an entry of this chunk, also beyond the last item
    const int r = lane - (started ? (int)(m & 63u) : -carry_r);  // the item's row within its entry
    const bool live = q < total;
    if (SHORT) {
      const int4 ea = ent[t * 2], eb = ent[t * 2 + 1];
      const uint2 s89 = reinterpret_cast<const uint2 *>(ent)[256 + t];
      raster_item<TWL, TH, PAD>(keys, entry_view(make_uint2(ea.x, ea.y), make_uint2(ea.z, ea.w), make_uint2(eb.x, eb.y),
                                                 make_uint2(eb.z, eb.w), s89), r, live);
    } else {
      const int4 e0 = ent[t * 3], e1 = ent[t * 3 + 1], e2 = ent[t * 3 + 2];
      raster_item<TWL, TH, PAD>(keys, entry_view(e0, e1, e2), r, live);
    }
  }
  return (total + 63) >> 6;
}

// ids-only epilogue.  16-byte stores where the rows allow it: a lane owns 4 consecutive pixels of a row (16 lanes per
// 64-pixel row, 16 rows per pass); the four low dwords sit 8 bytes apart in LDS (two ds_read2_b32), id = ~low (0 for an
// empty pixel -> -1).  Images whose width is not a multiple of 4 take one pixel per lane.
template <int TWL, int TH, int NT, int PAD>
__device__ __forceinline__ void store_ids(const unsigned long long *keys, const BinArgs &a, int32_t *ids_plane, int te,
                                          int px0, int py0) {
  // (exchanging every key with the empty one here -- ds_wrxchg_rtn_b64, so that the workgroup's next tile needs no fill --
  // was measured: returning LDS atomics are slow, 17.0 vs 15.6 us per C2 view)
  const uint32_t *klo = reinterpret_cast<const uint32_t *>(keys);
  const int rows_here = min(TH, a.h - py0);
  const bool vec = ((a.w & 3) == 0) && ((reinterpret_cast<uintptr_t>(ids_plane) & 15) == 0);
  if (vec) {
    const int c4 = (te & 15) * 4, rr = te >> 4;
    const int gx4 = px0 + c4;
    if (gx4 >= a.w) return;
    int32_t *dst = ids_plane + (int64_t)(py0 + rr) * a.w + gx4;
    const int64_t dstep = (int64_t)(NT / 16) * a.w;
    for (int row = rr; row < rows_here; row += NT / 16, dst += dstep) {
      const uint32_t *kr = klo + 2 * lds_idx<TWL, PAD>(row, c4);
      *reinterpret_cast<int4 *>(dst) = make_int4((int)~kr[0], (int)~kr[2], (int)~kr[4], (int)~kr[6]);
    }
  } else {
    constexpr int TW = 1 << TWL;
    const int col = te & (TW - 1), gx = px0 + col;
    if (gx >= a.w) return;
    int32_t *dst = ids_plane + (int64_t)(py0 + (te >> TWL)) * a.w + gx;
    const int64_t dstep = (int64_t)(NT / TW) * a.w;
    for (int row = te >> TWL; row < rows_here; row += NT / TW, dst += dstep) *dst = (int32_t)~klo[2 * lds_idx<TWL, PAD>(row, col)];
  }
}

// Fused projection epilogue (aggregate_projected_images fast path): the last pixel, in row-major order, of every face
// the tile shows goes to winner[face] with a global atomicMax of pixel + 1 -- meshes.py:1987-2001, where numpy's fancy
// assignment lets the last pixel of a face win.  A pixel can only be that last pixel if none of right / below-left /
// below / below-right shows the same face (a face's consecutive scanlines touch at least diagonally unless it is a steep
// sliver; extra candidates are harmless): 1.7 candidates per visible face on C2.  The fused kernel has almost no memory
// traffic, so this epilogue is priced in INSTRUCTIONS: a lane owns 4 consecutive pixels of TWO consecutive rows (three
// row reads serve both), all LDS reads are issued up front, the neighbours across lanes come from DPP row shifts (a
// 16-lane DPP row is exactly one 64-pixel tile row: lanes outside keep the `old` operand), every comparison is made on
// the RAW low dword of the key (~face: negative for a face, 0 for plain background; "differs" sentinels 1 and 2 can never
// equal one), the candidate conditions are plain mask arithmetic, and nothing waits on global memory: the label of the
// winning pixel is looked up by the vote kernel.  Background needs no mapping: the tile was filled with the id that
// background aliases (F - 1 with GR_FLAG_NEG1_IS_LAST_FACE, else -1 = raw 0, which no candidate test accepts).
// Unknown neighbours count as "differs": 1 across a tile edge, 2 outside the image (EDGE tiles only).
template <int TWL, int TH, int NT, int PAD, bool EDGE>
__device__ __forceinline__ void fused_winners(const unsigned long long *keys, const BinArgs &a, uint32_t *__restrict__ win,
                                              int te, int px0, int py0, int dbg) {
  static_assert(TWL == 6 && NT == 256 && TH % 32 == 0, "16 lanes x 4 pixels per tile row, 16 row pairs per pass");
  const uint32_t *klo = reinterpret_cast<const uint32_t *>(keys);
  const int c4 = (te & 15) * 4, rp = te >> 4;  // row pair 0 .. 15 of a pass
#pragma unroll
  for (int pass = 0; pass < TH / 32; ++pass) {
    const int r0 = 2 * rp + 32 * pass;          // rows r0, r0 + 1; the row below them is r0 + 2
    int c[3][6];                                // c[k][j + 1]: raw key of row r0 + k, column c4 + j, j = -1 .. 4
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      if (k < 2 || r0 + 2 < TH) {
        const uint32_t *kr = klo + 2 * lds_idx<TWL, PAD>(r0 + k, c4);
#pragma unroll
        for (int j = 0; j < 4; ++j) c[k][j + 1] = (int)kr[2 * j];
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) c[k][j + 1] = 1;  // the tile below: unknown
      }
    }
    const int gy = py0 + r0;
    if (EDGE) {
      if (gy >= a.h) continue;
      if (gy + 1 >= a.h) {
#pragma unroll
        for (int j = 0; j < 4; ++j) c[1][j + 1] = 2;
      }
      if (gy + 2 >= a.h) {
#pragma unroll
        for (int j = 0; j < 4; ++j) c[2][j + 1] = 2;
      }
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      c[k][5] = __builtin_amdgcn_update_dpp(1, c[k][1], 0x101 /* row_shl:1: lane + 1 */, 0xf, 0xf, false);
      if (k > 0) c[k][0] = __builtin_amdgcn_update_dpp(1, c[k][4], 0x111 /* row_shr:1: lane - 1 */, 0xf, 0xf, false);
    }
    const uint32_t p1 = (uint32_t)(gy * a.w + px0 + c4 + 1);  // linear pixel index + 1 of the lane's first pixel (h, w <= 16384)
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      if (EDGE && gy + k >= a.h) continue;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int f = c[k][j + 1];
        bool cand = (f < 0) & (c[k + 1][j + 1] != f) & (c[k + 1][j] != f);
        if (EDGE) {
          const int gx = px0 + c4 + j;
          cand = cand & (gx < a.w) & ((gx + 1 >= a.w) | ((c[k][j + 2] != f) & (c[k + 1][j + 2] != f)));
        } else {
          cand = cand & (c[k][j + 2] != f) & (c[k + 1][j + 2] != f);
        }
        if (cand && !(dbg & 8)) atomicMax(win + ~f, p1 + (uint32_t)(k * a.w + j));
      }
    }
  }
}

// the tile's entry list: count and first slot (single-pass binning: the tile's fixed segment; exact binning: the scan's offset)
__device__ __forceinline__ void tile_list(const BinArgs &a, const uint32_t *__restrict__ ctrl, int tile, uint32_t &cnt, int64_t &beg) {
  if (a.cap_tile > 0) {
    cnt = min(ctrl[GR_CTRL_HDR + tile], (uint32_t)a.cap_tile);
    beg = (int64_t)tile * a.cap_tile;
  } else {
    cnt = ctrl[GR_CTRL_HDR + tile] + ctrl[GR_CTRL_HDR + a.Tcap + tile];
    beg = ctrl[GR_CTRL_HDR + 2 * a.Tcap + tile];
    if (beg >= a.ent_cap) cnt = 0;
    else if (beg + cnt > a.ent_cap) cnt = (uint32_t)(a.ent_cap - beg);
  }
}

// Wave priority: a wave raises its priority for the scanline phase (the VALU-bound part) and drops it for the phases that
// wait on memory and barriers (tile fill, chunk loads, epilogue), so that the SIMD's issue slots go to the waves that can use
// them.  Builds alternated on one box (profiles/r03_ab/prio.log): plain 15.26 -> 15.00 us per C2 view, fused 16.93 -> 16.45;
// the reverse order loses 1-2 %, equal priorities are neutral.
#define GR_PRIO_MEM() __builtin_amdgcn_s_setprio(0)
#define GR_PRIO_ITEMS() __builtin_amdgcn_s_setprio(3)

// 16-byte piece q (0 .. 159) of a chunk that holds n (1 .. 64) entries in the short form: the front of the 32-byte parts or the
// front of the 8-byte parts (store_entry) -- is it needed?
__device__ __forceinline__ bool short_piece_needed(uint32_t q, uint32_t n) {
  return (q < 2 * n) | ((q >= 128) & (q < 128 + ((n + 1) >> 1)));  // no short-circuit: one predicate, one branch around the load
}

// One tile: keys in LDS -> chunks of entries -> scanline items -> epilogue.  nr_first / ex: the tile's first chunk (row
// counts and this lane's 16 bytes of the 3 KiB (2.5 KiB) of entries), requested by the caller -- and waited for by the caller
// (a chain), or here behind the fill of the key tile (WAIT: one tile per workgroup -- the request's latency overlaps the fill).
template <int TWL, int THL, int NT, bool FUSE, int PAD, bool SHORT, bool WAIT>
__device__ __forceinline__ void raster_one_tile(const BinArgs &a, const RasterOut &out, unsigned long long *keys, const int slot,
                                                const int tile, uint32_t cnt, const int64_t beg, uint32_t nr_first, v4i ex) {
  constexpr int TW = 1 << TWL, TH = 1 << THL;
  constexpr int NKEYS = (TW + PAD) * TH;
  constexpr int NW = NT / 64;
  constexpr int NMAIL = NW * 32;  // u64 units: 64 mailbox words per wave
  int4 *ent_lds = reinterpret_cast<int4 *>(keys + NKEYS + NMAIL);
  v4i *ent_st = reinterpret_cast<v4i *>(ent_lds);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  constexpr int ROWS_PER_PASS = NT / TW;
  const int64_t P = (int64_t)a.h * a.w;
  const int64_t plane = (int64_t)slot * P;
  const int tx = tile % a.TX, ty = tile / a.TX;
  const int px0 = tx << TWL, py0 = ty << THL;
  constexpr int EL = SHORT ? 40 : 48;  // 16-byte pieces of a 64-entry chunk per wave (4 waves): 40 or 48 bytes per entry
  // the tile's list: the slot's entry memory is laid out for 48-byte entries; the short form packs chunks of 64 40-byte
  // entries at the front of the tile's segment (tile * cap_tile is a multiple of 64 whenever the short form is chosen)
  const int4 *comp = SHORT ? reinterpret_cast<const int4 *>(reinterpret_cast<const char *>(a.comp + slot * a.ent_cap * GR_ENT_Q) + beg * 40)
                           : a.comp + (slot * a.ent_cap + beg) * GR_ENT_Q;
  const uint8_t *nr8 = a.nrow8 + slot * a.ent_cap + beg;

  if (!FUSE && cnt == 0) {  // empty tile (a view that overhangs the mesh): background, without the LDS round trip
    const int col = tid & (TW - 1), gx = px0 + col;
    if (gx < a.w && !(a.dbg & 2)) {
      for (int row = tid >> TWL; row < TH && py0 + row < a.h; row += ROWS_PER_PASS) {
        const int64_t p = plane + (int64_t)(py0 + row) * a.w + gx;
        if (out.ids) out.ids[p] = -1;
        if (out.depth) out.depth[p] = INFINITY;
      }
    }
    return;
  }
  const int tab_base = NKEYS * 8 + wv * 256;  // byte offset of the wave's 64 mailbox words, behind the keys
  const int tab_self = tab_base + lane * 4;
  {  // fill the tile (16-byte LDS stores): depth 0 | the id background stands for; mailboxes zero
    const int bg = (FUSE && out.compat) ? (int)out.F - 1 : -1;
    const unsigned long long fill = (unsigned long long)(uint32_t)~bg;
    ulonglong2 *k2 = reinterpret_cast<ulonglong2 *>(keys);
#pragma unroll
    for (int i = 0; i < (NKEYS / 2 + NT - 1) / NT; ++i)
      if (i * NT + tid < NKEYS / 2) k2[i * NT + tid] = make_ulonglong2(fill, fill);
    for (int i = tid; i < NMAIL / 2; i += NT) k2[NKEYS / 2 + i] = make_ulonglong2(0ull, 0ull);
  }
  uint32_t gen = 0;
  int rot = wv;  // this wave's first batch of the current chunk
  {  // first chunk: in registers already, complete (k_raster_tile waits for every request of the chain before its first
     // tile: a wait on the memory counter here would wait for the previous tile's stores)
    if (WAIT) asm volatile("" : "+v"(ex), "+v"(nr_first));
    if (lane < EL) ent_st[wv * EL + lane] = ex;
    __syncthreads();  // keys filled, chunk visible
    GR_PRIO_ITEMS();
    const int nrows = (uint32_t)lane < cnt ? (int)nr_first : 0;
    const int nb = raster_chunk_gather<TWL, TH, NW, PAD, SHORT>(keys, tab_base, tab_self, gen, ent_lds, nrows, lane, rot, a.dbg);
    rot = (rot - nb) & (NW - 1);
  }
#pragma unroll 1
  for (uint32_t c0 = 64; c0 < cnt; c0 += 64) {
    GR_PRIO_MEM();
    __syncthreads();  // every wave is done with the previous chunk before it is overwritten
    if (lane < EL) {
      const uint32_t qc = wv * EL + lane;  // piece of the chunk
      const uint32_t q = (SHORT ? (c0 >> 1) * 5 : c0 * GR_ENT_Q) + qc;
      if (SHORT ? short_piece_needed(qc, min(cnt - c0, 64u)) : q < cnt * GR_ENT_Q) ex = reinterpret_cast<const v4i *>(comp)[q];
      ent_st[wv * EL + lane] = ex;
    }
    __syncthreads();
    GR_PRIO_ITEMS();
    const uint32_t e = c0 + (uint32_t)lane;
    const int nrows = e < cnt ? (int)nr8[e] : 0;
    const int nb = raster_chunk_gather<TWL, TH, NW, PAD, SHORT>(keys, tab_base, tab_self, gen, ent_lds, nrows, lane, rot, a.dbg);
    rot = (rot - nb) & (NW - 1);
  }

  int te = tid;
  asm volatile("" : "+v"(te));  // the epilogue's addresses are derived here, not hoisted above the scanline phase
  GR_PRIO_MEM();
  __syncthreads();              // keys complete
  if (a.dbg & 2) return;
  if (FUSE) {
    uint32_t *win = out.winner + slot * out.F;
    const bool edge = px0 + TW > a.w || py0 + TH + 1 > a.h;
    if (edge) fused_winners<TWL, TH, NT, PAD, true>(keys, a, win, te, px0, py0, a.dbg);
    else fused_winners<TWL, TH, NT, PAD, false>(keys, a, win, te, px0, py0, a.dbg);
    if (out.ids) {  // the id image as well (rare): background is where no fragment landed (depth bits 0)
      const int col = te & (TW - 1), gx = px0 + col;
      if (gx < a.w)
        for (int row = te >> TWL; row < TH && py0 + row < a.h; row += ROWS_PER_PASS) {
          const unsigned long long key = keys[lds_idx<TWL, PAD>(row, col)];
          out.ids[plane + (int64_t)(py0 + row) * a.w + gx] = (key >> 32) ? (int32_t)~(uint32_t)key : -1;
        }
    }
  } else if (out.ids && !out.depth) {
    store_ids<TWL, TH, NT, PAD>(keys, a, out.ids + plane, te, px0, py0);
  } else {
    const int col = te & (TW - 1), gx = px0 + col;
    if (gx < a.w)
      for (int row = te >> TWL; row < TH && py0 + row < a.h; row += ROWS_PER_PASS) {
        const unsigned long long key = keys[lds_idx<TWL, PAD>(row, col)];
        const int64_t p = plane + (int64_t)(py0 + row) * a.w + gx;
        if (out.ids) out.ids[p] = (int32_t)~(uint32_t)key;  // low dword = ~face, 0 when empty: ~0 = -1
        if (out.depth) out.depth[p] = key ? 1.0f / __int_as_float((int)(key >> 32)) : INFINITY;
      }
  }
}

// K3  the tile kernel.  KT = 4: a workgroup takes four consecutive tiles one after the other.  The four counts are read
//     first (scalar loads), then the first chunks of all four tiles are requested EXACTLY, together, and waited for together
//     before the first tile starts: one wait for memory per chain instead of four, no stale slots fetched.  (Waiting for
//     tile k's chunk only when tile k starts would wait for tile k - 1's id stores: loads and stores share one in-order
//     counter.)  KT = 1 -- heavy scenes, small launches --: the first chunk is requested before the count is known (the
//     segment address is static; slots beyond the count hold stale data that nobody reads).
// The ids-only kernel asks the compiler for 7 waves per SIMD -- what its LDS allows anyway: the schedule the compiler picks
// under that hint is 3-4 % faster (15.8 -> 15.2 us per C2 view, builds alternated on one box with tools/ab_builds.sh); the
// fused kernel is not (left at the default).  Work items of two consecutive rows (look-up, unpack and the reciprocals paid
// once per two rows: -16 % VALU instructions) were measured as well: 74 VGPRs and half as many batches per tile for four
// waves -- 15.6 vs 16.0 without the hint, 16.3 vs 15.3 with it, fused 18.3 vs 17.2 -- dropped.
template <int TWL, int THL, int NT, bool FUSE, int KT, int PAD, bool SHORT>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(FUSE ? 1 : (THL == 5 ? 7 : 4), 8))) void k_raster_tile(BinArgs a, RasterOut out) {
  constexpr int TW = 1 << TWL, TH = 1 << THL;
  constexpr int NKEYS = (TW + PAD) * TH;
  constexpr int NW = NT / 64;
  constexpr int NMAIL = NW * 32;
  // the kernel's only LDS: keys (17.25 KiB for 64x32) + mailboxes (1 KiB) + one chunk of entries (3 KiB) -> 7 workgroups/CU
  // (20 KiB -- 4 padding keys per row with the mailboxes inside the padding -- gives 8, and loses more to LDS bank
  // conflicts than it gains: plain 16.6 vs 16.3 us per C2 view, fused 19.8 vs 17.9)
  __shared__ __attribute__((aligned(16))) unsigned long long keys[NKEYS + NMAIL + 64 * (SHORT ? 5 : 6)];
  static_assert(NT == 256, "the entry copy deals 48 int4 to each of 4 waves");
  static_assert(NKEYS % 2 == 0 && TH % 32 == 0, "key pairs; two 16-row passes per fused group");
  static_assert(KT == 1 || KT == 4, "one tile per workgroup, or a chain of four");
  const int slot = blockIdx.y;
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  // one tile per workgroup: the first chunk is requested before the count is known (one round trip less).  A chain waits
  // for the exact requests of its tiles 1 - 3 anyway before it starts: requesting its first tile's chunk early saves
  // nothing there (14.9 us per C2 view either way) and fetches 1.6 MB of stale slots per view -- not done
  const bool spec = KT == 1 && a.cap_tile >= 64 && !(a.var & 8);
  const int tile0 = KT * (int)blockIdx.x;
  const int n_tiles = min(KT, a.T - tile0);
  constexpr int EL = SHORT ? 40 : 48;
  const uint32_t q = wv * EL + lane;  // this thread's 16-byte piece of a 3 KiB (2.5 KiB) chunk (lanes 0 .. 47 (39) of every wave)
  // 16-byte pieces of the view's entry memory from entry `first` on, and the number of pieces `n` entries take
  auto pieces = [&](int64_t first) {
    const v4i *base = reinterpret_cast<const v4i *>(a.comp + (int64_t)slot * a.ent_cap * GR_ENT_Q);
    return SHORT ? reinterpret_cast<const v4i *>(reinterpret_cast<const char *>(base) + first * 40) : base + first * GR_ENT_Q;
  };
  auto needed = [](uint32_t q, uint32_t n) { return SHORT ? short_piece_needed(q, min(n, 64u)) : q < n * GR_ENT_Q; };
  uint32_t nr0 = 0, nr1 = 0, nr2 = 0, nr3 = 0;
  v4i ex0, ex1, ex2, ex3;  // whole 16-byte register tuples (the wait macro of raster_one_tile names them as such: with the
                           // components of an int4 struct named one by one the compiler split the tuples after the load -- and
                           // waited for each load right behind its request)
  if (spec) {
    const int64_t seg = (int64_t)tile0 * a.cap_tile;
    nr0 = a.nrow8[slot * a.ent_cap + seg + lane];
    if (lane < EL) ex0 = pieces(seg)[q];
  }
  const uint32_t *ctrl = a.ctrl + slot * a.ctrl_stride;
  uint32_t cnt0, cnt1 = 0, cnt2 = 0, cnt3 = 0;
  int64_t beg0, beg1 = 0, beg2 = 0, beg3 = 0;
  if (KT == 4 && a.cap_tile > 0) {
    // single-pass binning: the chain's four counters sit side by side, 16-byte aligned -- ONE scalar load instead of four
    // dependent ones, each behind its own wait (words behind the last tile's belong to the next counter array: valid memory)
    const uint4 c4 = *reinterpret_cast<const uint4 *>(ctrl + GR_CTRL_HDR + tile0);
    const uint32_t cap = (uint32_t)a.cap_tile;
    cnt0 = min(c4.x, cap);
    cnt1 = n_tiles > 1 ? min(c4.y, cap) : 0u;
    cnt2 = n_tiles > 2 ? min(c4.z, cap) : 0u;
    cnt3 = n_tiles > 3 ? min(c4.w, cap) : 0u;
    beg0 = (int64_t)tile0 * a.cap_tile; beg1 = beg0 + a.cap_tile; beg2 = beg1 + a.cap_tile; beg3 = beg2 + a.cap_tile;
  } else {
    tile_list(a, ctrl, tile0, cnt0, beg0);
    if (KT > 1) {
      if (n_tiles > 1) tile_list(a, ctrl, tile0 + 1, cnt1, beg1);
      if (n_tiles > 2) tile_list(a, ctrl, tile0 + 2, cnt2, beg2);
      if (n_tiles > 3) tile_list(a, ctrl, tile0 + 3, cnt3, beg3);
    }
  }
  if (a.dbg & 4) cnt0 = cnt1 = cnt2 = cnt3 = 0;
  const int64_t sbase = slot * a.ent_cap;
  if (!spec) {  // exact binning (or segments under 64 slots): the first chunk can only be requested now
    if ((uint32_t)lane < cnt0) nr0 = a.nrow8[sbase + beg0 + lane];
    if ((lane < EL) & needed(q, cnt0)) ex0 = pieces(beg0)[q];
  }
  if (KT > 1) {
    if ((uint32_t)lane < cnt1) nr1 = a.nrow8[sbase + beg1 + lane];
    if ((lane < EL) & needed(q, cnt1)) ex1 = pieces(beg1)[q];
    if ((uint32_t)lane < cnt2) nr2 = a.nrow8[sbase + beg2 + lane];
    if ((lane < EL) & needed(q, cnt2)) ex2 = pieces(beg2)[q];
    if ((uint32_t)lane < cnt3) nr3 = a.nrow8[sbase + beg3 + lane];
    if ((lane < EL) & needed(q, cnt3)) ex3 = pieces(beg3)[q];
  }
  // ONE wait for everything requested above, named as whole register tuples and BEFORE the first tile: behind this statement
  // the values are the statement's outputs, not loads in flight, so the compiler's bookkeeping of the (single, in-order)
  // memory counter has nothing left to wait for in the loop over tiles 1 .. 3 -- where a wait means waiting for the
  // previous tile's id stores (tests/test_isa_waits.py)
  if (KT > 1) asm volatile("" : "+v"(ex0), "+v"(ex1), "+v"(ex2), "+v"(ex3), "+v"(nr0), "+v"(nr1), "+v"(nr2), "+v"(nr3));
  raster_one_tile<TWL, THL, NT, FUSE, PAD, SHORT, KT == 1>(a, out, keys, slot, tile0, cnt0, beg0, nr0, ex0);
  if (KT > 1) {
#pragma unroll 1
    for (int k = 1; k < n_tiles; ++k) {  // ONE copy of the tile code for tiles 1 .. 3: the chunks rotate through ex1
      __syncthreads();                   // every wave has read the previous tile's keys
      raster_one_tile<TWL, THL, NT, FUSE, PAD, SHORT, false>(a, out, keys, slot, tile0 + k, cnt1, beg1, nr1, ex1);
      cnt1 = cnt2; cnt2 = cnt3; beg1 = beg2; beg2 = beg3;
      nr1 = nr2; nr2 = nr3; ex1 = ex2; ex2 = ex3;
    }
  }
}

// K1b  (variant bit 64 only: the default expands big faces inside K1) single-pass binning of the view's big list: a wave takes
//      64 big faces, one per lane (records recomputed from the soup: same code as K1, same bits) and expands their
//      (face, tile) pairs with bin_big_pairs.
__global__ __launch_bounds__(256) void k_bin_big(const float *__restrict__ cams, BinArgs a) {
  const int slot = blockIdx.y;
  const float *cam = cams + (int64_t)slot * GR_CAM_FLOATS;
  uint32_t *ctrl = a.ctrl + slot * a.ctrl_stride;
  const int64_t n_big = min((int64_t)ctrl[5], a.F);
  const int lane = threadIdx.x & 63;
  const int64_t wave0 = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), wstep = (int64_t)gridDim.x * 4;
  if (wave0 * 64 >= n_big) return;  // the usual case for terrain: nothing to do
  for (int64_t i0 = wave0 * 64; i0 < n_big; i0 += wstep * 64) {
    int4 r0 = {0, 0, 0, 0}, r1 = {0, 0, 0, 0}, r2 = {0, 0, 0, 0};
    int tx0 = 0, tx1 = -1, ty0 = 0, ty1 = -1;
    bool clip_me, keep = false;
    if (i0 + lane < n_big) keep = face_setup(a, cam, a.clip[(int64_t)slot * a.F + (a.F - 1 - (i0 + lane))], r0, r1, r2, tx0, tx1, ty0, ty1, clip_me);
    bin_big_pairs(a, ctrl, slot, lane, keep, r0, r1, r2, tx0, tx1, ty0, ty1);
  }
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
__global__ __launch_bounds__(256) void k_vote_labels(uint32_t *__restrict__ winner, const uint8_t *__restrict__ labels,
                                                     int n_views, int64_t F, int64_t P, int C,
                                                     uint32_t *__restrict__ votes, uint32_t *__restrict__ counts,
                                                     const unsigned long long *__restrict__ stats, int group,
                                                     const uint32_t *__restrict__ touched, int tw, int last_face_aliases_bg) {
  const int64_t f = (int64_t)blockIdx.x * 256 + threadIdx.x;
  // which views of the group can hold a winner for this workgroup's 256 faces (chunk = blockIdx.x): the bit the cull pass
  // set for the chunk, or the view's "all" word; without the bitmap (ids given by the caller) every view can.  Lane v of
  // every wave looks at view v: the ballot is the same in all four waves.
  unsigned long long dirty = ~0ull;
  if (touched) {
    const int v = threadIdx.x & 63;
    bool d = false;
    if (v < n_views) {
      const uint32_t *tv = touched + (int64_t)v * tw;
      d = (((tv[blockIdx.x >> 5] >> (blockIdx.x & 31u)) | tv[tw - 1]) & 1u) != 0u;
    }
    dirty = __ballot(d);
    if (last_face_aliases_bg && (int64_t)blockIdx.x == ((F - 1) >> 8)) dirty = ~0ull;  // background pixels vote for face F - 1
  }
  if (f >= F) return;
  // a launch group whose binning overflowed (and every group after it) must not vote: its winners are incomplete.  The
  // caller learns how many views were folded in (gr_raster_status: views_done) and repeats the call for the rest.
  const bool skip = stats != nullptr && stats[4] <= (unsigned long long)group;
  uint32_t c = 0;
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
      if ((int)lab[k] < C) votes[f * C + lab[k]] += 1u;
      ++c;
    }
  }
  if (c) counts[f] += c;
}

__global__ __launch_bounds__(256) void k_vote_values(uint32_t *__restrict__ winner, const double *__restrict__ img,
                                                     int n_views, int64_t F, int64_t P, int C,
                                                     double *__restrict__ sums, uint32_t *__restrict__ counts) {
  const int64_t f = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (f >= F) return;
  uint32_t c = 0;
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
        if (!isnan(x)) sums[f * C + ch] += x;  // nansum: NaN counts as 0 (meshes.py:2060-2062)
      }
      if (any_finite) ++c;
    }
  }
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

__global__ __launch_bounds__(256) void k_validate_faces(const int32_t *__restrict__ faces, int64_t n, int64_t V,
                                                        int *__restrict__ bad) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int v = faces[i];
  if (v < 0 || v >= V) atomicOr(bad, 1);
}

}  // namespace

// ------------------------------------------------------------------------------------------------------------------
// host side: context, scratch, C ABI
// ------------------------------------------------------------------------------------------------------------------
struct gr_ctx {
  int device = 0;
  const float *verts = nullptr;
  const int32_t *faces = nullptr;
  int64_t V = 0, F = 0;
  // bin scratch
  uint32_t *ctrl = nullptr;
  int4 *rec = nullptr;
  int4 *comp = nullptr;
  uint8_t *nrow8 = nullptr;
  int64_t nrow_have = 0;
  uint32_t *work = nullptr;
  int64_t work_stride = 0;
  uint32_t *clip = nullptr;   // [slot][F] clip lists (R7)
  int64_t clip_have = 0;
  float4 *blk = nullptr;
  uint32_t *blk_chunks = nullptr;  // [blk_cap][GR_CHUNK_LIST + 1]
  uint32_t *touched = nullptr;     // fused aggregation: [2][slots][tw] chunk bitmaps of the launch groups in flight
  int64_t touched_have = 0;
  uint32_t *cur_touched = nullptr; // the bitmap the next bin_batch fills (null: none)
  int cur_tw = 0;
  int64_t blk_cap = 0;
  float *soup = nullptr;
  int32_t *orig = nullptr;   // soup position -> caller's face id (Morton order)
  int64_t soup_cap = 0;
  unsigned long long *stats = nullptr;
  int *flag = nullptr;
  int64_t ctrl_stride = 0, rec_stride = 0, ent_cap = 0, ent_cap_request = 0;
  int64_t ctrl_have = 0, comp_have = 0, work_have = 0, rec_have = 0;  // allocated element counts
  int Tcap = 0, slots = 0;
  int64_t rec_F = 0;
  // tuning knobs (gr_set_option)
  int opt_thl = 5;      // log2 tile height (5 or 6); width is 64.  64x32 tiles: 16 KiB of LDS, 8 workgroups per CU
  int opt_batch = GR_MAX_BATCH;
  int opt_dbg = 0;
  int opt_var = 0;
  int opt_lds_pad = 0;   // extra dynamic LDS bytes per tile workgroup (occupancy experiments, GR_OPT_DEBUG_LDS)
  int opt_direct_cap = 512;  // single-pass binning: entry slots per tile (0 = always use the exact two-pass path)
  struct Learned { int64_t F; int T, cap; bool full; };
  Learned learned[8] = {};             // slots per tile learned from overflows -- and whether the image has faces the 40-byte entry
                                       // form cannot hold --, per (mesh size, tile count); [n_learned % 8] is replaced next
  int n_learned = 0;
  bool share_learned = true;           // consult / feed the process-wide table (off once GR_OPT_DIRECT_CAP was set by hand)
  int last_T = 0, last_B = 0;          // tile count and launch-group size of the last raster call
  int last_n_views = 0;
  bool direct_ok = true;     // cleared when a tile overflowed its slots: later calls take the exact path
  bool last_direct = false;
  // winner scratch
  void *winner = nullptr;
  size_t winner_bytes = 0;
  // fused aggregation: the vote kernel of launch group g runs on a side stream beside the binning of group g + 1
  hipStream_t side = nullptr;
  hipEvent_t ev_raster[2] = {nullptr, nullptr}, ev_vote[2] = {nullptr, nullptr};
  void *sort_tmp = nullptr;
  size_t sort_bytes = 0;
  hipStream_t last_stream = nullptr;
  // profiling
  bool profiling = false;
  struct Span { hipEvent_t a, b; int stage; };
  std::vector<Span> spans;
  std::vector<hipEvent_t> pool;
  int prof_views = 0, prof_raster_launches = 0;
  char err[512] = {0};
};

namespace {

// Slots per tile of the single-pass binning for an image of T tiles (0 = exact two-pass binning).  A call that
// overflowed the configured slots teaches the context the size that image needs (gr_raster_status), as long as the
// entry memory of a launch group stays within GR_DIRECT_BUDGET; other image sizes keep the configured value.
#define GR_DIRECT_BUDGET (24ll << 30)
// What one context has learned is kept process-wide as well, keyed by (face count, tile count): a second context for the
// same mesh and image size -- another camera set, another thread -- starts with segments that fit, without an overflowed
// first call.
std::mutex g_learned_mu;
gr_ctx::Learned g_learned[32];
int g_n_learned = 0;

// what is known about images of T tiles of the current mesh: slots per tile (0: nothing learned) and the entry form
void lookup_learned(const gr_ctx *c, int T, int &cap, bool &full) {
  cap = 0; full = false;
  for (int i = 0; i < std::min(c->n_learned, 8); ++i)
    if (c->learned[i].F == c->F && c->learned[i].T == T) { cap = c->learned[i].cap; full = c->learned[i].full; return; }
  if (c->share_learned) {
    std::lock_guard<std::mutex> lk(g_learned_mu);
    for (int i = 0; i < std::min(g_n_learned, 32); ++i)
      if (g_learned[i].F == c->F && g_learned[i].T == T) { cap = g_learned[i].cap; full = g_learned[i].full; return; }
  }
}

int direct_cap(const gr_ctx *c, int T) {
  if (c->opt_direct_cap <= 0 || !c->direct_ok) return 0;
  int cap; bool full;
  lookup_learned(c, T, cap, full);
  return std::max(cap, c->opt_direct_cap);
}

// The 40-byte entry form (store_entry) is the default of the single-pass binning; images with faces it cannot hold (93 px
// and more) are remembered like the slots per tile.  Variant bit 128: always 48 bytes.  The short form is laid out in chunks
// of 64 entries (store_entry): a tile's segment must be a whole number of chunks -- slots per tile set by hand to anything
// else: 48 bytes.
bool entry_short(const gr_ctx *c, int T) {
  const int dcap = direct_cap(c, T);
  if (dcap <= 0 || (dcap & 63) || (c->opt_var & 128)) return false;
  int cap; bool full;
  lookup_learned(c, T, cap, full);
  return !full;
}

// cap > 0: the slots per tile the image needs; full: it needs 48-byte entries (both are kept once learned)
void learn(gr_ctx *c, int T, int cap, bool full) {
  int old_cap; bool old_full;
  lookup_learned(c, T, old_cap, old_full);
  const gr_ctx::Learned v = {c->F, T, std::max(cap, old_cap), full || old_full};
  int i = 0;
  for (; i < std::min(c->n_learned, 8); ++i)
    if (c->learned[i].F == c->F && c->learned[i].T == T) break;
  if (i == std::min(c->n_learned, 8)) { i = c->n_learned % 8; c->n_learned += 1; }
  c->learned[i] = v;
  if (!c->share_learned) return;
  std::lock_guard<std::mutex> lk(g_learned_mu);
  int j = 0;
  for (; j < std::min(g_n_learned, 32); ++j)
    if (g_learned[j].F == c->F && g_learned[j].T == T) break;
  if (j == std::min(g_n_learned, 32)) { j = g_n_learned % 32; g_n_learned += 1; }
  g_learned[j] = v;
}

int fail(gr_ctx *c, int code, const char *fmt, ...) {
  if (c) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(c->err, sizeof(c->err), fmt, ap);
    va_end(ap);
  }
  return code;
}

#define GR_HIP(ctx, call)                                                                          \
  do {                                                                                             \
    hipError_t e_ = (call);                                                                        \
    if (e_ != hipSuccess) return fail(ctx, GR_EHIP, "%s: %s", #call, hipGetErrorString(e_));       \
  } while (0)

enum { ST_SETUP = 0, ST_SCAN, ST_FILL, ST_RASTER, ST_PROJECT, ST_VOTE, ST_GATHER, ST_N };

hipEvent_t take_event(gr_ctx *c) {
  hipEvent_t e;
  if (!c->pool.empty()) { e = c->pool.back(); c->pool.pop_back(); return e; }
  if (hipEventCreate(&e) != hipSuccess) return nullptr;
  return e;
}

struct Timed {  // RAII span around a kernel group when profiling is on
  gr_ctx *c; hipStream_t s; int stage; hipEvent_t a = nullptr, b = nullptr;
  Timed(gr_ctx *c_, hipStream_t s_, int st) : c(c_), s(s_), stage(st) {
    if (c->profiling) { a = take_event(c); b = take_event(c); if (a) (void)hipEventRecord(a, s); }
  }
  ~Timed() {
    if (c->profiling && a && b) { (void)hipEventRecord(b, s); c->spans.push_back({a, b, stage}); }
  }
};

inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Grow-only scratch: every buffer keeps its own capacity and is re-allocated only when it is too small (a new mesh or
// image size does not touch buffers that are already large enough).
template <typename T>
int grow(gr_ctx *c, T *&ptr, int64_t &have, int64_t want, const char *what) {
  if (ptr && have >= want) return GR_OK;
  (void)hipDeviceSynchronize();
  if (ptr) (void)hipFree(ptr);
  ptr = nullptr; have = 0;
  if (hipMalloc(&ptr, sizeof(T) * (size_t)want) != hipSuccess)
    return fail(c, GR_ENOMEM, "%s scratch allocation failed (%lld bytes)", what, (long long)(sizeof(T) * (size_t)want));
  have = want;
  return GR_OK;
}

int ensure_bins(gr_ctx *c, int n_slots, int T) {
  const int64_t F = c->F > 0 ? c->F : 1;
  const int dcap = direct_cap(c, T);
  const bool direct = dcap > 0;
  int64_t cap = c->ent_cap_request > 0 ? c->ent_cap_request : (F / 2 + 65536);
  if (direct) cap = std::max<int64_t>(cap, (int64_t)T * dcap);
  // The layout (strides) is that of THIS call; a buffer is re-allocated only when the call needs more elements than the
  // buffer has (a huge image with a small launch group and a small image with a full one share the same memory).
  const int64_t ctrl_stride = ((GR_CTRL_HDR + 4 * (int64_t)T) + 63) / 64 * 64;
  const int64_t work_stride = ceil_div(F, GR_BLOCK) + 4;
  int rc = grow(c, c->ctrl, c->ctrl_have, ctrl_stride * n_slots, "bin control");
  if (!rc) rc = grow(c, c->comp, c->comp_have, GR_ENT_Q * cap * n_slots, "entry list");
  if (!rc) rc = grow(c, c->nrow8, c->nrow_have, cap * n_slots + 64, "entry row counts");
  if (!rc) rc = grow(c, c->work, c->work_have, work_stride * n_slots, "work list");
  if (!rc) rc = grow(c, c->clip, c->clip_have, F * n_slots, "clip list");
  if (!rc && !direct) rc = grow(c, c->rec, c->rec_have, 4 * F * n_slots, "record planes");  // exact path only
  if (rc) return rc;
  c->slots = n_slots; c->Tcap = T; c->ent_cap = cap; c->ctrl_stride = ctrl_stride; c->work_stride = work_stride;
  c->rec_F = F; c->rec_stride = 4 * F;
  return GR_OK;
}

int ensure_winner(gr_ctx *c, size_t bytes) {
  if (c->winner && c->winner_bytes >= bytes) return GR_OK;
  (void)hipDeviceSynchronize();
  if (c->winner) (void)hipFree(c->winner);
  c->winner = nullptr; c->winner_bytes = 0;
  if (hipMalloc(&c->winner, bytes) != hipSuccess) return fail(c, GR_ENOMEM, "winner scratch allocation failed");
  if (hipMemset(c->winner, 0, bytes) != hipSuccess) return fail(c, GR_EHIP, "winner memset failed");
  c->winner_bytes = bytes;
  return GR_OK;
}

BinArgs make_args(gr_ctx *c, int h, int w, int slot0) {
  BinArgs a;
  a.ctrl_stride = c->ctrl_stride; a.rec_stride = c->rec_stride; a.ent_cap = c->ent_cap; a.F = c->F;
  a.work_stride = c->work_stride;
  a.ctrl = c->ctrl + slot0 * a.ctrl_stride; a.rec = c->rec + slot0 * a.rec_stride;
  a.comp = c->comp + slot0 * a.ent_cap * GR_ENT_Q; a.work = c->work + slot0 * a.work_stride;
  a.nrow8 = c->nrow8 + slot0 * a.ent_cap;
  a.stats = c->stats; a.blk = c->blk; a.soup = c->soup; a.orig = c->orig;
  a.blk_chunks = c->blk_chunks; a.touched = c->cur_touched; a.tw = c->cur_tw;
  a.clip = c->clip + slot0 * c->F;
  a.twl = GR_TILE_LOG2; a.thl = c->opt_thl;
  a.TX = (w + (1 << a.twl) - 1) >> a.twl; a.TY = (h + (1 << a.thl) - 1) >> a.thl; a.T = a.TX * a.TY; a.Tcap = c->Tcap;
  a.h = h; a.w = w; a.dbg = c->opt_dbg; a.var = c->opt_var;
  a.cap_tile = direct_cap(c, a.T);
  a.ent40 = entry_short(c, a.T) ? 1 : 0;
  a.group = 0;
  return a;
}

// stage 1 of a launch group: cull, set up and bin `nb` views (camera records `cams`) into scratch slots slot0..
int bin_batch(gr_ctx *c, const float *cams, int nb, int h, int w, int slot0, int group, hipStream_t s) {
  BinArgs a = make_args(c, h, w, slot0);
  a.group = group;
  GR_HIP(c, hipMemsetAsync(a.ctrl, 0, sizeof(uint32_t) * c->ctrl_stride * nb, s));
  {
    Timed t(c, s, ST_SETUP);
    const int nblk = (int)ceil_div(c->F, GR_BLOCK);
    hipLaunchKernelGGL(k_cull_blocks, dim3((unsigned)ceil_div(nblk, 256), nb), dim3(256), a.touched ? sizeof(uint32_t) * a.tw : 0, s,
                       cams, a, nblk);
    // k_setup_cull: a wave per surviving 64-face block would mostly pay for starting waves (a survey view keeps a tenth of
    // the blocks: C2 7.5 -> 6.1 us per view with an eighth of the workgroups): about nblk / 32 waves per view take a few
    // blocks each -- but never fewer than 16 k waves per launch, so that a call with a few views still fills the machine.
    // (Requesting the next block's soup one iteration ahead was measured on top of this: 100 VGPRs, no gain.)
    const int gmax = std::min((nblk + 3) / 4, 1024);
    const unsigned gsetup = (unsigned)std::max(1, std::min(gmax, std::max(nblk / 128, 4096 / std::max(nb, 1))));
    if (a.cap_tile > 0) {
      hipLaunchKernelGGL(k_setup_cull<true>, dim3(gsetup, nb), dim3(256), 0, s, cams, a);
      if (a.var & 64) hipLaunchKernelGGL(k_bin_big, dim3(256, nb), dim3(256), 0, s, cams, a);
      hipLaunchKernelGGL(k_clip_faces<true>, dim3(8, nb), dim3(64), 0, s, cams, a);
    } else {
      hipLaunchKernelGGL(k_setup_cull<false>, dim3(gsetup, nb), dim3(256), 0, s, cams, a);
      hipLaunchKernelGGL(k_clip_faces<false>, dim3(8, nb), dim3(64), 0, s, cams, a);
    }
  }
  c->last_direct = a.cap_tile > 0;
  if (a.cap_tile > 0) {
    Timed t(c, s, ST_SCAN);
    hipLaunchKernelGGL(k_bin_stats, dim3(nb), dim3(1024), 0, s, a);
  } else {
    {
      Timed t(c, s, ST_SCAN);
      hipLaunchKernelGGL(k_scan_tiles, dim3(nb), dim3(1024), 0, s, a);
    }
    {
      Timed t(c, s, ST_FILL);
      const unsigned g = (unsigned)std::min<int64_t>(ceil_div(c->F, 256), 1024);
      hipLaunchKernelGGL(k_fill_compile, dim3(g, nb), dim3(256), 0, s, a);
    }
  }
  GR_HIP(c, hipGetLastError());
  return GR_OK;
}

// stage 2: rasterize the binned views of scratch slots slot0.. into out (already offset to the group's first view)
int tile_batch(gr_ctx *c, int nb, int h, int w, int slot0, RasterOut out, hipStream_t s) {
  BinArgs a = make_args(c, h, w, slot0);
  {
    Timed t(c, s, ST_RASTER);
    // Four consecutive tiles per workgroup -- unless the image needed more than the default 512 slots per tile (a scene
    // with heavy tiles: chains of them make a few workgroups very long; hostile workload 57.9 vs 30.4 us per view at
    // 1000x750) or the launch has too few tiles to keep every CU busy with chains.
    const bool chain = (a.var & 1) == 0 && ((a.var & 16) != 0 || (a.cap_tile > 0 && a.cap_tile <= 512 && (int64_t)a.T * nb >= 16384));
    const dim3 grid(chain ? (unsigned)((a.T + 3) >> 2) : (unsigned)a.T, nb), block(256);
    const size_t pad = (size_t)c->opt_lds_pad;
#define GR_LAUNCH_TILE(THL_, FUSE_)                                                                                   \
  do {                                                                                                                \
    if (a.ent40) {                                                                                                    \
      if (chain) hipLaunchKernelGGL((k_raster_tile<6, THL_, 256, FUSE_, 4, GR_LDS_PAD, true>), grid, block, pad, s, a, out);  \
      else hipLaunchKernelGGL((k_raster_tile<6, THL_, 256, FUSE_, 1, GR_LDS_PAD, true>), grid, block, pad, s, a, out);        \
    } else if (chain) hipLaunchKernelGGL((k_raster_tile<6, THL_, 256, FUSE_, 4, GR_LDS_PAD, false>), grid, block, pad, s, a, out);  \
    else hipLaunchKernelGGL((k_raster_tile<6, THL_, 256, FUSE_, 1, GR_LDS_PAD, false>), grid, block, pad, s, a, out);        \
  } while (0)
    if (out.winner) {
      if (a.thl == 6) GR_LAUNCH_TILE(6, true);
      else GR_LAUNCH_TILE(5, true);
    } else if (a.thl == 6) GR_LAUNCH_TILE(6, false);
    else GR_LAUNCH_TILE(5, false);
#undef GR_LAUNCH_TILE
    c->prof_raster_launches += 1;
  }
  c->prof_views += nb;
  GR_HIP(c, hipGetLastError());
  return GR_OK;
}

int check_common(gr_ctx *c, int n_views, int h, int w) {
  if (!c) return GR_EINVAL;
  if (n_views < 0 || h <= 0 || w <= 0 || h > GR_MAX_DIM || w > GR_MAX_DIM)
    return fail(c, GR_EINVAL, "bad shape n_views=%d h=%d w=%d (limit %d)", n_views, h, w, GR_MAX_DIM);
  return GR_OK;
}

// unfused label projection for id images already in memory: winner pass + vote pass per launch group
int project_labels(gr_ctx *c, const int32_t *ids, const uint8_t *labels, int n_views, int h, int w, int C, uint32_t *votes,
                   uint32_t *counts, int flags, hipStream_t s) {
  const int64_t P = (int64_t)h * w, F = c->F;
  const int B = n_views < GR_MAX_BATCH ? n_views : GR_MAX_BATCH;
  int rc = ensure_winner(c, sizeof(uint32_t) * (size_t)F * B);
  if (rc) return rc;
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
                         votes, counts, (const unsigned long long *)nullptr, 0, (const uint32_t *)nullptr, 0, 0);
    }
  }
  GR_HIP(c, hipGetLastError());
  return GR_OK;
}

// pix2face for n_views cameras, optionally fused with the label projection (labels != nullptr): per launch group the
// tile kernel's epilogue feeds the per-face winners straight from LDS and k_vote_labels folds them into votes/counts.
int raster_views(gr_ctx *c, const float *cams, int n_views, int h, int w, int32_t *ids, float *depth,
                 const uint8_t *labels, int C, uint32_t *votes, uint32_t *counts, int flags, hipStream_t s) {
  int rc = check_common(c, n_views, h, w);
  if (rc) return rc;
  if (!c->verts) return fail(c, GR_ENOMESH, "gr_mesh_upload has not been called");
  if (!cams) return fail(c, GR_EINVAL, "null cams");
  if (n_views == 0) return GR_OK;
  GR_HIP(c, hipSetDevice(c->device));
  int B = n_views < c->opt_batch ? n_views : c->opt_batch;
  const int thl = c->opt_thl;
  const int T = ((w + GR_TILE - 1) >> GR_TILE_LOG2) * ((h + (1 << thl) - 1) >> thl);
  {  // very large images: fewer views per launch group, so that the fixed tile segments stay within the scratch budget
    const int64_t per_slot = (int64_t)T * direct_cap(c, T) * (16 * GR_ENT_Q);
    if (per_slot > 0) B = (int)std::max<int64_t>(1, std::min<int64_t>(B, GR_DIRECT_BUDGET / per_slot));
  }
  rc = ensure_bins(c, B, T);
  if (rc) return rc;
  c->last_T = T; c->last_B = B;
  const int64_t P = (int64_t)h * w, F = c->F;
  // Fused aggregation over several launch groups: the vote kernel of group g (a light, latency-bound pass over F winners)
  // runs on a side stream beside the binning of group g + 1 (also light); the tile kernels in between fill the machine on
  // their own.  Two winner buffers alternate; votes are still added group by group, in order (one side stream).
  const bool overlap = labels && n_views > B && !(c->opt_var & 4);
  // chunk bitmaps for the vote kernel (k_block_chunks / k_cull_blocks): one per launch group in flight; meshes beyond
  // 33 M faces do without (the bitmap of a view would not fit the cull kernel's LDS)
  const int tw = (int)(ceil_div(ceil_div(F, 256), 32) + 1);
  const bool use_touched = labels && tw <= 4096 && !(c->opt_var & 32);
  if (labels) {
    rc = ensure_winner(c, sizeof(uint32_t) * (size_t)F * B * (overlap ? 2 : 1));
    if (rc) return rc;
    if (use_touched) {
      rc = grow(c, c->touched, c->touched_have, (int64_t)2 * B * tw, "chunk bitmaps");
      if (rc) return rc;
    }
    if (overlap && !c->side) {
      GR_HIP(c, hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking));
      for (int i = 0; i < 2; ++i) {
        GR_HIP(c, hipEventCreateWithFlags(&c->ev_raster[i], hipEventDisableTiming));
        GR_HIP(c, hipEventCreateWithFlags(&c->ev_vote[i], hipEventDisableTiming));
      }
    }
  }
  c->last_stream = s;
  GR_HIP(c, hipMemsetAsync(c->stats, 0, sizeof(unsigned long long) * 4, s));
  GR_HIP(c, hipMemsetAsync(c->stats + 4, 0xFF, sizeof(unsigned long long), s));  // first overflowed group: none
  GR_HIP(c, hipMemsetAsync(c->stats + 5, 0, sizeof(unsigned long long), s));     // short-form miss: none
  c->last_n_views = n_views;
  int g = 0;
  for (int v0 = 0; v0 < n_views; v0 += B, ++g) {
    const int nb = (n_views - v0) < B ? (n_views - v0) : B;
    const int buf = overlap ? (g & 1) : 0;
    uint32_t *tch = use_touched ? c->touched + (int64_t)buf * B * tw : nullptr;
    if (tch) {
      // (the votes of group g - 2, which read this bitmap, are waited for below, before the tile kernel -- but the bitmap
      // is rewritten here already: wait now)
      if (overlap && g >= 2) GR_HIP(c, hipStreamWaitEvent(s, c->ev_vote[buf], 0));
      GR_HIP(c, hipMemsetAsync(tch, 0, sizeof(uint32_t) * (size_t)nb * tw, s));
    }
    c->cur_touched = tch; c->cur_tw = tw;
    rc = bin_batch(c, cams + (int64_t)v0 * GR_CAM_FLOATS, nb, h, w, 0, v0 / B, s);
    c->cur_touched = nullptr;
    if (rc) return rc;
    uint32_t *win = labels ? (uint32_t *)c->winner + (int64_t)buf * F * B : nullptr;
    RasterOut out;
    out.ids = ids ? ids + v0 * P : nullptr;
    out.depth = depth ? depth + v0 * P : nullptr;
    out.winner = win;
    out.F = F;
    out.compat = (flags & GR_FLAG_NEG1_IS_LAST_FACE) ? 1 : 0;
    if (overlap && g >= 2) GR_HIP(c, hipStreamWaitEvent(s, c->ev_vote[buf], 0));  // the votes of group g - 2 have read this buffer
    rc = tile_batch(c, nb, h, w, 0, out, s);
    if (rc) return rc;
    if (labels) {
      hipStream_t vs = s;
      if (overlap) {
        GR_HIP(c, hipEventRecord(c->ev_raster[buf], s));
        GR_HIP(c, hipStreamWaitEvent(c->side, c->ev_raster[buf], 0));
        vs = c->side;
      }
      {
        Timed t(c, vs, ST_VOTE);
        hipLaunchKernelGGL(k_vote_labels, dim3((unsigned)ceil_div(F, 256)), dim3(256), 0, vs, win, labels + v0 * P, nb, F, P, C,
                           votes, counts, (const unsigned long long *)c->stats, v0 / B, (const uint32_t *)tch, tw,
                           (flags & GR_FLAG_NEG1_IS_LAST_FACE) ? 1 : 0);
      }
      GR_HIP(c, hipGetLastError());
      if (overlap) GR_HIP(c, hipEventRecord(c->ev_vote[buf], c->side));
    }
  }
  if (overlap) {  // the caller's stream continues after the last votes
    GR_HIP(c, hipStreamWaitEvent(s, c->ev_vote[(g - 1) & 1], 0));
    if (g >= 2) GR_HIP(c, hipStreamWaitEvent(s, c->ev_vote[g & 1], 0));
  }
  return GR_OK;
}

}  // namespace

extern "C" {

int gr_version(void) { return GR_VERSION; }

int gr_ctx_create(int device, gr_ctx **out) {
  if (!out) return GR_EINVAL;
  *out = nullptr;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device < 0 || device >= n) return GR_ENODEVICE;
  if (hipSetDevice(device) != hipSuccess) return GR_ENODEVICE;
  gr_ctx *c = new (std::nothrow) gr_ctx();
  if (!c) return GR_ENOMEM;
  c->device = device;
  if (hipMalloc(&c->stats, sizeof(unsigned long long) * 8) != hipSuccess ||
      hipMalloc(&c->flag, sizeof(int) * 8) != hipSuccess) {  // flag word + upload scratch (vertex bounds)
    delete c;
    return GR_ENOMEM;
  }
  (void)hipMemset(c->stats, 0, sizeof(unsigned long long) * 8);
  *out = c;
  return GR_OK;
}

int gr_ctx_destroy(gr_ctx *c) {
  if (!c) return GR_OK;
  (void)hipSetDevice(c->device);
  (void)hipDeviceSynchronize();
  for (auto &sp : c->spans) { (void)hipEventDestroy(sp.a); (void)hipEventDestroy(sp.b); }
  for (auto e : c->pool) (void)hipEventDestroy(e);
  if (c->ctrl) (void)hipFree(c->ctrl);
  if (c->rec) (void)hipFree(c->rec);
  if (c->comp) (void)hipFree(c->comp);
  if (c->nrow8) (void)hipFree(c->nrow8);
  if (c->work) (void)hipFree(c->work);
  if (c->clip) (void)hipFree(c->clip);
  if (c->winner) (void)hipFree(c->winner);
  for (int i = 0; i < 2; ++i) {
    if (c->ev_raster[i]) (void)hipEventDestroy(c->ev_raster[i]);
    if (c->ev_vote[i]) (void)hipEventDestroy(c->ev_vote[i]);
  }
  if (c->side) (void)hipStreamDestroy(c->side);
  if (c->sort_tmp) (void)hipFree(c->sort_tmp);
  if (c->blk) (void)hipFree(c->blk);
  if (c->blk_chunks) (void)hipFree(c->blk_chunks);
  if (c->touched) (void)hipFree(c->touched);
  if (c->soup) (void)hipFree(c->soup);
  if (c->orig) (void)hipFree(c->orig);
  if (c->stats) (void)hipFree(c->stats);
  if (c->flag) (void)hipFree(c->flag);
  delete c;
  return GR_OK;
}

const char *gr_last_error(const gr_ctx *c) { return c ? c->err : "null context"; }

int gr_set_profiling(gr_ctx *c, int enabled) {
  if (!c) return GR_EINVAL;
  c->profiling = enabled != 0;
  for (auto &sp : c->spans) { c->pool.push_back(sp.a); c->pool.push_back(sp.b); }
  c->spans.clear();
  c->prof_views = 0; c->prof_raster_launches = 0;
  return GR_OK;
}

int gr_set_option(gr_ctx *c, int key, int value) {
  if (!c) return GR_EINVAL;
  switch (key) {
    case GR_OPT_TILE_H_LOG2:
      if (value != 5 && value != 6) return fail(c, GR_EINVAL, "tile height log2 must be 5 or 6");
      c->opt_thl = value; return GR_OK;
    case GR_OPT_BATCH:
      if (value < 1 || value > 64) return fail(c, GR_EINVAL, "batch must be in [1, 64]");
      c->opt_batch = value; return GR_OK;
    case GR_OPT_DEBUG:
      c->opt_dbg = value; return GR_OK;
    case GR_OPT_VARIANT:
      c->opt_var = value; return GR_OK;
    case GR_OPT_DEBUG_LDS:
      if (value < 0 || value > 65536) return fail(c, GR_EINVAL, "extra LDS bytes must be in [0, 65536]");
      c->opt_lds_pad = value; return GR_OK;
    case GR_OPT_DIRECT_CAP:
      if (value < 0 || value > 65536) return fail(c, GR_EINVAL, "slots per tile must be in [0, 65536]");
      c->opt_direct_cap = value; c->direct_ok = true; c->n_learned = 0; c->share_learned = false; return GR_OK;
    default: return fail(c, GR_EINVAL, "unknown option %d", key);
  }
}

int gr_get_stage_times(gr_ctx *c, gr_stage_times *o) {
  if (!c || !o) return GR_EINVAL;
  float acc[ST_N] = {0};
  for (auto &sp : c->spans) {
    GR_HIP(c, hipEventSynchronize(sp.b));
    float ms = 0.f;
    GR_HIP(c, hipEventElapsedTime(&ms, sp.a, sp.b));
    acc[sp.stage] += ms;
  }
  o->setup_ms = acc[ST_SETUP]; o->scan_ms = acc[ST_SCAN]; o->fill_ms = acc[ST_FILL]; o->raster_ms = acc[ST_RASTER];
  o->project_ms = acc[ST_PROJECT]; o->vote_ms = acc[ST_VOTE]; o->gather_ms = acc[ST_GATHER];
  o->raster_launches = c->prof_raster_launches; o->views = c->prof_views;
  for (auto &sp : c->spans) { c->pool.push_back(sp.a); c->pool.push_back(sp.b); }
  c->spans.clear();
  c->prof_views = 0; c->prof_raster_launches = 0;
  return GR_OK;
}

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
  int ax0 = 0, ax1 = 1, axs = 2;  // ax0, ax1: largest extents
  if (ext[axs] > ext[ax0]) std::swap(axs, ax0);
  if (ext[axs] > ext[ax1]) std::swap(axs, ax1);
  const float inv0 = ext[ax0] > 0.0f ? 65535.0f / ext[ax0] : 0.0f, inv1 = ext[ax1] > 0.0f ? 65535.0f / ext[ax1] : 0.0f;

  const int64_t nblk = ceil_div(F, GR_BLOCK);
  if (c->blk_cap < nblk) {
    if (c->blk) (void)hipFree(c->blk);
    c->blk = nullptr; c->blk_cap = 0;
    if (hipMalloc(&c->blk, sizeof(float4) * nblk) != hipSuccess) return fail(c, GR_ENOMEM, "block bounds allocation failed");
    if (c->blk_chunks) (void)hipFree(c->blk_chunks);
    c->blk_chunks = nullptr;
    if (hipMalloc(&c->blk_chunks, sizeof(uint32_t) * (GR_CHUNK_LIST + 1) * nblk) != hipSuccess)
      return fail(c, GR_ENOMEM, "block chunk list allocation failed");
    c->blk_cap = nblk;
  }
  if (c->soup_cap < F) {
    if (c->soup) (void)hipFree(c->soup);
    if (c->orig) (void)hipFree(c->orig);
    c->soup = nullptr; c->orig = nullptr; c->soup_cap = 0;
    if (hipMalloc(&c->soup, sizeof(float) * 9 * F) != hipSuccess) return fail(c, GR_ENOMEM, "face soup allocation failed");
    if (hipMalloc(&c->orig, sizeof(int32_t) * F) != hipSuccess) return fail(c, GR_ENOMEM, "face order allocation failed");
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
  hipLaunchKernelGGL(k_block_chunks, dim3((unsigned)ceil_div(nblk, 4)), dim3(256), 0, s, c->orig, F, c->blk_chunks);
  GR_HIP(c, hipGetLastError());
  c->verts = verts; c->faces = faces; c->V = V; c->F = F;
  return GR_OK;
}

int gr_raster_face_ids(gr_ctx *c, const float *cams, int n_views, int h, int w, int32_t *ids, float *depth,
                       void *stream) {
  if (c && !ids && !depth) return fail(c, GR_EINVAL, "null outputs");
  return raster_views(c, cams, n_views, h, w, ids, depth, nullptr, 0, nullptr, nullptr, 0, (hipStream_t)stream);
}

int gr_raster_status(gr_ctx *c, gr_raster_stats *o) {
  if (!c || !o) return GR_EINVAL;
  unsigned long long st[6] = {0, 0, 0, 0, 0, 0};
  GR_HIP(c, hipMemcpyAsync(st, c->stats, sizeof(st), hipMemcpyDeviceToHost, c->last_stream));
  GR_HIP(c, hipStreamSynchronize(c->last_stream));
  o->records = (int64_t)st[0]; o->entries = (int64_t)st[1]; o->max_entries = (int64_t)st[2];
  o->entry_cap = c->ent_cap; o->overflow = (int32_t)st[3];
  // views of the last call whose results are complete: every launch group in front of the first one that overflowed
  o->views_done = st[3] ? (int32_t)std::min<unsigned long long>(st[4] * (unsigned long long)std::max(c->last_B, 1),
                                                                (unsigned long long)c->last_n_views)
                        : c->last_n_views;
  if (st[3] && c->last_direct) {
    // A tile outgrew its fixed segment, or a face did not fit the 40-byte entry form (st[5]).  The counters kept counting,
    // so the need is known: the retry uses segments of that size if a launch group's entry memory stays within budget, and
    // bins exactly (count, scan, fill) otherwise; it uses 48-byte entries if the short form was missed.
    const int used = direct_cap(c, c->last_T);
    const bool grow = (int64_t)st[2] > used;
    if (st[5] != 0) learn(c, c->last_T, 0, true);
    if (grow) {
      const int64_t need = ((int64_t)st[2] + (int64_t)st[2] / 8 + 16 + 63) / 64 * 64;
      const int64_t bytes = need * (16 * GR_ENT_Q) * (int64_t)c->last_T * (int64_t)std::max(c->last_B, 1);
      if (need <= 16384 && bytes <= GR_DIRECT_BUDGET) learn(c, c->last_T, (int)need, false);
      else c->direct_ok = false;
      return fail(c, GR_EOVERFLOW, "single-pass binning overflow: a tile received %llu entries (slots per tile %d); "
                  "retry the call", st[2], used);
    }
    return fail(c, GR_EOVERFLOW, "single-pass binning: a face does not fit the 40-byte entry form; retry the call "
                "(48-byte entries from now on)");
  }
  if (st[3]) {
    // grow on the next call: exact need is known
    c->ent_cap_request = (int64_t)st[2] + (int64_t)st[2] / 8 + 65536;
    return fail(c, GR_EOVERFLOW, "bin list overflow: a view needs %llu entries, capacity %lld; retry the call",
                st[2], (long long)c->ent_cap);
  }
  return GR_OK;
}

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
  rc = ensure_winner(c, sizeof(uint32_t) * (size_t)F * B);
  if (rc) return rc;
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
  rc = ensure_winner(c, sizeof(uint32_t) * (size_t)F);
  if (rc) return rc;
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

int gr_raster_project_labels_u8(gr_ctx *c, const float *cams, const uint8_t *labels, int n_views, int h, int w, int C,
                                uint32_t *votes, uint32_t *counts, int32_t *ids_or_null, int flags, void *stream) {
  if (!c) return GR_EINVAL;
  if (!labels || !votes || !counts || C <= 0 || C > 255) return fail(c, GR_EINVAL, "bad fused project args C=%d", C);
  return raster_views(c, cams, n_views, h, w, ids_or_null, nullptr, labels, C, votes, counts, flags, (hipStream_t)stream);
}

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
  rc = ensure_winner(c, sizeof(uint32_t) * (size_t)F * B);
  if (rc) return rc;
  uint32_t *win = (uint32_t *)c->winner;
  // the "a value is no class index" flag: the context's flag word, read back below -- or, deferred, the high dword of the
  // caller's 64-bit pair counter (the count stays below 2^31: bit 32 is free)
  const bool defer = (flags & GR_FLAG_DEFER_CHECK) != 0;
  int *bad_flag = defer ? reinterpret_cast<int *>(key_count) + 1 : c->flag;
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
    (void)hipDeviceSynchronize();
    if (c->sort_tmp) (void)hipFree(c->sort_tmp);
    c->sort_tmp = nullptr; c->sort_bytes = 0;
    if (hipMalloc(&c->sort_tmp, need) != hipSuccess) return fail(c, GR_ENOMEM, "sort scratch allocation failed");
    c->sort_bytes = need;
  }
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
