// geograypher_amd/csrc/binning.hip -- per view: frustum cull of 64-face blocks, face set-up (R1 / R2 / R4), tile counting and
// -- single-pass binning -- entry compilation straight into the tiles' segments; R7 clipping; the exact two-pass path.
//   k_cull_blocks -> k_setup_cull (+ k_clip_faces) -> k_bin_stats            (exact: -> k_scan_tiles -> k_fill_compile)
// Compile with -ffp-contract=off: every floating-point operation below is individually rounded on purpose.
#include "gr_internal.hpp"
#include "dev_common.hpp"

using namespace grimpl;

namespace {

struct Vtx {
  int X, Y;
  float iz;
  bool valid;
  bool front, finite;  // q_z > near; camera-space point finite (R7: which invalid faces are clipped instead of dropped)
};

// the second half of R1: perspective divide, principal point, snap to 1/256 px -- in the rule-set's own order of operations, or
// (gl_order, GR_OPT_VERTEX_ORDER; oracle_raster.c R1-GL) in an OpenGL pipeline's: clip = P q, ndc = clip * (1 / w),
// window = fma(ndc, size / 2, size / 2), fixed = rint(256 (window - 0.5)), rows bottom-up -- op for op what Mesa's llvmpipe
// executes (95 % of the pixels on which R1 and llvmpipe disagree are vertices that land on the neighbouring step under that order)
__device__ __forceinline__ bool snap_vertex(float qx, float qy, float iz, const float *__restrict__ cam, int gl_order, int h, int w,
                                            int &X, int &Y) {
  if (!gl_order) {
    const float fx = cam[12] * qx;
    const float fy = cam[12] * qy;
    const float sx = cam[13] + fx * iz;
    const float sy = cam[14] + fy * iz;
    X = (int)floorf(sx * 256.0f + 0.5f);
    Y = (int)floorf(sy * 256.0f + 0.5f);
    return (fabsf(sx) < 16384.0f) && (fabsf(sy) < 16384.0f);
  }
  const float two_f = 2.0f * cam[12];
  const float px = two_f / (float)w, py = -(two_f / (float)h);   // correctly rounded divisions (-fhip-fp32-correctly-rounded-divide-sqrt)
  const float hw = 0.5f * (float)w, hh = 0.5f * (float)h;
  const float xc = px * qx, yc = py * qy;
  const float xn = xc * iz, yn = yc * iz;
  const float xw = __builtin_fmaf(xn, hw, hw), yw = __builtin_fmaf(yn, hh, hh);   // the ONE fused operation of the path, on purpose
  const float fx = (xw - 0.5f) * 256.0f, fy = (yw - 0.5f) * 256.0f;
  X = (int)rintf(fx) + 128;                    // v_rndne_f32: round half to even, like lrintf
  Y = 256 * h - 128 - (int)rintf(fy);
  return (fabsf(xw) < 16384.0f) && (fabsf(yw) < 16384.0f);
}

// R1 -- vertex transform, fp32, each operation individually rounded
__device__ __forceinline__ Vtx project_vertex(const float *__restrict__ p, const float *__restrict__ cam, int gl_order, int h, int w) {
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
  const bool inside = snap_vertex(qx, qy, iz, cam, gl_order, h, w, v.X, v.Y);
  v.valid = v.valid && inside;
  v.iz = iz;
  return v;
}

// ------------------------------------------------------------------------------------------------------------------
// K1  transform + cull + per-tile counts (+ compiled entries in single-pass mode).   grid (<= 1024, views)
//     (a) work list: the 64-face blocks whose bounding sphere passed k_cull_blocks (~87 % of a survey mesh is rejected
//         per view before a single face is read); every wave takes its own blocks;
//     (b) R1 once per DISTINCT vertex of the block (bvert / bidx of the upload), every face picks its three by position;
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

// The same with WEIGHTS: lane i asks for w_i in {1, 2, 4} consecutive list positions (a micro pair cut into 4 x 4 boxes); rank =
// the weights of the group's lanes below this one, size = the weights of the whole group.  w - 1 is 0, 1 or 3: two ballots of
// its bits turn the weighted sums into population counts.
__device__ __forceinline__ void wave_group_weighted(int t, int w, int lane, int &leader, int &rank, int &size) {
  leader = lane; rank = 0; size = 0;
  unsigned long long rem = __ballot(t >= 0);
  const unsigned long long b0 = __ballot(((w - 1) & 1) != 0), b1 = __ballot(((w - 1) & 2) != 0);
  const unsigned long long low = (1ull << lane) - 1ull;
  while (rem) {
    const int l = __ffsll((long long)rem) - 1;
    const int tl = __builtin_amdgcn_readlane(t, l);
    const unsigned long long m = __ballot(t == tl);
    if (t == tl) {
      leader = l;
      rank = __popcll(m & low) + __popcll(m & low & b0) + 2 * __popcll(m & low & b1);
      size = __popcll(m) + __popcll(m & b0) + 2 * __popcll(m & b1);
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
}

struct FaceForm {
  int Xf, Yf, dxf, dyf, tf, rf;   // FIRST edge (a = -dy > 0: it bounds a span from the left): origin vertex, direction,
  int Xm, Ym, dxm, dym, tm, rm;   //   top-left bias (0 / -1), reach = (TW/2)|a| + (TH/2)|b|;  MIDDLE edge;
  int Xl, Yl, dxl, dyl, tl, rl;   //   LAST edge (a < 0: from the right)
  int X0, Y0;                     // vertex 0: the anchor of the 1/z plane, whatever the edge order
  int w3, w4, w5;                 // slope words (16- or 24-bit packing)
  int ywf;                        // flag bits of the Yw word: bit 31 = 24-bit slopes, bit 30 = the span solver must correct its floor
  bool fast;                      // snapped bounding box below GR_FAST_EXT: 24-bit products, int32 everywhere
  int jmin, jmax, imin, imax;     // pixel bounding box (R2), clamped to the image
  int iz0, A, B, nface;           // 1/z at vertex 0 and its gradients (float bits), ~face
};

__device__ __forceinline__ FaceForm face_form(const int4 p0, const int4 p1, const int4 p2, int TW, int TH);
__device__ __forceinline__ bool tile_entry(const FaceForm &ff, int px0, int py0, int TW, int TH, int4 &e0, int4 &e1, int4 &e2,
                                           int &rows);
__device__ __forceinline__ bool compile_entry(const BinArgs &a, uint32_t *__restrict__ ctrl, int4 *__restrict__ comp,
                                              uint8_t *__restrict__ nr8, int64_t idx, const int4 p0, const int4 p1,
                                              const int4 p2, int px0, int py0, int TW, int TH);
__device__ __forceinline__ bool build_entry(const int4 p0, const int4 p1, const int4 p2, int px0, int py0, int TW, int TH,
                                            int4 &e0, int4 &e1, int4 &e2, int &rows);
__device__ __forceinline__ void store_entry(const BinArgs &a, uint32_t *__restrict__ ctrl, int4 *__restrict__ comp,
                                            uint8_t *__restrict__ nr8, int64_t idx, const int4 e0, const int4 e1, const int4 e2,
                                            int rows);
__device__ __forceinline__ void shifted_entry(const int4 b0, const int4 b1, const int4 b2, int boxx, int boxy, int px0, int py0, int dX,
                                              int dY, int TW, int TH, int4 &e0, int4 &e1, int4 &e2, int &rows);
// Faces whose snapped bounding box is smaller than GR_FAST_EXT sub-pixels (93 px) take a short form of the set-up (build_entry),
// fit the 40-byte entry (store_entry) and the 32-byte micro record (k_setup_cull)
#define GR_FAST_EXT 24000
__device__ __forceinline__ int pack16(int lo, int hi) { return (lo & 0xFFFF) | (hi << 16); }

// R1 / R2 / R4 for one face of the soup: the record (three int4) that compile_entry turns into per-tile entries, and the
// range of tiles its pixel bounding box touches.  Returns false for faces that draw nothing in this view; clip_me: the face
// straddles the near plane or the guard band (R7).
__device__ __forceinline__ bool face_setup_tail(const BinArgs &a, int face_id, Vtx v0, Vtx v1, Vtx v2, int4 &r0, int4 &r1,
                                                int4 &r2, int &tx0, int &tx1, int &ty0, int &ty1, bool &clip_me) {
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
  const double d1 = (double)v1.iz - (double)v0.iz;
  const double d2 = (double)v2.iz - (double)v0.iz;
  const double a2 = (double)area2;
  double n1, n2;
  n1 = d1 * (double)(v2.Y - v0.Y); n2 = d2 * (double)(v1.Y - v0.Y);
  const float A = (float)((n1 - n2) / a2);
  n1 = d2 * (double)(v1.X - v0.X); n2 = d1 * (double)(v2.X - v0.X);
  const float B = (float)((n1 - n2) / a2);
  r0 = make_int4(v0.X, v0.Y, v1.X, v1.Y);
  r1 = make_int4(v2.X, v2.Y, __float_as_int(v0.iz), face_id);
  r2 = make_int4(__float_as_int(A), __float_as_int(B), jmin | (jmax << 16), imin | (imax << 16));
  tx0 = jmin >> a.twl; tx1 = jmax >> a.twl;
  ty0 = imin >> a.thl; ty1 = imax >> a.thl;
  return true;
}

// a transformed vertex as the waves of K1 keep it in LDS: {X, Y, 1/z, valid | front << 1 | finite << 2}
__device__ __forceinline__ int4 pack_vtx(const Vtx &v) {
  return make_int4(v.X, v.Y, __float_as_int(v.iz), (v.valid ? 1 : 0) | (v.front ? 2 : 0) | (v.finite ? 4 : 0));
}
__device__ __forceinline__ Vtx unpack_vtx(const int4 q) {
  Vtx v;
  v.X = q.x; v.Y = q.y; v.iz = __int_as_float(q.z);
  v.valid = (q.w & 1) != 0; v.front = (q.w & 2) != 0; v.finite = (q.w & 4) != 0;
  return v;
}

// Single-pass binning of the wave's faces that reach over more than 2 x 2 tiles (`big`: this lane holds one, records r0 .. r2,
// tile rectangle tx0 .. ty1).  A per-lane walk over the tiles would leave 63 lanes waiting for the largest face -- 112 us per
// view on a scene with 20 000 trees seen obliquely (canopy and trunk faces of 300 x 40 pixels), where the terrain alone
// takes 7.  Instead the wave prefix-sums the tile counts of its faces and EXPANDS: the (face, tile) pairs are taken 64 at a
// time, a pair finds its face by a 6-step search over the prefix sums and pulls the record out of the owning lane's
// registers (ds_bpermute).  A tile the triangle does not touch takes no list slot.  The pairs of a step that name the same
// tile (neighbouring faces of one tree do) share ONE returning counter atomic (wave_group_capped: at most 16 groups are
// looked for, left-over pairs stand alone; 8 or 32 measure the same, 4: forest set-up +14 % -- setup_big_pairs_group_cap.log);
// all atomics of a step are in flight together.  (Round 5: the steps software-pipelined -- a step's positions and stores after
// the NEXT step's build_entry, built entries parked in LDS -- hide the atomics' round trip from the wave: forest set-up 39.6 ->
// 38.5 us per view, C2 +1.3 %: the other four waves of the SIMD hide it already.  Log and patch:
// profiles/r05_ab/setup_big_pairs_pipelined.*)
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
    if (tile >= 0 && lane == ld) base = atomicAdd(&cntS[cidx(a, tile)], (uint32_t)sz);
    const uint32_t pos = __shfl(base, ld) + (uint32_t)rk;
    if (tile >= 0) {
      if (pos < (uint32_t)a.cap_tile) {
        store_entry(a, ctrl, comp, nr8, (int64_t)tile * a.cap_tile + pos, e0, e1, e2, rows);
      } else atomicOr(&ctrl[2], 1u);
    }
  }
}

// Diagnostic build only (-DGR_STAMPS, tools/setup_phases.py): where a wave of K1 spends its life.  A stamp first waits for the
// wave's outstanding memory operations (vmcnt / lgkmcnt 0), so that a latency is charged to the phase that waited for it.
//   0 the block's loads (vertex list, positions)   1 transform + LDS exchange + face set-up (incl. the face-id load)
//   2 clip list + tile groups   3 counter atomics (issue + return)   4 entries of small faces   5 big faces, exact-path records
#ifdef GR_STAMPS
#define GR_SSTAMP(k) do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); const unsigned long long t_ = __builtin_amdgcn_s_memtime(); \
                          sacc[k] += t_ - st_; st_ = t_; } while (0)
#else
#define GR_SSTAMP(k) do { } while (0)
#endif

// DIRECT = true: single-pass binning.  Every tile owns a fixed segment of a.cap_tile entries; the list position
// returned by the (wave-aggregated) tile counter is final, so the compiled entry is written straight from here and
// the record planes, k_scan_tiles and k_fill_compile are skipped.  A tile that receives more than cap_tile entries
// raises the view's overflow word; the caller then repeats the call with the exact two-pass path (DIRECT = false).
#ifndef GR_SETUP_BPW
#define GR_SETUP_BPW 4u   // surviving blocks per wave of k_setup_cull, at least
#endif
#ifndef GR_SETUP_BPW_MAX
#define GR_SETUP_BPW_MAX 8u   // ... and at most (where the blocks outnumber the tiles)
#endif
// MICRO: the kernel of a call that keeps micro lists (a build of its own, like the tile kernel's: the ordinary kernel carries none
// of it)
template <bool DIRECT, bool MICRO = false>
__global__ __launch_bounds__(256)
__attribute__((amdgpu_waves_per_eu(5, 5)))  // at most 96 VGPRs: five waves per SIMD (three: +14.5 %, six -- 80 VGPRs, scratch -- +7 %: the kernel lives on latency hiding)
void k_setup_cull(const float *__restrict__ cams, BinArgs a) {
  const int lane = threadIdx.x & 63;
  uint32_t n_rec = 0;                    // single-pass binning: the wave's record count (a statistic), added when the wave leaves a view
  uint32_t n_mic = 0;                    // ... and its count of micro faces (pixel box at most 4 x 4)
#ifdef GR_STAMPS
  unsigned long long sacc[6] = {0, 0, 0, 0, 0, 0}, st_ = __builtin_amdgcn_s_memtime(), siter = 0;
  const unsigned long long st0_ = st_, sr0_ = __builtin_amdgcn_s_memrealtime();
#endif
  __shared__ int4 vt_s[4][GR_BLOCK_VERTS];  // the block's transformed vertices, one set of rows per wave (12 KiB per workgroup)
  int4 *const vt = vt_s[threadIdx.x >> 6];
  const int slot = blockIdx.y;
  const float *cam = cams + (int64_t)slot * GR_CAM_FLOATS;
  uint32_t *ctrl = a.ctrl + slot * a.ctrl_stride;
  const uint32_t *work = a.work + (int64_t)slot * a.work_stride;
  // every wave takes its own 64-face block from the view's work list (wave-uniform control flow, no workgroup barrier)
  const uint32_t wave0 = blockIdx.x * 4 + (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const uint32_t n_work = ctrl[3];       // (a) blocks that passed k_cull_blocks for this view
  if (wave0 >= n_work) return;
  // The grid is sized by the MESH (bin_batch: a workgroup per 128 blocks before culling); how many blocks of the view survive
  // the cull is only known here.  The view's list goes to its first ceil(n_work / GR_SETUP_BPW) waves -- at least that many
  // blocks per wave -- and the other waves leave: a view that sees 5 % of a 5 M-face mesh (config 5: 1.6 surviving blocks per
  // wave of the grid) no longer pays 2440 wave start-ups for 4000 blocks (set-up 10.7 -> 8.x us per view,
  // profiles/r05_ab/setup_grid_*.log: a grid sized per launch instead -- 6144 workgroups -- helps config 5 as much but costs C2 2 %); a view that keeps its waves busy anyway (C2: 4.2 blocks per wave) is not touched.
  // ... at least GR_SETUP_BPW = 4, and as many more as the view has surviving blocks per TILE, up to 8: where the blocks
  // outnumber the tiles -- a mesh rendered at a quarter of its photos' size: the hostile forest at 1000 x 750 has 25 blocks per
  // tile, C2 at that size 6.5 -- every tile counter is hit from many blocks at once and fewer waves in flight get through
  // faster.  (Measured with the counters of such an image packed into twelve 128-byte lines: forest at 1000 x 750 27.5 -> 24.5 us
  // per view at 16 blocks per wave, C2 at 1000 x 750 3.84 -> 3.76 at 6, full-size images -- under 2 blocks per tile -- 5-9 %
  // slower at 12-16.  With ONE counter per line for small images (ensure_bins; forest 24.9 -> 20.3 us) the forest's optimum is
  // back at 4-8 -- 19.5 us -- and 16 costs it 5 %: setup_grid_blocks_per_wave_*.log, setup_counter_per_line_small_images.log.)
  const uint32_t bpw = min((uint32_t)GR_SETUP_BPW_MAX, max((uint32_t)GR_SETUP_BPW, n_work / (uint32_t)max(a.T, 1)));
  const uint32_t wstep = min(gridDim.x * 4u, (n_work + bpw - 1u) / bpw);
  if (wave0 >= wstep) return;
  // (block indices through readfirstlane: loaded with a uniform address, but into a vector register -- every address derived
  // from them would be 64-bit VALU arithmetic instead of a scalar base.  The same for the wave's index above: the compiler
  // cannot see that threadIdx.x >> 6 is wave-uniform, and the loop's control flow and the work-list loads were vector code:
  // round 5, set-up stage 5.14 -> 4.78 us per C2 view together with the face-id load moved up beside the other two.)
  // (readfirstlane where the index is USED: applied to the load itself it would make the wave wait for the next block's index
  // at the top of every iteration instead of leaving the load in flight for the whole of it: +5 %)
  uint32_t blk_next = work[wave0];
  for (uint32_t wi = wave0; wi < n_work; wi += wstep) {
  const uint32_t blk_cur = (uint32_t)__builtin_amdgcn_readfirstlane((int)blk_next);
  const int64_t f = (int64_t)blk_cur * GR_BLOCK + lane;
  if (wi + wstep < n_work) blk_next = work[wi + wstep];

  bool keep = false, clip_me = false;
  int4 r0 = {0, 0, 0, 0}, r1 = {0, 0, 0, 0}, r2 = {0, 0, 0, 0};
  int tx0 = 0, tx1 = -1, ty0 = 0, ty1 = -1;
#ifdef GR_STAMPS
  st_ = __builtin_amdgcn_s_memtime(); ++siter;
#endif
  {
    // (b) R1 once per DISTINCT vertex of the block (k_block_vertices: about 48 for the 192 corners of a manifold patch, one
    //     round of the wave; a face soup takes three), results through the wave's own LDS rows -- LDS operations of one wave
    //     complete in order, so the reads below need no barrier --, then every face picks its three by position
    // everything the block needs from memory is requested here, together: the face's vertex-position word and id, this lane's
    // vertex of the block's list (every slot of the list is valid memory).  (Requesting them one block AHEAD -- a software
    // pipeline, 5 more live VGPRs: the fifth wave per SIMD or 12 bytes of scratch -- measured no better: 4.75 vs 4.78 us.)
    const uint32_t bi = f < a.F ? a.bidx[f] : 0u;
    const int face_id = f < a.F ? a.orig[f] : 0;
    const float *bv0 = a.bvert + ((int64_t)blk_cur * GR_BLOCK_VERTS + lane) * 3;
    const float vx = bv0[0], vy = bv0[1], vz = bv0[2];
    const int nv = (int)((uint32_t)__builtin_amdgcn_readfirstlane((int)bi) >> 24) + 1;  // lane 0 of a listed block is a face
#ifdef GR_STAMPS
    GR_SSTAMP(0);
#endif
    {
      const float p0[3] = {vx, vy, vz};
      if (lane < nv) vt[lane] = pack_vtx(project_vertex(p0, cam, a.gl_order, a.h, a.w));
    }
    if (nv > 64) {  // a face soup: two more rounds
      // the address is made from an opaque copy of the lane number: left to itself the compiler keeps `a.bvert + 12 * (lane + 64)`
      // as a loop invariant of the kernel's block loop and, at the 96 registers of five waves per SIMD, in SCRATCH -- a scratch
      // reload per block in front of these loads (tests/test_isa_waits.py holds the kernels to no scratch; C2 set-up 4.89 ->
      // 4.80 us per view, profiles/r05_ab/setup_scratch_fix.log)
      uint32_t l = lane;
      asm volatile("" : "+v"(l));
      const float *bvi = a.bvert + ((int64_t)blk_cur * GR_BLOCK_VERTS + l) * 3;
      for (int i = lane + 64; i < nv; i += 64) vt[i] = pack_vtx(project_vertex(bvi += 3 * 64, cam, a.gl_order, a.h, a.w));
    }
    const int4 q0 = vt[bi & 255u], q1 = vt[(bi >> 8) & 255u], q2 = vt[(bi >> 16) & 255u];
    if (f < a.F) keep = face_setup_tail(a, face_id, unpack_vtx(q0), unpack_vtx(q1), unpack_vtx(q2), r0, r1, r2, tx0, tx1, ty0, ty1, clip_me);
  }
  GR_SSTAMP(1);
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
  // (single-pass binning: ... and whose snapped bounding box stays below GR_FAST_EXT -- 93 px; a face of 93 to 128 px over 2 x 2
  // tiles goes with the big ones --: the entries of their second to fourth tiles are DERIVED from the first tile's, which the
  // short form of the edge constants allows exactly (shifted_entry))
  bool small_fp = keep && (tx1 - tx0 <= 1) && (ty1 - ty0 <= 1);
  if (DIRECT) {
    const int ext = max(imax3(r0.x, r0.z, r1.x) - imin3(r0.x, r0.z, r1.x), imax3(r0.y, r0.w, r1.y) - imin3(r0.y, r0.w, r1.y));
    small_fp = small_fp && ext < GR_FAST_EXT;
  }
  uint32_t *cntS = ctrl + GR_CTRL_HDR;
  uint32_t *cntB = cntS + a.Tcap;
  // MICRO records (round 5: lists; round 6: lean records).  In a view whose faces are mostly a few pixels wide -- a survey mesh at
  // render_img_scale 0.25, the reference's operating point (examples/aggregate_predictions.ipynb:61) -- every (face, tile) pair
  // whose part of the pixel bounding box in its tile is at most 8 x 8 goes to a SECOND list of the tile, filled from the back of
  // the tile's segment and counted in the otherwise unused cntB array, as one record per 4 x 4 box of that part (1, 2 or 4
  // records; 90 % of such pairs: one).  The tile kernel takes the list one record per LANE: sixteen point-sampled pixels, no span
  // solver, no mailboxes, no staging barriers (raster_tile.hip: micro_item).  A record is 32 bytes and costs this kernel about
  // thirty instructions -- three snapped vertices relative to the tile's centre pixel, the plane of 1/z, the key, the box --
  // where a compiled entry (build_entry: edge constants at the tile centre, edge order, slope packing, correction flags, 40
  // bytes + a row count) costs three hundred: the tile kernel derives its three edge functions from the vertices itself, once
  // per record, exactly (round 5 stored a compiled entry for every micro pair: 36 % of a wave's life in this kernel at 1000 x
  // 750, profiles/r05_ab/setup_phases_c2q.log).  Bit 30 of a tile id marks the class, so that the wave's groups keep the two
  // lists' positions apart.
  constexpr int GR_MICRO_BIT = 1 << 30;
  int mcls = 0;  // bit k: tile slot k (0 first, 1 right, 2 below, 3 below right) is a micro pair
  int mrec = 0;  // its records, 3 bits per slot (0, 1, 2 or 4)
  if (DIRECT && a.count_micro) {
    // how many faces of the view are micro faces (whole box at most 4 x 4): the statistic that switches micro lists on for
    // the NEXT call on this mesh and image size (gr_raster_status).  At full size only the clipped corners of ordinary faces
    // would qualify (2 % of the pairs) and give nearly every tile a list of a handful of entries -- a whole extra phase per
    // tile: ids kernel +7 % on C2 and C5 (profiles/r05_ab/micro_lists_per_tile_part.log) -- so the lists exist only where
    // micro FACES are the rule.  Counted only by the calls that can still learn from it (BinArgs::count_micro: the call that
    // looks at its first launch group -- the first for this mesh and image size --, or every call under the status-call protocol
    // of variant bit 16384; not with 48-byte entries, not once the lists are on).
    const int jmin = r2.z & 0xFFFF, jmax = (int)((uint32_t)r2.z >> 16), imin = r2.w & 0xFFFF, imax = (int)((uint32_t)r2.w >> 16);
    n_mic += (uint32_t)__popcll(__ballot(small_fp && jmax - jmin < 4 && imax - imin < 4));
  }
  if (DIRECT && MICRO && small_fp) {
    const int jmin = r2.z & 0xFFFF, jmax = (int)((uint32_t)r2.z >> 16), imin = r2.w & 0xFFFF, imax = (int)((uint32_t)r2.w >> 16);
    const int xb = (tx0 + 1) << a.twl, yb = (ty0 + 1) << a.thl;   // first column / row of the right / lower tiles
    const int wl = min(jmax, xb - 1) - jmin + 1, wr = jmax - xb + 1, ht = min(imax, yb - 1) - imin + 1, hb = imax - yb + 1;
    const int cl = wl <= 8 ? (wl + 3) >> 2 : 0, cr = wr <= 8 ? (wr + 3) >> 2 : 0;   // 4-pixel columns of the parts (0: not micro)
    const int rt = ht <= 8 ? (ht + 3) >> 2 : 0, rb = hb <= 8 ? (hb + 3) >> 2 : 0;
    const int m0 = cl * rt, m1 = tx1 > tx0 ? cr * rt : 0, m2 = ty1 > ty0 ? cl * rb : 0, m3 = (tx1 > tx0 && ty1 > ty0) ? cr * rb : 0;
    mrec = m0 | (m1 << 3) | (m2 << 6) | (m3 << 9);
    mcls = (m0 ? 1 : 0) | (m1 ? 2 : 0) | (m2 ? 4 : 0) | (m3 ? 8 : 0);
  }
  const int t00 = small_fp ? (ty0 * a.TX + tx0) | ((mcls & 1) ? GR_MICRO_BIT : 0) : -1;
  const int t01 = (small_fp && tx1 > tx0) ? (ty0 * a.TX + tx1) | ((mcls & 2) ? GR_MICRO_BIT : 0) : -1;
  const int t10 = (small_fp && ty1 > ty0) ? (ty1 * a.TX + tx0) | ((mcls & 4) ? GR_MICRO_BIT : 0) : -1;
  const int t11 = (small_fp && tx1 > tx0 && ty1 > ty0) ? (ty1 * a.TX + tx1) | ((mcls & 8) ? GR_MICRO_BIT : 0) : -1;
  // a tile's counter: cntS, or cntB for its micro list
  auto counter_of = [&](int t) { return (t & GR_MICRO_BIT) ? &cntB[cidx(a, t & ~GR_MICRO_BIT)] : &cntS[cidx(a, t)]; };
  int l0, k0, n0, l1 = lane, k1 = 0, n1 = 0, l2 = lane, k2 = 0, n2 = 0, l3 = lane, k3 = 0, n3 = 0;
  if (DIRECT && MICRO) {   // a micro pair takes as many positions as it has records, any other pair one
    wave_group_weighted(t00, max(mrec & 7, 1), lane, l0, k0, n0);
    if (__ballot(t01 >= 0)) wave_group_weighted(t01, max((mrec >> 3) & 7, 1), lane, l1, k1, n1);
    if (__ballot(t10 >= 0)) wave_group_weighted(t10, max((mrec >> 6) & 7, 1), lane, l2, k2, n2);
    if (__ballot(t11 >= 0)) wave_group_weighted(t11, max((mrec >> 9) & 7, 1), lane, l3, k3, n3);
  } else {
    wave_group(t00, lane, l0, k0, n0);
    if (__ballot(t01 >= 0)) wave_group(t01, lane, l1, k1, n1);
    if (__ballot(t10 >= 0)) wave_group(t10, lane, l2, k2, n2);
    if (__ballot(t11 >= 0)) wave_group(t11, lane, l3, k3, n3);
  }
  GR_SSTAMP(2);
  uint32_t base = 0, b0 = 0, b1 = 0, b2 = 0, b3 = 0;
  // record count: a list position for the exact path; a statistic otherwise, kept in a register until the wave is done (one
  // atomic per block on the view's one address made every wave of the view queue there: same-address atomics are served
  // one after the other, tools/ubench/atomic_rate.hip)
  if (DIRECT) n_rec += (uint32_t)n;
  else if (lane == leader) base = atomicAdd(&ctrl[0], (uint32_t)n);
  if (t00 >= 0 && lane == l0) b0 = atomicAdd(counter_of(t00), (uint32_t)n0);
  if (t01 >= 0 && lane == l1) b1 = atomicAdd(counter_of(t01), (uint32_t)n1);
  if (t10 >= 0 && lane == l2) b2 = atomicAdd(counter_of(t10), (uint32_t)n2);
  if (t11 >= 0 && lane == l3) b3 = atomicAdd(counter_of(t11), (uint32_t)n3);
  if (!DIRECT) base = __shfl(base, leader);
  int4 r3;
  r3.x = (int)(__shfl(b0, l0) + (uint32_t)k0);
  r3.y = (int)(__shfl(b1, l1) + (uint32_t)k1);
  r3.z = (int)(__shfl(b2, l2) + (uint32_t)k2);
  r3.w = (int)(__shfl(b3, l3) + (uint32_t)k3);
  GR_SSTAMP(3);
  if (DIRECT && MICRO && __ballot(mcls != 0)) {
    // the micro records, in the face's own lane: slot by slot (second to fourth tiles are the exception: rounds that few lanes
    // take part in, thirty instructions each), box by box
    const int X0 = r0.x, Y0 = r0.y, X1 = r0.z, Y1 = r0.w, X2 = r1.x, Y2 = r1.y;
    const int jmin = r2.z & 0xFFFF, jmax = (int)((uint32_t)r2.z >> 16), imin = r2.w & 0xFFFF, imax = (int)((uint32_t)r2.w >> 16);
    // the record holds the vertices as 16-bit offsets from the tile's centre pixel: they fit because a micro pair belongs to a
    // face below GR_FAST_EXT (small_fp)
    char *const segs = reinterpret_cast<char *>(a.comp + slot * a.ent_cap * GR_ENT_Q);
    const int TWh = 1 << (a.twl - 1), THh = 1 << (a.thl - 1);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int nk = (mrec >> (3 * k)) & 7;
      if (!__ballot(nk != 0)) continue;
      if (nk != 0) {
        const int tx = tx0 + (k & 1), ty = ty0 + (k >> 1);
        const int px0 = tx << a.twl, py0 = ty << a.thl;
        const int jlo = max(jmin - px0, 0), jhi = min(jmax - px0, (1 << a.twl) - 1);
        const int ilo = max(imin - py0, 0), ihi = min(imax - py0, (1 << a.thl) - 1);
        const int Pxc = (px0 + TWh) * 256 + 128, Pyc = (py0 + THh) * 256 + 128;   // centre of the tile's centre pixel
        const uint32_t pos = (uint32_t)(k == 0 ? r3.x : k == 1 ? r3.y : k == 2 ? r3.z : r3.w);
        const int ncx = (jhi - jlo + 4) >> 2;   // 4-pixel columns of the part: 1 or 2
        if ((pos + (uint32_t)nk) * 4u <= (uint32_t)a.cap_tile * 5u) {   // inside the segment (cap slots of 40 bytes; whether the two lists met: k_bin_stats)
          // record p of the tile's list: the 32 bytes that end 32 p bytes before the end of the tile's segment
          char *const rec_end = segs + ((int64_t)(ty * a.TX + tx) + 1) * a.cap_tile * 40 - (int64_t)pos * 32;
          const int4 va = make_int4(pack16(X0 - Pxc, Y0 - Pyc), pack16(X1 - Pxc, Y1 - Pyc), pack16(X2 - Pxc, Y2 - Pyc), r1.z);
          for (int sb = 0; sb < nk; ++sb) {
            const int sx = sb & (ncx - 1), sy = sb >> (ncx - 1);
            const int c0 = jlo + 4 * sx, c1 = min(c0 + 3, jhi), q0 = ilo + 4 * sy, q1 = min(q0 + 3, ihi);
            int4 *dst = reinterpret_cast<int4 *>(rec_end - 32 * (sb + 1));
            dst[0] = va;
            dst[1] = make_int4(r2.x, r2.y, (int)~(uint32_t)r1.w, c0 | ((c1 - c0) << 6) | (q0 << 8) | ((q1 - q0 + 1) << 14));
          }
        } else atomicOr(&ctrl[2], 1u);
      }
    }
  }
  if (DIRECT) {
    // faces over at most 2x2 tiles: positions came from the wave-aggregated counters; the lanes of a group hold
    // consecutive positions of the same tile segment, so their entries are written side by side.  Every such face has a
    // FIRST tile -- one dense round of build_entry in the face's own lane --; second to fourth tiles are the exception (0.5 per
    // face): those (face, tile) pairs are compacted -- prefix sum, 6-step search for the owning lane -- and their entries
    // DERIVED from the owner's first-tile entry, read back from LDS (shifted_entry: three multiply-adds per edge constant
    // instead of the whole set-up.  Rounds 2-5 pulled the owner's records by ds_bpermute and ran build_entry again: that round
    // was 23 % of the kernel on C2, profiles/r06_ab/setup_removal_probes.log; three mostly idle rounds in the face's own lane
    // from one FaceForm had measured +4 %: profiles/r05_ab/setup_own_lane_tiles_vs_compaction.log.)  With micro lists both rounds
    // see only the pairs that are no micro pairs -- in a view of micro faces none: the wave skips them.
    int4 *comp = a.comp + slot * a.ent_cap * GR_ENT_Q;
    uint8_t *nr8 = a.nrow8 + slot * a.ent_cap;
    const int TW = 1 << a.twl, TH = 1 << a.thl;
    const bool first_general = small_fp && !(mcls & 1);
    const int shape = small_fp ? ((tx1 > tx0 ? 1 : 0) | (ty1 > ty0 ? 2 : 0)) : 0;  // which neighbours exist: 1 right, 2 below
    // the face's extra tile slots that take a compiled entry: bit k - 1 for slot k (1 right, 2 below, 3 below right)
    const int extra = (shape == 3 ? 7 : shape) & ~(mcls >> 1);
    // the FIRST tile's entry: stored if that pair is no micro pair, and parked in the wave's LDS rows (the block's vertices are
    // done with) as the base of the face's other entries
    const bool base_needed = first_general || extra != 0;
    if (!MICRO || __ballot(base_needed)) {
      if (base_needed) {
        int4 e0, e1, e2;
        int rows;
        build_entry(r0, r1, r2, tx0 << a.twl, ty0 << a.thl, TW, TH, e0, e1, e2, rows);
        if (first_general) {
          if ((uint32_t)r3.x < (uint32_t)a.cap_tile) store_entry(a, ctrl, comp, nr8, (int64_t)t00 * a.cap_tile + (uint32_t)r3.x, e0, e1, e2, rows);
          else atomicOr(&ctrl[2], 1u);
        }
        vt[lane] = e0;                                          // c_first c_mid c_last | slopes a
        vt[64 + lane] = make_int4(e1.x, e1.z, e1.w, e2.x);      // slopes b | iz0 A B
        vt[128 + lane] = make_int4(e2.y, e2.z, e2.w, 0);        // Xw | ~face | Yw + flags
      }
    }
    const int ne = __popc((unsigned)extra);
    const int incl_e = wave_incl_scan(ne);
    const int total_e = __builtin_amdgcn_readlane(incl_e, 63);
    const int geo = tx0 | (ty0 << 12) | (extra << 24);
    for (int k0 = 0; k0 < total_e; k0 += 64) {
      const int q = k0 + lane;
      int t = 0;  // the face of pair q: the first lane whose inclusive sum exceeds q
#pragma unroll
      for (int step = 32; step >= 1; step >>= 1) t += (__shfl(incl_e, t + step - 1) <= q) ? step : 0;
      t = min(t, 63);
      const int g = __shfl(geo, t);
      int ex = (g >> 24) & 7;
      const int which = q - (__shfl(incl_e, t) - __popc((unsigned)ex));  // 0 .. 2: the face's which-th extra tile
      if (which >= 1) ex &= ex - 1;
      if (which >= 2) ex &= ex - 1;
      const int k = __ffs(ex);                                         // tile slot 1 (right), 2 (below), 3 (below right)
      const int4 b0 = vt[t], b1 = vt[64 + t], b2 = vt[128 + t];        // the owner's first-tile entry (LDS operations of a wave complete in order)
      const int boxx = __shfl(r2.z, t), boxy = __shfl(r2.w, t);
      const int py = __shfl(r3.y, t), pz = __shfl(r3.z, t), pw = __shfl(r3.w, t);
      if (q < total_e) {
        const uint32_t pos = (uint32_t)(k == 1 ? py : k == 2 ? pz : pw);
        const int tx = (g & 0xFFF) + (k & 1), ty = ((g >> 12) & 0xFFF) + (k >> 1);
        if (pos < (uint32_t)a.cap_tile) {
          int4 e0, e1, e2;
          int rows;
          shifted_entry(b0, b1, b2, boxx, boxy, tx << a.twl, ty << a.thl, (k & 1) ? TW : 0, (k >> 1) ? TH : 0, TW, TH, e0, e1, e2, rows);
          store_entry(a, ctrl, comp, nr8, (int64_t)(ty * a.TX + tx) * a.cap_tile + pos, e0, e1, e2, rows);
        } else atomicOr(&ctrl[2], 1u);
      }
    }
  }
  GR_SSTAMP(4);
  if (DIRECT) {
    // faces over more than 2 x 2 tiles: the wave expands their (face, tile) pairs right here, from the records it holds
    // (bin_big_pairs).  (Round 2 sent them through a per-view list and a second kernel -- one returning atomic per block on
    // ONE address per view: forest set-up 51.5 vs 39.0 us per view.)
    const bool big_fp = keep && !small_fp;
    if (__ballot(big_fp)) bin_big_pairs(a, ctrl, slot, lane, big_fp, r0, r1, r2, tx0, tx1, ty0, ty1);
  }
  if (keep && !DIRECT) {
    int4 *rec = a.rec + slot * a.rec_stride;
    const int64_t s = (int64_t)base + prefix, RP = a.rec_stride >> 2;  // four planes of RP >= F records per slot
    rec[s] = r0;
    rec[RP + s] = r1;
    rec[2 * RP + s] = r2;
    rec[3 * RP + s] = r3;
    if (!small_fp)
      for (int ty = ty0; ty <= ty1; ++ty)
        for (int tx = tx0; tx <= tx1; ++tx) atomicAdd(&cntB[ty * a.TX + tx], 1u);
  }
  GR_SSTAMP(5);
  }  // work list loop
  if (DIRECT && lane == 0 && n_rec) atomicAdd(&ctrl[0], n_rec);
  if (DIRECT && lane == 0 && n_mic) atomicAdd(&ctrl[6], n_mic);
#ifdef GR_STAMPS
  if (lane == 0 && a.stamps) {  // the second half of the stamp buffer: 1024 slots of 16 words
    unsigned long long *sd = a.stamps + 16 * 1024 + 16 * ((blockIdx.x * 4 + (threadIdx.x >> 6) + blockIdx.y * 977) & 1023);
    for (int k = 0; k < 6; ++k) atomicAdd(&sd[k], sacc[k]);
    atomicAdd(&sd[12], __builtin_amdgcn_s_memtime() - st0_);
    atomicAdd(&sd[13], __builtin_amdgcn_s_memrealtime() - sr0_);
    atomicAdd(&sd[14], siter);
    atomicAdd(&sd[15], 1ull);
  }
#endif
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
  // (with micro lists a tile's segment holds both lists, one from each end: compiled entries in whole chunks of 64 x 40 bytes
  // from the front, 32-byte micro records from the back -- together, in 40-byte slots, they must fit)
  for (int t = threadIdx.x; t < a.T; t += 1024) {
    const int64_t i = cidx(a, t);
    uint32_t c = cnt[i];
    if (a.micro) { const uint32_t cm = cnt[a.Tcap + i]; if (cm) c = ((c + 63u) & ~63u) + (cm * 32u + 39u) / 40u; }
    sum += c; mx = max(mx, c);
  }
  for (int o = 32; o > 0; o >>= 1) { sum += __shfl_xor(sum, o); mx = max(mx, (uint32_t)__shfl_xor((int)mx, o)); }
  if ((threadIdx.x & 63) == 0) { part[threadIdx.x >> 6] = sum; pmax[threadIdx.x >> 6] = mx; }
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long total = 0; uint32_t m = 0;
    for (int k = 0; k < 16; ++k) { total += part[k]; m = max(m, pmax[k]); }
    ctrl[1] = (uint32_t)total;
    atomicAdd(&a.stats[6], (unsigned long long)ctrl[3]);   // 64-face blocks that passed the frustum cull
    const bool ovf = m > (uint32_t)a.cap_tile || ctrl[2] != 0;
    atomicAdd(&a.stats[0], (unsigned long long)ctrl[0]);
    atomicAdd(&a.stats[1], total);
    atomicAdd(&a.stats[8], (unsigned long long)ctrl[6]);  // micro faces (pixel box at most 4 x 4)
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
    atomicMax(&a.stats[9], (unsigned long long)ctrl[0]);   // records the view needs (clipped faces: several each)
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
#define GR_FLOOR_NOCORR_MAX 16000  // largest slope magnitude for which edge_floor<false> is exact (see there)
// build_entry in two halves.  face_form: everything about an entry that does not depend on the tile -- the three edges in the
// order the tile kernel wants them (each with its origin vertex, direction, fill-rule bias and "reach"), the slope words, the
// flag bits, the plane of 1/z -- computed ONCE per face; tile_entry: the part that does -- the three edge constants at the
// tile's centre pixel, the rows of the entry, the anchors of the row / column words -- about a quarter of the whole.  K1
// compiles the first and the second-to-fourth tiles of a small face in the face's own lane from one FaceForm (round 5;
// rounds 2-4 dealt the extra (face, tile) pairs to lanes through 15 ds_bpermute per pair and ran all of build_entry again).
__device__ __forceinline__ FaceForm face_form(const int4 p0, const int4 p1, const int4 p2, int TW, int TH) {
  FaceForm ff;
  const int X0 = p0.x, Y0 = p0.y, X1 = p0.z, Y1 = p0.w, X2 = p1.x, Y2 = p1.y;
  const int dx0 = X1 - X0, dy0 = Y1 - Y0, dx1 = X2 - X1, dy1 = Y2 - Y1, dx2 = X0 - X2, dy2 = Y0 - Y2;
  const int t0 = ((dy0 < 0) || (dy0 == 0 && dx0 < 0)) ? 0 : -1;  // R3 tie rule as a bias: left and bottom edges own their pixels
  const int t1 = ((dy1 < 0) || (dy1 == 0 && dx1 < 0)) ? 0 : -1;
  const int t2 = ((dy2 < 0) || (dy2 == 0 && dx2 < 0)) ? 0 : -1;
  const int ext = max(imax3(X0, X1, X2) - imin3(X0, X1, X2), imax3(Y0, Y1, Y2) - imin3(Y0, Y1, Y2));
  ff.fast = ext < GR_FAST_EXT;
  const int a0 = -dy0, a1 = -dy1, a2 = -dy2;
  // The edges are stored in an order the tile kernel relies on: FIRST an edge with a > 0 (it bounds the span from the left),
  // LAST one with a < 0 (from the right), the remaining one in between -- a triangle of non-zero area has both kinds (the a_k
  // sum to zero; so do the b_k: the last edge's slopes are not stored).
  const int kf = a0 > 0 ? 0 : (a1 > 0 ? 1 : 2);   // first: a > 0
  const int kl = a0 < 0 ? 0 : (a1 < 0 ? 1 : 2);   // last: a < 0
  const int km = 3 - kf - kl;
  auto pick = [](int k, int v0, int v1, int v2) { return k == 0 ? v0 : (k == 1 ? v1 : v2); };
  ff.Xf = pick(kf, X0, X1, X2); ff.Yf = pick(kf, Y0, Y1, Y2); ff.dxf = pick(kf, dx0, dx1, dx2); ff.dyf = pick(kf, dy0, dy1, dy2);
  ff.Xm = pick(km, X0, X1, X2); ff.Ym = pick(km, Y0, Y1, Y2); ff.dxm = pick(km, dx0, dx1, dx2); ff.dym = pick(km, dy0, dy1, dy2);
  ff.Xl = pick(kl, X0, X1, X2); ff.Yl = pick(kl, Y0, Y1, Y2); ff.dxl = pick(kl, dx0, dx1, dx2); ff.dyl = pick(kl, dy0, dy1, dy2);
  ff.tf = pick(kf, t0, t1, t2); ff.tm = pick(km, t0, t1, t2); ff.tl = pick(kl, t0, t1, t2);
  ff.rf = (TW / 2) * abs(ff.dyf) + (TH / 2) * abs(ff.dxf);
  ff.rm = (TW / 2) * abs(ff.dym) + (TH / 2) * abs(ff.dxm);
  ff.rl = (TW / 2) * abs(ff.dyl) + (TH / 2) * abs(ff.dxl);
  const int af = -ff.dyf, am = -ff.dym, bf = ff.dxf, bm = ff.dxm;
  // slopes: four values (the last edge's are -(first + middle)).  Two packings: 16 bits each when every slope of the face
  // fits (faces below 128 pixels: nearly all of them), else 24 bits each, flagged in bit 31 of the Yw word
  const bool narrow = max(max(abs(a0), abs(a1)), max(abs(a2), max(abs(dx0), max(abs(dx1), abs(dx2))))) <= 32767;
  if (narrow) {
    ff.w3 = pack16(af, am); ff.w4 = pack16(bf, bm);
    ff.w5 = ff.fast ? 0 : 1;  // never read for 16-bit slopes; non-zero tells store_entry that the short form does not fit
  } else {
    ff.w3 = (af & 0xFFFFFF) | (am << 24);
    ff.w4 = ((am >> 8) & 0xFFFF) | (bf << 16);
    ff.w5 = ((bf >> 16) & 0xFF) | (bm << 8);
  }
  // bit 31: 24-bit slopes; bit 30: some slope magnitude beyond GR_FLOOR_NOCORR_MAX (the span solver must correct its floor)
  const bool corr = !narrow || max(abs(a0), max(abs(a1), abs(a2))) > GR_FLOOR_NOCORR_MAX;
  ff.ywf = (narrow ? 0 : (int)0x80000000) | (corr ? 0x40000000 : 0);
  ff.X0 = X0; ff.Y0 = Y0;
  ff.jmin = p2.z & 0xFFFF; ff.jmax = (int)((uint32_t)p2.z >> 16);
  ff.imin = p2.w & 0xFFFF; ff.imax = (int)((uint32_t)p2.w >> 16);
  ff.iz0 = p1.z; ff.A = p2.x; ff.B = p2.y; ff.nface = (int)~(uint32_t)p1.w;
  return ff;
}

__device__ __forceinline__ bool tile_entry(const FaceForm &ff, int px0, int py0, int TW, int TH, int4 &e0, int4 &e1, int4 &e2,
                                           int &rows) {
  const int Pxo = px0 * 256 + 128, Pyo = py0 * 256 + 128;  // centre of the tile's first pixel
  const int jlo = max(ff.jmin - px0, 0), jhi = min(ff.jmax - px0, TW - 1);
  const int ilo = max(ff.imin - py0, 0), ihi = min(ff.imax - py0, TH - 1);
  // row word, CENTRED like everything else the tile kernel reads: float(P_y - Y0) of centred row y_c = y - TH/2 is
  // float(256 y_c + Yw); the entry's first row as y_c (6 bits, signed).  |Pyo - Y0| + 8192 < 2^23 inside the guard band
  const int yw = ((Pyo - ff.Y0 + (TH / 2) * 256) & 0xFFFFFF) | (((ilo - TH / 2) & 0x3F) << 24);
  int nr = (jhi >= jlo) ? max(ihi - ilo + 1, 0) : 0;         // rows of the entry in this tile (<= 64)
  // ONE form for every face, however large: the three edge functions in units of 256 around the tile CENTRE,
  //   E'_k(x_c, y_c) = C'_k + a_k x_c + b_k y_c,   x_c = x - TW/2, y_c = y - TH/2,   a_k = -dy_k, b_k = dx_k (|.| < 2^23),
  //   C'_k = floor(C_k / 256) with C_k the exact edge value (fill-rule bias included) at the centre pixel.
  // Exact because A_k = 256 a_k and B_k = 256 b_k are multiples of 256: E_k >= 0 <=> floor(E_k / 256) >= 0 <=> E'_k >= 0.
  // C'_k can be as large as 2^39 for a face that spans the guard band, but inside the tile (|x_c| <= TW/2 + 2 with the
  // solver's reach, |y_c| <= TH/2) the sum a x_c + b y_c stays within M_k = (TW/2 + 2)|a_k| + (TH/2)|b_k|: a C'_k beyond
  // +-M_k cannot change sign in the tile, so it is CLAMPED to +-(M_k + 1) -- same coverage, and every value the tile
  // kernel forms fits int32 (M_k < 2^29.1).  The plane of 1/z refers to vertex 0 whatever the edge order.
  const int Pxc = Pxo + (TW / 2) * 256, Pyc = Pyo + (TH / 2) * 256;  // centre of the tile's centre pixel
  int cf, cm, cl;
  if (ff.fast) {  // every product has 24-bit factors and every value fits int32: no 64-bit arithmetic, no clamp
    cf = (__mul24(ff.dxf, Pyc - ff.Yf) - __mul24(ff.dyf, Pxc - ff.Xf) + ff.tf) >> 8;
    cm = (__mul24(ff.dxm, Pyc - ff.Ym) - __mul24(ff.dym, Pxc - ff.Xm) + ff.tm) >> 8;
    cl = (__mul24(ff.dxl, Pyc - ff.Yl) - __mul24(ff.dyl, Pxc - ff.Xl) + ff.tl) >> 8;
  } else {
    const long long Cf = ((long long)ff.dxf * (Pyc - ff.Yf) - (long long)ff.dyf * (Pxc - ff.Xf) + (long long)ff.tf) >> 8;
    const long long Cm = ((long long)ff.dxm * (Pyc - ff.Ym) - (long long)ff.dym * (Pxc - ff.Xm) + (long long)ff.tm) >> 8;
    const long long Cl = ((long long)ff.dxl * (Pyc - ff.Yl) - (long long)ff.dyl * (Pxc - ff.Xl) + (long long)ff.tl) >> 8;
    const long long hx = TW / 2 + 2, hy = TH / 2;
    const long long Mf = hx * abs(ff.dyf) + hy * abs(ff.dxf) + 1, Mm = hx * abs(ff.dym) + hy * abs(ff.dxm) + 1,
                    Ml = hx * abs(ff.dyl) + hy * abs(ff.dxl) + 1;
    cf = (int)min(max(Cf, -Mf), Mf);
    cm = (int)min(max(Cm, -Mm), Mm);
    cl = (int)min(max(Cl, -Ml), Ml);
  }
  // The bounding box reaches this tile; the triangle itself may not (the far corner of a diagonal face).  An edge whose
  // value is negative even at the tile corner most in its favour, C' + (TW/2)|a| + (TH/2)|b| < 0, excludes every pixel of
  // the tile: the entry is DEAD (0 rows: the tile kernel never looks at it); bin_big_pairs asks before it takes a list slot.
  const bool touches = nr > 0 && cf + ff.rf >= 0 && cm + ff.rm >= 0 && cl + ff.rl >= 0;
  if (!touches) nr = 0;
  rows = nr;
  // float(P_x - X0) of the pixel with CENTRED column x_c = x - TW/2 is float(256 x_c + Xw)
  const int xw = ((Pxo - ff.X0 + (TW / 2) * 256) & 0xFFFFFF) | (nr << 24);
  e0 = make_int4(cf, cm, cl, ff.w3);
  e1 = make_int4(ff.w4, ff.w5, ff.iz0, ff.A);
  // ~face sits in an EVEN word: the tile kernel forms the 64-bit key (depth << 32 | ~face) in the register pair the entry
  // was read into, without a move
  e2 = make_int4(ff.B, xw, ff.nface, yw | ff.ywf);
  return touches;
}

// The entry of a face below GR_FAST_EXT in the tile dX pixels to the right of / dY pixels below the tile whose entry is given
// (b0 = {c_first, c_mid, c_last, slopes a}, b1 = {slopes b, iz0, A, B}, b2 = {Xw, ~face, Yw + flags, -}; boxx / boxy: the face's
// pixel box, px0 / py0: the new tile's first pixel).  Exact: in the short form the edge constants are the unclamped
// c_k = floor(E_k(centre pixel) / 256) and E_k is linear with slopes 256 a_k, 256 b_k, so the constant at a centre dX, dY pixels
// away is c_k + a_k dX + b_k dY -- the integer the full set-up (face_form + tile_entry, ten times the instructions) computes;
// slopes, plane and key do not depend on the tile, the anchors of the row / column words move by the same step.
__device__ __forceinline__ void shifted_entry(const int4 b0, const int4 b1, const int4 b2, int boxx, int boxy, int px0, int py0, int dX,
                                              int dY, int TW, int TH, int4 &e0, int4 &e1, int4 &e2, int &rows) {
  const int af = (b0.w << 16) >> 16, am = b0.w >> 16, bf = (b1.x << 16) >> 16, bm = b1.x >> 16;   // 16-bit slopes (pack16)
  const int al = -(af + am), bl = -(bf + bm);                                                     // the a_k sum to zero; so do the b_k
  const int cf = b0.x + __mul24(af, dX) + __mul24(bf, dY);
  const int cm = b0.y + __mul24(am, dX) + __mul24(bm, dY);
  const int cl = b0.z + __mul24(al, dX) + __mul24(bl, dY);
  const int rf = (TW / 2) * abs(af) + (TH / 2) * abs(bf), rm = (TW / 2) * abs(am) + (TH / 2) * abs(bm),
            rl = (TW / 2) * abs(al) + (TH / 2) * abs(bl);
  const int jmin = boxx & 0xFFFF, jmax = (int)((uint32_t)boxx >> 16), imin = boxy & 0xFFFF, imax = (int)((uint32_t)boxy >> 16);
  const int jlo = max(jmin - px0, 0), jhi = min(jmax - px0, TW - 1);
  const int ilo = max(imin - py0, 0), ihi = min(imax - py0, TH - 1);
  int nr = (jhi >= jlo) ? max(ihi - ilo + 1, 0) : 0;
  if (!(nr > 0 && cf + rf >= 0 && cm + rm >= 0 && cl + rl >= 0)) nr = 0;   // a dead entry, as in tile_entry
  rows = nr;
  const int xw = ((b2.x + 256 * dX) & 0xFFFFFF) | (nr << 24);
  const int yw = ((b2.z + 256 * dY) & 0xFFFFFF) | (((ilo - TH / 2) & 0x3F) << 24) | (b2.z & (int)0xC0000000);
  e0 = make_int4(cf, cm, cl, b0.w);
  e1 = make_int4(b1.x, 0, b1.y, b1.z);
  e2 = make_int4(b1.w, xw, b2.y, yw);
}

__device__ __forceinline__ bool build_entry(const int4 p0, const int4 p1, const int4 p2, int px0, int py0, int TW, int TH,
                                            int4 &e0, int4 &e1, int4 &e2, int &rows) {
  const FaceForm ff = face_form(p0, p1, p2, TW, TH);
  return tile_entry(ff, px0, py0, TW, TH, e0, e1, e2, rows);
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
    // (the slot was handed out already: it must not keep stale bytes -- an older view's entry, or 48-byte data read as a
    // 40-byte entry.  Zero rows: no work item of the tile kernel ever looks at it; the view is repeated anyway)
    if (e1.y != 0 || e2.w < 0) { atomicOr(&ctrl[2], 2u); nr8[idx] = 0; return; }
    // a chunk of 64 entries (2560 bytes) holds the 64 x {s0 .. s7} first, then the 64 x {s8, s9}: the tile kernel copies the
    // chunk to LDS as it is and reads an entry with two 16-byte reads and one 8-byte read, all aligned.  (Three planes -- 64 x
    // {s0 .. s3}, 64 x {s4 .. s7}, 64 x {s8, s9}: a tile group's lanes write consecutive bytes with every store -- take 4.5 % off
    // this kernel and add 1.7 % to the tile kernel, 3 % if it re-orders the pieces as it stages them: a wash,
    // profiles/r06_ab/setup_entry_planes.log)
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
  const int64_t RP = a.rec_stride >> 2;                                  // records per plane
  const uint32_t n_rec = (uint32_t)min((int64_t)ctrl[0], RP);            // (the counter keeps counting past the planes: the call is repeated)
  const uint32_t *cntS = ctrl + GR_CTRL_HDR;
  const uint32_t *off = ctrl + GR_CTRL_HDR + 2 * a.Tcap;
  uint32_t *cur = ctrl + GR_CTRL_HDR + 3 * a.Tcap;
  const int4 *rec0 = a.rec + slot * a.rec_stride;
  int4 *comp = a.comp + slot * a.ent_cap * GR_ENT_Q;
  uint8_t *nr8 = a.nrow8 + slot * a.ent_cap;
  const int TW = 1 << a.twl, TH = 1 << a.thl;
  for (uint32_t r = blockIdx.x * 256 + threadIdx.x; r < n_rec; r += gridDim.x * 256) {
    const int4 p0 = rec0[r], p1 = rec0[RP + r], p2 = rec0[2 * RP + r];
    const int tx0 = (p2.z & 0xFFFF) >> a.twl, tx1 = (int)((uint32_t)p2.z >> 16) >> a.twl;
    const int ty0 = (p2.w & 0xFFFF) >> a.thl, ty1 = (int)((uint32_t)p2.w >> 16) >> a.thl;
    const bool small_fp = (tx1 - tx0 <= 1) && (ty1 - ty0 <= 1);
    int4 pos = {0, 0, 0, 0};
    if (small_fp) pos = rec0[3 * RP + r];
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
        const uint32_t pos = atomicAdd(&cntS[cidx(a, t)], 1u);
        if (pos < (uint32_t)a.cap_tile) {
          const int64_t idx = (int64_t)t * a.cap_tile + pos;
          compile_entry(a, ctrl, comp, nr8, idx, r0, r1, r2, tx << a.twl, ty << a.thl, 1 << a.twl, 1 << a.thl);
        } else atomicOr(&ctrl[2], 1u);
      }
  } else {
    // a clipped face becomes up to six triangles, each a record: more records than the planes hold (F, unless an earlier call
    // asked for more) -> the view is reported like any overflow; the counter keeps counting, so the retry knows the need
    // (k_scan_tiles: stats[9]; found by tools/fuzz_parity.py seed 934669: 18 faces around the camera, exact binning)
    const uint32_t s = atomicAdd(&ctrl[0], 1u);
    const int64_t RP = a.rec_stride >> 2;
    if ((int64_t)s >= RP) { atomicMax(&a.stats[3], 1ull); atomicMin(&a.stats[4], (unsigned long long)a.group); return; }
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
    rec[s] = r0; rec[RP + s] = r1; rec[2 * RP + s] = r2; rec[3 * RP + s] = r3;
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
      int Xs, Ys;
      if (!snap_vertex(qx, qy, iz, cam, a.gl_order, a.h, a.w, Xs, Ys)) { bad = true; break; }
      sX[e][tid] = Xs;
      sY[e][tid] = Ys;
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

// K0i  the zeroes a launch group's bin pass starts from (bin_batch); stats != null: the call's first group
__global__ __launch_bounds__(256) void k_bin_init(uint4 *__restrict__ ctrl16, int64_t n16, uint32_t *__restrict__ touched, int64_t nt,
                                                  unsigned long long *__restrict__ stats) {
  const int64_t i0 = (int64_t)blockIdx.x * 256 + threadIdx.x, step = (int64_t)gridDim.x * 256;
  for (int64_t i = i0; i < n16; i += step) ctrl16[i] = make_uint4(0u, 0u, 0u, 0u);
  for (int64_t i = i0; i < nt; i += step) touched[i] = 0u;
  // [0..3] records, entries, largest count, overflow; [4] first overflowed launch group: none; [5..9] short-form miss, blocks,
  // chunk visits, micro faces, records of a view (gr_raster_status)
  if (stats && i0 < 10) stats[i0] = i0 == 4 ? ~0ull : 0ull;
}

// (view, 64-face group) pairs the vote passes of the last fused call visited, added up for gr_raster_status (k_vote_labels keeps a
// slot per group: no contention there; this runs once per status call)
__global__ __launch_bounds__(256) void k_sum_visits(const uint32_t *__restrict__ visits, int64_t n, unsigned long long *__restrict__ out) {
  unsigned long long sum = 0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) sum += visits[i];
  for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
  if ((threadIdx.x & 63) == 0 && sum) atomicAdd(out, sum);
}

}  // namespace

namespace grimpl {

// stage 1 of a launch group: cull, set up and bin `nb` views (camera records `cams`) into scratch slots slot0..
int bin_batch(gr_ctx *c, const float *cams, int nb, int h, int w, int slot0, int group, hipStream_t s) {
  BinArgs a = make_args(c, h, w, slot0);
  a.group = group;
  {
    // ONE kernel zeroes what the group's passes count in -- the control words of its views, the fused path's group maps (bytes
    // the tile kernel's epilogue sets) and, in the call's first group, the call's statistics -- where rounds 1-5 issued four to five fills: a fill of 40 bytes
    // costs as much as a kernel launch (4.9 us each in the rocprof trace of a C2 step of 940 us; the step: -0.9 %, at quarter
    // scale -2 %: profiles/r05_ab/step_deferred_stats_vs_init_kernel_vs_fills.log, builds in rotated order).  (Folding k_bin_stats into the last k_clip_faces block of each view, to
    // save that launch too, measured no gain: one wave adding up a view's counters takes as long as the launch it saves --
    // step_clip_stats_merged*.)
    const int64_t n16 = c->ctrl_stride * nb / 4;   // ctrl_stride is a multiple of 64 words
    const int64_t nt = a.touched ? (int64_t)nb * a.tw : 0;
    const unsigned blocks = (unsigned)std::max<int64_t>(1, std::min<int64_t>(ceil_div(std::max(n16, nt), 256), 2048));
    unsigned long long *const st = c->stats_pending ? c->stats : nullptr;
    GR_LAUNCH_EV((hipEvent_t) nullptr, chain_begin(c), k_bin_init, dim3(blocks), dim3(256), 0, s, reinterpret_cast<uint4 *>(a.ctrl), n16, a.touched, nt, st);
    c->stats_pending = false;
  }
  if (a.dbg & 512) {  // test hook: every entry slot and row count starts as garbage (0xFF), like scratch that an earlier call left behind
    GR_HIP(c, hipMemsetAsync(a.comp, 0xFF, sizeof(int4) * GR_ENT_Q * (size_t)c->ent_cap * nb, s));
    GR_HIP(c, hipMemsetAsync(a.nrow8, 0xFF, (size_t)c->ent_cap * nb, s));
  }
  {
    // (the stage's span: from the end of k_bin_init to the end of k_clip_faces -- chain_stop)
    const int nblk = (int)ceil_div(c->F, GR_BLOCK);
    hipLaunchKernelGGL(k_cull_blocks, dim3((unsigned)ceil_div(nblk, 256), nb), dim3(256), 0, s,
                       cams, a, nblk);
    // about nblk / 32 waves per view take a few blocks each -- but never fewer than 16 k waves per launch, so that a call with
    // a few views still fills the machine.
    const int gmax = std::min((nblk + 3) / 4, 1024);
    const dim3 gsetup((unsigned)std::max(1, std::min(gmax, std::max(nblk / 128, 4096 / std::max(nb, 1)))), nb);
    if (a.cap_tile > 0) {
      if (a.micro) hipLaunchKernelGGL((k_setup_cull<true, true>), gsetup, dim3(256), 0, s, cams, a);
      else hipLaunchKernelGGL((k_setup_cull<true, false>), gsetup, dim3(256), 0, s, cams, a);
      GR_LAUNCH_EV((hipEvent_t) nullptr, chain_stop(c, ST_SETUP), k_clip_faces<true>, dim3(8, nb), dim3(64), 0, s, cams, a);
    } else {
      hipLaunchKernelGGL((k_setup_cull<false, false>), gsetup, dim3(256), 0, s, cams, a);
      GR_LAUNCH_EV((hipEvent_t) nullptr, chain_stop(c, ST_SETUP), k_clip_faces<false>, dim3(8, nb), dim3(64), 0, s, cams, a);
    }
  }
  c->last_direct = a.cap_tile > 0;
  if (a.cap_tile > 0) {
    // the view totals behind gr_raster_status.  Nothing on the device waits for them unless the call is fused (the vote kernel
    // skips overflowed groups), has more launch groups to come (they reuse the counters) or looks at its first group:
    // otherwise they are added up when -- if -- the status call asks (bin_stats_deferred), and a caller that runs unchecked
    // calls back to back (check=False: bench.py's timed loops) does not pay a launch per call for numbers nobody reads (the C2
    // step -1.5 %, at quarter scale -3.5 %: same log)
    if (c->defer_stats) {
      c->stats_deferred = true; c->deferred_args = a; c->deferred_nb = nb;   // (the arguments as they are: options may change before the status call)
    } else {
      GR_LAUNCH_EV((hipEvent_t) nullptr, chain_stop(c, ST_SCAN), k_bin_stats, dim3(nb), dim3(1024), 0, s, a);
    }
  } else {
    GR_LAUNCH_EV((hipEvent_t) nullptr, chain_stop(c, ST_SCAN), k_scan_tiles, dim3(nb), dim3(1024), 0, s, a);
    {
      const unsigned g = (unsigned)std::min<int64_t>(ceil_div(c->F, 256), 1024);
      GR_LAUNCH_EV((hipEvent_t) nullptr, chain_stop(c, ST_FILL), k_fill_compile, dim3(g, nb), dim3(256), 0, s, a);
    }
  }
  GR_HIP(c, hipGetLastError());
  return GR_OK;
}

// gr_raster_stats.chunk_visits of the last fused call: the vote passes' visit counters, summed into the call's statistics
int sum_visits(gr_ctx *c, hipStream_t s) {
  if (!c->visits_pending) return GR_OK;
  c->visits_pending = false;
  hipLaunchKernelGGL(k_sum_visits, dim3(64), dim3(256), 0, s, c->visits, ceil_div(c->F, 64), c->stats + 7);
  GR_HIP(c, hipGetLastError());
  return GR_OK;
}

// the deferred k_bin_stats of the last raster call (see bin_batch), on the stream of that call
int bin_stats_deferred(gr_ctx *c, hipStream_t s) {
  if (!c->stats_deferred) return GR_OK;
  c->stats_deferred = false;
  hipLaunchKernelGGL(k_bin_stats, dim3(c->deferred_nb), dim3(1024), 0, s, c->deferred_args);
  GR_HIP(c, hipGetLastError());
  return GR_OK;
}

}  // namespace grimpl
