// geograypher_amd/csrc/raster_tile.hip -- k_raster_tile, the dominant kernel: a 64 x 32 tile of depth|id keys in LDS, one
// scanline of one triangle per lane, ds_max_u64 resolve; ids or fused per-face winners out.
// Compile with -ffp-contract=off: every floating-point operation below is individually rounded on purpose.
#include "gr_internal.hpp"
#include "dev_common.hpp"

using namespace grimpl;

// EIGHT workgroups per CU (round 4): 20 480 bytes of LDS at most -- an eighth of the CU's --, at most 80 SGPRs (the hardware
// admits floor(800 / (ceil(sgpr / 16) * 16 + 16)) workgroups of 256 threads: 81 SGPRs would make it seven), at most 64 VGPRs.
//   40-byte entries: 64 x 32 keys with 3 padding keys per row (16.75 KiB) + mailboxes (0.5 KiB) + one chunk (2.5 KiB) = 20 224
//   48-byte entries: 2 padding keys per row (16.5 KiB) + mailboxes (0.5 KiB) + one chunk (3 KiB)                      = 20 480
// The closed-network model of DESIGN.md section 5 gives an eighth workgroup +2-3 % VALU utilisation; measured, builds
// alternated on one box (profiles/r04_ab/ab_wg8.log): ids kernel 13.85 -> 13.55 us per C2 view with 40-byte entries; the
// 48-byte kernels reached eight when the mailbox words went from 32 to 16 bits (raster_chunk_gather): hostile workload 119.1
// -> 114.9 us of tile kernel per view (profiles/r04_ab/mailbox16_forest.log).  The same 512 bytes gave the 40-byte kernels
// their third padding key: an ODD row stride is what the epilogues' 8-byte reads want (store_ids) -- fused kernel 15.8 ->
// 14.85 us per C2 view, 32.0 -> 30.4 on config 5; ids kernel unchanged; 4 keys: no better than 2 (profiles/r04_ab/lds_pad_mailbox16.log).
// (Round 2's 8-workgroup variant -- 4 padding keys, mailboxes inside the padding -- lost to the bank conflicts of the
// epilogue's dword reads, which are gone: DESIGN.md.)
#ifndef GR_WPE
#define GR_WPE 8       // waves per SIMD the 64 x 32 kernels ask the compiler for (what their LDS allows)
#endif
#ifndef GR_NUM_SGPR
#define GR_NUM_SGPR 80
#endif
#ifndef GR_ROLL_KT
#define GR_ROLL_KT 16   // tiles per workgroup of the rolling-chain kernel
#endif

namespace {

#ifdef GR_STAMPS
// diagnostic build: work-item statistics -- [0] items, [1] items whose span is empty, [2] pixels drawn, [3] batches
// (1024 slots: same-address atomics are served one per 11 ns)
__device__ unsigned long long g_item_stats[1024][4];
#endif

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
//     What bounds it: VALU issue (81-85 % of the SIMD cycles), then the LDS pipe (62-67 %); 8 workgroups
//     fit a CU (20 KiB of LDS each).
//     Epilogues: ids -> 16-byte stores (4 pixels per lane); fused projection -> per-face winners (see fused_winners).
// ------------------------------------------------------------------------------------------------------------------
// LDS image of a tile: rows of TW keys padded by PAD keys (stride 67 keys = 536 B with 40-byte entries, 66 with 48-byte ones).
// The rows of one triangle walk their spans in step; with a row offset of p key-banks a pile-up on one bank needs a left edge
// that recedes p px per row (a pad of 1 piled up every 45-degree edge: 530 of 1820 LDS cycles per tile were bank conflicts;
// rounds 1-3 used 5; 3 / 5 / 7 measured alike, 9 and 11 worse: DESIGN.md), and a pixel's address advances by a plain +8 bytes
// along the scanline (no wrap arithmetic in the inner loop).  3 and 2 are what leaves room for an eighth workgroup per CU (above).
#ifndef GR_LDS_PAD
#define GR_LDS_PAD 3       // kernels with 40-byte entries
#endif
#ifndef GR_LDS_PAD48
#define GR_LDS_PAD48 2     // kernels with 48-byte entries
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
#ifdef GR_STAMPS
  {
    const unsigned long long ml = __ballot(live), me = __ballot(live && xs > xe);
    int px = (live && xs <= xe) ? xe - xs + 1 : 0;
    for (int o = 32; o > 0; o >>= 1) px += __shfl_xor(px, o);
    if ((threadIdx.x & 63) == 0) {
      unsigned long long *st = g_item_stats[(blockIdx.x * 4 + (threadIdx.x >> 6) + blockIdx.y * 977) & 1023];
      atomicAdd(&st[0], (unsigned long long)__popcll(ml));
      atomicAdd(&st[1], (unsigned long long)__popcll(me));
      atomicAdd(&st[2], (unsigned long long)px);
      atomicAdd(&st[3], 1ull);
    }
  }
#endif
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

// MICRO records (round 5: lists; round 6: lean records): one lane = one record = a box of at most 4 x 4 pixels of one face.  K1
// sends a (face, tile) pair whose part of the pixel bounding box in the tile is at most 8 x 8 pixels to the tile's second list
// (binning.hip), one record per 4 x 4 box; such a face is not worth scanline items -- three rows of two pixels at render_img_scale
// 0.25, each paying the span solver's three reciprocals, the mailbox look-up and its share of two barriers per chunk.  The record
// holds what R2 - R4 produced and nothing derived from it: the three snapped vertices as 16-bit offsets from the centre of the
// tile's centre pixel (positive orientation, vertex 0 first), 1/z at vertex 0 and its gradients, ~face, and the box.  Here the lane
// forms the three edge functions of R3 at the box's first pixel, exactly, in int32 (|d| < 24 000, |P - V| < 41 000: products
// below 2^30) -- E_k = dx_k (P_y - Y_k) - dy_k (P_x - X_k) + t_k with the tie rule of R3 (left and bottom edges own their pixels) as the bias t_k, covered <=> all E_k >= 0 --
// and point-samples the sixteen pixel centres (three adds per pixel) with the same 1/z expression as raster_item, op for op (R4):
// identical coverage and identical depth bits, whichever list a face was put in.  Rows and columns outside the tile can only come
// from a torn record of an overflowed pass (the view is repeated): they are masked, never written.
template <int TWL, int TH, int PAD>
__device__ __forceinline__ void micro_item(unsigned long long *keys, const int4 ea, const int4 eb, const bool valid) {
  constexpr int TW = 1 << TWL;
  const int X0 = __builtin_amdgcn_sbfe(ea.x, 0, 16), Y0 = ea.x >> 16, X1 = __builtin_amdgcn_sbfe(ea.y, 0, 16), Y1 = ea.y >> 16;
  const int X2 = __builtin_amdgcn_sbfe(ea.z, 0, 16), Y2 = ea.z >> 16;
  const float iz0 = __int_as_float(ea.w), A = __int_as_float(eb.x), B = __int_as_float(eb.y);
  const uint32_t key = (uint32_t)eb.z;
  const int box = eb.w;
  const int xc0 = (box & 63) - TW / 2, last_col = (box >> 6) & 3;          // centred first column; columns - 1
  const int yc0 = ((box >> 8) & 63) - TH / 2;                              // centred first row
  const int nrows = valid ? min((box >> 14) & 7, 4) : 0;
  const int dx0 = X1 - X0, dy0 = Y1 - Y0, dx1 = X2 - X1, dy1 = Y2 - Y1, dx2 = X0 - X2, dy2 = Y0 - Y2;
  const int t0 = ((dy0 < 0) || (dy0 == 0 && dx0 < 0)) ? 0 : -1;           // R3 tie rule as a bias (left and bottom edges own)
  const int t1 = ((dy1 < 0) || (dy1 == 0 && dx1 < 0)) ? 0 : -1;
  const int t2 = ((dy2 < 0) || (dy2 == 0 && dx2 < 0)) ? 0 : -1;
  const int Px = xc0 * 256, Py = yc0 * 256;                                // the box's first pixel centre, same frame as the vertices
  int r0 = __mul24(dx0, Py - Y0) - __mul24(dy0, Px - X0) + t0;             // edge values at that pixel
  int r1 = __mul24(dx1, Py - Y1) - __mul24(dy1, Px - X1) + t1;
  int r2 = __mul24(dx2, Py - Y2) - __mul24(dy2, Px - X2) + t2;
  const int ax0 = -dy0 * 256, ax1 = -dy1 * 256, ax2 = -dy2 * 256;         // a step of one pixel along the row
  const int ay0 = dx0 * 256, ay1 = dx1 * 256, ay2 = dx2 * 256;            // ... down the column
  const float fx0 = (float)(Px - X0);
#pragma unroll 1   // (unrolled, the sixteen pixels' temporaries cost the chain kernels their 64-register budget)
  for (int r = 0; r < 4; ++r) {
    const bool rl = r < nrows;
    if (!__ballot(rl)) break;
    const int yc = yc0 + r;
    int e0 = r0, e1 = r1, e2 = r2;
    // a fourth "edge": columns left in the box (negative beyond it, and for a row this lane does not have or that lies outside the tile)
    int e3 = (rl && (uint32_t)(yc + TH / 2) < (uint32_t)TH) ? last_col : -1;
    const float m1 = B * (float)(yc * 256 - Y0);
    unsigned long long *row = keys + (__mul24(yc + TH / 2, TW + PAD) + xc0 + TW / 2);
    float fx = fx0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const bool in = ((e0 | e1) | (e2 | e3)) >= 0;
      float t = A * fx;
      t = m1 + t;
      const float z = iz0 + t;
      const int zb = max(__float_as_int(z), 1);
      if (in) atomicMax(row + j, ((unsigned long long)(uint32_t)zb << 32) | key);
      e0 += ax0; e1 += ax1; e2 += ax2; e3 -= 1;
      fx += 256.0f;
    }
    r0 += ay0; r1 += ay1; r2 += ay2;
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
                                                   const int first_b) {
  char *const lds = reinterpret_cast<char *>(keys);
  const int incl = wave_incl_scan(nrows);
  int total = __builtin_amdgcn_readlane(incl, 63);
  const int excl = incl - nrows;
  for (int k0 = first_b * 64; k0 < total; k0 += 64 * NW) {
    const int q = k0 + lane;
    const int slot = excl - k0;
    // 16-bit mailbox words (512 bytes of mailboxes: with 1 KiB of them the 48-byte kernels did not fit eight workgroups on a
    // CU, and the 40-byte ones had no room for a third padding key).  A 4-bit generation: when it wraps -- and at the tile's
    // first batch -- the wave wipes its mailbox
    gen += 1u << 12;
    if (!(gen & 0xF000u)) {
      *reinterpret_cast<uint16_t *>(lds + tab_self) = (uint16_t)0;
      gen = 1u << 12;
    }
    if (nrows > 0 && slot >= 0 && slot < 64)
      *reinterpret_cast<uint16_t *>(lds + tab_base + slot * 2) = (uint16_t)(gen | (uint32_t)(lane << 6) | (uint32_t)slot);
    const uint32_t m = wave_incl_max((uint32_t)*reinterpret_cast<const uint16_t *>(lds + tab_self));
    const int carry_t = __popcll(__ballot(incl <= k0));  // the entry that holds item k0: it exists (k0 < total), <= 63
    const int carry_r = k0 - __builtin_amdgcn_readlane(excl, carry_t);  // row of item k0 within that entry
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
// Stores to the output images are NON-TEMPORAL (global_store ... nt): an image is written once and never read by this kernel, and
// as ordinary stores its dirty lines sat in L2 and the 256 MB Infinity Cache until the NEXT kernel's traffic pushed them out -- the
// set-up kernel of the following launch group found the mesh evicted and paid for the write-back (a build of the tile kernel that
// stores nothing made the SET-UP 15 % faster: profiles/r06_ab/tile_removal_probes.log).  Round 6, builds alternated on one box: C2
// set-up 4.74 -> 4.23 us per view, ids kernel unchanged; config 5 set-up -6 %, ids kernel -7 %; C2 at 1000 x 750 ids kernel -13 %
// (profiles/r06_ab/nontemporal_image_stores.log).
typedef int gr_v4i __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store_ids16(int32_t *dst, const int4 v) {
  gr_v4i x = {v.x, v.y, v.z, v.w};
  __builtin_nontemporal_store(x, reinterpret_cast<gr_v4i *>(dst));
}
template <typename T>
__device__ __forceinline__ void store_px(T *dst, const T v) { __builtin_nontemporal_store(v, dst); }

template <int TWL, int TH, int NT, int PAD, bool PLAIN>
__device__ __forceinline__ void store_ids(const unsigned long long *keys, const BinArgs &a, int32_t *ids_plane, int te,
                                          int px0, int py0) {
  // (exchanging every key with the empty one here -- ds_wrxchg_rtn_b64, so that the workgroup's next tile needs no fill --
  // was measured: returning LDS atomics are slow, 17.0 vs 15.6 us per C2 view)
  const uint32_t *klo = reinterpret_cast<const uint32_t *>(keys);
  const int rows_here = min(TH, a.h - py0);
  const bool vec = PLAIN || (((a.w & 3) == 0) && ((reinterpret_cast<uintptr_t>(ids_plane) & 15) == 0));
  if (vec) {
    const int c4 = (te & 15) * 4, rr = te >> 4;
    const int gx4 = px0 + c4;
    if (gx4 >= a.w) return;
    int32_t *dst = ids_plane + (int64_t)(py0 + rr) * a.w + gx4;
    const int64_t dstep = (int64_t)(NT / 16) * a.w;
    // Whole keys through ds_read_b64.  The four low dwords of a lane's pixels sit 8 bytes apart and the 16 lanes of a row
    // 32 bytes apart: as ds_read2_b32 (32 banks of 4 bytes; low dwords only ever touch the 16 even ones) every access is a
    // 4-way bank conflict, 32 LDS cycles per wave and pass.  ds_read_b64 uses all 64 banks, two per lane, in groups of 32
    // lanes (two tile rows, an odd number of keys apart: even and odd key slots): 2-way, 16 cycles.  (Conflict-FREE, 8
    // cycles, is possible -- the upper half of a row's lanes reads its keys in the order 2, 3, 0, 1 -- at the price of four
    // v_cndmask per pass to put the ids back in order: measured -3.3 % on the ids kernel against the dword reads, but the
    // kernel's VALU pipes are the busier ones (84 % against 55 %): the plain order is the default.)
    const uint32_t kb = (uint32_t)(uintptr_t)reinterpret_cast<const char *>(keys);
    auto read4 = [&](int row, unsigned long long (&k)[4]) {
      const uint32_t base = kb + 8u * (uint32_t)lds_idx<TWL, PAD>(row, c4);
      asm volatile("ds_read_b64 %0, %4\n\tds_read_b64 %1, %4 offset:8\n\tds_read_b64 %2, %4 offset:16\n\tds_read_b64 %3, %4 offset:24"
                   : "=&v"(k[0]), "=&v"(k[1]), "=&v"(k[2]), "=&v"(k[3]) : "v"(base) : "memory");
    };
    auto ids_of = [&](const unsigned long long (&k)[4]) {
      return make_int4((int)~(uint32_t)k[0], (int)~(uint32_t)k[1], (int)~(uint32_t)k[2], (int)~(uint32_t)k[3]);
    };
    if (TH == 2 * (NT / 16) && rows_here == TH) {  // a whole tile: both passes' reads in flight together, ONE wait
      unsigned long long k0[4], k1[4];
      read4(rr, k0);
      read4(rr + NT / 16, k1);
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(k0[0]), "+v"(k0[1]), "+v"(k0[2]), "+v"(k0[3]), "+v"(k1[0]), "+v"(k1[1]), "+v"(k1[2]), "+v"(k1[3]) : : "memory");
      store_ids16(dst, ids_of(k0));
      store_ids16(dst + dstep, ids_of(k1));
    } else {
      for (int row = rr; row < rows_here; row += NT / 16, dst += dstep) {
        unsigned long long k[4];
        read4(row, k);
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(k[0]), "+v"(k[1]), "+v"(k[2]), "+v"(k[3]) : : "memory");
        store_ids16(dst, ids_of(k));
      }
    }
  } else if (!PLAIN) {
    constexpr int TW = 1 << TWL;
    const int col = te & (TW - 1), gx = px0 + col;
    if (gx >= a.w) return;
    int32_t *dst = ids_plane + (int64_t)(py0 + (te >> TWL)) * a.w + gx;
    const int64_t dstep = (int64_t)(NT / TW) * a.w;
    for (int row = te >> TWL; row < rows_here; row += NT / TW, dst += dstep) store_px(dst, (int32_t)~klo[2 * lds_idx<TWL, PAD>(row, col)]);
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
                                              uint8_t *__restrict__ mark, int te, int px0, int py0) {
  static_assert(TWL == 6 && NT == 256 && TH % 32 == 0, "16 lanes x 4 pixels per tile row, 16 row pairs per pass");
  const int c4 = (te & 15) * 4, rp = te >> 4;  // row pair 0 .. 15 of a pass
#pragma unroll
  for (int pass = 0; pass < TH / 32; ++pass) {
    const int r0 = 2 * rp + 32 * pass;          // rows r0, r0 + 1; the row below them is r0 + 2
    int c[3][6];                                // c[k][j + 1]: raw key of row r0 + k, column c4 + j, j = -1 .. 4
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      if (k < 2 || r0 + 2 < TH) {
        // whole keys through ds_read_b64 (see store_ids): the 4-way bank conflict of the dword reads becomes a 2-way one
        const uint32_t kbase = (uint32_t)(uintptr_t)(reinterpret_cast<const char *>(keys) + 8 * lds_idx<TWL, PAD>(r0 + k, c4));
        unsigned long long k0, k1, k2, k3;
        asm volatile("ds_read_b64 %0, %4\n\tds_read_b64 %1, %4 offset:8\n\tds_read_b64 %2, %4 offset:16\n\tds_read_b64 %3, %4 offset:24\n\t"
                     "s_waitcnt lgkmcnt(0)"
                     : "=&v"(k0), "=&v"(k1), "=&v"(k2), "=&v"(k3) : "v"(kbase) : "memory");
        c[k][1] = (int)(uint32_t)k0; c[k][2] = (int)(uint32_t)k1; c[k][3] = (int)(uint32_t)k2; c[k][4] = (int)(uint32_t)k3;
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
        if (cand) {
          atomicMax(win + ~f, p1 + (uint32_t)(k * a.w + j));
          if (mark) mark[(uint32_t)~f >> 6] = 1;   // the face's group of 64 holds a winner in this view: what the vote pass reads
        }
      }
    }
  }
}

template <int TWL, int TH, int NT, int PAD, bool EDGE>
__device__ __forceinline__ void fused_winners_rows(const unsigned long long *keys, const BinArgs &a, uint32_t *__restrict__ win,
                                              uint8_t *__restrict__ mark, int te, int px0, int py0) {
  static_assert(TWL == 6 && NT == 256 && TH % 32 == 0, "a wave per 64-pixel tile row, TH / 4 consecutive rows per wave");
  // The same candidates with a LANE PER COLUMN (round 6; the kernels with micro lists use it): a wave owns TH / 4 consecutive rows of
  // the tile and reads them one row per instruction -- 64 consecutive keys, every key of the tile once (+ one row of the wave below)
  // --; the neighbours to the right and below-left / below-right come from DPP wave shifts (wave_shl:1 / wave_shr:1: GFX9 has them;
  // the lane at the end keeps `old` = 1, "unknown across the tile edge").  What it buys is the ATOMICS: the 64 lanes of one
  // instruction are 64 neighbouring pixels of a row -- in a view of 3-pixel faces some twenty faces next to each other, whose
  // winner words share a few cache lines -- where the form above spreads an instruction over four row pairs and every fourth pixel.
  // A view of micro faces has a candidate every 4.7 pixels and its fused kernel was winner traffic (probes: 3.55 us per C2 view at
  // 1000 x 750, 1.77 with the candidates computed and nothing written, 1.56 without the epilogue): 3.54 -> 2.23 us.  Full-size views
  // (a candidate every 57 pixels) are 2.5 % slower this way and keep the form above.  profiles/r06_ab/fused_epilogue.log
  constexpr int RW = TH / 4;
  const int lane = te & 63;
  const int wv = __builtin_amdgcn_readfirstlane(te >> 6);
  const uint32_t *klo = reinterpret_cast<const uint32_t *>(keys);
  const int gx = px0 + lane;
#pragma unroll
  for (int base = 0; base < RW; base += 8) {
    const int r0 = wv * RW + base;
    int c[9], cr[9], cl[9];   // raw low dwords (~face; 0: background) of rows r0 .. r0 + 8; their right / left neighbours
#pragma unroll
    for (int k = 0; k < 9; ++k) c[k] = (k < 8 || r0 + 8 < TH) ? (int)klo[2 * lds_idx<TWL, PAD>(min(r0 + k, TH - 1), lane)] : 1;
    if (r0 + 8 >= TH) c[8] = 1;   // the tile below: unknown
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      cr[k] = __builtin_amdgcn_update_dpp(1, c[k], 0x130 /* wave_shl:1: lane + 1 */, 0xf, 0xf, false);
      cl[k] = k > 0 ? __builtin_amdgcn_update_dpp(1, c[k], 0x138 /* wave_shr:1: lane - 1 */, 0xf, 0xf, false) : 1;
    }
    const uint32_t p1 = (uint32_t)((py0 + r0) * a.w + gx + 1);   // linear pixel index + 1 of the lane's pixel in row r0 (h, w <= 16384)
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int f = c[k];
      bool cand;
      if (EDGE) {   // outside the image: 2 ("differs"), like the rows and columns the old form never read
        const int gy = py0 + r0 + k;
        const bool below = gy + 1 < a.h, right = gx + 1 < a.w;
        cand = (f < 0) & (gx < a.w) & (gy < a.h) & (!below | ((c[k + 1] != f) & (cl[k + 1] != f))) &
               (!right | ((cr[k] != f) & (!below | (cr[k + 1] != f))));
      } else {
        cand = (f < 0) & (c[k + 1] != f) & (cl[k + 1] != f) & (cr[k] != f) & (cr[k + 1] != f);
      }
      if (cand) {
        atomicMax(win + ~f, p1 + (uint32_t)(k * a.w));
        if (mark) mark[(uint32_t)~f >> 6] = 1;   // the face's group of 64 holds a winner in this view: what the vote pass reads
      }
    }
  }
}

// the tile's entry list: count and first slot (single-pass binning: the tile's fixed segment; exact binning: the scan's offset)
__device__ __forceinline__ void tile_list(const BinArgs &a, const uint32_t *__restrict__ ctrl, int tile, uint32_t &cnt, int64_t &beg) {
  if (a.cap_tile > 0) {
    cnt = min(ctrl[GR_CTRL_HDR + cidx(a, tile)], (uint32_t)a.cap_tile);
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
#ifndef GR_PRIO_I
#define GR_PRIO_I 3   // scanline phase
#endif
#ifndef GR_PRIO_M
#define GR_PRIO_M 0   // fill, chunk loads, barriers
#endif
#ifndef GR_PRIO_E
#define GR_PRIO_E 0   // epilogue
#endif
#define GR_PRIO_MEM() __builtin_amdgcn_s_setprio(GR_PRIO_M)
#define GR_PRIO_ITEMS() __builtin_amdgcn_s_setprio(GR_PRIO_I)

// Diagnostic build only (-DGR_STAMPS, tools/tile_phases.py; the production library has none of this): every wave reads the
// shader clock (s_memtime) at the phase boundaries of a tile and adds the cycles of each phase to a per-wave accumulator;
// at the end of the kernel lane 0 of every wave adds them to a.stamps[phase] (and the wave count to a.stamps[15]).  A stamp
// waits for the wave's outstanding LDS operations (lgkmcnt): the latency of an operation is charged to the phase that issued it.
//   0 prologue (counters, chunk requests, the ONE wait)   1 key fill + chunk staging   2 barrier behind the fill
//   3 scanline items, first chunk   4 later chunks (barriers, loads, staging, items)   5 barrier behind the items
//   6 epilogue (key reads, id stores / winner atomics)   7 barrier between the tiles of a chain   8 empty-tile path
#ifdef GR_STAMPS
struct StampAcc { unsigned long long t, acc[9], t0, r0; };  // t0 / r0: shader clock and 100 MHz real-time clock at the wave's start
#define GR_STAMP_ARG , StampAcc &sa
#define GR_STAMP_PASS , sa
#define GR_STAMP(k) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); sa.acc[k] += t_ - sa.t; sa.t = t_; } while (0)
#else
#define GR_STAMP_ARG
#define GR_STAMP_PASS
#define GR_STAMP(k) do { } while (0)
#endif

// 16-byte piece q (0 .. 159) of a chunk that holds n (1 .. 64) entries in the short form: the front of the 32-byte parts or the
// front of the 8-byte parts (store_entry) -- is it needed?
__device__ __forceinline__ bool short_piece_needed(uint32_t q, uint32_t n) {
  return (q < 2 * n) | ((q >= 128) & (q < 128 + ((n + 1) >> 1)));  // no short-circuit: one predicate, one branch around the load
}

// One tile: keys in LDS -> chunks of entries -> scanline items -> epilogue.  nr_first / ex: the tile's first chunk (row
// counts and this lane's 16 bytes of the 3 KiB (2.5 KiB) of entries), requested by the caller -- and waited for by the caller
// (a chain), or here behind the fill of the key tile (WAIT: one tile per workgroup -- the request's latency overlaps the fill).
// PLAIN: the ids-only kernel of the usual call -- ids to an image whose rows take 16-byte stores, no depth image: the epilogue's
// other forms (depth, one pixel per lane) are not in the kernel at all (the cold paths cost the hot one registers and schedule)
// The four first-chunk requests a rolling chain keeps in flight (k_raster_tile_roll): whole 16-byte tuples + the row counts.
struct ChunkRing { v4i e0; uint32_t n0; };

template <int TWL, int THL, int NT, bool FUSE, int PAD, bool SHORT, bool WAIT, bool PLAIN, bool ROLL = false, bool MICRO = false>
__device__ __forceinline__ void raster_one_tile(const BinArgs &a, const RasterOut &out, unsigned long long *keys, const int slot,
                                                const int tile, uint32_t cnt, const int64_t beg, uint32_t nr_first, v4i ex GR_STAMP_ARG,
                                                ChunkRing *ring = nullptr, uint32_t cntm = 0u) {
  constexpr int TW = 1 << TWL, TH = 1 << THL;
  constexpr int NKEYS = (TW + PAD) * TH;
  constexpr int NW = NT / 64;
  constexpr int NMAIL = NW * 16;  // u64 units: 64 16-bit mailbox words per wave
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

  // the tile's MICRO list (single-pass binning with 40-byte entries): cntm entries from the back of the segment; the scanline
  // list must not reach into it (the two can only collide in a pass whose tile outgrew its segment: the view is repeated)
  if (!(SHORT && MICRO)) cntm = 0u;
  if (SHORT && MICRO) {   // both lists inside the segment (a view whose lists met is repeated: k_bin_stats)
    cntm = min(cntm, (uint32_t)a.cap_tile * 5u / 4u);
    cnt = min(cnt, (uint32_t)a.cap_tile);
  }
  if (!FUSE && cnt == 0 && cntm == 0) {  // empty tile (a view that overhangs the mesh): background, without the LDS round trip
    // (a rolling chain waits for its ring here as well, before the stores: a path that left the function with a request in
    // flight made the compiler wait for EVERYTHING -- the previous tile's stores included -- at the next tile's first request)
    if (ROLL) asm volatile("" : "+v"(ring->e0), "+v"(ring->n0));
    if (PLAIN) {  // 16-byte stores: a lane owns 4 consecutive pixels of a row, 16 rows per pass
      const int gx4 = px0 + (tid & 15) * 4;
      if (gx4 < a.w)
        for (int row = tid >> 4; row < TH && py0 + row < a.h; row += NT / 16)
          store_ids16(out.ids + plane + (int64_t)(py0 + row) * a.w + gx4, make_int4(-1, -1, -1, -1));
      GR_STAMP(8);
      return;
    }
    const int col = tid & (TW - 1), gx = px0 + col;
    if (gx < a.w) {
      for (int row = tid >> TWL; row < TH && py0 + row < a.h; row += ROWS_PER_PASS) {
        const int64_t p = plane + (int64_t)(py0 + row) * a.w + gx;
        if (out.ids) store_px(out.ids + p, (int32_t)-1);
        if (out.depth) store_px(out.depth + p, INFINITY);
      }
    }
    GR_STAMP(8);
    return;
  }
  const int tab_base = NKEYS * 8 + wv * 128;  // byte offset of the wave's 64 mailbox words, behind the keys
  const int tab_self = tab_base + lane * 2;
  {  // fill the tile (16-byte LDS stores): depth 0 | the id background stands for (the waves wipe their mailboxes themselves)
    const int bg = (FUSE && out.compat) ? (int)out.F - 1 : -1;
    const unsigned long long fill = (unsigned long long)(uint32_t)~bg;
    ulonglong2 *k2 = reinterpret_cast<ulonglong2 *>(keys);
#pragma unroll
    for (int i = 0; i < (NKEYS / 2 + NT - 1) / NT; ++i)
      if (i * NT + tid < NKEYS / 2) k2[i * NT + tid] = make_ulonglong2(fill, fill);
  }
  uint32_t gen = 0xF000u;  // the first batch finds the generation wrapped and wipes the wave's mailbox
  int rot = wv;  // this wave's first batch of the current chunk
  {  // first chunk: in registers already, complete (k_raster_tile waits for every request of the chain before its first
     // tile: a wait on the memory counter here would wait for the previous tile's stores)
    if (WAIT) asm volatile("" : "+v"(ex), "+v"(nr_first));
    if (!ROLL && lane < EL) ent_st[wv * EL + lane] = ex;   // a rolling chain staged it before it re-used the registers
    GR_STAMP(1);
    __syncthreads();  // keys filled, chunk visible
    GR_STAMP(2);
    GR_PRIO_ITEMS();
    // LATER CHUNKS ARE REQUESTED ONE CHUNK AHEAD (round 5).  A tile with more than 64 entries -- every tile of a down-scaled
    // or a hostile view: 3-6 chunks at 1000 x 750 on the terrain, dozens under the forest -- used to load each later chunk and
    // its row counts right where it staged them, behind the barrier that frees the chunk buffer: two exposed round trips
    // per chunk, 80 % (terrain at quarter scale) to 94 % (forest) of a wave's life (tools/tile_phases.py).  Now the request
    // for chunk c + 1 goes out before chunk c's items are walked and has that whole phase to arrive.  The row counts come as
    // the low byte of an unaligned dword load: a byte load is zero-extended at once, i.e. waited for at once.
    typedef uint32_t __attribute__((aligned(1))) u32_unaligned;
    auto request_chunk = [&](uint32_t c0, v4i &exn, uint32_t &nrn) {
      if (c0 >= cnt) return;
      nrn = *reinterpret_cast<const u32_unaligned *>(nr8 + c0 + lane);   // bytes behind the tile's entries: its segment / padding
      if (lane < EL) {
        const uint32_t qc = wv * EL + lane;  // piece of the chunk
        const uint32_t q = (SHORT ? (c0 >> 1) * 5 : c0 * GR_ENT_Q) + qc;
        if (SHORT ? short_piece_needed(qc, min(cnt - c0, 64u)) : q < cnt * GR_ENT_Q) exn = reinterpret_cast<const v4i *>(comp)[q];
      }
    };
    uint32_t nrn = 0u;
    request_chunk(64u, ex, nrn);
    const int nrows = (uint32_t)lane < cnt ? (int)nr_first : 0;
    const int nb = raster_chunk_gather<TWL, TH, NW, PAD, SHORT>(keys, tab_base, tab_self, gen, ent_lds, nrows, lane, rot);
    rot = (rot - nb) & (NW - 1);
    GR_STAMP(3);
#pragma unroll 1
    for (uint32_t c0 = 64; c0 < cnt; c0 += 64) {
      GR_PRIO_MEM();
      __syncthreads();  // every wave is done with the previous chunk before it is overwritten
      if (lane < EL) ent_st[wv * EL + lane] = ex;
      const uint32_t e = c0 + (uint32_t)lane;
      const int nrows_c = e < cnt ? (int)(nrn & 0xFFu) : 0;
      request_chunk(c0 + 64u, ex, nrn);   // into the registers just staged: in flight through this chunk's items
      __syncthreads();
      GR_PRIO_ITEMS();
      const int nbc = raster_chunk_gather<TWL, TH, NW, PAD, SHORT>(keys, tab_base, tab_self, gen, ent_lds, nrows_c, lane, rot);
      rot = (rot - nbc) & (NW - 1);
      GR_STAMP(4);
    }
  }
  if (SHORT && MICRO && cntm > 0) {
    // micro records: every wave reads ITS chunks of 64 straight from the segment (no staging, no barrier), one record per lane.
    // Record k of the list: the 32 bytes that end 32 k bytes before the end of the tile's segment -- a wave's chunk is 2 KiB of
    // consecutive memory.
    const char *seg_end = reinterpret_cast<const char *>(comp) + (size_t)a.cap_tile * 40;
#pragma unroll 1
    for (uint32_t c = (uint32_t)wv; c * 64u < cntm; c += NW) {
      const uint32_t k = c * 64u + (uint32_t)lane;
      const int4 *rec = reinterpret_cast<const int4 *>(seg_end - 32u * (min(k, cntm - 1u) + 1u));
      const int4 ea = rec[0], eb = rec[1];
      micro_item<TWL, TH, PAD>(keys, ea, eb, k < cntm);
    }
  }
  int te = tid;
  asm volatile("" : "+v"(te));  // the epilogue's addresses are derived here, not hoisted above the scanline phase
  GR_PRIO_MEM();
  // a rolling chain: the request for the NEXT tile (issued at this tile's start, an items phase ago) is waited for HERE,
  // before this tile's id stores are issued -- loads and stores share one in-order counter, and behind the stores the wait
  // would be a wait for them.  Named as whole register tuples: behind this statement nothing of the ring is in flight.
  if (ROLL) asm volatile("" : "+v"(ring->e0), "+v"(ring->n0));
  __syncthreads();              // keys complete
  if (GR_PRIO_E != GR_PRIO_M) __builtin_amdgcn_s_setprio(GR_PRIO_E);
  GR_STAMP(5);
  if (FUSE) {
    uint32_t *win = out.winner + slot * out.F;
    uint8_t *mark = out.touched ? out.touched + slot * out.tb : nullptr;
    const bool edge = px0 + TW > a.w || py0 + TH + 1 > a.h;
    if (MICRO) {   // views of micro faces: a lane per column (neighbouring faces' winner atomics in one instruction)
      if (edge) fused_winners_rows<TWL, TH, NT, PAD, true>(keys, a, win, mark, te, px0, py0);
      else fused_winners_rows<TWL, TH, NT, PAD, false>(keys, a, win, mark, te, px0, py0);
    } else if (edge) fused_winners<TWL, TH, NT, PAD, true>(keys, a, win, mark, te, px0, py0);
    else fused_winners<TWL, TH, NT, PAD, false>(keys, a, win, mark, te, px0, py0);
    if (out.ids) {  // the id image as well (rare): background is where no fragment landed (depth bits 0)
      const int col = te & (TW - 1), gx = px0 + col;
      if (gx < a.w)
        for (int row = te >> TWL; row < TH && py0 + row < a.h; row += ROWS_PER_PASS) {
          const unsigned long long key = keys[lds_idx<TWL, PAD>(row, col)];
          store_px(out.ids + plane + (int64_t)(py0 + row) * a.w + gx, (key >> 32) ? (int32_t)~(uint32_t)key : (int32_t)-1);
        }
    }
  } else if (PLAIN || (out.ids && !out.depth)) {
    store_ids<TWL, TH, NT, PAD, PLAIN>(keys, a, out.ids + plane, te, px0, py0);
  } else if (!PLAIN) {
    const int col = te & (TW - 1), gx = px0 + col;
    if (gx < a.w)
      for (int row = te >> TWL; row < TH && py0 + row < a.h; row += ROWS_PER_PASS) {
        const unsigned long long key = keys[lds_idx<TWL, PAD>(row, col)];
        const int64_t p = plane + (int64_t)(py0 + row) * a.w + gx;
        if (out.ids) store_px(out.ids + p, (int32_t)~(uint32_t)key);  // low dword = ~face, 0 when empty: ~0 = -1
        if (out.depth) store_px(out.depth + p, key ? 1.0f / __int_as_float((int)(key >> 32)) : INFINITY);
      }
  }
  if (GR_PRIO_E != GR_PRIO_M) __builtin_amdgcn_s_setprio(GR_PRIO_M);
  GR_STAMP(6);
}

// K3  the tile kernel.  KT = 4: a workgroup takes four consecutive tiles one after the other.  The four counts are read
//     first (scalar loads), then the first chunks of all four tiles are requested EXACTLY, together, and waited for together
//     before the first tile starts: one wait for memory per chain instead of four, no stale slots fetched.  (Waiting for
//     tile k's chunk only when tile k starts would wait for tile k - 1's id stores: loads and stores share one in-order
//     counter.)  KT = 1 -- heavy scenes, small launches --: the first chunk is requested before the count is known (the
//     segment address is static; slots beyond the count hold stale data that nobody reads).
// The 64 x 32 kernels ask the compiler for GR_WPE = 8 waves per SIMD -- what their LDS allows (round 3: 7; the schedule the
// compiler picked under that hint was 3-4 % faster than without, 15.8 -> 15.2 us per C2 view, builds alternated on one box); the
// fused kernel is not (left at the default).  Work items of two consecutive rows (look-up, unpack and the reciprocals paid
// once per two rows: -16 % VALU instructions) were measured as well: 74 VGPRs and half as many batches per tile for four
// waves -- 15.6 vs 16.0 without the hint, 16.3 vs 15.3 with it, fused 18.3 vs 17.2 -- dropped.
// MICRO: the kernels of a call that keeps micro lists (binning.hip; chosen per mesh and image size, gr_raster_status) -- a build of
// their own: the second list's counters and phase cost the ordinary kernels 6 of their 64 registers and 1-2.6 % on C2 even
// when every micro list is empty (profiles/r05_ab/micro_lists_whole_face.log).
template <int TWL, int THL, int NT, bool FUSE, int KT, int PAD, bool SHORT, bool PLAIN = false, bool MICRO = false>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(THL == 5 ? GR_WPE : (FUSE ? 1 : 4), 8)))
__attribute__((amdgpu_num_sgpr(GR_NUM_SGPR))) void k_raster_tile(BinArgs a, RasterOut out) {
  constexpr int TW = 1 << TWL, TH = 1 << THL;
  constexpr int NKEYS = (TW + PAD) * TH;
  constexpr int NW = NT / 64;
  constexpr int NMAIL = NW * 16;
  // the kernel's only LDS: keys + 0.5 KiB of mailboxes + one chunk of entries (2.5 or 3 KiB): 8 workgroups/CU for 64 x 32 (top)
  __shared__ __attribute__((aligned(16))) unsigned long long keys[NKEYS + NMAIL + 64 * (SHORT ? 5 : 6)];
  static_assert(NT == 256, "the entry copy deals 48 int4 to each of 4 waves");
  static_assert(NKEYS % 2 == 0 && TH % 32 == 0, "key pairs; two 16-row passes per fused group");
  static_assert(KT == 1 || KT == 4, "one tile per workgroup, or a chain of four");
  const int slot = blockIdx.y;
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
#ifdef GR_STAMPS
  StampAcc sa;
  for (int k = 0; k < 9; ++k) sa.acc[k] = 0;
  sa.t = sa.t0 = __builtin_amdgcn_s_memtime();
  sa.r0 = __builtin_amdgcn_s_memrealtime();
#endif
  // one tile per workgroup: the first chunk is requested before the count is known (one round trip less).  A chain waits
  // for the exact requests of its tiles 1 - 3 anyway before it starts: requesting its first tile's chunk early saves
  // nothing there (14.9 us per C2 view either way) and fetches 1.6 MB of stale slots per view -- not done
  const bool spec = KT == 1 && a.cap_tile >= 64;
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
  // fused projection: a view whose binning did not finish (a tile outgrew its segment, a face missed the 40-byte form) is
  // repeated by the caller and its launch group does not vote -- its winners are not wanted, and a list with a hole in it is
  // not walked at all (one scalar load beside the counters')
  if (FUSE && ctrl[2] != 0u) return;
  uint32_t cnt0, cnt1 = 0, cnt2 = 0, cnt3 = 0;
  uint32_t cm0 = 0, cm1 = 0, cm2 = 0, cm3 = 0;   // the tiles' micro lists (counted in the cntB array: binning.hip)
  int64_t beg0, beg1 = 0, beg2 = 0, beg3 = 0;
  if (SHORT && MICRO) {
    if (KT == 4) {
      uint4 m4;
      if (a.clg < 0) m4 = *reinterpret_cast<const uint4 *>(ctrl + GR_CTRL_HDR + a.Tcap + tile0);
      else {   // counters spread over lines (ensure_bins); the three tiles behind the image's last have counters too (zero)
        const uint32_t *m = ctrl + GR_CTRL_HDR + a.Tcap;
        m4 = make_uint4(m[cidx(a, tile0)], m[cidx(a, tile0 + 1)], m[cidx(a, tile0 + 2)], m[cidx(a, tile0 + 3)]);
      }
      cm0 = m4.x; cm1 = n_tiles > 1 ? m4.y : 0u; cm2 = n_tiles > 2 ? m4.z : 0u; cm3 = n_tiles > 3 ? m4.w : 0u;
    } else cm0 = ctrl[GR_CTRL_HDR + a.Tcap + cidx(a, tile0)];
  }
  if (KT == 4 && a.cap_tile > 0) {
    // single-pass binning: the chain's four counters sit side by side, 16-byte aligned -- ONE scalar load instead of four
    // dependent ones, each behind its own wait (words behind the last tile's belong to the next counter array: valid memory)
    uint4 c4;
    if (a.clg < 0) c4 = *reinterpret_cast<const uint4 *>(ctrl + GR_CTRL_HDR + tile0);
    else {
      const uint32_t *cp = ctrl + GR_CTRL_HDR;
      c4 = make_uint4(cp[cidx(a, tile0)], cp[cidx(a, tile0 + 1)], cp[cidx(a, tile0 + 2)], cp[cidx(a, tile0 + 3)]);
    }
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
  GR_STAMP(0);
  static_assert(!(FUSE && PLAIN), "the plain kernel writes ids only");
  raster_one_tile<TWL, THL, NT, FUSE, PAD, SHORT, KT == 1, PLAIN, false, MICRO>(a, out, keys, slot, tile0, cnt0, beg0, nr0, ex0 GR_STAMP_PASS, nullptr, cm0);
  if (KT > 1) {
#pragma unroll 1
    for (int k = 1; k < n_tiles; ++k) {  // ONE copy of the tile code for tiles 1 .. 3: the chunks rotate through ex1
      __syncthreads();                   // every wave has read the previous tile's keys
      GR_STAMP(7);
      raster_one_tile<TWL, THL, NT, FUSE, PAD, SHORT, false, PLAIN, false, MICRO>(a, out, keys, slot, tile0 + k, cnt1, beg1, nr1, ex1 GR_STAMP_PASS, nullptr, cm1);
      cnt1 = cnt2; cnt2 = cnt3; beg1 = beg2; beg2 = beg3; cm1 = cm2; cm2 = cm3;
      nr1 = nr2; nr2 = nr3; ex1 = ex2; ex2 = ex3;
    }
  }
#ifdef GR_STAMPS
  if (lane == 0 && a.stamps) {  // 1024 slots of 16 words: same-address atomics are served one per 11 ns
    unsigned long long *st = a.stamps + 16 * ((blockIdx.x * 4 + wv + blockIdx.y * 977) & 1023);
    for (int k = 0; k < 9; ++k) atomicAdd(&st[k], sa.acc[k]);
    atomicAdd(&st[12], __builtin_amdgcn_s_memtime() - sa.t0);      // wave lifetime in shader cycles ...
    atomicAdd(&st[13], __builtin_amdgcn_s_memrealtime() - sa.r0);  // ... and in 10 ns ticks: their ratio x 100 MHz is the clock under load
    atomicAdd(&st[15], 1ull);                 // waves
    atomicAdd(&st[14], (unsigned long long)n_tiles);  // tile visits x waves
  }
#endif
}

// K3r  ROLLING chains (round 5).  A chain of four pays its prologue -- two dependent round trips: the tiles' counters, then
//      their first chunks -- once per four tiles, and doubling that prologue costs C2 3.9 % and C5 7.0 % of the kernel
//      (profiles/r05_ab/tile_prologue_doubled.log): that is what hiding it can buy.  Here a workgroup takes KTL consecutive
//      tiles and keeps ONE first-chunk request in flight all the time (five live registers; a chain of four holds fifteen
//      through its first tile): the prologue asks for tile 0; every tile copies its chunk from the request registers to LDS
//      and at once re-uses them for the NEXT tile's request (all the chain's counters were read in the prologue), which has
//      the tile's whole items phase to arrive and is waited for before the tile's id stores go out (raster_one_tile<ROLL>).
//      The exposed prologue is paid once per KTL tiles.  Single-pass binning only (the counters of consecutive tiles sit side
//      by side).
template <int TWL, int THL, int NT, bool FUSE, int PAD, bool SHORT, bool PLAIN, int KTL, bool MICRO = false>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(THL == 5 ? GR_WPE : (FUSE ? 1 : 4), 8)))
__attribute__((amdgpu_num_sgpr(GR_NUM_SGPR))) void k_raster_tile_roll(BinArgs a, RasterOut out) {
  constexpr int TW = 1 << TWL, TH = 1 << THL;
  constexpr int NKEYS = (TW + PAD) * TH;
  constexpr int NW = NT / 64;
  constexpr int NMAIL = NW * 16;
  __shared__ __attribute__((aligned(16))) unsigned long long keys[NKEYS + NMAIL + 64 * (SHORT ? 5 : 6)];
  static_assert(NT == 256 && KTL >= 2 && KTL <= 63, "one lane per counter, one more reads as zero");
  const int slot = blockIdx.y;
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
#ifdef GR_STAMPS
  StampAcc sa;
  for (int k = 0; k < 9; ++k) sa.acc[k] = 0;
  sa.t = sa.t0 = __builtin_amdgcn_s_memtime();
  sa.r0 = __builtin_amdgcn_s_memrealtime();
#endif
  const int tile0 = KTL * (int)blockIdx.x;
  const int n_tiles = min(KTL, a.T - tile0);
  constexpr int EL = SHORT ? 40 : 48;
  const uint32_t q = wv * EL + lane;
  const v4i *base = reinterpret_cast<const v4i *>(a.comp + (int64_t)slot * a.ent_cap * GR_ENT_Q);
  auto needed = [](uint32_t qq, uint32_t n) { return SHORT ? short_piece_needed(qq, min(n, 64u)) : qq < n * GR_ENT_Q; };
  const uint32_t *ctrl = a.ctrl + slot * a.ctrl_stride;
  if (FUSE && ctrl[2] != 0u) return;
  const uint32_t cap = (uint32_t)a.cap_tile;
  const int64_t sbase = slot * a.ent_cap;
  // the chain's counters: lane k holds tile k's (ONE vector register for the whole chain, read with v_readlane: no counter is
  // loaded inside the tile loop, where a load is a wait)
  const uint32_t *cntp = ctrl + GR_CTRL_HDR;
  uint32_t cvec = lane < n_tiles ? min(cntp[cidx(a, tile0 + lane)], cap) : 0u;
  uint32_t mvec = (SHORT && MICRO && lane < n_tiles) ? cntp[a.Tcap + cidx(a, tile0 + lane)] : 0u;   // the tiles' micro lists (cntB array)
  asm volatile("" : "+v"(cvec), "+v"(mvec));
  auto count_at = [&](int tt) { return (uint32_t)__builtin_amdgcn_readlane((int)cvec, tt); };   // tt < 64; lanes >= n_tiles hold 0
  ChunkRing r;
  r.n0 = 0u;
  typedef uint32_t __attribute__((aligned(1))) u32_unaligned;
  auto request = [&](uint32_t cnt, int64_t beg, v4i &e, uint32_t &n) {
    // The lane's row count as the low byte of an (unaligned) DWORD load, unconditionally (the bytes read lie in the tile's
    // segment or the 64 bytes of padding behind the array; entries beyond the count are never looked at).  Nothing may touch
    // the loaded register before the request is waited for: a byte load is followed by its zero extension at once, a register
    // that is zeroed and then loaded under a mask by a wait for every store in flight -- each a full wait at the tile's start.
    n = *reinterpret_cast<const u32_unaligned *>(a.nrow8 + sbase + beg + lane);
    const char *seg = reinterpret_cast<const char *>(base) + beg * (SHORT ? 40 : 48);
    if ((lane < EL) & needed(q, cnt)) e = *reinterpret_cast<const v4i *>(seg + (q << 4));
  };
  const int64_t beg0 = (int64_t)tile0 * a.cap_tile;
  request(count_at(0), beg0, r.e0, r.n0);
  asm volatile("" : "+v"(r.e0), "+v"(r.n0));
  GR_STAMP(0);
  v4i *ent_st = reinterpret_cast<v4i *>(keys + NKEYS + NMAIL);
#pragma unroll 1
  for (int t = 0; t < n_tiles; ++t) {
    if (t > 0) {
      __syncthreads();                   // every wave has read the previous tile's keys (and is done with its last chunk)
      GR_STAMP(7);
    }
    // stage the tile's first chunk from the request registers and re-use them at once for the NEXT tile's request
    const uint32_t cnt = count_at(t), cnt1 = count_at(t + 1);
    const int64_t beg = (int64_t)(tile0 + t) * a.cap_tile;
    if (lane < EL) ent_st[wv * EL + lane] = r.e0;
    const uint32_t nr_first = r.n0 & 0xFFu;
    request(cnt1, beg + a.cap_tile, r.e0, r.n0);
    // (the by-value chunk argument is the later chunks' temporary: NOT the request register, which is in flight again)
    raster_one_tile<TWL, THL, NT, FUSE, PAD, SHORT, false, PLAIN, true, MICRO>(a, out, keys, slot, tile0 + t, cnt, beg, nr_first, v4i{0, 0, 0, 0} GR_STAMP_PASS, &r,
                                                                                 (uint32_t)__builtin_amdgcn_readlane((int)mvec, t));
  }
#ifdef GR_STAMPS
  if (lane == 0 && a.stamps) {
    unsigned long long *st = a.stamps + 16 * ((blockIdx.x * 4 + wv + blockIdx.y * 977) & 1023);
    for (int kk = 0; kk < 9; ++kk) atomicAdd(&st[kk], sa.acc[kk]);
    atomicAdd(&st[12], __builtin_amdgcn_s_memtime() - sa.t0);
    atomicAdd(&st[13], __builtin_amdgcn_s_memrealtime() - sa.r0);
    atomicAdd(&st[15], 1ull);
    atomicAdd(&st[14], (unsigned long long)n_tiles);
  }
#endif
}

}  // namespace

#ifdef GR_STAMPS
extern "C" int gr_debug_read_item_stats(unsigned long long *out4_h) {  // diagnostic build: read and clear (tools/tile_phases.py)
  if (hipDeviceSynchronize() != hipSuccess) return GR_EHIP;
  static unsigned long long all[1024][4];
  if (hipMemcpyFromSymbol(all, HIP_SYMBOL(g_item_stats), sizeof(all)) != hipSuccess) return GR_EHIP;
  for (int k = 0; k < 4; ++k) out4_h[k] = 0;
  for (int i = 0; i < 1024; ++i)
    for (int k = 0; k < 4; ++k) out4_h[k] += all[i][k];
  memset(all, 0, sizeof(all));
  if (hipMemcpyToSymbol(HIP_SYMBOL(g_item_stats), all, sizeof(all)) != hipSuccess) return GR_EHIP;
  return GR_OK;
}
#endif

namespace grimpl {

// stage 2: rasterize the binned views of scratch slots slot0.. into out (already offset to the group's first view)
int tile_batch(gr_ctx *c, int nb, int h, int w, int slot0, RasterOut out, hipStream_t s) {
  BinArgs a = make_args(c, h, w, slot0);
  {
    // the kernel's span: from the end of the kernel in front of it (the chain of launch-attached stop events bin_batch began)
    // to its own end; a call whose bin pass left no chain: recorded events
    hipEvent_t ev_stop = chain_stop(c, ST_RASTER);
    Timed t(c, s, ev_stop ? -1 : ST_RASTER);
    const hipEvent_t ev_none = nullptr;
    // Four consecutive tiles per workgroup -- unless the image needed more than the default 512 slots per tile (a scene
    // with heavy tiles: chains of them make a few workgroups very long; hostile workload 57.9 vs 30.4 us per view at
    // 1000x750) or the launch has too few tiles to keep every CU busy with chains.
    // (A view of micro faces has large segments -- hundreds of 32-byte records per tile -- but uniform, light tiles: its ids kernel
    // takes chains as well, 2.52 -> 2.42 us per C2 view at 1000 x 750; its FUSED kernel does not: rolling chains of 16 leave a
    // 1000 x 750 launch with 24 workgroups per view, 3.56 -> 4.13 us -- profiles/r06_ab/micro_chains.log.)
    const bool light = a.cap_tile > 0 && (a.cap_tile <= 512 || (a.micro && out.winner == nullptr));
    const bool chain = (a.var & 1) == 0 && ((a.var & 16) != 0 || (light && (int64_t)a.T * nb >= 16384));
    // Rolling chains of GR_ROLL_KT tiles (k_raster_tile_roll) where a chain of four would run -- for the FUSED kernel, whose
    // epilogue stores nothing: C2 15.07 -> 14.22 us per view, C5 29.99 -> 28.67 (profiles/r05_ab/rolling_chains.log).  The ids
    // kernels lose with them (C2 13.21 -> 13.94, C5 28.5 -> 33.3): the wait for the next tile's request, placed before the
    // tile's id stores, is also a wait for the PREVIOUS tile's stores (one in-order counter), so a wave never has more than
    // one tile's stores in flight -- a chain of four has up to four.  They keep chains of four.
    const bool roll = chain && a.cap_tile > 0 && a.thl == 5 && out.winner != nullptr;
    const dim3 grid(roll ? (unsigned)((a.T + GR_ROLL_KT - 1) / GR_ROLL_KT) : chain ? (unsigned)((a.T + 3) >> 2) : (unsigned)a.T, nb), block(256);
    const size_t pad = (size_t)c->opt_lds_pad;
#define GR_LAUNCH_TILE_M(THL_, FUSE_, PLAIN_, MICRO_)                                                                 \
  do {                                                                                                                \
    if (a.ent40) {                                                                                                    \
      if (chain) GR_LAUNCH_EV(ev_none, ev_stop, (k_raster_tile<6, THL_, 256, FUSE_, 4, GR_LDS_PAD, true, PLAIN_, MICRO_>), grid, block, pad, s, a, out);  \
      else GR_LAUNCH_EV(ev_none, ev_stop, (k_raster_tile<6, THL_, 256, FUSE_, 1, GR_LDS_PAD, true, PLAIN_, MICRO_>), grid, block, pad, s, a, out);        \
    } else if (chain) GR_LAUNCH_EV(ev_none, ev_stop, (k_raster_tile<6, THL_, 256, FUSE_, 4, GR_LDS_PAD48, false, PLAIN_, false>), grid, block, pad, s, a, out);  \
    else GR_LAUNCH_EV(ev_none, ev_stop, (k_raster_tile<6, THL_, 256, FUSE_, 1, GR_LDS_PAD48, false, PLAIN_, false>), grid, block, pad, s, a, out);        \
  } while (0)
#define GR_LAUNCH_TILE(THL_, FUSE_, PLAIN_)                                                                           \
  do {                                                                                                                \
    if (a.micro) GR_LAUNCH_TILE_M(THL_, FUSE_, PLAIN_, true);                                                         \
    else GR_LAUNCH_TILE_M(THL_, FUSE_, PLAIN_, false);                                                                \
  } while (0)
    // the usual ids-only call: rows of whole 16-byte pieces, every view's plane 16-byte aligned, no depth image
    // (variant bit 512: the general ids kernel also where the plain one would run)
    const bool plain = !out.winner && out.ids && !out.depth && (w & 3) == 0 && (reinterpret_cast<uintptr_t>(out.ids) & 15) == 0 &&
                       !(a.var & 512);
    if (roll) {  // fused, 64 x 32 tiles, single-pass binning
      if (a.ent40 && a.micro) GR_LAUNCH_EV(ev_none, ev_stop, (k_raster_tile_roll<6, 5, 256, true, GR_LDS_PAD, true, false, GR_ROLL_KT, true>), grid, block, pad, s, a, out);
      else if (a.ent40) GR_LAUNCH_EV(ev_none, ev_stop, (k_raster_tile_roll<6, 5, 256, true, GR_LDS_PAD, true, false, GR_ROLL_KT, false>), grid, block, pad, s, a, out);
      else GR_LAUNCH_EV(ev_none, ev_stop, (k_raster_tile_roll<6, 5, 256, true, GR_LDS_PAD48, false, false, GR_ROLL_KT, false>), grid, block, pad, s, a, out);
    } else if (out.winner) {
      if (a.thl == 6) GR_LAUNCH_TILE(6, true, false);
      else GR_LAUNCH_TILE(5, true, false);
    } else if (plain) {
      if (a.thl == 6) GR_LAUNCH_TILE(6, false, true);
      else GR_LAUNCH_TILE(5, false, true);
    } else if (a.thl == 6) GR_LAUNCH_TILE(6, false, false);
    else GR_LAUNCH_TILE(5, false, false);
#undef GR_LAUNCH_TILE
#undef GR_LAUNCH_TILE_M
    c->chain_ev = nullptr;
    c->prof_raster_launches += 1;
  }
  c->prof_views += nb;
  GR_HIP(c, hipGetLastError());
  return GR_OK;
}

}  // namespace grimpl
