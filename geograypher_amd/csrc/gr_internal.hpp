#pragma once
// geograypher_amd/csrc/gr_internal.hpp -- what the translation units of libgeograster share: constants, the kernel
// argument blocks, the context, and the host-side helpers (error text, HIP-event spans, grow-only scratch).
//   geograster.hip   context, options, learned binning table, the raster call (launch groups, side stream), status
//   mesh_upload.hip  gr_mesh_upload: Morton order, de-indexed soup, block bounds            (per upload)
//   binning.hip      k_cull_blocks, k_setup_cull, k_clip_faces, entry compilation, exact-path scan / fill  (per view)
//   raster_tile.hip  k_raster_tile: the dominant kernel
//   project.hip      winners, votes, gathers, sparse pairs, finalize, argmax + their entry points
//   warp.hip         distortion warp, lens inversion (row f1)
//   resize.hip       photo down-scale of get_image (anti-aliased resize)
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

#include "geograster.h"

// ------------------------------------------------------------------------------------------------------------------
// constants
// ------------------------------------------------------------------------------------------------------------------
#define GR_TILE 64          // tile width in pixels (a workgroup rasterizes 64x32 or 64x64 tiles out of LDS)
#define GR_TILE_LOG2 6
#define GR_MAX_BATCH 64     // views per launch group (amortises kernel boundaries and per-launch tails)
#define GR_ENT_Q 3          // int4 per compiled (face, tile) entry: 48 bytes, 12 words
#define GR_CTRL_HDR 8       // ctrl words before the tile arrays: rec_count, total_entries, overflow, work_count, clip_count, -, micro_count
#define GR_MAX_DIM 16384    // h, w limit (guard band and 16-bit bbox packing)
#define GR_BLOCK 64         // faces per block of the Morton-ordered soup: one wave, one bounding sphere
#define GR_BLOCK_VERTS 192  // distinct vertices a block can have (3 per face); a patch of a manifold mesh has about 48
namespace grimpl {

struct BinArgs {
  uint32_t *ctrl;        // [slot][GR_CTRL_HDR + 4*Tcap]  rec_count,total,overflow,work_count,clip_count,- | cntS[T] | cntB[T] | offset[T] | curB[T]
                         //   cntS: entries whose list position was handed out in k_setup_cull (faces touching <= 2x2 tiles)
                         //   cntB: entries of larger faces, placed by k_fill_compile behind the cntS block of their tile (exact path)
  int4 *rec;             // [slot][4][F]  plane0 {X0,Y0,X1,Y1} plane1 {X2,Y2,iz0,face} plane2 {A,B,jmin|jmax<<16,imin|imax<<16}
                         //               plane3 {list position in up to 4 tiles}
  const float *soup;     // [F][9] the three vertex positions of every face, in Morton order (built once per upload)
  const float *bvert;    // [ceil(F/64)][GR_BLOCK_VERTS][3] the DISTINCT vertices of every 64-face block, in order of first use
  const uint32_t *bidx;  // [F] positions of the face's three vertices in its block's list (8 bits each) | (distinct vertices - 1) << 24
  const int32_t *orig;   // [F] soup position -> face id of the caller's mesh
  const float4 *blk;     // [ceil(F/64)] bounding sphere (centre, radius) of each block of GR_BLOCK faces, local frame
  uint32_t *touched;     // fused aggregation: [slot][tw] a BYTE per group of 64 consecutive CALLER face ids that received a winner in the
                         // view (set by the tile kernel's epilogue beside its winner atomic; zeroed by the group's init kernel), or null
  int tw;                // words per slot of `touched`
  uint32_t *work;        // [slot][work_stride] blocks of this view that passed the frustum test (ctrl[3] = count)
  uint32_t *clip;        // [slot][F] soup faces that straddle the near plane / guard band (R7; ctrl[4] = count)
  int64_t work_stride;
  int4 *comp;            // [slot][ent_cap][GR_ENT_Q]  compiled (face, tile) entries grouped by tile, 48 bytes each (ent40: 40 bytes
                         //                            each at the front of the same slot memory)
  uint8_t *nrow8;        // [slot][ent_cap] rows of each entry inside its tile (the tile kernel's scan input: a compact stream)
  unsigned long long *stats;  // [6] records, entries, max_entries, overflow, first overflowed launch group (over the call),
                              //     short-form miss (a face the 40-byte entry cannot hold: the caller repeats with 48 bytes)
  int group;             // index of this launch group inside the call
  int64_t ctrl_stride;   // words per slot
  int64_t rec_stride;    // int4 per slot: four planes of rec_stride / 4 >= F records
  int64_t ent_cap;       // entries per slot
  int64_t F;
  int T, TX, TY, Tcap;   // Tcap: WORDS per counter array (the tile count rounded up to 4, times the counter stride)
  int clg;               // tile t's counters are cntS[cidx(a, t)], cntB[cidx(a, t)]: -1 = packed side by side; >= 0 = spread over 2^clg lines of
                         // 128 bytes, tile t in line t mod 2^clg at word t div 2^clg (2^clg >= T: every tile a line of its own)
  int h, w;
  int twl, thl;          // log2 of the tile width / height in pixels
  int cap_tile;          // > 0: single-pass binning, every tile owns cap_tile entry slots (list base = tile * cap_tile)
  int ent40;             // 1: entries are written in the SHORT form (40 bytes, store_entry below); single-pass binning only
  int count_micro;       // 1: K1 counts the view's micro faces (pixel box at most 4 x 4) for gr_raster_stats: calls that can still learn micro lists
  int micro;             // 1: (face, tile) pairs of at most 4 x 4 pixels go to the tile's second list (K1 / raster_one_tile); needs ent40
  int var;               // variant bits (GR_OPT_VARIANT, include/geograster.h)
  int gl_order;          // GR_OPT_VERTEX_ORDER: 1 = the second half of the vertex stage in an OpenGL pipeline's order of operations
#ifdef GR_STAMPS
  unsigned long long *stamps;  // diagnostic build: [16] cycles per tile-kernel phase, summed over waves (raster_tile.hip)
#endif
  int dbg;               // GR_OPT_DEBUG: 512 (a TEST hook, results stay right): entry slots and row counts are poisoned with 0xFF
                         // before every launch group is binned
};

// word of tile t's counter inside a counter array (BinArgs::clg)
__device__ __forceinline__ int64_t cidx(const BinArgs &a, int t) {
  return a.clg < 0 ? (int64_t)t : (int64_t)((((uint32_t)t & ((1u << a.clg) - 1u)) << 5) | ((uint32_t)t >> a.clg));
}

struct RasterOut {
  int32_t *ids;      // [slot][h][w] or null
  float *depth;      // [slot][h][w] or null
  uint32_t *winner;  // fused projection: [slot][F] keys = (last pixel of the face in the view) + 1, or null
  uint8_t *touched;  // ... and [slot][4 tw] the byte map of the 64-face groups that hold a winner (BinArgs::touched)
  int64_t tb;        // bytes per slot of `touched`
  int64_t F;
  int compat;        // GR_FLAG_NEG1_IS_LAST_FACE
};

}  // namespace grimpl

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
  uint32_t *touched = nullptr;     // fused aggregation: [2][slots][tw] group maps (a byte per 64 faces) of the launch groups in flight
  uint32_t *visits = nullptr;      // ... [F / 64] (view, group) pairs the vote passes of the current call visited (gr_raster_stats.chunk_visits)
  int64_t visits_have = 0;
  bool visits_pending = false;     // ... not added into the call's statistics yet (gr_raster_status does)
  int64_t touched_have = 0;
  uint32_t *cur_touched = nullptr; // the bitmap the next bin_batch fills (null: none)
  int cur_tw = 0;
  int64_t blk_cap = 0;
  float *soup = nullptr;
  float *bvert = nullptr;    // distinct vertices per 64-face block (k_block_vertices)
  uint32_t *bidx = nullptr;  // per soup face: its vertices' positions in the block's list
  int32_t *orig = nullptr;   // soup position -> caller's face id (Morton order)
  int64_t soup_cap = 0;
  unsigned long long *stats = nullptr;
  int *flag = nullptr;
  int64_t ctrl_stride = 0, rec_stride = 0, ent_cap = 0, ent_cap_request = 0;
  int64_t ctrl_have = 0, comp_have = 0, work_have = 0, rec_have = 0;  // allocated element counts
  int Tcap = 0, slots = 0, clg = -1;
  int64_t rec_F = 0;                   // records per plane and slot of the exact path (>= F)
  int64_t rec_cap_request = 0;         // ... asked for by a call whose clipped faces outgrew F records (gr_raster_status)
  // tuning knobs (gr_set_option)
  int opt_thl = 5;      // log2 tile height (5 or 6); width is 64.  64x32 tiles: 16 KiB of LDS, 8 workgroups per CU
  int opt_batch = GR_MAX_BATCH;
  int opt_dbg = 0;
  int opt_var = 0;
  int opt_gl_order = 0;  // GR_OPT_VERTEX_ORDER
  int opt_lds_pad = 0;   // extra dynamic LDS bytes per tile workgroup (occupancy experiments, GR_OPT_DEBUG_LDS)
  int opt_direct_cap = 512;  // single-pass binning: entry slots per tile (0 = always use the exact two-pass path)
  struct Learned { uint64_t mesh; int T, cap; bool full; bool micro = false; };
  Learned learned[8] = {};             // slots per tile learned from overflows -- and whether the image has faces the 40-byte entry
                                       // form cannot hold --, per (mesh signature, tile count); [n_learned % 8] is replaced next
  int n_learned = 0;
  bool share_learned = true;           // consult / feed the process-wide table (off once GR_OPT_DIRECT_CAP was set by hand, on
                                       // again with the next gr_mesh_upload)
  uint64_t mesh_sig = 0;               // signature of the uploaded mesh: face count, vertex count, vertex bounds (gr_mesh_upload)
  // The binning configuration of the CURRENT raster call, resolved once at its top (resolve_binning): the process-wide table
  // can change under a running call (another context, another thread) -- sizing, allocation, the bin pass and the tile pass
  // of every launch group must agree on the slots per tile and on the entry form.
  int cur_cap = 0;                     // single-pass binning: slots per tile (0: exact two-pass binning)
  bool cur_ent40 = false;              // 40-byte entries
  bool cur_micro = false;              // micro lists (faces of at most 4 x 4 pixels on a second list per tile) for this call
  bool cur_count_micro = false;        // ... and whether this call counts micro faces (it can still learn the lists)
  bool cur_look = false;               // nothing learned about this (mesh, image size): the first launch group's counts are read
                                       // before its tile kernel runs (raster_views)
  int rebinned = 0;                    // times the last raster call started over after that look
  bool stats_pending = false;          // the call's statistics have not been reset yet (the first launch group's init kernel does)
  bool defer_stats = false;            // this call's view totals (k_bin_stats) wait for gr_raster_status: one launch group, not fused, no look
  bool stats_deferred = false;         // ... and have not been added up yet
  grimpl::BinArgs deferred_args;               // ... with these arguments, for deferred_nb views
  int deferred_nb = 0;
  int64_t opt_budget_mb = 24 << 10;    // entry memory of one launch group (GR_OPT_DIRECT_BUDGET_MB)
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
#ifdef GR_STAMPS
  unsigned long long *stamps = nullptr;  // diagnostic build: phase cycles of the tile kernel, summed since the last read
#endif
  double *resize_tmp = nullptr;        // rows pass of gr_resize_image_f64: [2 h_out][w_in * C]
  int64_t resize_have = 0;
  hipStream_t last_stream = nullptr;   // of the last raster call (gr_raster_status reads its outcome there)
  std::vector<hipStream_t> used_streams;  // streams that work touching context scratch was enqueued on since the last quiesce
  // profiling
  bool profiling = false;
  struct Span { hipEvent_t a, b; int stage; bool own_a = true; };   // own_a false: `a` is the `b` of the span before (a chain of launch-attached stop events)
  hipEvent_t chain_ev = nullptr;       // the stop event of the latest instrumented launch of the running raster call (chain_stop)
  bool chain_first = false;
  std::vector<Span> spans;
  std::vector<hipEvent_t> pool;
  int prof_views = 0, prof_raster_launches = 0;
  char err[512] = {0};
};

namespace grimpl {

inline int fail(gr_ctx *c, int code, const char *fmt, ...) {
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

inline hipEvent_t take_event(gr_ctx *c) {
  hipEvent_t e;
  if (!c->pool.empty()) { e = c->pool.back(); c->pool.pop_back(); return e; }
  if (hipEventCreate(&e) != hipSuccess) return nullptr;
  return e;
}

struct Timed {  // RAII span around a kernel group when profiling is on
  gr_ctx *c; hipStream_t s; int stage; hipEvent_t a = nullptr, b = nullptr;
  Timed(gr_ctx *c_, hipStream_t s_, int st) : c(c_), s(s_), stage(st) {
    if (c->profiling && st >= 0) { a = take_event(c); b = take_event(c); if (a) (void)hipEventRecord(a, s); }   // (st < 0: no span)
  }
  ~Timed() {
    if (c->profiling && a && b) { (void)hipEventRecord(b, s); gr_ctx::Span sp; sp.a = a; sp.b = b; sp.stage = stage; c->spans.push_back(sp); }
  }
};

// Spans whose events ride on kernel LAUNCHES (hipExtLaunchKernelGGL's stop event: the END of that kernel) instead of being
// recorded between kernels: an event record is a packet of its own on the stream and costs the kernels behind it 3.8 us -- four
// of them 1.65 % of a C2 step, 4.3 % of a quarter-scale one (tools/event_cost.py).  (The launch's START event is no use: it is
// stamped when the packet is taken up, while the kernel before it still runs.)  So the stages of a raster call are timed as a
// CHAIN of ends: chain_begin -> the end of the kernel in front of the first stage (k_bin_init); chain_stop(stage) -> the end of
// the stage's last kernel, the span [previous end, this end] -- the launch gap in front of a kernel is part of its span.
inline hipEvent_t chain_begin(gr_ctx *c) {
  if (c->chain_ev && c->chain_first) c->pool.push_back(c->chain_ev);   // a chain that was begun and never continued
  c->chain_ev = c->profiling ? take_event(c) : nullptr;
  c->chain_first = true;
  return c->chain_ev;
}
inline hipEvent_t chain_stop(gr_ctx *c, int stage) {
  if (!c->profiling || !c->chain_ev) return nullptr;
  hipEvent_t e = take_event(c);
  if (!e) return nullptr;
  gr_ctx::Span sp; sp.a = c->chain_ev; sp.b = e; sp.stage = stage; sp.own_a = c->chain_first;
  c->spans.push_back(sp);
  c->chain_first = false;
  c->chain_ev = e;
  return e;
}
inline void release_spans(gr_ctx *c) {   // events back to the pool
  if (c->chain_ev && c->chain_first) c->pool.push_back(c->chain_ev);
  c->chain_ev = nullptr; c->chain_first = false;
  for (auto &sp : c->spans) { if (sp.own_a) c->pool.push_back(sp.a); c->pool.push_back(sp.b); }
  c->spans.clear();
}
// launch KERNEL with (optional) start / stop events attached
#define GR_LAUNCH_EV(EVA, EVB, KERNEL, GRID, BLOCK, SHMEM, STREAM, ...)                                        \
  do {                                                                                                         \
    const hipEvent_t eva_ = (EVA), evb_ = (EVB);   /* evaluated ONCE: chain_stop takes an event and pushes a span */ \
    if (eva_ || evb_) hipExtLaunchKernelGGL(KERNEL, GRID, BLOCK, SHMEM, STREAM, eva_, evb_, 0, __VA_ARGS__);   \
    else hipLaunchKernelGGL(KERNEL, GRID, BLOCK, SHMEM, STREAM, __VA_ARGS__);                                  \
  } while (0)

inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Grow-only scratch: every buffer keeps its own capacity and is re-allocated only when it is too small (a new mesh or
// image size does not touch buffers that are already large enough).
// Before scratch is freed: wait for the work that can still use it -- the stream of the context's last call and its side
// stream -- not for the whole device (other contexts, the caller's own streams keep running).
inline void quiesce(gr_ctx *c) {
  for (hipStream_t st : c->used_streams) (void)hipStreamSynchronize(st);
  c->used_streams.clear();
  if (c->side) (void)hipStreamSynchronize(c->side);
}

inline void note_stream(gr_ctx *c, hipStream_t s) {
  for (hipStream_t st : c->used_streams)
    if (st == s) return;
  c->used_streams.push_back(s);
}

template <typename T>
int grow(gr_ctx *c, T *&ptr, int64_t &have, int64_t want, const char *what) {
  if (ptr && have >= want) return GR_OK;
  if (ptr) quiesce(c);
  if (ptr) (void)hipFree(ptr);
  ptr = nullptr; have = 0;
  if (hipMalloc(&ptr, sizeof(T) * (size_t)want) != hipSuccess)
    return fail(c, GR_ENOMEM, "%s scratch allocation failed (%lld bytes)", what, (long long)(sizeof(T) * (size_t)want));
  have = want;
  return GR_OK;
}

// (the zeroes go out on the CALL's stream: a plain hipMemset runs on the null stream, which a non-blocking stream -- torch's
// side streams, one per device thread of a devices=[...] mesh -- does not wait for: the first view's winners could be wiped
// after they were written; found by tests/test_devices_kwarg.py with three contexts on one GPU)
inline int ensure_winner(gr_ctx *c, size_t bytes, hipStream_t s) {
  if (c->winner && c->winner_bytes >= bytes) return GR_OK;
  if (c->winner) quiesce(c);
  if (c->winner) (void)hipFree(c->winner);
  c->winner = nullptr; c->winner_bytes = 0;
  if (hipMalloc(&c->winner, bytes) != hipSuccess) return fail(c, GR_ENOMEM, "winner scratch allocation failed");
  if (hipMemsetAsync(c->winner, 0, bytes, s) != hipSuccess) return fail(c, GR_EHIP, "winner memset failed");
  c->winner_bytes = bytes;
  return GR_OK;
}

inline BinArgs make_args(gr_ctx *c, int h, int w, int slot0) {
  BinArgs a;
  a.ctrl_stride = c->ctrl_stride; a.rec_stride = c->rec_stride; a.ent_cap = c->ent_cap; a.F = c->F;
  a.work_stride = c->work_stride;
  a.ctrl = c->ctrl + slot0 * a.ctrl_stride; a.rec = c->rec + slot0 * a.rec_stride;
  a.comp = c->comp + slot0 * a.ent_cap * GR_ENT_Q; a.work = c->work + slot0 * a.work_stride;
  a.nrow8 = c->nrow8 + slot0 * a.ent_cap;
  a.stats = c->stats; a.blk = c->blk; a.soup = c->soup; a.orig = c->orig; a.bvert = c->bvert; a.bidx = c->bidx;
  a.touched = c->cur_touched; a.tw = c->cur_tw;
  a.clip = c->clip + slot0 * c->F;
  a.twl = GR_TILE_LOG2; a.thl = c->opt_thl;
  a.TX = (w + (1 << a.twl) - 1) >> a.twl; a.TY = (h + (1 << a.thl) - 1) >> a.thl; a.T = a.TX * a.TY; a.Tcap = c->Tcap; a.clg = c->clg;
  a.h = h; a.w = w; a.dbg = c->opt_dbg; a.var = c->opt_var; a.gl_order = c->opt_gl_order;
  a.cap_tile = c->cur_cap;          // the call's snapshot: every launch group, bin pass and tile pass alike
  a.ent40 = c->cur_ent40 ? 1 : 0;
  a.micro = c->cur_micro ? 1 : 0;   // resolved once per call (resolve_binning)
  a.count_micro = c->cur_count_micro ? 1 : 0;
  a.group = 0;
#ifdef GR_STAMPS
  a.stamps = c->stamps;
#endif
  return a;
}

inline int check_common(gr_ctx *c, int n_views, int h, int w) {
  if (!c) return GR_EINVAL;
  if (n_views < 0 || h <= 0 || w <= 0 || h > GR_MAX_DIM || w > GR_MAX_DIM)
    return fail(c, GR_EINVAL, "bad shape n_views=%d h=%d w=%d (limit %d)", n_views, h, w, GR_MAX_DIM);
  return GR_OK;
}

// launchers that live in other translation units
int bin_batch(gr_ctx *c, const float *cams, int nb, int h, int w, int slot0, int group, hipStream_t s);   // binning.hip
int tile_batch(gr_ctx *c, int nb, int h, int w, int slot0, RasterOut out, hipStream_t s);                 // raster_tile.hip
int bin_stats_deferred(gr_ctx *c, hipStream_t s);                                                         // binning.hip
int sum_visits(gr_ctx *c, hipStream_t s);                                                                 // binning.hip
int project_labels(gr_ctx *c, const int32_t *ids, const uint8_t *labels, int n_views, int h, int w, int C, uint32_t *votes,
                   uint32_t *counts, int flags, hipStream_t s);                                           // project.hip
void launch_vote_labels(gr_ctx *c, hipStream_t vs, uint32_t *win, const uint8_t *labels, int nb, int64_t F, int64_t P, int C,
                        uint32_t *votes, uint32_t *counts, int group, const uint32_t *touched, int tw, int flags);  // project.hip

}  // namespace grimpl
