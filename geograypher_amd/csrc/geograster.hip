// geograypher_amd/csrc/geograster.hip -- libgeograster: hand-written CDNA4 (gfx950, wave64) kernels + the C ABI of include/geograster.h.
// This file: context, options, the learned binning table, the raster call and its status (the kernels: gr_internal.hpp).
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

#include <unistd.h>

#include <mutex>
#include <new>

#include "gr_internal.hpp"

using namespace grimpl;

namespace {

// Slots per tile of the single-pass binning for an image of T tiles (0 = exact two-pass binning).  A call that
// overflowed the configured slots teaches the context the size that image needs (gr_raster_status), as long as the
// entry memory of a launch group stays within the budget (GR_OPT_DIRECT_BUDGET_MB, default 24 GiB); other image sizes keep
// the configured value.
// What one context has learned is kept process-wide as well, keyed by (mesh signature, tile count): a second context for the
// same mesh and image size -- another camera set, another thread -- starts with segments that fit, without an overflowed
// first call.  With gr_learned_cache_file the table is also read from / written to a small text file, so that a NEW PROCESS
// starts sized (one line per entry: signature, tiles, slots, entry form).  The signature is the face count, the vertex count
// and the vertex bounds of the upload: another mesh that happens to have as many faces does not inherit anything.
std::mutex g_learned_mu;
gr_ctx::Learned g_learned[64];
int g_n_learned = 0;
char g_learned_path[1024] = {0};

// cap == GR_LEARNED_EXACT: the image's tiles need more slots than one view's entry memory may take -- it bins exactly
#define GR_LEARNED_EXACT (-1)
#define GR_LEARNED_MAX_CAP 65536   // what GR_OPT_DIRECT_CAP accepts

inline int merge_cap(int a, int b) { return (a == GR_LEARNED_EXACT || b == GR_LEARNED_EXACT) ? GR_LEARNED_EXACT : std::max(a, b); }

void put_global_locked(const gr_ctx::Learned &v);

bool parse_learned_line(const char *line, gr_ctx::Learned &v) {
  unsigned long long m; int T, cap, full, micro = 0;
  if (line[0] == '#' || sscanf(line, "%llx %d %d %d %d", &m, &T, &cap, &full, &micro) < 4) return false;   // the fifth field came with 0.2.2
  if (T <= 0 || cap < GR_LEARNED_EXACT || cap > GR_LEARNED_MAX_CAP) return false;
  v = {(uint64_t)m, T, cap, full != 0, micro != 0};
  return true;
}

void save_learned_locked() {
  if (!g_learned_path[0]) return;
  // several processes (the ranks of one job) share the file: what the others wrote since this process read it is merged
  // in before the rewrite (entry-wise maximum), so that no rank's lesson is lost to another's rename
  if (FILE *f = fopen(g_learned_path, "r")) {
    char line[256];
    gr_ctx::Learned v;
    while (fgets(line, sizeof(line), f))
      if (parse_learned_line(line, v)) put_global_locked(v);
    fclose(f);
  }
  char tmp[1100];
  snprintf(tmp, sizeof(tmp), "%s.tmp.%d", g_learned_path, (int)getpid());
  FILE *f = fopen(tmp, "w");
  if (!f) return;
  fprintf(f, "# libgeograster: slots per tile (-1: exact binning) / 48-byte entries / micro lists learned per (mesh signature, tile count)\n");
  for (int i = 0; i < std::min(g_n_learned, 64); ++i)
    fprintf(f, "%016llx %d %d %d %d\n", (unsigned long long)g_learned[i].mesh, g_learned[i].T, g_learned[i].cap, g_learned[i].full ? 1 : 0,
            g_learned[i].micro ? 1 : 0);
  fclose(f);
  if (rename(tmp, g_learned_path) != 0) (void)remove(tmp);
}

void put_global_locked(const gr_ctx::Learned &v) {
  int j = 0;
  for (; j < std::min(g_n_learned, 64); ++j)
    if (g_learned[j].mesh == v.mesh && g_learned[j].T == v.T) break;
  if (j == std::min(g_n_learned, 64)) {
    j = g_n_learned % 64; g_n_learned += 1;
    g_learned[j] = v;
    return;
  }
  g_learned[j].cap = merge_cap(g_learned[j].cap, v.cap);
  g_learned[j].full = g_learned[j].full || v.full;
  g_learned[j].micro = g_learned[j].micro || v.micro;
}

// what is known about images of T tiles of the current mesh: slots per tile (0: nothing learned) and the entry form
void lookup_learned(const gr_ctx *c, int T, int &cap, bool &full, bool *micro = nullptr) {
  cap = 0; full = false;
  if (micro) *micro = false;
  for (int i = 0; i < std::min(c->n_learned, 8); ++i)
    if (c->learned[i].mesh == c->mesh_sig && c->learned[i].T == T) {
      cap = c->learned[i].cap; full = c->learned[i].full;
      if (micro) *micro = c->learned[i].micro;
      return;
    }
  if (c->share_learned) {
    std::lock_guard<std::mutex> lk(g_learned_mu);
    for (int i = 0; i < std::min(g_n_learned, 64); ++i)
      if (g_learned[i].mesh == c->mesh_sig && g_learned[i].T == T) {
        cap = g_learned[i].cap; full = g_learned[i].full;
        if (micro) *micro = g_learned[i].micro;
        return;
      }
  }
}

// The binning configuration of one raster call, from ONE look at the tables: slots per tile (0 = exact two-pass binning)
// and the entry form.  The 40-byte entry form (store_entry) is the default of the single-pass binning; images with faces it
// cannot hold (93 px and more) are remembered like the slots per tile.  Variant bit 128: always 48 bytes.  The short form is
// laid out in chunks of 64 entries (store_entry): a tile's segment must be a whole number of chunks -- slots per tile set by
// hand to anything else: 48 bytes.
void resolve_binning(gr_ctx *c, int T) {
  int cap = 0; bool full = false, micro = false;
  c->cur_cap = 0; c->cur_ent40 = false; c->cur_micro = false; c->cur_look = false; c->cur_count_micro = false;
  if (c->opt_direct_cap <= 0 || !c->direct_ok) return;
  lookup_learned(c, T, cap, full, &micro);
  if (cap == GR_LEARNED_EXACT) return;  // this (mesh, image size) bins exactly: one view's segments would not fit the budget
  // nothing is known about the slots this mesh and image size need: the call looks at the counts of its first launch group
  // before that group's tile kernel runs (raster_views).  Variant bit 16384: never (rounds 1-4: gr_raster_status reports it)
  c->cur_look = cap == 0 && !(c->opt_var & 16384);
  c->cur_cap = std::max(cap, c->opt_direct_cap);
  c->cur_ent40 = !(c->cur_cap & 63) && !(c->opt_var & 128) && !full;
  // micro lists: where an earlier call found most faces of the image at most 4 x 4 pixels (gr_raster_status; remembered like
  // the slots per tile), with 40-byte entries.  Variant bits: 8192 = always, 4096 = never.
  c->cur_micro = c->cur_ent40 && !(c->opt_var & 4096) && (micro || (c->opt_var & 8192));
  // micro faces are counted where the count can still switch the lists on: by the call that looks at its first launch group, and
  // by every call of the status-call protocol (variant bit 16384)
  c->cur_count_micro = c->cur_ent40 && !c->cur_micro && !(c->opt_var & 4096) && (c->cur_look || (c->opt_var & 16384));
}

// cap > 0: the slots per tile the image needs; full: it needs 48-byte entries (both are kept once learned)
void learn(gr_ctx *c, int T, int cap, bool full, bool micro = false) {
  int old_cap; bool old_full, old_micro;
  lookup_learned(c, T, old_cap, old_full, &old_micro);
  const gr_ctx::Learned v = {c->mesh_sig, T, merge_cap(cap, old_cap), full || old_full, micro || old_micro};
  int i = 0;
  for (; i < std::min(c->n_learned, 8); ++i)
    if (c->learned[i].mesh == c->mesh_sig && c->learned[i].T == T) break;
  if (i == std::min(c->n_learned, 8)) { i = c->n_learned % 8; c->n_learned += 1; }
  c->learned[i] = v;
  if (!c->share_learned) return;
  std::lock_guard<std::mutex> lk(g_learned_mu);
  put_global_locked(v);
  save_learned_locked();
}

// What the counters of an overflowed single-pass bin pass teach (they kept counting, so the need is known): segments of that
// size if ONE view's entry memory stays within budget -- raster_views shrinks the launch group until the segments fit; round 4
// priced the failed call's whole group and switched single-pass binning off for the CONTEXT, every image size, for good --,
// exact binning (count, scan, fill) otherwise.  max_tile: the largest count a tile reached.
void learn_slots(gr_ctx *c, int T, int64_t max_tile, bool full, bool micro) {
  const int64_t need = (max_tile + max_tile / 8 + 16 + 63) / 64 * 64;
  const int64_t bytes_one_view = need * (16 * GR_ENT_Q) * (int64_t)T;
  if (need <= GR_LEARNED_MAX_CAP && bytes_one_view <= (c->opt_budget_mb << 20)) learn(c, T, (int)need, full, micro);
  else learn(c, T, GR_LEARNED_EXACT, full, micro);   // this (mesh, image size) bins exactly from now on
}

int ensure_bins(gr_ctx *c, int n_slots, int T) {
  const int64_t F = c->F > 0 ? c->F : 1;
  const int dcap = c->cur_cap;  // resolved once per call (resolve_binning)
  const bool direct = dcap > 0;
  int64_t cap = c->ent_cap_request > 0 ? c->ent_cap_request : (F / 2 + 65536);
  if (direct) cap = std::max<int64_t>(cap, (int64_t)T * dcap);
  // The layout (strides) is that of THIS call; a buffer is re-allocated only when the call needs more elements than the
  // buffer has (a huge image with a small launch group and a small image with a full one share the same memory).
  // Small images: ONE tile counter per 128-byte line.  Atomics on the same line are served one after the other whatever their
  // addresses (32 counters in one line: 5.9 ns per atomic chip-wide, one per line: 0.37 -- profiles/r03_ubench_atomic_rate.txt),
  // and an image of 1000 x 750 has all its 384 counters in twelve lines, hit by every wave that bins the view.  (Images of
  // thousands of tiles spread their atomics over hundreds of lines anyway, and padding them would cost the init kernel 32 x the
  // bytes.)  Single-pass binning only; variant bit 131072: packed counters everywhere.
  int clg = -1;
  if (direct && !(c->opt_var & 131072)) {
    if (T <= 1024) { clg = 0; while ((1 << clg) < T + 3) ++clg; }          // a line per tile
  }
  const int Tcap = clg < 0 ? ((T + 3) & ~3) : (32 << clg);  // words per counter array; the arrays start 16-byte aligned (a chain reads four counters at once)
  const int64_t ctrl_stride = ((GR_CTRL_HDR + (direct ? 2 : 4) * (int64_t)Tcap) + 63) / 64 * 64;   // (exact binning: + offsets and cursors)
  const int64_t work_stride = ceil_div(F, GR_BLOCK) + 4;
  int rc = grow(c, c->ctrl, c->ctrl_have, ctrl_stride * n_slots, "bin control");
  if (!rc) rc = grow(c, c->comp, c->comp_have, GR_ENT_Q * cap * n_slots, "entry list");
  if (!rc) rc = grow(c, c->nrow8, c->nrow_have, cap * n_slots + 64, "entry row counts");
  if (!rc) rc = grow(c, c->work, c->work_have, work_stride * n_slots, "work list");
  if (!rc) rc = grow(c, c->clip, c->clip_have, F * n_slots, "clip list");
  const int64_t RF = std::max<int64_t>(F, c->rec_cap_request);  // records per view: a face each, a clipped face up to six
  if (!rc && !direct) rc = grow(c, c->rec, c->rec_have, 4 * RF * n_slots, "record planes");  // exact path only
  if (rc) return rc;
  c->slots = n_slots; c->Tcap = Tcap; c->clg = clg; c->ent_cap = cap; c->ctrl_stride = ctrl_stride; c->work_stride = work_stride;
  c->rec_F = RF; c->rec_stride = 4 * RF;
  return GR_OK;
}

// pix2face for n_views cameras, optionally fused with the label projection (labels != nullptr): per launch group the
// tile kernel's epilogue feeds the per-face winners straight from LDS and k_vote_labels folds them into votes/counts.
int raster_views(gr_ctx *c, const float *cams, int n_views, int h, int w, int32_t *ids, float *depth,
                 const uint8_t *labels, int C, uint32_t *votes, uint32_t *counts, int flags, hipStream_t s, int again = 0) {
  int rc = check_common(c, n_views, h, w);
  if (rc) return rc;
  c->rebinned = again;
  if (!c->verts) return fail(c, GR_ENOMESH, "gr_mesh_upload has not been called");
  if (!cams) return fail(c, GR_EINVAL, "null cams");
  if (n_views == 0) return GR_OK;
  GR_HIP(c, hipSetDevice(c->device));
  int B = n_views < c->opt_batch ? n_views : c->opt_batch;
  const int thl = c->opt_thl;
  const int T = ((w + GR_TILE - 1) >> GR_TILE_LOG2) * ((h + (1 << thl) - 1) >> thl);
  resolve_binning(c, T);
  {  // very large images: fewer views per launch group, so that the fixed tile segments stay within the scratch budget
    const int64_t per_slot = (int64_t)T * c->cur_cap * (16 * GR_ENT_Q);
    if (per_slot > 0) B = (int)std::max<int64_t>(1, std::min<int64_t>(B, (c->opt_budget_mb << 20) / per_slot));
  }
  rc = ensure_bins(c, B, T);
  if (rc) return rc;
  c->last_T = T; c->last_B = B;
  const int64_t P = (int64_t)h * w, F = c->F;
  // Fused aggregation over several launch groups: the vote kernel of group g (a light, latency-bound pass over F winners)
  // runs on a side stream beside the binning of group g + 1 (also light); the tile kernels in between fill the machine on
  // their own.  Two winner buffers alternate; votes are still added group by group, in order (one side stream).
  const bool overlap = labels && n_views > B && !(c->opt_var & 4);
  // group maps for the vote kernel: a byte per 64 consecutive caller face ids, set by the tile kernel's epilogue where it
  // issues a winner (zeroed by the launch group's init kernel); in words; one map per view of the launch groups in flight
  const int tw = (int)ceil_div(ceil_div(F, 64), 4);
  const bool use_touched = labels != nullptr;
  if (labels) {
    rc = ensure_winner(c, sizeof(uint32_t) * (size_t)F * B * (overlap ? 2 : 1), s);
    if (rc) return rc;
    if (use_touched) {
      rc = grow(c, c->touched, c->touched_have, (int64_t)2 * B * tw, "winner group maps");
      if (!rc) rc = grow(c, c->visits, c->visits_have, ceil_div(F, 64) + 4, "visit counters");
      if (rc) return rc;
      if (again == 0) GR_HIP(c, hipMemsetAsync(c->visits, 0, sizeof(uint32_t) * (size_t)(ceil_div(F, 64) + 4), s));
      c->visits_pending = true;
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
  note_stream(c, s);
  c->stats_pending = true;   // the call's statistics are reset by the first launch group's init kernel (bin_batch: k_bin_init)
  c->stats_deferred = false;
  if (!labels) c->visits_pending = false;
  c->defer_stats = !labels && n_views <= B && !(c->cur_look && again < 2);
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
      // (zeroed by the group's init kernel: bin_batch)
    }
    c->cur_touched = tch; c->cur_tw = tw;
    rc = bin_batch(c, cams + (int64_t)v0 * GR_CAM_FLOATS, nb, h, w, 0, v0 / B, s);
    c->cur_touched = nullptr;
    if (rc) return rc;
    if (c->cur_look && g == 0 && again < 2) {
      // The first launch group of a (mesh, image size) nothing has been learned about: its counts are read HERE, before its
      // tile kernel runs -- one host round trip per mesh and image size in the life of the learned table.  A tile that outgrew
      // its slots, a face the 40-byte entries cannot hold, a view of mostly micro faces: the lesson is taken and the call
      // starts over (nothing has been written yet; this group's bin pass -- a quarter of its time -- is what the lesson
      // costs, where rounds 1-4 paid the whole call once more after gr_raster_status).  Otherwise the slots in use are noted as
      // sufficient and no call looks again.  Later groups of the call, and later calls with more crowded views, keep the
      // GR_EOVERFLOW protocol.
      unsigned long long st[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
      GR_HIP(c, hipMemcpyAsync(st, c->stats, sizeof(st), hipMemcpyDeviceToHost, s));
      GR_HIP(c, hipStreamSynchronize(s));
      const bool grow = st[3] && (int64_t)st[2] > c->cur_cap, miss = st[3] && st[5] != 0;
      const bool micro = !c->cur_micro && !(c->opt_var & (128 | 4096)) && !miss && st[0] > 0 && 5 * st[8] > 2 * st[0];
      if (grow) learn_slots(c, T, (int64_t)st[2], miss, micro);
      else learn(c, T, c->cur_cap, miss, micro);
      if (grow || miss || micro) return raster_views(c, cams, n_views, h, w, ids, depth, labels, C, votes, counts, flags, s, again + 1);
    }
    uint32_t *win = labels ? (uint32_t *)c->winner + (int64_t)buf * F * B : nullptr;
    RasterOut out;
    out.ids = ids ? ids + v0 * P : nullptr;
    out.depth = depth ? depth + v0 * P : nullptr;
    out.winner = win;
    out.touched = reinterpret_cast<uint8_t *>(tch);
    out.tb = 4 * (int64_t)tw;
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
      launch_vote_labels(c, vs, win, labels + v0 * P, nb, F, P, C, votes, counts, v0 / B, (const uint32_t *)tch, tw, flags);
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
  if (hipMalloc(&c->stats, sizeof(unsigned long long) * 16) != hipSuccess ||
      hipMalloc(&c->flag, sizeof(int) * 8) != hipSuccess) {  // flag word + upload scratch (vertex bounds)
    delete c;
    return GR_ENOMEM;
  }
  (void)hipMemset(c->stats, 0, sizeof(unsigned long long) * 16);
  (void)hipDeviceSynchronize();   // (the null stream's memset: a non-blocking stream of the first call would not wait for it)
#ifdef GR_STAMPS
  if (hipMalloc(&c->stamps, sizeof(unsigned long long) * 32 * 1024) == hipSuccess) (void)hipMemset(c->stamps, 0, sizeof(unsigned long long) * 32 * 1024);
#endif
  *out = c;
  return GR_OK;
}

int gr_ctx_destroy(gr_ctx *c) {
  if (!c) return GR_OK;
  (void)hipSetDevice(c->device);
  (void)hipDeviceSynchronize();
  for (auto &sp : c->spans) { if (sp.own_a) (void)hipEventDestroy(sp.a); (void)hipEventDestroy(sp.b); }
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
  if (c->resize_tmp) (void)hipFree(c->resize_tmp);
  if (c->blk) (void)hipFree(c->blk);
  if (c->visits) (void)hipFree(c->visits);
  if (c->touched) (void)hipFree(c->touched);
  if (c->soup) (void)hipFree(c->soup);
  if (c->bvert) (void)hipFree(c->bvert);
  if (c->bidx) (void)hipFree(c->bidx);
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
  release_spans(c);
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
    case GR_OPT_SHARE_LEARNED:
      c->share_learned = value != 0; return GR_OK;
    case GR_OPT_VERTEX_ORDER:
      if (value != 0 && value != 1) return fail(c, GR_EINVAL, "vertex order must be 0 (rule R1) or 1 (OpenGL's order of operations)");
      c->opt_gl_order = value; return GR_OK;
    case GR_OPT_DIRECT_BUDGET_MB:
      if (value < 1) return fail(c, GR_EINVAL, "entry-memory budget must be at least 1 MiB");
      c->opt_budget_mb = value; return GR_OK;
    default: return fail(c, GR_EINVAL, "unknown option %d", key);
  }
}

int gr_learned_cache_file(const char *path_h) {
  std::lock_guard<std::mutex> lk(g_learned_mu);
  if (!path_h || !path_h[0]) { g_learned_path[0] = 0; return GR_OK; }
  if (strlen(path_h) >= sizeof(g_learned_path)) return GR_EINVAL;
  snprintf(g_learned_path, sizeof(g_learned_path), "%s", path_h);
  FILE *f = fopen(path_h, "r");
  if (!f) return GR_OK;  // nothing learned yet: the file appears with the first overflow
  char line[256];
  gr_ctx::Learned v;
  while (fgets(line, sizeof(line), f))
    if (parse_learned_line(line, v)) put_global_locked(v);
  fclose(f);
  return GR_OK;
}

int gr_learned_cache_clear(void) {
  std::lock_guard<std::mutex> lk(g_learned_mu);
  g_n_learned = 0;
  memset(g_learned, 0, sizeof(g_learned));
  return GR_OK;
}

#ifdef GR_STAMPS
// diagnostic build only (tools/tile_phases.py): read and clear the tile kernel's phase-cycle sums (synchronises the device)
int gr_debug_read_stamps(gr_ctx *c, unsigned long long *out16_h) {
  if (!c || !out16_h || !c->stamps) return GR_EINVAL;
  GR_HIP(c, hipDeviceSynchronize());
  std::vector<unsigned long long> all(16 * 1024);
  GR_HIP(c, hipMemcpy(all.data(), c->stamps, sizeof(unsigned long long) * all.size(), hipMemcpyDeviceToHost));
  GR_HIP(c, hipMemset(c->stamps, 0, sizeof(unsigned long long) * all.size()));
  for (int k = 0; k < 16; ++k) out16_h[k] = 0;
  for (size_t i = 0; i < all.size(); ++i) out16_h[i & 15] += all[i];
  return GR_OK;
}
#endif

#ifdef GR_STAMPS
// diagnostic build only (tools/setup_phases.py): the set-up kernel's phase-cycle sums (the second half of the stamp buffer)
int gr_debug_read_setup_stamps(gr_ctx *c, unsigned long long *out16_h) {
  if (!c || !out16_h || !c->stamps) return GR_EINVAL;
  GR_HIP(c, hipDeviceSynchronize());
  std::vector<unsigned long long> all(16 * 1024);
  GR_HIP(c, hipMemcpy(all.data(), c->stamps + 16 * 1024, sizeof(unsigned long long) * all.size(), hipMemcpyDeviceToHost));
  GR_HIP(c, hipMemset(c->stamps + 16 * 1024, 0, sizeof(unsigned long long) * all.size()));
  for (int k = 0; k < 16; ++k) out16_h[k] = 0;
  for (size_t i = 0; i < all.size(); ++i) out16_h[i & 15] += all[i];
  return GR_OK;
}
#endif

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
  release_spans(c);
  c->prof_views = 0; c->prof_raster_launches = 0;
  return GR_OK;
}

int gr_raster_face_ids(gr_ctx *c, const float *cams, int n_views, int h, int w, int32_t *ids, float *depth,
                       void *stream) {
  if (c && !ids && !depth) return fail(c, GR_EINVAL, "null outputs");
  return raster_views(c, cams, n_views, h, w, ids, depth, nullptr, 0, nullptr, nullptr, 0, (hipStream_t)stream);
}

int gr_raster_status(gr_ctx *c, gr_raster_stats *o) {
  if (!c || !o) return GR_EINVAL;
  unsigned long long st[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  {
    int rc = bin_stats_deferred(c, c->last_stream);   // the view totals of a call that left them for now
    if (!rc) rc = sum_visits(c, c->last_stream);      // ... and the visit counters of a fused call's vote passes
    if (rc) return rc;
  }
  GR_HIP(c, hipMemcpyAsync(st, c->stats, sizeof(st), hipMemcpyDeviceToHost, c->last_stream));
  GR_HIP(c, hipStreamSynchronize(c->last_stream));
  o->records = (int64_t)st[0]; o->entries = (int64_t)st[1]; o->max_entries = (int64_t)st[2];
  o->entry_cap = c->ent_cap; o->overflow = (int32_t)st[3];
  o->blocks = (int64_t)st[6]; o->chunk_visits = (int64_t)st[7];
  o->rebinned_groups = c->rebinned;
  // views of the last call whose results are complete: every launch group in front of the first one that overflowed
  o->views_done = st[3] ? (int32_t)std::min<unsigned long long>(st[4] * (unsigned long long)std::max(c->last_B, 1),
                                                                (unsigned long long)c->last_n_views)
                        : c->last_n_views;
  if (st[3] && c->last_direct) {
    // A tile outgrew its fixed segment, or a face did not fit the 40-byte entry form (st[5]).  The counters kept counting,
    // so the need is known: the retry uses segments of that size if a launch group's entry memory stays within budget, and
    // bins exactly (count, scan, fill) otherwise; it uses 48-byte entries if the short form was missed.
    const int used = c->cur_cap;  // what the call ran with
    const bool grow = (int64_t)st[2] > used;
    if (st[5] != 0) learn(c, c->last_T, 0, true);
    if (grow) {
      learn_slots(c, c->last_T, (int64_t)st[2], false, false);
      return fail(c, GR_EOVERFLOW, "single-pass binning overflow: a tile received %llu entries (slots per tile %d); "
                  "retry the call", st[2], used);
    }
    return fail(c, GR_EOVERFLOW, "single-pass binning: a face does not fit the 40-byte entry form; retry the call "
                "(48-byte entries from now on)");
  }
  // Most of the image's faces are at most 4 x 4 pixels (K1 counts them; a survey mesh rendered at a quarter of its photos'
  // resolution: 85-95 %): the next call for this mesh and image size keeps micro lists (no retry: this call's result stands)
  if (!st[3] && c->last_direct && c->cur_ent40 && !c->cur_micro && !(c->opt_var & 4096) && st[0] > 0 && 5 * st[8] > 2 * st[0])
    learn(c, c->last_T, 0, false, true);
  if (st[3] && (int64_t)st[9] > c->rec_F) {
    // exact binning: the clipped faces of a view became more triangles than the record planes hold (one record per face
    // unless a call asked for more): the planes grow on the next call, the need is known
    c->rec_cap_request = (int64_t)st[9] + (int64_t)st[9] / 8 + 64;
    if ((int64_t)st[2] > c->ent_cap) c->ent_cap_request = (int64_t)st[2] + (int64_t)st[2] / 8 + 65536;
    return fail(c, GR_EOVERFLOW, "record list overflow: a view needs %llu records (a clipped face becomes several triangles), "
                "capacity %lld; retry the call", st[9], (long long)c->rec_F);
  }
  if (st[3]) {
    // grow on the next call: exact need is known
    c->ent_cap_request = (int64_t)st[2] + (int64_t)st[2] / 8 + 65536;
    return fail(c, GR_EOVERFLOW, "bin list overflow: a view needs %llu entries, capacity %lld; retry the call",
                st[2], (long long)c->ent_cap);
  }
  return GR_OK;
}

int gr_raster_project_labels_u8(gr_ctx *c, const float *cams, const uint8_t *labels, int n_views, int h, int w, int C,
                                uint32_t *votes, uint32_t *counts, int32_t *ids_or_null, int flags, void *stream) {
  if (!c) return GR_EINVAL;
  if (!labels || !votes || !counts || C <= 0 || C > 255) return fail(c, GR_EINVAL, "bad fused project args C=%d", C);
  return raster_views(c, cams, n_views, h, w, ids_or_null, nullptr, labels, C, votes, counts, flags, (hipStream_t)stream);
}

}  // extern "C"

