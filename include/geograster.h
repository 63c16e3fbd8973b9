/*
 * include/geograster.h -- C ABI of libgeograster (MI355X / gfx950 HIP implementation of geograypher's
 * image<->mesh projection hot path).
 *
 * This is the drop-in boundary.  The reference (pure Python) has no FFI for this path; the seam it offers is the
 * subclass-override plugin `pix2face` (geograypher/meshes/derived_meshes.py:642-650, precedent
 * TexturedPhotogrammetryMeshPyTorch3dRendering) plus the numpy stages that consume its output.  Each entry point
 * below names the reference lines it replaces; INTEGRATION.md shows the ctypes stub a reference maintainer adds.
 *
 * Conventions
 *  - Every pointer is a DEVICE pointer (hipMalloc / torch tensor .data_ptr()) unless its name ends in _h.
 *  - The caller allocates and owns all inputs and outputs.  The library allocates only per-context scratch
 *    (bin lists, per-face winners), grown lazily, freed by gr_ctx_destroy.
 *  - All work is enqueued on `stream` (a hipStream_t passed as void*; NULL = the default stream) and is
 *    asynchronous.  No hidden synchronisation except: gr_ctx_destroy, gr_mesh_upload (index validation) and
 *    scratch growth (hipMalloc/hipFree when a larger batch, image or mesh is first seen).
 *  - Return value: 0 (GR_OK) or a negative GR_E* code; text via gr_last_error().  No C++ exception crosses the ABI.
 *  - A context belongs to one (device, host thread); distinct contexts are independent.
 */
#ifndef GEOGRASTER_H
#define GEOGRASTER_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GR_VERSION 123 /* 0.2.3: the first launch group of an unknown (mesh, image size) is looked at before its tile kernel runs (gr_raster_stats.rebinned_groups); 0.2.2: micro lists (a fifth field in the learned-table file); 0.2.1: gr_learned_cache_clear; 0.2.0: gr_resize_image_f64, gr_learned_cache_file, mesh-signature keyed learned table */

enum {
  GR_OK = 0,
  GR_EINVAL = -1,    /* bad argument / shape                                                  */
  GR_EHIP = -2,      /* HIP runtime error (text in gr_last_error)                             */
  GR_ENOMEM = -3,    /* scratch allocation failed                                             */
  GR_ENOMESH = -4,   /* no mesh uploaded                                                      */
  GR_EINDEX = -5,    /* face index outside [0, V) found by gr_mesh_upload                     */
  GR_EOVERFLOW = -6, /* gr_raster_status: a tile list outgrew its slots, or a face did not fit
                        the 40-byte entry form -- the library has noted what the image needs;
                        repeat the call from view `views_done` on                              */
  GR_ENODEVICE = -7  /* no usable gfx950 device                                               */
};

/* flags for the projection / aggregation entry points */
enum {
  GR_FLAG_NEG1_IS_LAST_FACE = 1, /* reproduce meshes.py:1998-2001: pix2face == -1 writes the LAST face */
  GR_FLAG_DEFER_CHECK = 2        /* gr_project_index_pairs: do not synchronise; key_count then points to TWO 64-bit words,
                                    {pair count, error flag}: a value outside [0, n_classes) makes key_count[1] non-zero
                                    instead of failing the call (the caller reads both words once, at the end)          */
};

/* camera record: 16 floats per view, see DESIGN.md R0.
 *  [0..8]  R   cam_to_world rotation, row-major       (cameras.py:84, 446-477)
 *  [9..11] t   camera position (chunk-local frame)
 *  [12]    f_eff   focal length in pixels of the RENDERED image  = f * h / image_height
 *  [13,14] cxp,cyp principal point in pixels of the rendered image (pyvista path: w/2, h/2)
 *  [15]    near    faces with any vertex at camera depth <= near are discarded                 */
#define GR_CAM_FLOATS 16

typedef struct gr_ctx gr_ctx;

/* per-stage device times of the most recent profiled call, milliseconds (see gr_set_profiling) */
typedef struct gr_stage_times {
  float setup_ms;   /* k_setup_cull : transform + cull + record + tile counts      */
  float scan_ms;    /* k_scan_tiles                                                */
  float fill_ms;    /* k_fill_compile                                              */
  float raster_ms;  /* k_raster_tile (the dominant kernel)                         */
  float project_ms; /* k_winner_* : last-writer-wins pixel -> face                 */
  float vote_ms;    /* k_vote_*   : per-face accumulate                            */
  float gather_ms;  /* k_gather_texture                                            */
  int32_t raster_launches;
  int32_t views;
} gr_stage_times;

typedef struct gr_raster_stats {
  int64_t records;      /* faces that survived culling, summed over the views of the last call */
  int64_t entries;      /* (face, tile) pairs, summed over views                               */
  int64_t max_entries;  /* largest per-view entry count seen (capacity needed)                 */
  int64_t entry_cap;    /* current per-view capacity                                           */
  int32_t overflow;     /* != 0: some view exceeded entry_cap, output incomplete               */
  int32_t views_done;   /* leading views of the last call whose outputs / votes are complete    */
  int64_t blocks;       /* 64-face blocks that passed the per-view frustum cull, summed over views (single-pass binning):
                           what a culled pass has to read of the mesh is blocks x 64 x 36 B + 16 B per block tested   */
  int64_t chunk_visits; /* fused aggregation: (view, group of 64 consecutive caller face ids) pairs the vote passes visited, summed over
                           the call (a group is visited in the views whose tile pass produced a winner in it; each visit reads and
                           resets 64 winners: 64 x 8 B of k_vote_labels' algorithmic bytes)                               */
  int64_t rebinned_groups; /* times the last call binned its FIRST launch group again: a call for a mesh and image size the
                           library has learned nothing about reads that group's counts before its tile kernel runs (one host
                           round trip, once per mesh and image size) and, if a tile outgrew its slots, a face needs 48-byte
                           entries or micro lists pay, starts over with what it learned -- instead of a GR_EOVERFLOW retry  */
} gr_raster_stats;

int gr_version(void);

/* context -------------------------------------------------------------------------------------------------- */
int gr_ctx_create(int device, gr_ctx **out);
int gr_ctx_destroy(gr_ctx *ctx);
const char *gr_last_error(const gr_ctx *ctx);

/* Turn per-stage hipEvent timing on (1) or off (0).  When on, HIP events on `stream` delimit every kernel group;
 * gr_get_stage_times synchronises on them.  The stages of a raster call (set-up, tile kernel; scan / fill of the exact
 * binning) are delimited by stop events attached to the kernel launches themselves (hipExtLaunchKernelGGL): a stage runs
 * from the end of the kernel in front of it to the end of its last kernel, the launch gap in front of a kernel included --
 * timing costs the call 0.2 % this way, where events recorded between the kernels cost it 1.65 % (4.3 % on small images). */
int gr_set_profiling(gr_ctx *ctx, int enabled);

/* Tuning knobs (results never depend on them -- GR_OPT_VERTEX_ORDER excepted, which selects between two documented rule-sets --;
 * tests run every setting against the oracle). */
enum {
  GR_OPT_TILE_H_LOG2 = 2,   /* tile height: 5 (64x32, default) or 6 (64x64)                                  */
  GR_OPT_BATCH = 3,         /* views per launch group, 1..64 (default 64)                                    */
  GR_OPT_DIRECT_CAP = 6,    /* single-pass binning: entry slots per tile (default 512); 0 = always bin exactly
                               (count, scan, fill).  A call for a mesh and image size nothing is known about reads the
                               counts of its FIRST launch group before that group's tile kernel runs and starts over by
                               itself if they say so (gr_raster_stats.rebinned_groups); in any later group, or any later
                               call with more crowded views, a tile that outgrows its slots is reported by
                               gr_raster_status (GR_EOVERFLOW); the retry uses segments of the size that image needs
                               (remembered per mesh size and tile count, in the context and -- unless this option was
                               set by hand -- process-wide, so that another context for the same mesh and image size
                               starts with segments that fit) or, beyond 65536 slots, or 24 GB of entry memory for ONE
                               view, bins exactly.  Setting the option forgets what the context learned          */
  GR_OPT_VARIANT = 7,       /* mode bits (results identical; the parity tests run every one against the oracle):
                                    1 = one tile per workgroup of the tile kernel (default: chains of four consecutive tiles in
                                        large launches of light tiles, rolling chains of 16 for the fused kernel);
                                    4 = fused votes on the caller's stream (default: a side stream beside the next group's binning;
                                        the profiling scripts use it: rocprofv3 counter passes do not survive the side stream);
                                   16 = chains whatever the size of the launch (tests: small images through the chain kernels);
                                  128 = 48-byte entries always (default: 40-byte entries; images with faces of 93 pixels and more
                                        fall back to 48 bytes, remembered like the slots per tile);
                                  512 = the general ids kernel (depth output, any width) also where the plain one would run;
                                 4096 = micro lists never, 8192 = always (default: a call whose views show mostly faces of at most
                                        4 x 4 pixels -- a mesh rendered at a fraction of its photos' resolution -- teaches the
                                        library to keep, for that mesh and image size, a second list per tile for such faces,
                                        which the tile kernel point-samples one face per lane);
                                16384 = no look at the first launch group's counts (every overflow goes through gr_raster_status);
                               131072 = tile counters packed side by side whatever the image size (default: images of at most
                                        1024 tiles keep one counter per 128-byte line -- atomics on one line are served one after
                                        the other) */
  GR_OPT_SHARE_LEARNED = 8, /* 1 (default): consult and feed the process-wide table of learned slots per tile / entry forms
                               (and its file, gr_learned_cache_file); 0: this context learns for itself only.  Setting
                               GR_OPT_DIRECT_CAP by hand switches it off; this option switches it back on            */
  GR_OPT_DIRECT_BUDGET_MB = 9, /* entry memory one launch group may take, MiB (default 24576).  Scratch of the single-pass
                               binning = views per launch group (<= 64) x tiles x slots per tile x 48 B -- 9.3 GB for 64
                               views of 4000 x 3000 at the default 512 slots --; a launch group shrinks until it fits, and
                               an image whose learned slots would not fit even ONE view bins exactly instead (remembered
                               per mesh and image size like the slots themselves; other image sizes are not affected)   */
  GR_OPT_VERTEX_ORDER = 10, /* 0 (default): rule R1 of DESIGN.md -- s = c + (f q) (1 / q_z), X = floor(256 s + 0.5).  1: the same
                               perspective divide, viewport transform and snap in the ORDER OF OPERATIONS of an OpenGL pipeline, as
                               Mesa's llvmpipe -- the software GL of the reference's Dockerfile:6-13 -- executes them behind the camera
                               transform: clip = P q with P = (2 f / w, -2 f / h), ndc = clip * (1 / q_z), window = fma(ndc, size / 2,
                               size / 2), fixed = rint(256 (window - 0.5)), rows bottom-up.  The two orders put 3-12 % of the vertices
                               of a view on neighbouring 1/256 px steps, which decides 0.004 % of its pixels (95 % of the pixels on
                               which R1 and llvmpipe differ; the rest are faces llvmpipe clips at the image border:
                               profiles/r06_gl_residue.txt).  With 1 the library differs from llvmpipe on 14 of the 12 000 000 pixels
                               of a C2 view.  The principal point must be the window centre (cxp = w / 2, cyp = h / 2: the pyvista
                               camera of cameras.py:446-477); results depend on this option by design -- the oracle has the same
                               switch (oracle_raster.c R1-GL) and the parity tests run both                              */
  GR_OPT_DEBUG_LDS = 98,    /* extra dynamic LDS bytes per tile workgroup: lowers occupancy (timing experiments)     */
  GR_OPT_DEBUG = 99         /* test hook: 512 = entry slots and row counts are poisoned with 0xFF before every launch group is
                               binned (results stay right: tests/test_overflow_protocol.py)                      */
};
int gr_set_option(gr_ctx *ctx, int key, int value);

/* Persist what overflowed calls taught the library -- slots per tile and entry form per (mesh signature, tile count); the
 * signature is the face count, vertex count and vertex bounds of the upload -- in a small text file, so that a NEW process
 * starts with segments that fit (no GR_EOVERFLOW retry on its first call).  Reads `path_h` now (a missing file is fine) and
 * rewrites it whenever something new is learned (atomic rename).  Process-wide; NULL or "" switches persistence off.
 * The reference keeps its caches under CACHE_FOLDER (constants.py:18; pix2face's cache_folder argument, meshes.py:1683):
 * the Python binding points this at CACHE_FOLDER/geograster_learned.txt. */
int gr_learned_cache_file(const char *path_h);
/* Forget everything the PROCESS-WIDE table holds (contexts keep what they learned themselves; the file is not touched and the
 * file name stays set).  For callers that need a cold start -- a benchmark's "first process that ever sees the scene" leg, a
 * hermetic test -- after which gr_learned_cache_file(path) reads a file's entries into the empty table. */
int gr_learned_cache_clear(void);
int gr_get_stage_times(gr_ctx *ctx, gr_stage_times *out_h);

/* mesh -- replaces the per-view mesh + colour upload of meshes.py:1776-1817 (plotter.clear/add_mesh) and the
 * coordinate hand-over of meshes.py:1641-1676.  verts: V x 3 fp32 in the cameras' local frame; faces: F x 3 int32.
 * Borrowed: the caller keeps both alive until the next upload or gr_ctx_destroy.  Validates 0 <= index < V
 * (GR_EINDEX).  The library keeps its own de-indexed copy of the faces, ordered along a Morton curve of their
 * centroids; every id it reports is an index into the caller's `faces`, whatever their order.  Synchronises `stream`. */
int gr_mesh_upload(gr_ctx *ctx, const float *verts, const int32_t *faces, int64_t V, int64_t F, void *stream);

/* pix2face -- replaces meshes.py:1776-1836 (encode ids, VTK render, decode, background mask) for n_views
 * cameras of equal image size.  ids: n_views x h x w int32, background -1.  depth (may be NULL): n_views x h x w
 * fp32 camera-space depth of the visible face, +inf for background.  Rule-set: DESIGN.md R0-R7. */
int gr_raster_face_ids(gr_ctx *ctx, const float *cams, int n_views, int h, int w, int32_t *ids, float *depth,
                       void *stream);
/* Outcome of the last raster call (synchronises its stream; a call of one launch group that is not fused leaves its view totals
 * to be added up here, by one small kernel on that stream, instead of paying for them in every call).  GR_EOVERFLOW: the single-pass binning could not finish a launch
 * group -- a tile received more entries than its segment holds, or a face is too large (93 px and more) for the 40-byte
 * entries the call started with.  The first `views_done` views are final; the context (and the process-wide table, see
 * GR_OPT_DIRECT_CAP) now knows the segment size / entry form this mesh and image size need: call again for the remaining
 * views.  At most one such retry per cause for a given (mesh, image size). */
int gr_raster_status(gr_ctx *ctx, gr_raster_stats *out_h);

/* render_flat gather -- replaces meshes.py:1921-1937: out[p,:] = face_tex[ids[p],:] where ids[p] != -1 else NaN.
 * ids: n_pix int32; face_tex: F x C f64; out: n_pix x C f64. */
int gr_gather_texture_f64(gr_ctx *ctx, const int32_t *ids, int64_t n_pix, const double *face_tex, int64_t F, int C,
                          double *out, void *stream);

/* project_images + aggregate step for index labels -- replaces meshes.py:1987-2002 and 2057-2067 for the
 * one-hot label images of cameras/segmentor.py:33-42 + predictors/segmentor.py:37-69.
 * ids: n_views x h x w int32; labels: n_views x h x w uint8 class indices (>= C: all-zero one-hot row, still an
 * observation).  Per view the LAST pixel (row-major) of each face wins; votes[f*C + label] += 1 and counts[f] += 1
 * are ACCUMULATED into the caller's buffers (zero them before the first call). */
int gr_project_labels_u8(gr_ctx *ctx, const int32_t *ids, const uint8_t *labels, int n_views, int h, int w, int C,
                         uint32_t *votes, uint32_t *counts, int flags, void *stream);

/* same for continuous images (cameras.py:154-177 float images): img n_views x h x w x C f64 (NaN allowed).
 * sums[f*C+c] = nz(sums[f*C+c]) + nz(value) view by view, nz(NaN) = 0 -- np.nansum([summed, projection], axis=0) of
 * meshes.py:2060-2062 to the letter: a running sum that went NaN (+inf met -inf) counts as 0 at the next view of the call
 * or of the next call on the same buffers, whether that view shows the face or not; counts[f] += any(isfinite(row))
 * (2064-2067).
 * CONTRACT WHEN VIEWS ARE SHARDED over processes (geograypher_amd/distributed.py: rank r accumulates views r, r + N, ... into
 * its own buffers, the buffers are added with one all-reduce): counts are exact for any N; sums of FINITE inputs equal the
 * serial result within 1e-12 relative (addition order); for inputs with +-inf the recurrence above runs per rank over that
 * rank's views and the per-rank sums are then added, so a face that met +inf and -inf in views of DIFFERENT ranks ends NaN
 * where the serial reference -- which drops the NaN at the following view -- ends finite
 * (tests/test_distributed_gloo.py::test_float_aggregation_contract_for_non_finite_inputs_at_world_two). */
int gr_project_values_f64(gr_ctx *ctx, const int32_t *ids, const double *img, int n_views, int h, int w, int C,
                          double *sums, uint32_t *counts, int flags, void *stream);

/* project_images for ONE view, materialised like the reference generator yields it (meshes.py:1991-2002):
 * tex: F x C f64, NaN for faces no pixel maps to.  img: h x w x C f64. */
int gr_project_view_f64(gr_ctx *ctx, const int32_t *ids, const double *img, int h, int w, int C, double *tex,
                        int flags, void *stream);

/* fused pix2face + project_labels (ids never leave the chip unless ids_or_null != NULL): the
 * aggregate_projected_images fast path, meshes.py:2004-2084 over n_views cameras.  The tile rasterizer's epilogue feeds
 * the per-face winners straight from its LDS tile.  If gr_raster_status afterwards reports GR_EOVERFLOW, the votes of the
 * first gr_raster_stats.views_done views HAVE been folded into votes/counts and those of the remaining views have
 * not (the launch group that overflowed and every later one are skipped on the device): call again with the camera
 * records and label images from view `views_done` on.  No rollback of votes/counts is needed. */
int gr_raster_project_labels_u8(gr_ctx *ctx, const float *cams, const uint8_t *labels, int n_views, int h, int w,
                                int C, uint32_t *votes, uint32_t *counts, int32_t *ids_or_null, int flags,
                                void *stream);

/* save_renders epilogue (row f2) -- replaces meshes.py:1921-1937 + 2325-2337 in one pass: out[p,c] = uint8(tex[ids[p],c]),
 * with `null_value` where the pixel has no face or the value is < 0, > 255 or not finite.  out: n_pix x C uint8. */
int gr_gather_texture_u8(gr_ctx *ctx, const int32_t *ids, int64_t n_pix, const double *face_tex, int64_t F, int C,
                         int null_value, uint8_t *out, void *stream);

/* sparse index aggregation (row f3) -- replaces the loop body of TexturedPhotogrammetryMeshIndexPredictions
 * .aggregate_projected_images (derived_meshes.py:470-520) for single-channel images whose finite values are class
 * indices: per view the last pixel of each face wins (as project_images); a finite value v adds one observation:
 * counts[f] += 1 and the pair key f * n_classes + int(v) is appended to keys[*key_count ...] (device counter, capacity
 * key_cap; pairs beyond it are dropped but still counted in *key_count).  Calls APPEND: the caller zeroes *key_count and may
 * collect the pairs of many calls in one buffer before counting them once (gr_count_pairs).  Synchronises `stream` and
 * returns GR_EINDEX when a value is outside [0, n_classes) -- unless GR_FLAG_DEFER_CHECK is set: then the call only enqueues
 * work, key_count must point to two 64-bit words (both zeroed by the caller) and such a value makes key_count[1] non-zero. */
int gr_project_index_pairs(gr_ctx *ctx, const int32_t *ids, const double *img, int n_views, int h, int w,
                           int64_t n_classes, uint32_t *counts, uint64_t *keys, int64_t key_cap, uint64_t *key_count,
                           int flags, void *stream);
/* multiplicity of every distinct pair key: radix sort + run-length encode (rocPRIM via hipcub) in context scratch.
 * unique_keys / pair_counts: capacity n.  *n_unique_h (host) receives the number of distinct keys.  Synchronises. */
int gr_count_pairs(gr_ctx *ctx, uint64_t *keys, int64_t n, uint64_t *unique_keys, uint32_t *pair_counts,
                   int64_t *n_unique_h, void *stream);

/* distortion warp of an image through a cached sampling map (row f1) -- replaces utils/image.py:72-126
 * (flexible_inputs_warp -> skimage.transform.warp, mode "constant") as called by cameras.py:1092-1156 for the face-id
 * image of pix2face (meshes.py:1842-1854).  map_rows/map_cols: h_out x w_out f64, the position to sample in `in` for
 * every output pixel (cameras.py:995-1062).  Nearest neighbour = floor(x + 0.5); samples outside `in` read `fill`.
 * reference_float_roundtrip != 0 reproduces the reference's rescale-to-[0,1]-and-back truncation bit for bit
 * (value_min = min(in.min(), fill), value_range = max(in.max(), fill) - value_min as the reference computes them). */
int gr_warp_nearest_i32(gr_ctx *ctx, const int32_t *in, int h_in, int w_in, const double *map_rows,
                        const double *map_cols, int h_out, int w_out, int32_t fill, int reference_float_roundtrip,
                        double value_min, double value_range, int32_t *out, void *stream);
/* same for float64 images with C interleaved channels; order 0 (nearest) or 1 (bilinear). */
int gr_warp_f64(gr_ctx *ctx, const double *in, int h_in, int w_in, int C, const double *map_rows, const double *map_cols,
                int h_out, int w_out, int order, double fill, double *out, void *stream);

/* inverse of the lens model (row f1) -- replaces the host-side inversion of the forward distortion map by
 * scipy.interpolate.griddata on every `inversion_downsample`-th pixel (cameras.py:1045-1062, utils/indexing.py:87-150;
 * minutes at full resolution) with a dense Newton solve on the device: for every pixel (i, j) of the warped image of size
 * h x w (= int(image_height * image_scale) x int(image_width * image_scale)) the fractional pixel (row, col) of the
 * ideal image that the Metashape frame-camera model (derived_cameras.py:163-208) sends there, `fill` where that lies
 * outside the ideal image.  par_h (HOST pointer, 13 doubles): f, cx, cy, image_width, image_height, k1, k2, k3, k4, p1,
 * p2, b1, b2.  The forward map follows cameras.py:1012-1043 (model evaluated at the pixel index at scale 1, at
 * (index + 0.5) / scale otherwise).  map_rows / map_cols: h x w f64, the layout gr_warp_* consume. */
int gr_invert_distortion_f64(gr_ctx *ctx, const double *par_h, int h, int w, double image_scale, int max_iters,
                             double fill, double *map_rows, double *map_cols, void *stream);

/* get_image(image_scale) behind the file read -- replaces cameras.py:154-174: `image / 255.0` for uint8 images, then
 * skimage.transform.resize(image, (int(h * s), int(w * s))) with its defaults (order 1, mode "reflect", anti-aliasing
 * Gaussian sigma = (n_in / n_out - 1) / 2 per axis through scipy.ndimage.gaussian_filter(mode="mirror"), half-pixel-centre
 * sampling) -- called per view by project_images (meshes.py:1988 via cameras.py:866-867).  src: h_in x w_in x C image in its
 * FILE dtype (GR_DTYPE_*: the photo crosses the link as uint8, not as float64); divide_by_255 != 0 (uint8 only): values are
 * divided by 255.0 first, as get_image does; out: h_out x w_out x C f64.  Equal sizes: the conversion alone.  Agrees with
 * scikit-image 0.18.3 and with the >= 0.19 formulation (the pinned 0.21.0) to 1e-12 (tests/test_photo_resize.py).  Uses
 * context scratch (2 x h_out x w_in x C doubles). */
enum { GR_DTYPE_U8 = 0, GR_DTYPE_F32 = 1, GR_DTYPE_F64 = 2 };
int gr_resize_image_f64(gr_ctx *ctx, const void *src, int dtype, int h_in, int w_in, int C, int divide_by_255, int h_out,
                        int w_out, double *out, void *stream);

/* finalise -- meshes.py:2069-2082: summed[counts==0] = NaN; average = summed / counts.
 * votes_u32 (F x C) is converted to f64 `summed`; average and summed are F x C f64, counts_f64 is F f64. */
int gr_finalize_votes(gr_ctx *ctx, const uint32_t *votes, const uint32_t *counts, int64_t F, int C, double *average,
                      double *summed, double *counts_f64, void *stream);
int gr_finalize_sums_f64(gr_ctx *ctx, double *sums_inout, const uint32_t *counts, int64_t F, int C, double *average,
                         double *counts_f64, void *stream);

/* find_argmax_nonzero_value -- utils/indexing.py:9-32 on an F x C f64 array: argmax per row as f64, NaN when the
 * row sums to zero or holds a non-finite value. */
int gr_argmax_nonzero_f64(gr_ctx *ctx, const double *array, int64_t F, int C, double *out, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* GEOGRASTER_H */
