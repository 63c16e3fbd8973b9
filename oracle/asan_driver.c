/*
 * oracle/asan_driver.c -- TEST INFRASTRUCTURE ONLY.  Edge-case driver of the CPU oracle for the sanitizer build
 * (`make -C oracle asan`: -fsanitize=address,undefined).  It feeds oracle_raster.c / oracle_envelope.c the inputs that
 * stress their index arithmetic -- 1x1 images, images that are not multiples of anything, faces far larger than the
 * image, faces behind and across the near plane (the R7 clipper and its 8-vertex polygon buffers), zero-area and
 * coincident faces, NaN / inf vertices, a camera with near <= 0, background-only views, labels >= C, face id -1 aliasing
 * the last face -- and checks orc_raster_spec == orc_raster_fast on every scene.  Exit code 0 = no sanitizer report and no
 * mismatch.  Deterministic (own LCG), no files.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

int orc_raster_spec(const float *, const int32_t *, int64_t, int64_t, const float *, int, int, int32_t *, float *, int32_t *);
int orc_raster_fast(const float *, const int32_t *, int64_t, int64_t, const float *, int, int, int32_t *, float *, int32_t *);
int orc_raster_views(const float *, const int32_t *, int64_t, int64_t, const float *, int, int, int, int32_t *, int);
int orc_project_labels(const int32_t *, const uint8_t *, int, int, int64_t, int, int, uint32_t *, uint32_t *, int64_t *);
int orc_envelope(const float *, const int32_t *, int64_t, const float *, int, int, double, double, uint8_t *, int32_t *,
                 double *, double *, int32_t *, uint8_t *, double *, double *);
int orc_raster_float(const float *, const int32_t *, int64_t, const float *, int, int, int32_t *, double *);

static uint64_t state = 0x9E3779B97F4A7C15ull;
static double rnd(void) { state = state * 6364136223846793005ull + 1442695040888963407ull; return (double)(state >> 11) / 9007199254740992.0; }

static void nadir_cam(float *cam, double x, double y, double z, double f, int h, int w, double near_) {
  const float R[9] = {1, 0, 0, 0, -1, 0, 0, 0, -1}; /* camera looks down -z, image y down */
  memcpy(cam, R, sizeof(R));
  cam[9] = (float)x; cam[10] = (float)y; cam[11] = (float)z;
  cam[12] = (float)f; cam[13] = (float)(w / 2.0); cam[14] = (float)(h / 2.0); cam[15] = (float)near_;
}

static int run_scene(int n, int h, int w, double size_lo, double size_hi, double zcam, double near_, int poison) {
  float *verts = malloc(sizeof(float) * 9 * (size_t)n);
  int32_t *faces = malloc(sizeof(int32_t) * 3 * (size_t)n);
  for (int f = 0; f < n; ++f) {
    const double cx = (rnd() - 0.5) * 8, cy = (rnd() - 0.5) * 8, cz = (rnd() - 0.5) * 1.0;
    const double s = exp(log(size_lo) + rnd() * (log(size_hi) - log(size_lo)));
    for (int k = 0; k < 3; ++k) {
      verts[9 * f + 3 * k + 0] = (float)(cx + (rnd() - 0.5) * s);
      verts[9 * f + 3 * k + 1] = (float)(cy + (rnd() - 0.5) * s);
      verts[9 * f + 3 * k + 2] = (float)(cz + (rnd() - 0.5) * s * 0.3);
      faces[3 * f + k] = 3 * f + k;
    }
  }
  if (poison && n >= 12) {
    verts[0] = NAN; verts[9 + 1] = INFINITY; verts[18 + 2] = -INFINITY;               /* non-finite vertices          */
    for (int k = 0; k < 3; ++k) verts[27 + 3 * k + 2] = (float)(zcam + 5);              /* behind the camera            */
    verts[36 + 2] = (float)(zcam + 5);                                                   /* straddles the near plane     */
    memcpy(verts + 45 + 3, verts + 45, 3 * sizeof(float));                               /* zero area                    */
    memcpy(verts + 54, verts + 63, 9 * sizeof(float));                                   /* coincident faces             */
    for (int k = 0; k < 3; ++k) { verts[72 + 3 * k] *= 1e4f; verts[72 + 3 * k + 1] *= 1e4f; } /* beyond the guard band */
    verts[81 + 2] = (float)(zcam - 1e-4);                                                /* a hair in front of the lens  */
  }
  float cam[16];
  nadir_cam(cam, (rnd() - 0.5), (rnd() - 0.5), zcam, 0.6 * (h > w ? h : w), h, w, near_);
  const size_t np = (size_t)h * w;
  int32_t *a = malloc(4 * np), *b = malloc(4 * np), *zb = malloc(4 * np);
  float *da = malloc(4 * np), *db = malloc(4 * np);
  orc_raster_spec(verts, faces, 3 * n, n, cam, h, w, a, da, zb);
  orc_raster_fast(verts, faces, 3 * n, n, cam, h, w, b, db, zb);
  int bad = memcmp(a, b, 4 * np) != 0 || memcmp(da, db, 4 * np) != 0;
  /* aggregation stage on these ids: labels include the ignore value, -1 aliases the last face */
  uint8_t *lab = malloc(np);
  for (size_t p = 0; p < np; ++p) lab[p] = (uint8_t)((p * 7u) % 6u == 5u ? 255u : (p * 7u) % 6u);
  uint32_t *votes = calloc((size_t)n * 4, 4), *counts = calloc((size_t)n, 4);
  int64_t *winner = malloc(sizeof(int64_t) * (size_t)n);
  bad |= orc_project_labels(a, lab, h, w, n, 4, 1, votes, counts, winner) != 0;
  bad |= orc_project_labels(a, lab, h, w, n, 4, 0, votes, counts, winner) != 0;
  /* envelope classifier and the second rasterizer */
  uint8_t *cls = malloc(np), *sure = malloc(np);
  int32_t *eid = malloc(4 * np), *fa = malloc(4 * np);
  double *za = malloc(8 * np), *zbb = malloc(8 * np), *zlo = malloc(8 * np), *zhi = malloc(8 * np);
  orc_envelope(verts, faces, n, cam, h, w, 1.0 / 256 + 2e-3, 1e-5, cls, eid, za, zbb, fa, sure, zlo, zhi);
  orc_raster_float(verts, faces, n, cam, h, w, eid, za);
  /* two views through the threaded entry point */
  float cams2[32];
  memcpy(cams2, cam, sizeof(cam)); memcpy(cams2 + 16, cam, sizeof(cam));
  int32_t *two = malloc(8 * np);
  orc_raster_views(verts, faces, 3 * n, n, cams2, 2, h, w, two, 1);
  bad |= memcmp(two, b, 4 * np) != 0 || memcmp(two + np, b, 4 * np) != 0;
  free(verts); free(faces); free(a); free(b); free(zb); free(da); free(db); free(lab); free(votes); free(counts);
  free(winner); free(cls); free(sure); free(eid); free(fa); free(za); free(zbb); free(zlo); free(zhi); free(two);
  return bad;
}

int main(void) {
  int bad = 0, scenes = 0;
  const int sizes[][2] = {{1, 1}, {3, 70}, {65, 33}, {64, 64}, {97, 131}, {200, 257}};
  for (int s = 0; s < 6; ++s)
    for (int rep = 0; rep < 4; ++rep) {
      const int n = rep == 0 ? 12 : (rep == 1 ? 200 : (rep == 2 ? 1500 : 40));
      const double hi = rep == 3 ? 400.0 : (rep == 2 ? 0.3 : 30.0);       /* faces far larger than the image at rep 3 */
      const double zcam = rep % 2 ? 6.0 : 0.3;                             /* the second one sits inside the scene      */
      const double near_ = (s == 5 && rep == 3) ? -1.0 : 0.02;             /* a camera the clipper must refuse          */
      bad |= run_scene(n, sizes[s][0], sizes[s][1], 0.01, hi, zcam, near_, 1);
      ++scenes;
    }
  printf("%s: %d scenes\n", bad ? "MISMATCH" : "ok", scenes);
  return bad ? 1 : 0;
}
