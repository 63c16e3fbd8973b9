/*
 * tests/golden/drisw_loader.c -- BUILD-CONTAINER TOOL (test infrastructure; not product code, never runs on the GPU box).
 *
 * A minimal loader for Mesa's software-rasterizer DRI driver (/usr/lib/x86_64-linux-gnu/dri/swrast_dri.so of the
 * libgl1-mesa-dri 23.2.1 package: llvmpipe), playing the part an X server / libGL plays for it, so that
 * tests/golden/make_golden_gl.py can render the reference's face-id image (geograypher/meshes/meshes.py:1776-1836) with the
 * software GL family the reference's own Dockerfile:6-13 (libgl1 + xvfb) renders with -- without an X server.
 *
 * Interface followed: /usr/include/GL/internal/dri_interface.h (mesa-common-dev): __driDriverGetExtensions_swrast ->
 * DRI_Core + DRI_SWRast; the loader offers DRI_SWRastLoader (drawable size, put/get image into a private pixel buffer).
 * All rendering of the golden generator goes to a framebuffer object, so the window-system drawable is a 16x16 dummy.
 * GL entry points come from the shared glapi (libglapi.so.0, which the driver links against): _glapi_get_proc_address.
 */
#include <GL/internal/dri_interface.h>
#include <dlfcn.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static char g_err[512];
static void *g_driver, *g_glapi;
static const __DRIcoreExtension *g_core;
static const __DRIswrastExtension *g_swrast;
static __DRIscreen *g_screen;
static __DRIcontext *g_ctx;
static __DRIdrawable *g_draw;
static const __DRIconfig **g_configs;
static void *(*g_get_proc)(const char *);

enum { DUMMY_W = 16, DUMMY_H = 16 };
static char g_pixels[DUMMY_W * DUMMY_H * 4];

static void ld_get_drawable_info(__DRIdrawable *d, int *x, int *y, int *w, int *h, void *priv) {
  (void)d; (void)priv;
  *x = 0; *y = 0; *w = DUMMY_W; *h = DUMMY_H;
}
static void ld_put_image(__DRIdrawable *d, int op, int x, int y, int w, int h, char *data, void *priv) {
  (void)d; (void)op; (void)x; (void)y; (void)w; (void)h; (void)data; (void)priv;   /* nothing is ever presented */
}
static void ld_get_image(__DRIdrawable *d, int x, int y, int w, int h, char *data, void *priv) {
  (void)d; (void)x; (void)y; (void)priv;
  memset(data, 0, (size_t)w * h * 4);
}
static void ld_put_image2(__DRIdrawable *d, int op, int x, int y, int w, int h, int stride, char *data, void *priv) {
  (void)d; (void)op; (void)x; (void)y; (void)w; (void)h; (void)stride; (void)data; (void)priv;
}
static void ld_get_image2(__DRIdrawable *d, int x, int y, int w, int h, int stride, char *data, void *priv) {
  (void)d; (void)x; (void)y; (void)w; (void)priv;
  memset(data, 0, (size_t)stride * h);
}

static const __DRIswrastLoaderExtension g_loader_ext = {
    .base = {__DRI_SWRAST_LOADER, 3},
    .getDrawableInfo = ld_get_drawable_info,
    .putImage = ld_put_image,
    .getImage = ld_get_image,
    .putImage2 = ld_put_image2,
    .getImage2 = ld_get_image2,
};
static const __DRIextension *g_loader_exts[] = {&g_loader_ext.base, NULL};

const char *drisw_error(void) { return g_err; }

void *drisw_get_proc(const char *name) { return g_get_proc ? g_get_proc(name) : NULL; }

/* core_profile != 0: OpenGL 3.3 core (what VTK 9 asks for); 0: compatibility profile.  Returns 0 on success. */
int drisw_open(const char *driver_path, int core_profile) {
  g_err[0] = 0;
  if (g_ctx) return 0;
  g_glapi = dlopen("libglapi.so.0", RTLD_NOW | RTLD_GLOBAL);
  if (!g_glapi) { snprintf(g_err, sizeof g_err, "dlopen libglapi.so.0: %s", dlerror()); return -1; }
  g_get_proc = (void *(*)(const char *))dlsym(g_glapi, "_glapi_get_proc_address");
  g_driver = dlopen(driver_path, RTLD_NOW | RTLD_GLOBAL);
  if (!g_driver) { snprintf(g_err, sizeof g_err, "dlopen %s: %s", driver_path, dlerror()); return -1; }
  const __DRIextension **(*get_exts)(void) =
      (const __DRIextension **(*)(void))dlsym(g_driver, __DRI_DRIVER_GET_EXTENSIONS "_swrast");
  if (!get_exts || !g_get_proc) { snprintf(g_err, sizeof g_err, "driver entry points not found"); return -1; }
  const __DRIextension **exts = get_exts();
  for (int i = 0; exts && exts[i]; ++i) {
    if (!strcmp(exts[i]->name, __DRI_CORE)) g_core = (const __DRIcoreExtension *)exts[i];
    if (!strcmp(exts[i]->name, __DRI_SWRAST)) g_swrast = (const __DRIswrastExtension *)exts[i];
  }
  if (!g_core || !g_swrast || g_swrast->base.version < 4) {
    snprintf(g_err, sizeof g_err, "driver lacks DRI_Core / DRI_SWRast >= 4");
    return -1;
  }
  g_screen = g_swrast->createNewScreen2(0, g_loader_exts, exts, &g_configs, NULL);
  if (!g_screen) { snprintf(g_err, sizeof g_err, "createNewScreen2 failed"); return -1; }
  /* an RGBA8 single-buffered config; depth of the window-system buffer is irrelevant (rendering goes to an FBO) */
  const __DRIconfig *pick = NULL;
  for (int i = 0; g_configs[i]; ++i) {
    unsigned r = 0, a = 0, db = 1, samples = 1;
    g_core->getConfigAttrib(g_configs[i], __DRI_ATTRIB_RED_SIZE, &r);
    g_core->getConfigAttrib(g_configs[i], __DRI_ATTRIB_ALPHA_SIZE, &a);
    g_core->getConfigAttrib(g_configs[i], __DRI_ATTRIB_DOUBLE_BUFFER, &db);
    g_core->getConfigAttrib(g_configs[i], __DRI_ATTRIB_SAMPLES, &samples);
    if (r == 8 && a == 8 && !db && samples == 0) { pick = g_configs[i]; break; }
  }
  if (!pick) pick = g_configs[0];
  unsigned err = 0;
  const uint32_t attribs[] = {__DRI_CTX_ATTRIB_MAJOR_VERSION, 3, __DRI_CTX_ATTRIB_MINOR_VERSION, 3};
  g_ctx = g_swrast->createContextAttribs(g_screen, core_profile ? __DRI_API_OPENGL_CORE : __DRI_API_OPENGL, pick, NULL,
                                         core_profile ? 2 : 0, attribs, &err, NULL);
  if (!g_ctx) { snprintf(g_err, sizeof g_err, "createContextAttribs failed (error %u)", err); return -1; }
  g_draw = g_swrast->createNewDrawable(g_screen, pick, NULL);
  if (!g_draw) { snprintf(g_err, sizeof g_err, "createNewDrawable failed"); return -1; }
  if (!g_core->bindContext(g_ctx, g_draw, g_draw)) { snprintf(g_err, sizeof g_err, "bindContext failed"); return -1; }
  (void)g_pixels;
  return 0;
}

void drisw_close(void) {
  if (g_ctx) { g_core->unbindContext(g_ctx); g_core->destroyContext(g_ctx); g_ctx = NULL; }
  if (g_draw) { g_core->destroyDrawable(g_draw); g_draw = NULL; }
  if (g_screen) { g_core->destroyScreen(g_screen); g_screen = NULL; }
}
