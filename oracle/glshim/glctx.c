/* Test infrastructure (never part of the product): an off-screen OpenGL 4.5 context on Mesa's software rasteriser
 * (llvmpipe) WITHOUT an X server, EGL or OSMesa -- this image ships libGL's Mesa DRI drivers but none of those.  The
 * swrast DRI driver is opened directly and driven through Mesa's public DRI interface (GL/internal/dri_interface.h):
 * DRI_SWRast createNewScreen2 / createContextAttribs / createNewDrawable + DRI_Core bindContext, with a do-nothing
 * swrast loader (all rendering goes to framebuffer objects).  GL entry points come from libglapi.
 *
 * Used by oracle/glshim/moderngl.py, the stand-in for the `moderngl` package that lets tests/golden/make_golden_gl.py
 * run the REFERENCE's own rgbd_3d/moderngl_renderer.py + GLSL shaders in the build container to generate golden vectors. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include <dlfcn.h>
#include <GL/internal/dri_interface.h>

static void getDrawableInfo(__DRIdrawable* d, int* x, int* y, int* w, int* h, void* lp) { *x = *y = 0; *w = 16; *h = 16; }
static void putImage(__DRIdrawable* d, int op, int x, int y, int w, int h, char* data, void* lp) {}
static void getImage(__DRIdrawable* d, int x, int y, int w, int h, char* data, void* lp) { memset(data, 0, (size_t)w * h * 4); }
static void putImage2(__DRIdrawable* d, int op, int x, int y, int w, int h, int stride, char* data, void* lp) {}
static void getImage2(__DRIdrawable* d, int x, int y, int w, int h, int stride, char* data, void* lp) { memset(data, 0, (size_t)stride * h); }
static const __DRIswrastLoaderExtension swrast_loader = {
    .base = {__DRI_SWRAST_LOADER, 3}, .getDrawableInfo = getDrawableInfo, .putImage = putImage, .getImage = getImage,
    .putImage2 = putImage2, .getImage2 = getImage2};
static const __DRIextension* loader_exts[] = {&swrast_loader.base, NULL};

static void* g_api = NULL;
static char g_err[256] = "";

const char* glctx_error(void) { return g_err; }

/* compat != 0: compatibility profile (accepts the reference's `#version 130` shaders); returns 0 on success */
int glctx_create(const char* driver_path, int major, int minor, int compat) {
  g_api = dlopen("libglapi.so.0", RTLD_NOW | RTLD_GLOBAL);
  void* h = dlopen(driver_path, RTLD_NOW | RTLD_GLOBAL);
  if (!h || !g_api) { snprintf(g_err, sizeof g_err, "dlopen: %s", dlerror()); return 1; }
  const __DRIextension** (*get)(void) = (const __DRIextension** (*)(void))dlsym(h, "__driDriverGetExtensions_swrast");
  if (!get) { snprintf(g_err, sizeof g_err, "no __driDriverGetExtensions_swrast"); return 2; }
  const __DRIextension** exts = get();
  const __DRIcoreExtension* core = NULL;
  const __DRIswrastExtension* sw = NULL;
  for (int i = 0; exts[i]; ++i) {
    if (!strcmp(exts[i]->name, __DRI_CORE)) core = (const __DRIcoreExtension*)exts[i];
    if (!strcmp(exts[i]->name, __DRI_SWRAST)) sw = (const __DRIswrastExtension*)exts[i];
  }
  if (!core || !sw || sw->base.version < 4) { snprintf(g_err, sizeof g_err, "DRI_Core / DRI_SWRast v4 missing"); return 3; }
  const __DRIconfig** configs = NULL;
  __DRIscreen* scr = sw->createNewScreen2(0, loader_exts, exts, &configs, NULL);
  if (!scr || !configs || !configs[0]) { snprintf(g_err, sizeof g_err, "createNewScreen2 failed"); return 4; }
  uint32_t attribs[] = {__DRI_CTX_ATTRIB_MAJOR_VERSION, (uint32_t)major, __DRI_CTX_ATTRIB_MINOR_VERSION, (uint32_t)minor};
  unsigned err = 0;
  __DRIcontext* ctx = sw->createContextAttribs(scr, compat ? __DRI_API_OPENGL : __DRI_API_OPENGL_CORE, configs[0], NULL, 2,
                                               attribs, &err, NULL);
  if (!ctx) { snprintf(g_err, sizeof g_err, "createContextAttribs failed (error %u)", err); return 5; }
  __DRIdrawable* dr = sw->createNewDrawable(scr, configs[0], NULL);
  if (!dr || !core->bindContext(ctx, dr, dr)) { snprintf(g_err, sizeof g_err, "bindContext failed"); return 6; }
  return 0;
}

void* glctx_proc(const char* name) {
  if (!g_api) return NULL;
  void* (*gpa)(const char*) = (void* (*)(const char*))dlsym(g_api, "_glapi_get_proc_address");
  return gpa ? gpa(name) : NULL;
}
