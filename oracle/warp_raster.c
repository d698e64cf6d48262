/* Software rasteriser + per-view shading/aggregation — TEST INFRASTRUCTURE (the warp oracle).
 *
 * CPU restatement of what the reference delegates to OpenGL (rgbd_3d/moderngl_renderer.py:198-202,
 * 307-315 and rgbd_3d/shaders/aggregation.{vsh,fsh,csh}): each source mesh is drawn ALONE with a
 * `<` depth test into a 24-bit depth buffer, fragments are shaded with the view-angle weight, and
 * the views are blended per pixel.  It follows the GL 4.3 rasterisation rules (near-plane clipping of
 * primitives in clip space, pixel centres, top-left fill rule, perspective-correct smooth varyings,
 * window-space-linear depth, gl_FrontFacing from the signed window area, NEAREST texel fetch) and is
 * PINNED TO REAL OPENGL: tests/golden/make_golden_gl.py runs the reference's own renderer + shaders on
 * Mesa llvmpipe (oracle/glshim/) and tests/test_warp_cpu.py compares this rasteriser with those outputs
 * (7 scenes incl. the 26-view `3x9` viewset and near-plane clipping: no mask pixel differs).
 *
 * Deliberately formulated differently from the HIP kernel (ivid_amd/csrc/warp.hip evaluates 2-D
 * homogeneous edge functions and never clips): here a triangle is CLIPPED against the near plane
 * z_clip >= -w_clip (Sutherland-Hodgman in clip space, double precision, carrying the barycentric
 * coordinates of the original triangle on every new vertex), the resulting polygon is projected to
 * window coordinates, fan-triangulated and each sub-triangle scanned with classic screen-space
 * barycentrics; perspective correction divides by w.  Triangles with a vertex behind the eye
 * (frustum skirt / discontinuity sheets seen from a far-away camera) are therefore checked too.
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

static void tri_vertices(int t, int P, int ft, int* vi) {
  int quad = t >> 1, Q = P - 1, qr = quad / Q, qc = quad % Q;
  int i00 = qr * P + qc, i01 = i00 + 1, i10 = i00 + P, i11 = i10 + 1;
  if ((t & 1) == 0) { vi[0] = i01; vi[1] = i00; vi[2] = ft ? i11 : i10; }
  else { vi[0] = i10; vi[1] = i11; vi[2] = ft ? i00 : i01; }
}

/* clip-space vertex + barycentric coordinates with respect to the ORIGINAL triangle */
typedef struct { double x, y, z, w, b[3]; } CV;
/* window-space vertex of the clipped polygon */
typedef struct { double sx, sy, zn, w, b[3]; } WV;

/* gl_Position = u_projection * u_modelview * vec4(i_position, 1): fp32 like the vertex shader */
static void clip_vertices(const float* V, const int* vi, const float* m, CV* c) {
  for (int k = 0; k < 3; ++k) {
    const float* p = V + (size_t)vi[k] * 9;
    c[k].x = m[0] * p[0] + m[1] * p[1] + m[2] * p[2] + m[3];
    c[k].y = m[4] * p[0] + m[5] * p[1] + m[6] * p[2] + m[7];
    c[k].z = m[8] * p[0] + m[9] * p[1] + m[10] * p[2] + m[11];
    c[k].w = m[12] * p[0] + m[13] * p[1] + m[14] * p[2] + m[15];
    c[k].b[0] = c[k].b[1] = c[k].b[2] = 0.0;
    c[k].b[k] = 1.0;
  }
}

/* Sutherland-Hodgman against the near plane  d(v) = z + w >= 0; returns the vertex count (0, 3 or 4) */
static int clip_near(const CV* in, int n, CV* out) {
  int m = 0;
  for (int i = 0; i < n; ++i) {
    const CV *a = &in[i], *b = &in[(i + 1) % n];
    double da = a->z + a->w, db = b->z + b->w;
    if (da >= 0.0) out[m++] = *a;
    if ((da >= 0.0) != (db >= 0.0)) {
      double t = da / (da - db);
      CV v;
      v.x = a->x + t * (b->x - a->x); v.y = a->y + t * (b->y - a->y);
      v.z = a->z + t * (b->z - a->z); v.w = a->w + t * (b->w - a->w);
      for (int k = 0; k < 3; ++k) v.b[k] = a->b[k] + t * (b->b[k] - a->b[k]);
      out[m++] = v;
    }
  }
  return m;
}

/* Edge function of (P -> Q) at C, window coordinates (y down); with the top-left rule for points ON the edge:
 * after orientation (inside positive) an edge owns its points iff it is a left edge (gradient points right) or a
 * top edge (horizontal, interior below = larger y). */
static int edge_inside(double px, double py, double qx, double qy, double cx, double cy, double orient) {
  double A = -(qy - py) * orient, B = (qx - px) * orient;
  double e = A * (cx - px) + B * (cy - py);
  if (e > 0.0) return 1;
  if (e < 0.0) return 0;
  return A > 0.0 || (A == 0.0 && B > 0.0);
}

/* Rasterise one triangle of the mesh into (depth24, tri, bary, front): `<` test, first drawn wins ties.
 * Returns 1 when the triangle had to be clipped against the near plane. */
static int raster_triangle(const float* V, const unsigned char* diag, int t, int P, const float* mvp, int R,
                           uint32_t* depth24, int32_t* tri, double* bary, unsigned char* front) {
  int vi[3];
  tri_vertices(t, P, diag[t >> 1], vi);
  CV c[3], poly[8];
  clip_vertices(V, vi, mvp, c);
  int clipped = (c[0].z + c[0].w < 0.0) || (c[1].z + c[1].w < 0.0) || (c[2].z + c[2].w < 0.0);
  int n = clip_near(c, 3, poly);
  if (n < 3) return clipped;
  float pad[3];
  for (int k = 0; k < 3; ++k) pad[k] = (float)((((int)V[(size_t)vi[k] * 9 + 8]) >> 1) & 1);
  WV w[8];
  for (int i = 0; i < n; ++i) {
    if (!(poly[i].w > 0.0)) return clipped; /* on the near plane w = near > 0; a degenerate projection is dropped */
    w[i].sx = (poly[i].x / poly[i].w + 1.0) * 0.5 * R;
    w[i].sy = (1.0 - poly[i].y / poly[i].w) * 0.5 * R; /* row 0 = top */
    w[i].zn = poly[i].z / poly[i].w;
    w[i].w = poly[i].w;
    memcpy(w[i].b, poly[i].b, sizeof(w[i].b));
  }
  /* signed area of the polygon in the y-down window frame; CCW in a y-up frame (= front face) is negative here */
  double area2 = 0.0;
  for (int i = 0; i < n; ++i) area2 += w[i].sx * w[(i + 1) % n].sy - w[(i + 1) % n].sx * w[i].sy;
  if (area2 == 0.0) return clipped;
  const int is_front = area2 < 0.0;
  const double orient = area2 > 0.0 ? 1.0 : -1.0;
  for (int f = 1; f + 1 < n; ++f) { /* fan (w[0], w[f], w[f+1]) */
    const WV* q[3] = {&w[0], &w[f], &w[f + 1]};
    double d = (q[1]->sy - q[2]->sy) * (q[0]->sx - q[2]->sx) + (q[2]->sx - q[1]->sx) * (q[0]->sy - q[2]->sy);
    if (d == 0.0) continue;
    double xmin = fmin(q[0]->sx, fmin(q[1]->sx, q[2]->sx)), xmax = fmax(q[0]->sx, fmax(q[1]->sx, q[2]->sx));
    double ymin = fmin(q[0]->sy, fmin(q[1]->sy, q[2]->sy)), ymax = fmax(q[0]->sy, fmax(q[1]->sy, q[2]->sy));
    if (xmax < 0.0 || ymax < 0.0 || xmin > R || ymin > R) continue;
    int x0 = xmin - 0.5 < 0.0 ? 0 : (int)floor(xmin - 0.5), y0 = ymin - 0.5 < 0.0 ? 0 : (int)floor(ymin - 0.5);
    int x1 = xmax - 0.5 > R - 1 ? R - 1 : (int)ceil(xmax - 0.5), y1 = ymax - 0.5 > R - 1 ? R - 1 : (int)ceil(ymax - 0.5);
    for (int y = y0; y <= y1; ++y)
      for (int x = x0; x <= x1; ++x) {
        double cx = x + 0.5, cy = y + 0.5;
        if (!edge_inside(q[0]->sx, q[0]->sy, q[1]->sx, q[1]->sy, cx, cy, orient)) continue;
        if (!edge_inside(q[1]->sx, q[1]->sy, q[2]->sx, q[2]->sy, cx, cy, orient)) continue;
        if (!edge_inside(q[2]->sx, q[2]->sy, q[0]->sx, q[0]->sy, cx, cy, orient)) continue;
        double l[3];
        l[0] = ((q[1]->sy - q[2]->sy) * (cx - q[2]->sx) + (q[2]->sx - q[1]->sx) * (cy - q[2]->sy)) / d;
        l[1] = ((q[2]->sy - q[0]->sy) * (cx - q[2]->sx) + (q[0]->sx - q[2]->sx) * (cy - q[2]->sy)) / d;
        l[2] = 1.0 - l[0] - l[1];
        double zn = l[0] * q[0]->zn + l[1] * q[1]->zn + l[2] * q[2]->zn;
        if (zn < -1.0 || zn > 1.0) continue; /* far plane (near was clipped; rounding may leave -1 - eps) */
        /* perspective-correct weights of the ORIGINAL vertices */
        double pb[3] = {0, 0, 0}, qs = 0.0;
        for (int i = 0; i < 3; ++i) {
          double qi = l[i] / q[i]->w;
          qs += qi;
          for (int k = 0; k < 3; ++k) pb[k] += qi * q[i]->b[k];
        }
        for (int k = 0; k < 3; ++k) pb[k] /= qs;
        if (!is_front) { /* aggregation.fsh:22-23: back-facing fragment with interpolated padding flag > 0.001 -> discard */
          double pv = pb[0] * pad[0] + pb[1] * pad[1] + pb[2] * pad[2];
          if (pv > 0.001) continue;
        }
        float depth = (float)(0.5 * zn + 0.5);
        if (depth < 0.f) depth = 0.f;
        if (depth > 1.f) depth = 1.f;
        uint32_t d24 = (uint32_t)(depth * 16777215.0f + 0.5f);
        int i = y * R + x;
        if (d24 < depth24[i]) { /* '<': first drawn wins ties */
          depth24[i] = d24;
          tri[i] = t;
          front[i] = (unsigned char)is_front;
          bary[3 * (size_t)i] = pb[0]; bary[3 * (size_t)i + 1] = pb[1]; bary[3 * (size_t)i + 2] = pb[2];
        }
      }
  }
  return clipped;
}

/* Draw one mesh alone.  Outputs per pixel: 24-bit depth, winning triangle (-1 = none), the perspective-correct
 * barycentrics of that fragment and its facing.  Returns the number of triangles that crossed the near plane. */
int oracle_raster(const float* V, const unsigned char* diag, int S, const float* mvp, int R, uint32_t* depth24,
                  int32_t* tri, double* bary, unsigned char* front) {
  int P = S + 2, ntri = 2 * (P - 1) * (P - 1), clipped = 0;
  for (int i = 0; i < R * R; ++i) { depth24[i] = 0xffffffffu; tri[i] = -1; front[i] = 0; }
  for (int t = 0; t < ntri; ++t) clipped += raster_triangle(V, diag, t, P, mvp, R, depth24, tri, bary, front);
  return clipped;
}

/* acc: float [R*R][8] = colour rgba sums, depth sum/weight, mask depth/colour counts (aggregation.csh) */
void oracle_shade_aggregate(const float* V, const unsigned char* diag, const float* colors, const float* campos, int S,
                            int R, const uint32_t* depth24, const int32_t* tri, const double* bary,
                            const unsigned char* front, float* acc) {
  int P = S + 2;
  for (int i = 0; i < R * R; ++i) {
    if (tri[i] < 0) continue;
    int vi[3];
    tri_vertices(tri[i], P, diag[tri[i] >> 1], vi);
    float depth = (float)depth24[i] / 16777215.0f;
    float col[3] = {0, 0, 0}, wgt = 0.f;
    if (front[i]) {
      float at[8] = {0}, fe = 0, fp = 0, fr = 0;
      for (int k = 0; k < 3; ++k) {
        float bw = (float)bary[3 * (size_t)i + k];
        const float* p = V + (size_t)vi[k] * 9;
        float nl = 1.0f / sqrtf(p[3] * p[3] + p[4] * p[4] + p[5] * p[5]);
        at[0] += bw * p[0]; at[1] += bw * p[1]; at[2] += bw * p[2];
        at[3] += bw * p[3] * nl; at[4] += bw * p[4] * nl; at[5] += bw * p[5] * nl;
        at[6] += bw * p[6]; at[7] += bw * p[7];
        int fl = (int)p[8];
        fe += bw * (fl & 1); fp += bw * ((fl >> 1) & 1); fr += bw * ((fl >> 2) & 1);
      }
      int tx = (int)floorf(at[6] * S), ty = (int)floorf(at[7] * S);
      if (tx < 0) tx = 0;
      if (tx > S - 1) tx = S - 1;
      if (ty < 0) ty = 0;
      if (ty > S - 1) ty = S - 1;
      const float* tc = colors + ((size_t)ty * S + tx) * 3;
      col[0] = tc[0]; col[1] = tc[1]; col[2] = tc[2];
      float dx = campos[0] - at[0], dy = campos[1] - at[1], dz = campos[2] - at[2];
      float dl = 1.0f / sqrtf(dx * dx + dy * dy + dz * dz), nl = 1.0f / sqrtf(at[3] * at[3] + at[4] * at[4] + at[5] * at[5]);
      float wv = (dx * at[3] + dy * at[4] + dz * at[5]) * dl * nl;
      wv = fminf(fmaxf(wv, 0.f), 1.f);
      wv = expf(fmaxf(-acosf(wv) * 20.f, -50.f));
      wv = fmaxf(wv, 1e-4f);
      if (!(fr < 0.999f)) wv *= 1e-8f;
      if (fp > 0.001f || fe > 0.999f) wv = 1e-16f;
      wgt = fmaxf(wv, 1e-16f);
    }
    float* a = acc + (size_t)i * 8;
    float wd = wgt > 1e-14f ? 1.0f : (wgt > 0.0f ? 1e-8f : 0.0f);
    if (fabsf(a[5] - 1e-8f) < 1e-8f && fabsf(wd - 1e-8f) < 1e-8f) {
      if (depth * 1e-8f > a[4]) {
        a[4] = depth * 1e-8f; a[5] = 1e-8f;
        a[0] = col[0] * wgt; a[1] = col[1] * wgt; a[2] = col[2] * wgt; a[3] = wgt;
      }
    } else {
      a[4] += depth * wd; a[5] += wd;
      a[0] += col[0] * wgt; a[1] += col[1] * wgt; a[2] += col[2] * wgt; a[3] += wgt;
    }
    a[6] += wgt > 1e-14f ? 1.0f : 0.0f;
    a[7] += wgt > 1e-6f ? 1.0f : 0.0f;
  }
}
