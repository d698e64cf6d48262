/* Software rasteriser + per-view shading/aggregation — TEST INFRASTRUCTURE (the warp oracle).
 *
 * CPU restatement of what the reference delegates to OpenGL (rgbd_3d/moderngl_renderer.py:198-202,
 * 307-315 and rgbd_3d/shaders/aggregation.{vsh,fsh,csh}): each source mesh is drawn ALONE with a
 * `<` depth test into a 24-bit depth buffer, fragments are shaded with the view-angle weight, and
 * the views are blended per pixel.  PARITY UNPINNED: no OpenGL/EGL exists in the build container, so
 * this follows the GL 4.3 rasterisation rules (pixel centres, perspective-correct smooth varyings,
 * window-space-linear depth, gl_FrontFacing from the signed window area, NEAREST texel fetch) rather
 * than outputs of the reference itself.
 *
 * Deliberately formulated differently from the HIP kernel (ivid_amd/csrc/warp.hip uses 2-D homogeneous
 * edge functions): here vertices are projected to window coordinates, barycentrics are the classic
 * screen-space ones and perspective correction divides by w — valid because every vertex has w > 0
 * for cameras on the unit sphere looking at the origin (asserted via the return value).
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

typedef struct { double sx[3], sy[3], zn[3], w[3]; double area; int vi[3]; int ok; } Tri;

static Tri setup(const float* V, const unsigned char* diag, int t, int P, const float* m, int R) {
  Tri s; s.ok = 1;
  tri_vertices(t, P, diag[t >> 1], s.vi);
  for (int k = 0; k < 3; ++k) {
    const float* p = V + (size_t)s.vi[k] * 9;
    float cx = m[0] * p[0] + m[1] * p[1] + m[2] * p[2] + m[3];
    float cy = m[4] * p[0] + m[5] * p[1] + m[6] * p[2] + m[7];
    float cz = m[8] * p[0] + m[9] * p[1] + m[10] * p[2] + m[11];
    float cw = m[12] * p[0] + m[13] * p[1] + m[14] * p[2] + m[15];
    if (!(cw > 1e-6f)) s.ok = 0;
    s.w[k] = cw;
    s.sx[k] = ((double)cx / cw + 1.0) * 0.5 * R;
    s.sy[k] = (1.0 - (double)cy / cw) * 0.5 * R; /* row 0 = top */
    s.zn[k] = (double)cz / cw;
  }
  /* signed area in a y-up frame: CCW = front (sy is y-down, so flip the sign) */
  s.area = -((s.sx[1] - s.sx[0]) * (s.sy[2] - s.sy[0]) - (s.sx[2] - s.sx[0]) * (s.sy[1] - s.sy[0]));
  return s;
}

/* screen-space barycentrics of pixel centre (px,py); returns 0 when outside */
static int bary(const Tri* s, double px, double py, double* l) {
  double d = (s->sy[1] - s->sy[2]) * (s->sx[0] - s->sx[2]) + (s->sx[2] - s->sx[1]) * (s->sy[0] - s->sy[2]);
  if (d == 0.0) return 0;
  l[0] = ((s->sy[1] - s->sy[2]) * (px - s->sx[2]) + (s->sx[2] - s->sx[1]) * (py - s->sy[2])) / d;
  l[1] = ((s->sy[2] - s->sy[0]) * (px - s->sx[2]) + (s->sx[0] - s->sx[2]) * (py - s->sy[2])) / d;
  l[2] = 1.0 - l[0] - l[1];
  return l[0] >= 0.0 && l[1] >= 0.0 && l[2] >= 0.0;
}

/* returns the number of triangles skipped because a vertex had w <= 0 (must be 0 for a valid comparison) */
int oracle_raster(const float* V, const unsigned char* diag, int S, const float* mvp, int R, uint32_t* depth24,
                  int32_t* tri) {
  int P = S + 2, ntri = 2 * (P - 1) * (P - 1), skipped = 0;
  for (int i = 0; i < R * R; ++i) { depth24[i] = 0xffffffffu; tri[i] = -1; }
  for (int t = 0; t < ntri; ++t) {
    Tri s = setup(V, diag, t, P, mvp, R);
    if (!s.ok) { skipped++; continue; }
    if (s.area == 0.0) continue;
    double xmin = fmin(s.sx[0], fmin(s.sx[1], s.sx[2])), xmax = fmax(s.sx[0], fmax(s.sx[1], s.sx[2]));
    double ymin = fmin(s.sy[0], fmin(s.sy[1], s.sy[2])), ymax = fmax(s.sy[0], fmax(s.sy[1], s.sy[2]));
    int x0 = (int)floor(xmin - 0.5), x1 = (int)ceil(xmax - 0.5), y0 = (int)floor(ymin - 0.5), y1 = (int)ceil(ymax - 0.5);
    if (x0 < 0) x0 = 0; if (y0 < 0) y0 = 0; if (x1 > R - 1) x1 = R - 1; if (y1 > R - 1) y1 = R - 1;
    float pad[3];
    for (int k = 0; k < 3; ++k) pad[k] = (float)((((int)V[(size_t)s.vi[k] * 9 + 8]) >> 1) & 1);
    for (int y = y0; y <= y1; ++y)
      for (int x = x0; x <= x1; ++x) {
        double l[3];
        if (!bary(&s, x + 0.5, y + 0.5, l)) continue;
        double zn = l[0] * s.zn[0] + l[1] * s.zn[1] + l[2] * s.zn[2];
        if (zn < -1.0 || zn > 1.0) continue;
        if (s.area < 0.0) { /* back face: discard when the interpolated padding flag > 0.001 */
          double q0 = l[0] / s.w[0], q1 = l[1] / s.w[1], q2 = l[2] / s.w[2];
          double pv = (q0 * pad[0] + q1 * pad[1] + q2 * pad[2]) / (q0 + q1 + q2);
          if (pv > 0.001) continue;
        }
        float depth = (float)(0.5 * zn + 0.5);
        if (depth < 0.f) depth = 0.f; if (depth > 1.f) depth = 1.f;
        uint32_t d24 = (uint32_t)(depth * 16777215.0f + 0.5f);
        int i = y * R + x;
        if (d24 < depth24[i]) { depth24[i] = d24; tri[i] = t; } /* '<': first drawn wins ties */
      }
  }
  return skipped;
}

/* acc: float [R*R][8] = colour rgba sums, depth sum/weight, mask depth/colour counts (aggregation.csh) */
void oracle_shade_aggregate(const float* V, const unsigned char* diag, const float* colors, const float* campos, int S,
                            const float* mvp, int R, const uint32_t* depth24, const int32_t* tri, float* acc) {
  int P = S + 2;
  for (int i = 0; i < R * R; ++i) {
    if (tri[i] < 0) continue;
    int y = i / R, x = i % R;
    Tri s = setup(V, diag, tri[i], P, mvp, R);
    float depth = (float)depth24[i] / 16777215.0f;
    float col[3] = {0, 0, 0}, wgt = 0.f;
    if (s.area > 0.0) {
      double l[3];
      bary(&s, x + 0.5, y + 0.5, l);
      double q[3] = {l[0] / s.w[0], l[1] / s.w[1], l[2] / s.w[2]};
      double qs = q[0] + q[1] + q[2];
      float at[8] = {0}, fe = 0, fp = 0, fr = 0;
      for (int k = 0; k < 3; ++k) {
        float bw = (float)(q[k] / qs);
        const float* p = V + (size_t)s.vi[k] * 9;
        float nl = 1.0f / sqrtf(p[3] * p[3] + p[4] * p[4] + p[5] * p[5]);
        at[0] += bw * p[0]; at[1] += bw * p[1]; at[2] += bw * p[2];
        at[3] += bw * p[3] * nl; at[4] += bw * p[4] * nl; at[5] += bw * p[5] * nl;
        at[6] += bw * p[6]; at[7] += bw * p[7];
        int fl = (int)p[8];
        fe += bw * (fl & 1); fp += bw * ((fl >> 1) & 1); fr += bw * ((fl >> 2) & 1);
      }
      int tx = (int)floorf(at[6] * S), ty = (int)floorf(at[7] * S);
      if (tx < 0) tx = 0; if (tx > S - 1) tx = S - 1; if (ty < 0) ty = 0; if (ty > S - 1) ty = S - 1;
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
