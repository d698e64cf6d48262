// RGBD depth-warp conditioning on the GPU: depth -> textured height-field mesh, z-buffered
// forward projection into a new camera, per-pixel weighted aggregation over all source views and
// the 3x SSAA resolve (8-bit LANCZOS colour, centre-sample depth, masks, depth edges, erosion).
//
// Replaces, batched over (sample, source view) and without leaving the device:
//   rgbd_3d/utils.py  linearize_depth :38-58, unproject :89-110, triangulate :113-134,
//                     mask_discontinuity :137-141, depth_to_mesh(padding='frustum', cal_normal=True)
//                     :144-260, cal_depth_normal :263-274, depth_edge :311-332,
//                     aggregate_conditions :420-477
//   rgbd_3d/moderngl_renderer.py AggregationRenderer.render :260-340 (one OpenGL draw + one compute
//                     dispatch per source view, three read-backs per target view)
//   rgbd_3d/shaders/aggregation.vsh / aggregation.fsh / clear.csh / aggregation.csh
//
// Rasterisation is a scatter/z-buffer kernel over the IMPLICIT grid mesh (no index buffer): one
// thread per triangle walks its screen bounding box with 2-D homogeneous edge functions (no near-
// plane clipping needed) and resolves visibility with a packed 64-bit atomicMin
// (24-bit window depth << 32 | triangle id) — deterministic, and equal depths resolve to the lower
// primitive id exactly like GL's in-order `<` depth test.  Shading happens afterwards, once per
// pixel, on the winning triangle only.  All of it is tiny next to the UNet (HBM/latency bound).
#include "common.h"
#include "internal.h"

// The mesh / raster arithmetic mirrors numpy (separately rounded float32 / float64 operations): no FMA contraction,
// otherwise a 1-ulp change of a depth shows up as ~1e-5 in the Sobel normals (differences of neighbouring points).
#pragma clang fp contract(off)

namespace {

struct MeshParams {
  int B, S, P;              // samples, image side, padded side (S+2)
  double focal, ppp;        // 0.5/tan(fov/2); image_plane_size / S
  float nearv, farv;        // z-buffer encoding of the network's depth channel (CLI near/far)
  float atol, rtol;
  int erode;
  int frustum;              // 1: padding='frustum' (sample.py), 0: numeric padding of `padpx` pixels (load_scene, render.py)
  int nopad;                // 1: padding=None (forward_backward_warp's second mesh): the ring of the padded grid holds COPIES
                            //    of the border vertices (degenerate triangles, never rasterised) and carries no padding flag
  double padpx;
  int metric;               // 1: input holds RGB in [0,1] and METRIC depth (a stored scene), 0: network output in [-1,1]
};

// linear (metric) depth of pixel (r,c) in fp32, as inference/sample.py:83,126 + linearize_depth do
__device__ __forceinline__ float lin_depth(const float* rgbd, const MeshParams& m, int b, int r, int c) {
  const int S = m.S;
  if (m.metric) return rgbd[((size_t)b * 4 + 3) * S * S + (size_t)r * S + c];
  // every operation rounded separately like numpy's float32 arithmetic (no FMA contraction): neighbouring depths
  // differ by ~1e-3, so a 1-ulp change here shows up as 1e-5 in the normals
  float d = __fadd_rn(__fmul_rn(rgbd[((size_t)b * 4 + 3) * S * S + (size_t)r * S + c], 0.5f), 0.5f);
  d = fminf(fmaxf(d, 1e-6f), 1.0f - 1e-6f);
  return __fdiv_rn(__fmul_rn(m.nearv, m.farv), __fsub_rn(m.farv, __fmul_rn(__fsub_rn(m.farv, m.nearv), d)));
}

struct D3 { double x, y, z; };
__device__ __forceinline__ D3 sub(D3 a, D3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ D3 add(D3 a, D3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ D3 mul(D3 a, double s) { return {a.x * s, a.y * s, a.z * s}; }
__device__ __forceinline__ D3 cross(D3 a, D3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
__device__ __forceinline__ double norm(D3 a) { return sqrt(a.x * a.x + a.y * a.y + a.z * a.z); }

// camera-space point of UNPADDED pixel (r,c), indices clamped (np.pad 'edge'); rows flipped so image top = +Y
__device__ __forceinline__ D3 cam_point(const float* rgbd, const MeshParams& m, int b, int r, int c) {
  const int S = m.S;
  r = min(max(r, 0), S - 1);
  c = min(max(c, 0), S - 1);
  const double d = (double)lin_depth(rgbd, m, b, r, c);
  const double u = (c + 0.5) / S, v = (S - 1 - r + 0.5) / S;
  return {(u - 0.5) / m.focal * d, (v - 0.5) / m.focal * d, -d};
}

// camera-space point of PADDED vertex (pr,pc) including the frustum skirt (utils.py:184-199)
__device__ __forceinline__ D3 pad_point(const float* rgbd, const MeshParams& m, int b, int pr, int pc) {
  const int P = m.P;
  D3 p = cam_point(rgbd, m, b, pr - 1, pc - 1);
  const double d = -p.z;
  const bool top = pr == 0, bot = pr == P - 1, lef = pc == 0, rig = pc == P - 1;
  const double ppp = m.frustum ? m.ppp : m.padpx * m.ppp;  // utils.py:186 / :201
  if (top) p.y += ppp * d;
  if (bot) p.y -= ppp * d;
  if (lef) p.x -= ppp * d;
  if (rig) p.x += ppp * d;
  if (m.frustum && (top || bot || lef || rig)) p = mul(p, -0.1 / p.z);
  return p;
}

__global__ __launch_bounds__(256) void mesh_points_kernel(const float* __restrict__ rgbd, MeshParams m,
                                                          const float* __restrict__ inv_mv,  // [B][16] row-major
                                                          float* __restrict__ verts, float* __restrict__ dpad,
                                                          int* __restrict__ flags, float* __restrict__ colors) {
  const int P = m.P, S = m.S, b = blockIdx.y;
  const int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= P * P) return;
  const int pr = v / P, pc = v - pr * P;
  const int r = min(max(pr - 1, 0), S - 1), c = min(max(pc - 1, 0), S - 1);
  // Sobel-smoothed normal of the unpadded point grid (cal_depth_normal, utils.py:263-274)
  D3 ex = {0, 0, 0}, ey = {0, 0, 0};
#pragma unroll
  for (int k = -1; k <= 1; ++k) {
    const double w = k == 0 ? 2.0 : 1.0;
    ex = add(ex, mul(sub(cam_point(rgbd, m, b, r + k, c + 1), cam_point(rgbd, m, b, r + k, c - 1)), w));
    ey = add(ey, mul(sub(cam_point(rgbd, m, b, r - 1, c + k), cam_point(rgbd, m, b, r + 1, c + k)), w));
  }
  D3 n = cross(mul(ex, 0.25), mul(ey, 0.25));
  n = mul(n, 1.0 / norm(n));
  const D3 p = pad_point(rgbd, m, b, pr, pc);
  const float* M = inv_mv + b * 16;
  float* o = verts + ((size_t)b * P * P + v) * 9;
  o[0] = (float)((double)M[0] * p.x + (double)M[1] * p.y + (double)M[2] * p.z + (double)M[3]);
  o[1] = (float)((double)M[4] * p.x + (double)M[5] * p.y + (double)M[6] * p.z + (double)M[7]);
  o[2] = (float)((double)M[8] * p.x + (double)M[9] * p.y + (double)M[10] * p.z + (double)M[11]);
  o[3] = (float)((double)M[0] * n.x + (double)M[1] * n.y + (double)M[2] * n.z);
  o[4] = (float)((double)M[4] * n.x + (double)M[5] * n.y + (double)M[6] * n.z);
  o[5] = (float)((double)M[8] * n.x + (double)M[9] * n.y + (double)M[10] * n.z);
  o[6] = (float)((c + 0.5) / S);
  o[7] = (float)((r + 0.5) / S);
  dpad[(size_t)b * P * P + v] = lin_depth(rgbd, m, b, r, c);
  flags[(size_t)b * P * P + v] = (!m.nopad && (pr == 0 || pr == P - 1 || pc == 0 || pc == P - 1)) ? 2 : 0;
  if (pr >= 1 && pr <= S && pc >= 1 && pc <= S) {  // texture = RGB in [0,1], row 0 = image top
    const size_t px = (size_t)(pr - 1) * S + (pc - 1);
#pragma unroll
    for (int ch = 0; ch < 3; ++ch)
      colors[((size_t)b * S * S + px) * 3 + ch] =
          m.metric ? rgbd[((size_t)b * 4 + ch) * S * S + px]
                   : __fadd_rn(__fmul_rn(rgbd[((size_t)b * 4 + ch) * S * S + px], 0.5f), 0.5f);
  }
}

// vertex ids of triangle t (two per quad; triangulate, utils.py:113-134): A = (01, 00, ft?11:10), B = (10, 11, ft?00:01)
__device__ __forceinline__ void tri_vertices(int t, int P, int ft, int* vi) {
  const int quad = t >> 1, Q = P - 1;
  const int qr = quad / Q, qc = quad - qr * Q;
  const int i00 = qr * P + qc, i01 = i00 + 1, i10 = i00 + P, i11 = i10 + 1;
  if ((t & 1) == 0) { vi[0] = i01; vi[1] = i00; vi[2] = ft ? i11 : i10; }
  else              { vi[0] = i10; vi[1] = i11; vi[2] = ft ? i00 : i01; }
}

__global__ __launch_bounds__(256) void mesh_faces_kernel(const float* __restrict__ rgbd, MeshParams m,
                                                         const float* __restrict__ dpad, int* __restrict__ flags,
                                                         unsigned char* __restrict__ diag) {
  const int P = m.P, Q = P - 1, b = blockIdx.y;
  const int quad = blockIdx.x * blockDim.x + threadIdx.x;
  if (quad >= Q * Q) return;
  const int qr = quad / Q, qc = quad - qr * Q;
  const D3 p00 = pad_point(rgbd, m, b, qr, qc), p01 = pad_point(rgbd, m, b, qr, qc + 1);
  const D3 p10 = pad_point(rgbd, m, b, qr + 1, qc), p11 = pad_point(rgbd, m, b, qr + 1, qc + 1);
  const int ft = norm(sub(p00, p11)) < norm(sub(p01, p10)) ? 1 : 0;  // shorter 3-D diagonal
  diag[(size_t)b * Q * Q + quad] = (unsigned char)ft;
  const float* dp = dpad + (size_t)b * P * P;
  int* fl = flags + (size_t)b * P * P;
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    int vi[3];
    tri_vertices(2 * quad + k, P, ft, vi);
    const float d0 = dp[vi[0]], d1 = dp[vi[1]], d2 = dp[vi[2]];
    const float dmax = fmaxf(d0, fmaxf(d1, d2)), dmin = fminf(d0, fminf(d1, d2));
    const float i0 = 1.0f / d0, i1 = 1.0f / d1, i2 = 1.0f / d2;
    const float imax = fmaxf(i0, fmaxf(i1, i2)), imin = fminf(i0, fminf(i1, i2));
    if ((dmax - dmin > m.atol) && (imax - imin > m.rtol)) {  // mask_discontinuity, utils.py:137-141
      atomicOr(&fl[vi[0]], 1);
      atomicOr(&fl[vi[1]], 1);
      atomicOr(&fl[vi[2]], 1);
    }
  }
}

__global__ __launch_bounds__(256) void mesh_flags_kernel(MeshParams m, const int* __restrict__ flags,
                                                         float* __restrict__ verts) {
  const int P = m.P, b = blockIdx.y;
  const int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= P * P) return;
  const int pr = v / P, pc = v - pr * P;
  const int* fl = flags + (size_t)b * P * P;
  int f = fl[v];
  if (m.erode > 0) {  // cv2.erode(1-disc, ones(2e+1)) == 0: a discontinuity vertex within Chebyshev radius e
    bool er = false;
    for (int dr = -m.erode; dr <= m.erode; ++dr)
      for (int dc = -m.erode; dc <= m.erode; ++dc) {
        const int rr = pr + dr, cc = pc + dc;
        if (rr >= 0 && rr < P && cc >= 0 && cc < P && (fl[rr * P + cc] & 1)) er = true;
      }
    if (er) f |= 4;
  }
  verts[((size_t)b * P * P + v) * 9 + 8] = (float)f;
}

// ---------------------------------------------------------------------------------------------
// rasterisation
struct TriSetup {
  int vi[3];
  double a[3], b[3], c[3];  // lambda'_i(X,Y) = a_i X + b_i Y + c_i  (already divided by det)
  double zc[3];             // clip-space z of the vertices
  float w[3], xn[3], yn[3];
  bool valid, front, all_front_w;
};

__device__ __forceinline__ TriSetup tri_setup(const float* __restrict__ V, const unsigned char* __restrict__ diag,
                                              int t, int P, const float* __restrict__ mvp) {
  TriSetup s;
  tri_vertices(t, P, diag[t >> 1], s.vi);
  double x[3], y[3], w[3];
  s.all_front_w = true;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const float* p = V + (size_t)s.vi[k] * 9;
    // gl_Position = u_projection * u_modelview * vec4(i_position, 1), fp32 like the vertex shader
    const float cx = mvp[0] * p[0] + mvp[1] * p[1] + mvp[2] * p[2] + mvp[3];
    const float cy = mvp[4] * p[0] + mvp[5] * p[1] + mvp[6] * p[2] + mvp[7];
    const float cz = mvp[8] * p[0] + mvp[9] * p[1] + mvp[10] * p[2] + mvp[11];
    const float cw = mvp[12] * p[0] + mvp[13] * p[1] + mvp[14] * p[2] + mvp[15];
    x[k] = cx; y[k] = cy; w[k] = cw; s.zc[k] = cz; s.w[k] = cw;
    if (!(cw > 1e-6f)) s.all_front_w = false;
    s.xn[k] = cx / cw; s.yn[k] = cy / cw;
  }
  double a[3], b[3], c[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const int i1 = (k + 1) % 3, i2 = (k + 2) % 3;
    a[k] = y[i1] * w[i2] - y[i2] * w[i1];
    b[k] = x[i2] * w[i1] - x[i1] * w[i2];
    c[k] = x[i1] * y[i2] - x[i2] * y[i1];
  }
  const double det = x[0] * a[0] + y[0] * b[0] + w[0] * c[0];
  s.valid = det != 0.0 && isfinite(det);
  s.front = det > 0.0;  // CCW in NDC (y up) = front face (ctx.front_face = 'ccw')
  const double inv = s.valid ? 1.0 / det : 0.0;
#pragma unroll
  for (int k = 0; k < 3; ++k) { s.a[k] = a[k] * inv; s.b[k] = b[k] * inv; s.c[k] = c[k] * inv; }
  return s;
}

struct Frag { double l[3]; double sum; float depth; bool inside; };

__device__ __forceinline__ Frag eval_frag(const TriSetup& s, double X, double Y) {
  Frag f;
#pragma unroll
  for (int k = 0; k < 3; ++k) f.l[k] = s.a[k] * X + s.b[k] * Y + s.c[k];
  f.sum = f.l[0] + f.l[1] + f.l[2];  // = 1 / w_clip at this pixel
  // top-left fill rule for pixel centres exactly ON an edge (NDC, y up; the gradient (a, b) of an edge function points
  // inside for either orientation because the functions are divided by det): a left edge (a > 0) or a horizontal top
  // edge (a == 0, interior below: b < 0) owns its points, the others do not -- a shared edge yields exactly one fragment
  bool in = f.sum > 0.0;
#pragma unroll
  for (int k = 0; k < 3; ++k)
    in = in && (f.l[k] > 0.0 || (f.l[k] == 0.0 && (s.a[k] > 0.0 || (s.a[k] == 0.0 && s.b[k] < 0.0))));
  f.inside = in;
  const double zndc = f.l[0] * s.zc[0] + f.l[1] * s.zc[1] + f.l[2] * s.zc[2];  // z_clip / w_clip, affine in screen space
  f.inside = f.inside && zndc >= -1.0 && zndc <= 1.0;
  f.depth = (float)(0.5 * zndc + 0.5);
  return f;
}

constexpr unsigned long long ZEMPTY = ~0ull;

// Screen bounding box of a triangle; triangles with a vertex at / behind the eye plane get the whole target (the 2-D
// homogeneous edge functions need no clipping).  Returns false when it lies outside the target.
__device__ __forceinline__ bool tri_bbox(const TriSetup& s, int R, int& x0, int& x1, int& r0, int& r1) {
  x0 = 0; x1 = R - 1; r0 = 0; r1 = R - 1;
  if (s.all_front_w) {
    const float xmin = fminf(s.xn[0], fminf(s.xn[1], s.xn[2])), xmax = fmaxf(s.xn[0], fmaxf(s.xn[1], s.xn[2]));
    const float ymin = fminf(s.yn[0], fminf(s.yn[1], s.yn[2])), ymax = fmaxf(s.yn[0], fmaxf(s.yn[1], s.yn[2]));
    if (xmax < -1.f || xmin > 1.f || ymax < -1.f || ymin > 1.f) return false;
    const float h = 0.5f * R;
    x0 = max(0, (int)floorf((xmin + 1.f) * h - 0.5f) - 0);
    x1 = min(R - 1, (int)ceilf((xmax + 1.f) * h - 0.5f));
    r0 = max(0, (int)floorf((1.f - ymax) * h - 0.5f));
    r1 = min(R - 1, (int)ceilf((1.f - ymin) * h - 0.5f));
  }
  return x0 <= x1 && r0 <= r1;
}

// depth test '<' + in-order draw of one fragment: packed 64-bit atomicMin (24-bit window depth << 32 | triangle id)
__device__ __forceinline__ void raster_pixel(const TriSetup& s, const float* pad, int t, int r, int x, int R, double step,
                                             unsigned long long* zb) {
  const double Y = 1.0 - (r + 0.5) * step, X = (x + 0.5) * step - 1.0;
  const Frag f = eval_frag(s, X, Y);
  if (!f.inside) return;
  if (!s.front) {  // aggregation.fsh:22-23: back-facing skirt fragments are discarded (no depth write)
    const double isum = 1.0 / f.sum;
    const double pv = (f.l[0] * pad[0] + f.l[1] * pad[1] + f.l[2] * pad[2]) * isum;
    if (pv > 0.001) return;
  }
  const unsigned d24 = (unsigned)(fminf(fmaxf(f.depth, 0.f), 1.f) * 16777215.0f + 0.5f);
  atomicMin(&zb[(size_t)r * R + x], ((unsigned long long)d24 << 32) | (unsigned)t);
}

constexpr int RASTER_SMALL = 256;  // bounding boxes up to this many pixels are walked by the triangle's own thread

// Pass 1: one thread per triangle.  Small triangles (the height field proper: a few pixels each) are rasterised in
// place; large ones -- skirt and discontinuity sheets seen from another camera, up to the whole target when a vertex is
// behind the eye; EVERY triangle of a noisy depth map -- are queued for pass 2 so that no thread walks a big box alone.
// work[0] = counter, work[2 + i] = mesh * ntri + triangle.  The push is wave-aggregated (one atomic per wave).
__global__ __launch_bounds__(256) void raster_kernel(const float* __restrict__ verts, const unsigned char* __restrict__ diag,
                                                     int B, int P, const float* __restrict__ mvp, int R,
                                                     unsigned long long* __restrict__ zbuf, int* __restrict__ work,
                                                     int work_cap, int nodiscard) {
  const int Q = P - 1, ntri = 2 * Q * Q;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int view = blockIdx.y, b = blockIdx.z;
  if (t >= ntri) return;
  const size_t mb = (size_t)view * B + b;
  const float* V = verts + mb * P * P * 9;
  const TriSetup s = tri_setup(V, diag + mb * Q * Q, t, P, mvp + b * 16);
  if (!s.valid) return;
  int x0, x1, r0, r1;
  if (!tri_bbox(s, R, x0, x1, r0, r1)) return;
  const bool big = (x1 - x0 + 1) * (r1 - r0 + 1) > RASTER_SMALL && work_cap > 0;
  const unsigned long long bigs = __ballot(big);
  if (big) {
    const int lane = threadIdx.x & 63, leader = __ffsll((long long)bigs) - 1;
    int base = 0;
    if (lane == leader) base = atomicAdd(&work[0], __popcll(bigs));
    base = __shfl(base, leader);
    const int slot = base + __popcll(bigs & ((1ull << lane) - 1ull));
    if (slot < work_cap) {
      work[2 + slot] = (int)(mb * ntri + t);
      return;
    }  // queue full: fall through and walk it here (correct, only slower)
  }
  float pad[3];   // nodiscard (SimpleRenderer: simple.fsh never discards): the padding test is switched off
#pragma unroll
  for (int k = 0; k < 3; ++k) pad[k] = nodiscard ? 0.f : (float)((((int)V[(size_t)s.vi[k] * 9 + 8]) >> 1) & 1);
  unsigned long long* zb = zbuf + mb * R * R;
  const double step = 2.0 / R;
  for (int r = r0; r <= r1; ++r)
    for (int x = x0; x <= x1; ++x) raster_pixel(s, pad, t, r, x, R, step, zb);
}

// Pass 2: one WAVE per queued triangle (grid-stride over the queue).  The box is cut into 8x8-pixel tiles; every lane
// tests one tile (rejected when one of the three affine edge functions is negative on all of it), the survivors are
// visited one after the other with one pixel per lane.  A sliver that crosses 100 pixels costs ~25 such visits instead
// of a 10^4-pixel box walk.  Results are order-independent (atomicMin on (depth, id)).
__global__ __launch_bounds__(256) void raster_big_kernel(const float* __restrict__ verts,
                                                         const unsigned char* __restrict__ diag, int B, int P,
                                                         const float* __restrict__ mvp, int R,
                                                         unsigned long long* __restrict__ zbuf,
                                                         const int* __restrict__ work, int work_cap, int nodiscard) {
  const int Q = P - 1, ntri = 2 * Q * Q;
  const int count = min(work[0], work_cap);
  const int lane = threadIdx.x & 63;
  const int wid = (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6), nwaves = (int)((gridDim.x * blockDim.x) >> 6);
  const double step = 2.0 / R;
  for (int i = wid; i < count; i += nwaves) {
    const unsigned entry = (unsigned)work[2 + i];
    const size_t mb = entry / (unsigned)ntri;
    const int t = (int)(entry - (unsigned)mb * (unsigned)ntri);
    const int b = (int)(mb % B);
    const float* V = verts + mb * P * P * 9;
    const TriSetup s = tri_setup(V, diag + mb * Q * Q, t, P, mvp + b * 16);
    int x0, x1, r0, r1;
    tri_bbox(s, R, x0, x1, r0, r1);
    float pad[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) pad[k] = nodiscard ? 0.f : (float)((((int)V[(size_t)s.vi[k] * 9 + 8]) >> 1) & 1);
    unsigned long long* zb = zbuf + mb * R * R;
    const int tx0 = x0 >> 3, ty0 = r0 >> 3;
    const int ntx = (x1 >> 3) - tx0 + 1, nty = (r1 >> 3) - ty0 + 1, nt = ntx * nty;
    for (int base = 0; base < nt; base += 64) {
      const int ti = base + lane;
      bool keep = ti < nt;
      if (keep) {
        const int ty = ty0 + ti / ntx, tx = tx0 + ti % ntx;
        const int px0 = max(x0, tx << 3), px1 = min(x1, (tx << 3) + 7), pr0 = max(r0, ty << 3), pr1 = min(r1, (ty << 3) + 7);
        // pixel-centre extent of the tile in NDC
        const double X0 = (px0 + 0.5) * step - 1.0, X1 = (px1 + 0.5) * step - 1.0;
        const double Y0 = 1.0 - (pr1 + 0.5) * step, Y1 = 1.0 - (pr0 + 0.5) * step;
#pragma unroll
        for (int k = 0; k < 3; ++k) {  // max of the affine edge function over the tile
          const double m = s.a[k] * (s.a[k] >= 0.0 ? X1 : X0) + s.b[k] * (s.b[k] >= 0.0 ? Y1 : Y0) + s.c[k];
          keep = keep && !(m < 0.0);
        }
      }
      unsigned long long live = __ballot(keep);
      while (live) {
        const int j = __ffsll((long long)live) - 1;
        live &= live - 1;
        const int tj = base + j;
        const int pr = ((ty0 + tj / ntx) << 3) + (lane >> 3), px = ((tx0 + tj % ntx) << 3) + (lane & 7);
        if (px >= x0 && px <= x1 && pr >= r0 && pr <= r1) raster_pixel(s, pad, t, pr, px, R, step, zb);
      }
    }
  }
}

// One thread per (pixel, sample): walk the source views in order, shade the visible fragment of each
// (aggregation.fsh) and blend with aggregation.csh's rules; then AggregationRenderer.render's read-back math.
__global__ __launch_bounds__(256) void aggregate_kernel(const float* __restrict__ verts,
                                                        const unsigned char* __restrict__ diag,
                                                        const float* __restrict__ colors,
                                                        const float* __restrict__ campos, int NV, int B, int S,
                                                        const float* __restrict__ mvp, int R,
                                                        const unsigned long long* __restrict__ zbuf, float rnear,
                                                        float rfar, unsigned char* __restrict__ color8,
                                                        float* __restrict__ color_f32,
                                                        float* __restrict__ depth_lin,
                                                        unsigned char* __restrict__ mask_c,
                                                        unsigned char* __restrict__ mask_d) {
  const int P = S + 2, Q = P - 1;
  const int pix = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
  if (pix >= R * R) return;
  const int r = pix / R, x = pix - r * R;
  const double step = 2.0 / R;
  const double X = (x + 0.5) * step - 1.0, Y = 1.0 - (r + 0.5) * step;
  float cr = 0.f, cg = 0.f, cb = 0.f, ca = 0.f, dr = 0.f, dg = 0.f, md = 0.f, mc = 0.f;
  for (int view = 0; view < NV; ++view) {
    const size_t mb = (size_t)view * B + b;
    const unsigned long long key = zbuf[mb * R * R + pix];
    if (key == ZEMPTY) continue;  // clear colour (0,0,0,0): every term of aggregation.csh is a no-op
    const int t = (int)(key & 0xffffffffu);
    const float depth = (float)(unsigned)(key >> 32) / 16777215.0f;  // 24-bit depth attachment read as float
    const float* V = verts + mb * P * P * 9;
    const TriSetup s = tri_setup(V, diag + mb * Q * Q, t, P, mvp + b * 16);
    float col[3] = {0.f, 0.f, 0.f}, wgt = 0.f;
    if (s.front) {
      const Frag f = eval_frag(s, X, Y);
      const double isum = 1.0 / f.sum;
      float at[9];
#pragma unroll
      for (int j = 0; j < 9; ++j) at[j] = 0.f;
      float fe = 0.f, fp = 0.f, fr = 0.f;
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const float bw = (float)(f.l[k] * isum);  // perspective-correct barycentric
        const float* p = V + (size_t)s.vi[k] * 9;
        const float nl = rsqrtf(p[3] * p[3] + p[4] * p[4] + p[5] * p[5]);  // vertex shader: normalize(i_normal)
        at[0] += bw * p[0]; at[1] += bw * p[1]; at[2] += bw * p[2];
        at[3] += bw * p[3] * nl; at[4] += bw * p[4] * nl; at[5] += bw * p[5] * nl;
        at[6] += bw * p[6]; at[7] += bw * p[7];
        const int fl = (int)p[8];
        fe += bw * (fl & 1); fp += bw * ((fl >> 1) & 1); fr += bw * ((fl >> 2) & 1);
      }
      // texture(colortex, uv): NEAREST, clamp to edge; texture row = image row
      const int tx = min(max((int)floorf(at[6] * S), 0), S - 1), ty = min(max((int)floorf(at[7] * S), 0), S - 1);
      const float* tc = colors + ((mb * S + ty) * S + tx) * 3;
      col[0] = tc[0]; col[1] = tc[1]; col[2] = tc[2];
      const float* cp = campos + mb * 3;
      float dx = cp[0] - at[0], dy = cp[1] - at[1], dz = cp[2] - at[2];
      const float dl = rsqrtf(dx * dx + dy * dy + dz * dz);
      const float nl = rsqrtf(at[3] * at[3] + at[4] * at[4] + at[5] * at[5]);
      float wv = (dx * at[3] + dy * at[4] + dz * at[5]) * dl * nl;
      wv = fminf(fmaxf(wv, 0.f), 1.f);
      wv = acosf(wv);
      wv = fmaxf(-wv * 20.f, -50.f);
      wv = expf(wv);
      wv = fmaxf(wv, 1e-4f);
      if (!(fr < 0.999f)) wv *= 1e-8f;
      if (fp > 0.001f || fe > 0.999f) wv = 1e-16f;
      wgt = fmaxf(wv, 1e-16f);
    }
    // aggregation.csh:18-41
    const float wd = wgt > 1e-14f ? 1.0f : (wgt > 0.0f ? 1e-8f : 0.0f);
    const float mcol = wgt > 1e-6f ? 1.0f : 0.0f, mdep = wgt > 1e-14f ? 1.0f : 0.0f;
    if (fabsf(dg - 1e-8f) < 1e-8f && fabsf(wd - 1e-8f) < 1e-8f) {
      if (depth * 1e-8f > dr) {
        dr = depth * 1e-8f; dg = 1e-8f;
        cr = col[0] * wgt; cg = col[1] * wgt; cb = col[2] * wgt; ca = wgt;
      }
    } else {
      dr += depth * wd; dg += wd;
      cr += col[0] * wgt; cg += col[1] * wgt; cb += col[2] * wgt; ca += wgt;
    }
    md += mdep; mc += mcol;
  }
  // moderngl_renderer.py:317-331
  const size_t o = (size_t)b * R * R + pix;
  const float rgb[3] = {cr, cg, cb};
#pragma unroll
  for (int ch = 0; ch < 3; ++ch) {
    const float v = ca > 0.0f ? rgb[ch] / fmaxf(ca, 1e-24f) : 0.0f;
    color8[o * 3 + ch] = (unsigned char)(fminf(fmaxf(v, 0.f), 1.f) * 255.0f);  // to8b (truncating cast), utils.py:34-35
    if (color_f32) color_f32[o * 3 + ch] = v;   // AggregationRenderer.render's own return value (float colour)
  }
  const float dz = dg > 0.0f ? dr / fmaxf(dg, 1e-24f) : 0.0f;
  depth_lin[o] = rnear * rfar / (rfar - dz * (rfar - rnear));
  mask_c[o] = mc > 0.5f;
  mask_d[o] = md > 0.5f;
}

// SimpleRenderer (moderngl_renderer.py:11-148, shaders/simple.vsh / simple.fsh): ONE mesh, one depth-tested draw.
// Per pixel: front-facing fragment -> (NEAREST texel, alpha = edge flag interpolated > 0.999 ? 0 : 1); back-facing ->
// (0,0,0,0) (its depth still occludes); nothing drawn -> clear colour 0 and clear depth 1.  Read-back math of
// SimpleRenderer.render (:127-138): mask = alpha > 0.5, depth = near*far / (far - d*(far-near)).
__global__ __launch_bounds__(256) void simple_shade_kernel(const float* __restrict__ verts,
                                                           const unsigned char* __restrict__ diag,
                                                           const float* __restrict__ colors, int B, int S,
                                                           const float* __restrict__ mvp, int R,
                                                           const unsigned long long* __restrict__ zbuf, float rnear,
                                                           float rfar, float* __restrict__ color_f32,
                                                           float* __restrict__ depth_lin,
                                                           unsigned char* __restrict__ mask) {
  const int P = S + 2, Q = P - 1;
  const int pix = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
  if (pix >= R * R) return;
  const int r = pix / R, x = pix - r * R;
  const double step = 2.0 / R;
  const double X = (x + 0.5) * step - 1.0, Y = 1.0 - (r + 0.5) * step;
  const unsigned long long key = zbuf[(size_t)b * R * R + pix];
  float col[3] = {0.f, 0.f, 0.f}, alpha = 0.f, depth = 1.0f;
  if (key != ZEMPTY) {
    const int t = (int)(key & 0xffffffffu);
    depth = (float)(unsigned)(key >> 32) / 16777215.0f;
    const float* V = verts + (size_t)b * P * P * 9;
    const TriSetup s = tri_setup(V, diag + (size_t)b * Q * Q, t, P, mvp + b * 16);
    if (s.front) {
      const Frag f = eval_frag(s, X, Y);
      const double isum = 1.0 / f.sum;
      float u = 0.f, v = 0.f, fe = 0.f;
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const float bw = (float)(f.l[k] * isum);
        const float* p = V + (size_t)s.vi[k] * 9;
        u += bw * p[6]; v += bw * p[7];
        fe += bw * (float)(((int)p[8]) & 1);   // v_is_edge = mod(i_flag, 2)
      }
      const int tx = min(max((int)floorf(u * S), 0), S - 1), ty = min(max((int)floorf(v * S), 0), S - 1);
      const float* tc = colors + (((size_t)b * S + ty) * S + tx) * 3;
      col[0] = tc[0]; col[1] = tc[1]; col[2] = tc[2];
      alpha = fe > 0.999f ? 0.0f : 1.0f;
    }
  }
  const size_t o = (size_t)b * R * R + pix;
  color_f32[o * 3] = col[0]; color_f32[o * 3 + 1] = col[1]; color_f32[o * 3 + 2] = col[2];
  depth_lin[o] = rnear * rfar / (rfar - depth * (rfar - rnear));
  mask[o] = alpha > 0.5f;
}

__global__ __launch_bounds__(256) void to8b_kernel(const float* __restrict__ in, unsigned char* __restrict__ out, long long n) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = (unsigned char)(fminf(fmaxf(in[i], 0.f), 1.f) * 255.0f);   // to8b, utils.py:34-35
}

// ---------------------------------------------------------------------------------------------
// SSAA resolve (aggregate_conditions, utils.py:450-467)

// One pass of Pillow's 8-bit separable resampler (ImagingResampleHorizontal/Vertical_8bpc): integer
// coefficients with 22 fractional bits, rounding constant 1<<21, clip to [0,255].  `in` is [B][H][W][3];
// horizontal: out [B][H][S][3]; vertical: out [B][S][W][3].
__global__ __launch_bounds__(256) void resample8_kernel(const unsigned char* __restrict__ in, int H, int W, int vertical,
                                                        int S, const int* __restrict__ bounds,
                                                        const int* __restrict__ kk, int ksize,
                                                        unsigned char* __restrict__ out) {
  const int b = blockIdx.y;
  const int OH = vertical ? S : H, OW = vertical ? W : S;
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= OH * OW) return;
  const int oy = idx / OW, ox = idx - oy * OW;
  const int o = vertical ? oy : ox;
  const int xmin = bounds[o * 2], xcnt = bounds[o * 2 + 1];
  const int* k = kk + o * ksize;
  int acc[3] = {1 << 21, 1 << 21, 1 << 21};
  for (int j = 0; j < xcnt; ++j) {
    const int sy = vertical ? xmin + j : oy, sx = vertical ? ox : xmin + j;
    const unsigned char* p = in + (((size_t)b * H + sy) * W + sx) * 3;
    acc[0] += p[0] * k[j]; acc[1] += p[1] * k[j]; acc[2] += p[2] * k[j];
  }
  unsigned char* q = out + (((size_t)b * OH + oy) * OW + ox) * 3;
#pragma unroll
  for (int ch = 0; ch < 3; ++ch) q[ch] = (unsigned char)min(max(acc[ch] >> 22, 0), 255);
}

// centre sub-pixel depth -> project_depth (utils.py:61-67); masks = more than 75% of the ssaa^2 sub-pixels
__global__ __launch_bounds__(256) void cond_gather_kernel(const float* __restrict__ depth_lin,
                                                          const unsigned char* __restrict__ mask_c,
                                                          const unsigned char* __restrict__ mask_d, int S, int ssaa,
                                                          float nearv, float farv, float* __restrict__ dproj,
                                                          unsigned char* __restrict__ m0, unsigned char* __restrict__ mr0) {
  const int b = blockIdx.y, p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= S * S) return;
  const int r = p / S, c = p - r * S, R = S * ssaa, off = (ssaa - 1) / 2;
  const size_t base = (size_t)b * R * R;
  float d = depth_lin[base + (size_t)(r * ssaa + off) * R + c * ssaa + off];
  d = fminf(fmaxf(d, nearv), farv);
  dproj[(size_t)b * S * S + p] = (1.0f / nearv - 1.0f / d) / (1.0f / nearv - 1.0f / farv);
  int cd = 0, cc = 0;
  for (int i = 0; i < ssaa; ++i)
    for (int j = 0; j < ssaa; ++j) {
      const size_t q = base + (size_t)(r * ssaa + i) * R + c * ssaa + j;
      cd += mask_d[q];
      cc += mask_c[q];
    }
  const float thr = 0.75f * ssaa * ssaa;
  m0[(size_t)b * S * S + p] = (float)cd > thr;
  mr0[(size_t)b * S * S + p] = (float)cc > thr;
}

__device__ __forceinline__ bool ddiff(float x, float y, float atol, float rtol) {  // depth_edge.depth_diff
  x = fmaxf(x, 1e-6f);
  y = fmaxf(y, 1e-6f);
  return fabsf(x - y) > atol && fabsf(1.0f / x - 1.0f / y) > rtol;
}

// mask &= (fewer than 3 of the 8 neighbours lie across a depth discontinuity)   (depth_edge, utils.py:311-332)
__global__ __launch_bounds__(256) void cond_edge_kernel(const float* __restrict__ dproj,
                                                        const unsigned char* __restrict__ m0, int S, float atol,
                                                        float rtol, unsigned char* __restrict__ m1) {
  const int b = blockIdx.y, p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= S * S) return;
  const int r = p / S, c = p - r * S;
  const float* d = dproj + (size_t)b * S * S;
  int cnt = 0;
  for (int dr = -1; dr <= 1; ++dr)
    for (int dc = -1; dc <= 1; ++dc) {
      if (!dr && !dc) continue;
      const int rr = r + dr, cc = c + dc;
      if (rr < 0 || rr >= S || cc < 0 || cc >= S) continue;
      cnt += ddiff(d[p], d[rr * S + cc], atol, rtol);
    }
  m1[(size_t)b * S * S + p] = m0[(size_t)b * S * S + p] && cnt < 3;
}

// mask_rgb &= erode(mask, (2e-1)^2 ones) (cv2 border = +inf: out-of-image never erodes); apply masks
__global__ __launch_bounds__(256) void cond_final_kernel(const unsigned char* __restrict__ c8small,
                                                         const float* __restrict__ dproj,
                                                         const unsigned char* __restrict__ m1,
                                                         const unsigned char* __restrict__ mr0, int S, int erode,
                                                         const float* __restrict__ lut255, float* __restrict__ color,
                                                         float* __restrict__ depth, float* __restrict__ mask,
                                                         float* __restrict__ mask_rgb, float* __restrict__ convex) {
  const int b = blockIdx.y, p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= S * S) return;
  const int r = p / S, c = p - r * S;
  const size_t o = (size_t)b * S * S;
  const int rad = erode - 1;  // kernel (2*erode-1) x (2*erode-1)
  bool keep = true;
  if (rad >= 0) {
    for (int dr = -rad; dr <= rad; ++dr)
      for (int dc = -rad; dc <= rad; ++dc) {
        const int rr = r + dr, cc = c + dc;
        if (rr >= 0 && rr < S && cc >= 0 && cc < S && !m1[o + rr * S + cc]) keep = false;
      }
  }
  const float m = m1[o + p] ? 1.f : 0.f;
  const float mr = (mr0[o + p] && keep) ? 1.f : 0.f;
  const float d = dproj[o + p];
#pragma unroll
  for (int ch = 0; ch < 3; ++ch) color[((size_t)b * 3 + ch) * S * S + p] = lut255[c8small[(o + p) * 3 + ch]] * mr;
  depth[o + p] = d * m;
  mask[o + p] = m;
  mask_rgb[o + p] = mr;
  convex[o + p] = d;
}

}  // namespace

extern "C" int ivid_mesh_build(const float* rgbd, int B, int S, const float* inv_modelview, float fov_deg, float nearv,
                               float farv, float atol, float rtol, int erode, float padding, int input_mode, float* verts,
                               unsigned char* diag, float* colors, float* scratch_depth, int* scratch_flags, void* stream) {
  if (B <= 0 || S < 2) return ivid_set_error("mesh_build: bad size", hipSuccess);
  if (input_mode != 0 && input_mode != 1) return ivid_set_error("mesh_build: bad input_mode", hipSuccess);
  MeshParams m;
  m.B = B; m.S = S; m.P = S + 2;
  const double fov = (double)fov_deg * 3.14159265358979323846 / 180.0;
  m.focal = 0.5 / tan(0.5 * fov);
  m.ppp = 2.0 * tan(0.5 * fov) / S;
  m.nearv = nearv; m.farv = farv; m.atol = atol; m.rtol = rtol; m.erode = erode;
  m.nopad = padding == -2.f ? 1 : 0;
  m.frustum = (padding < 0.f && !m.nopad) ? 1 : 0; m.padpx = m.nopad ? 0.0 : padding; m.metric = input_mode;
  hipStream_t s = (hipStream_t)stream;
  const int PP = m.P * m.P, QQ = (m.P - 1) * (m.P - 1);
  hipLaunchKernelGGL(mesh_points_kernel, dim3((PP + 255) / 256, B), dim3(256), 0, s, rgbd, m, inv_modelview, verts,
                     scratch_depth, scratch_flags, colors);
  hipLaunchKernelGGL(mesh_faces_kernel, dim3((QQ + 255) / 256, B), dim3(256), 0, s, rgbd, m, scratch_depth,
                     scratch_flags, diag);
  hipLaunchKernelGGL(mesh_flags_kernel, dim3((PP + 255) / 256, B), dim3(256), 0, s, m, scratch_flags, verts);
  return ivid_check_launch("mesh_build");
}

extern "C" int ivid_warp_render(const float* verts, const unsigned char* diag, const float* colors, const float* campos,
                                int NV, int B, int S, const float* mvp, int R, float rnear, float rfar,
                                unsigned long long* zbuf, unsigned char* color8, float* depth_lin,
                                unsigned char* mask_color, unsigned char* mask_depth, int* work, int work_cap,
                                float* color_f32, void* stream) {
  if (NV <= 0 || B <= 0 || R <= 0) return ivid_set_error("warp_render: bad size", hipSuccess);
  if (work_cap < 0 || (work_cap > 0 && !work)) return ivid_set_error("warp_render: bad work queue", hipSuccess);
  if ((size_t)NV * B * 2 * (S + 1) * (S + 1) >= ((size_t)1 << 32)) return ivid_set_error("warp_render: too many triangles for the queue's 32-bit ids", hipSuccess);
  hipStream_t s = (hipStream_t)stream;
  hipError_t e = hipMemsetAsync(zbuf, 0xff, (size_t)NV * B * R * R * sizeof(unsigned long long), s);
  if (e != hipSuccess) return ivid_set_error("warp_render: memset", e);
  if (work_cap > 0) {
    e = hipMemsetAsync(work, 0, 2 * sizeof(int), s);
    if (e != hipSuccess) return ivid_set_error("warp_render: memset", e);
  }
  const int P = S + 2, ntri = 2 * (P - 1) * (P - 1);
  hipLaunchKernelGGL(raster_kernel, dim3((ntri + 255) / 256, NV, B), dim3(256), 0, s, verts, diag, B, P, mvp, R, zbuf,
                     work, work_cap, 0);
  if (work_cap > 0)
    hipLaunchKernelGGL(raster_big_kernel, dim3(work_cap < 8192 ? work_cap : 8192), dim3(256), 0, s, verts, diag, B, P, mvp, R,
                       zbuf, work, work_cap, 0);
  hipLaunchKernelGGL(aggregate_kernel, dim3((R * R + 255) / 256, B), dim3(256), 0, s, verts, diag, colors, campos, NV, B,
                     S, mvp, R, zbuf, rnear, rfar, color8, color_f32, depth_lin, mask_color, mask_depth);
  return ivid_check_launch("warp_render");
}

extern "C" int ivid_warp_resolve(const unsigned char* color8, const float* depth_lin, const unsigned char* mask_color,
                                 const unsigned char* mask_depth, int B, int S, int ssaa, const int* bounds,
                                 const int* coeffs, int ksize, const float* lut255, float nearv, float farv, float atol,
                                 float rtol, int erode, unsigned char* tmp_h, unsigned char* tmp_small, float* tmp_dproj,
                                 unsigned char* tmp_masks, float* color, float* depth, float* mask, float* mask_rgb,
                                 float* convex, void* stream) {
  if (B <= 0 || S <= 0 || ssaa <= 0) return ivid_set_error("warp_resolve: bad size", hipSuccess);
  hipStream_t s = (hipStream_t)stream;
  const int R = S * ssaa, SS = S * S;
  // Pillow resizes horizontally first, then vertically (Resample.c ImagingResampleInner)
  hipLaunchKernelGGL(resample8_kernel, dim3((R * S + 255) / 256, B), dim3(256), 0, s, color8, R, R, 0, S, bounds, coeffs,
                     ksize, tmp_h);
  hipLaunchKernelGGL(resample8_kernel, dim3((SS + 255) / 256, B), dim3(256), 0, s, tmp_h, R, S, 1, S, bounds, coeffs,
                     ksize, tmp_small);
  unsigned char* m0 = tmp_masks;
  unsigned char* mr0 = tmp_masks + (size_t)B * SS;
  unsigned char* m1 = tmp_masks + (size_t)2 * B * SS;
  hipLaunchKernelGGL(cond_gather_kernel, dim3((SS + 255) / 256, B), dim3(256), 0, s, depth_lin, mask_color, mask_depth, S,
                     ssaa, nearv, farv, tmp_dproj, m0, mr0);
  hipLaunchKernelGGL(cond_edge_kernel, dim3((SS + 255) / 256, B), dim3(256), 0, s, tmp_dproj, m0, S, atol, rtol, m1);
  hipLaunchKernelGGL(cond_final_kernel, dim3((SS + 255) / 256, B), dim3(256), 0, s, tmp_small, tmp_dproj, m1, mr0, S,
                     erode, lut255, color, depth, mask, mask_rgb, convex);
  return ivid_check_launch("warp_resolve");
}

extern "C" int ivid_simple_render(const float* verts, const unsigned char* diag, const float* colors, int B, int S,
                                  const float* mvp, int R, float rnear, float rfar, unsigned long long* zbuf, int* work,
                                  int work_cap, float* color_f32, float* depth_lin, unsigned char* mask, void* stream) {
  if (B <= 0 || R <= 0 || S < 2) return ivid_set_error("simple_render: bad size", hipSuccess);
  if (work_cap < 0 || (work_cap > 0 && !work)) return ivid_set_error("simple_render: bad work queue", hipSuccess);
  if (!color_f32 || !depth_lin || !mask || !zbuf) return ivid_set_error("simple_render: null output", hipSuccess);
  hipStream_t s = (hipStream_t)stream;
  hipError_t e = hipMemsetAsync(zbuf, 0xff, (size_t)B * R * R * sizeof(unsigned long long), s);
  if (e != hipSuccess) return ivid_set_error("simple_render: memset", e);
  if (work_cap > 0) {
    e = hipMemsetAsync(work, 0, 2 * sizeof(int), s);
    if (e != hipSuccess) return ivid_set_error("simple_render: memset", e);
  }
  const int P = S + 2, ntri = 2 * (P - 1) * (P - 1);
  // simple.fsh never discards (aggregation.fsh drops back-facing skirt fragments): nodiscard = 1
  hipLaunchKernelGGL(raster_kernel, dim3((ntri + 255) / 256, 1, B), dim3(256), 0, s, verts, diag, B, P, mvp, R, zbuf, work,
                     work_cap, 1);
  if (work_cap > 0)
    hipLaunchKernelGGL(raster_big_kernel, dim3(work_cap < 8192 ? work_cap : 8192), dim3(256), 0, s, verts, diag, B, P, mvp, R,
                       zbuf, work, work_cap, 1);
  hipLaunchKernelGGL(simple_shade_kernel, dim3((R * R + 255) / 256, B), dim3(256), 0, s, verts, diag, colors, B, S, mvp, R,
                     zbuf, rnear, rfar, color_f32, depth_lin, mask);
  return ivid_check_launch("simple_render");
}

extern "C" int ivid_resample8_lanczos(const float* color_f32, int B, int R, int S, const int* bounds, const int* coeffs,
                                      int ksize, unsigned char* tmp_hi8, unsigned char* tmp_h, unsigned char* out8,
                                      void* stream) {
  if (B <= 0 || R <= 0 || S <= 0) return ivid_set_error("resample8: bad size", hipSuccess);
  hipStream_t s = (hipStream_t)stream;
  const long long n = (long long)B * R * R * 3;
  hipLaunchKernelGGL(to8b_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, color_f32, tmp_hi8, n);
  hipLaunchKernelGGL(resample8_kernel, dim3((R * S + 255) / 256, B), dim3(256), 0, s, tmp_hi8, R, R, 0, S, bounds, coeffs,
                     ksize, tmp_h);
  hipLaunchKernelGGL(resample8_kernel, dim3((S * S + 255) / 256, B), dim3(256), 0, s, tmp_h, R, S, 1, S, bounds, coeffs,
                     ksize, out8);
  return ivid_check_launch("resample8");
}
