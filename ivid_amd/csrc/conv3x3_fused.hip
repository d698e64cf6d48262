// Fused  GroupNorm-apply (+FiLM) + SiLU [+ nearest x2 upsample] + 3x3 convolution  with an LDS-staged halo tile.
//
// Replaces, for the 3x3 convolutions of ResBlock2d at spatial widths >= 32 (87 % of the model's FLOPs):
//   in_layers  = GroupNorm32 -> SiLU -> [h_upd] -> Conv2d 3x3        (adm.py:157-161, 203-208)
//   out_layers = GroupNorm32*(1+scale)+shift -> SiLU -> Conv2d 3x3   (adm.py:177-183, 214-219)
// including the skip concat of the decoder (two source tensors, adm.py:563) and the residual epilogue.
//
// Compared with gn_apply + conv_igemm this removes the activated tensor's HBM round trip (one write + one read of
// every conv input) and loads each activation slab from L2 ONCE per channel chunk instead of once per tap:
//
//   tile      : 8 x 32 output pixels of ONE image (M = 256) x 256 output channels, 8 waves (2 x 4), wave = 4 image rows
//               x 64 channels (the 32-lane MFMA fragment is one 32-pixel image-row segment)
//   A operand : per 128-byte channel chunk, the (8+2) x (32+2) pixel HALO of the tile is read raw from HBM/L2 into
//               registers (one 16-byte piece = 8 bf16 channels of one pixel per lane), transformed ONCE
//               y = silu(x*a + b) with the per-(image, channel) GroupNorm/FiLM coefficients (gn_finalize's output),
//               zeroed outside the image (the conv pads the ACTIVATED tensor), and written to a double-buffered,
//               XOR-swizzled LDS image.  The 9 taps then read their fragments from that image at shifted rows.
//   B operand : the [256 cout x 128 B] weight slab of (chunk, tap) streams through a 2-stage LDS ring with
//               global_load_lds, exactly as in conv_igemm.hip.
//   schedule  : K-step = (chunk, tap); taps are unrolled, each tap issues the raw load of ONE halo piece of the NEXT
//               chunk and, three taps later (right after the step's barrier, when everything in flight has landed),
//               transforms + stores it: at most 3 pieces (12 VGPRs) are in flight and the VALU work is spread evenly.
#include "common.h"
#include "internal.h"

namespace {

struct FusedArgs {
  const char* src0;
  const char* src1;
  const float* ab;     // [N][C0+C1][2] GroupNorm(+FiLM) scale / offset per (image, channel)
  const char* w;       // [Cout][9][C0+C1]
  const float* bias;
  char* out;
  const char* res;
  const char* zero;
  float* stats;
  int C0, C1;
  int N, H, W;         // OUTPUT spatial dims (source is H/2 x W/2 when up == 1)
  int Cout;
  int up;              // 0: source has the output size; 1: nearest x2 upsample of the activated source
  int res_mode;        // 0 none, 1 same, 2 residual source is (H/2, W/2) nearest-up
  int tiles_x, tiles_y, ntiles_n, ntiles_total;
};

template <typename T> struct Mma2;
template <> struct Mma2<__bf16> {
  static __device__ __forceinline__ void run(const bf16x8& a, const bf16x8& b, f32x16& c) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  }
};
template <> struct Mma2<float> {
  static __device__ __forceinline__ void run(const f32x4& a, const f32x4& b, f32x16& c) {
#pragma unroll
    for (int j = 0; j < 4; ++j) c = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], b[j], c, 0, 0, 0);
  }
};

constexpr int TH = 8, TW = 32;             // output tile (pixels)
constexpr int HW_ = TW + 2, HH_ = TH + 2;  // halo
constexpr int HROWS = HH_ * HW_;           // 340 halo pixels
constexpr int BN = 256, NT = 512;
constexpr int A_BYTES = HROWS * 128;       // one halo image
constexpr int B_BYTES = BN * 128;
constexpr int PIECES = (HROWS + 63) / 64;  // halo pieces per thread (64 halo pixels per pass of 512 threads)
constexpr int AB_BYTES = 512;                 // (a,b) pairs of one 128-byte channel chunk: 64 ch x 2 floats (32 ch for fp32)
constexpr int LDS_BYTES = 2 * A_BYTES + 2 * B_BYTES + 2 * AB_BYTES;

template <typename T>
__global__ __launch_bounds__(NT) void conv3x3_fused_kernel(const FusedArgs p) {
  typedef typename Elem<T>::vec vec_t;
  constexpr int VE = Elem<T>::VE;
  constexpr int BKE = 128 / (int)sizeof(T);
  constexpr int MI = 4, NI = 2, WTN = 64;
  static_assert(PIECES == 6, "schedule below assumes 6 halo pieces per thread");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const sA0 = smem;
  char* const sB0 = smem + 2 * A_BYTES;
  char* const sAB0 = smem + 2 * A_BYTES + 2 * B_BYTES;  // double-buffered GroupNorm coefficients of a chunk

  const int tile = xcd_remap(blockIdx.x, p.ntiles_total);
  // tile id -> (image, tile row, tile col, cout tile); cout tiles of one pixel tile are neighbours (shared A in L2)
  const int tn = tile % p.ntiles_n;
  int rest = tile / p.ntiles_n;
  const int tx = rest % p.tiles_x;
  rest /= p.tiles_x;
  const int ty = rest % p.tiles_y;
  const int img = rest / p.tiles_y;
  const int y0 = ty * TH, x0 = tx * TW, n0 = tn * BN;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int Ctot = p.C0 + p.C1;
  const int chunks = Ctot / BKE;
  const size_t Ktot = (size_t)9 * Ctot;
  const int Hs = p.up ? p.H >> 1 : p.H, Ws = p.up ? p.W >> 1 : p.W;

  // ---- halo staging: thread handles channel piece cpc = tid&7 of halo pixels hrow = j*64 + (tid>>3), j = 0..5.
  //      Nothing is precomputed per piece (6 pieces x pointers would cost ~40 VGPRs): the pixel decode is a handful of
  //      integer ops, executed once per piece and chunk. ----
  const int cpc = tid & 7;
  const int hrow0 = tid >> 3;
  struct Piece { size_t pix; int lds; bool ok, act; };
  auto piece_desc = [&](int j) -> Piece {
    Piece d;
    const int hrow = j * 64 + hrow0;
    d.act = hrow < HROWS;
    const int hy = hrow / HW_, hx = hrow - hy * HW_;
    const int y = y0 + hy - 1, x = x0 + hx - 1;
    d.ok = d.act && (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W;
    const int ys = p.up ? y >> 1 : y, xs = p.up ? x >> 1 : x;
    d.pix = ((size_t)img * Hs + ys) * Ws + xs;
    d.lds = hrow * 128 + ((cpc ^ ((hrow >> 1) & 7)) << 4);
    return d;
  };
  // wave-uniform description of where channel chunk ch lives (src0 or the skip tensor src1)
  struct ChunkSrc { const char* base; int C, coff; };
  auto chunk_src = [&](int ch) -> ChunkSrc {
    const int cbase = ch * BKE;
    ChunkSrc c;
    if (cbase >= p.C0) { c.base = p.src1; c.C = p.C1; c.coff = cbase - p.C0; }
    else               { c.base = p.src0; c.C = p.C0; c.coff = cbase; }
    return c;
  };
  const float* abn = p.ab + (size_t)img * Ctot * 2;

  // ---- weight staging (as conv_igemm): thread owns 4 pieces of the [256][128 B] slab, rows 64 apart (same swizzle) ----
  const int b_row = tid >> 3;
  const char* b_ptr0 = p.w + ((size_t)(n0 + b_row) * Ktot + (((tid & 7) ^ ((b_row >> 1) & 7)) * VE)) * sizeof(T);
  const size_t b_stride = (size_t)64 * Ktot * sizeof(T);
  auto issue_b = [&](int stage, int ch, int tap) {
    char* sB = sB0 + stage * B_BYTES;
    const size_t koff = ((size_t)tap * Ctot + (size_t)ch * BKE) * sizeof(T);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const char* g = (n0 + b_row + 64 * i < p.Cout) ? b_ptr0 + i * b_stride + koff : p.zero;
      glds16(g, sB + (i * NT + wave * 64) * 16);
    }
  };
  // GroupNorm coefficients of chunk ch -> LDS by LDS-DMA from wave 0 (BKE channels x (a,b) x 4 B = 512 B for bf16,
  // 256 B for fp32: exactly BKE/2 lanes x 16 B, never past the end of the coefficient buffer): kept out of the VGPRs
  auto issue_ab = [&](int ch) {
    if (wave == 0 && lane < BKE / 2)
      glds16((const char*)(abn + (size_t)ch * BKE * 2) + lane * 16, sAB0 + (ch & 1) * AB_BYTES);
  };

  // raw load of halo piece j of channel chunk ch (zero page for padding / idle lanes)
  auto load_piece = [&](int j, const ChunkSrc& cs) -> vec_t {
    const Piece d = piece_desc(j);
    const char* g = cs.base + (d.pix * cs.C + cs.coff) * sizeof(T) + cpc * 16;
    g = d.ok ? g : p.zero;
    return *(const vec_t*)g;
  };
  // y = silu(x*a + b), zero outside the image, stored to the swizzled halo image of chunk ch
  auto store_piece = [&](int j, const vec_t& raw, int ch, char* sA) {
    const Piece d = piece_desc(j);
    if (!d.act) return;
    const float* abl = (const float*)(sAB0 + (ch & 1) * AB_BYTES) + cpc * VE * 2;
    float f[VE];
    vec_to_f32<T>(raw, f);
#pragma unroll
    for (int e = 0; e < VE; e += 2) {
      const f32x4 q = *(const f32x4*)(abl + e * 2);  // a[e], b[e], a[e+1], b[e+1]
      const float v0 = silu_f(f[e] * q[0] + q[1]), v1 = silu_f(f[e + 1] * q[2] + q[3]);
      f[e] = d.ok ? v0 : 0.f;
      f[e + 1] = d.ok ? v1 : 0.f;
    }
    *(vec_t*)(sA + d.lds) = f32_to_vec<T>(f);
  };

  f32x16 acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

  const int frow = lane & 31, fhalf = lane >> 5;
  const int b_off0 = (wn * WTN + frow) * 128;          // fragment ni adds 32 rows (swizzle unchanged: rows 32 apart)
  const int b_sw0 = ((wn * WTN + frow) >> 1) & 7;
  // halo row of (fragment 0, lane pixel) for the CENTRE tap; fragment mi adds mi*HW_, tap (dy,dx) adds dy*HW_ + dx
  const int a_hrow0 = (wm * 4 + 1) * HW_ + frow + 1;

  // ---------------- prologue: halo of chunk 0, weights of (chunk 0, tap 0) ----------------
  issue_ab(0);
  wait_vmcnt0();
  __syncthreads();
  {
    const ChunkSrc cs0 = chunk_src(0);
#pragma unroll 1
    for (int j = 0; j < PIECES; ++j) {
      const vec_t raw = load_piece(j, cs0);
      store_piece(j, raw, 0, sA0);
    }
  }
  issue_b(0, 0, 0);

  // ---------------- main loop ----------------
  // K-step = (chunk, tap), tap = 3*g + t with (dy, dx) = (g-1, t-1).  The t loop is unrolled so the three in-flight halo
  // pieces live in statically indexed registers: at tap 3g+t slot t is consumed (piece 3(g-1)+t, issued three taps
  // ago) and refilled (piece 3g+t of the next chunk).
  int kstep = 0;  // running K-step index (selects the weight stage)
  for (int ch = 0; ch < chunks; ++ch) {
    const char* sA = sA0 + (ch & 1) * A_BYTES;
    char* sAn = sA0 + ((ch + 1) & 1) * A_BYTES;
    const bool more = ch + 1 < chunks;
    const ChunkSrc csn = chunk_src(more ? ch + 1 : ch);
    vec_t raw[3];
#pragma unroll 1
    for (int g = 0; g < 3; ++g) {
#pragma unroll
      for (int t = 0; t < 3; ++t, ++kstep) {
        wait_vmcnt0();
        __syncthreads();  // weights of this step landed; halo writes of earlier steps visible; previous reads finished
        const char* sB = sB0 + (kstep & 1) * B_BYTES;
        // ---- first fragments of this step (requested first: their LDS latency hides behind the transform below) ----
        const int tapoff = (g - 1) * HW_ + (t - 1);
        vec_t a[MI], b[NI];
        int a_addr[MI];
        // opaque to the optimiser: otherwise it hoists all 9 taps x 4 fragments x 4 k-pieces of swizzled LDS addresses
        // out of the chunk loop (144 values -> spills); recomputing them costs 3 VALU ops per ds_read
        int hbase = a_hrow0;
        asm volatile("" : "+v"(hbase));
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
          const int hr = hbase + mi * HW_ + tapoff;
          a_addr[mi] = hr * 128;
          a[mi] = *(const vec_t*)(sA + a_addr[mi] + ((fhalf ^ ((hr >> 1) & 7)) << 4));
        }
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) b[ni] = *(const vec_t*)(sB + b_off0 + ni * 4096 + ((fhalf ^ b_sw0) << 4));
        // ---- consume the halo piece issued three taps ago (landed: everything in flight was drained above) ----
        if (more && g >= 1) store_piece(3 * (g - 1) + t, raw[t], ch + 1, sAn);
        // ---- issue: weights of the next K-step, one raw halo piece of the next chunk ----
        {
          const bool last = (g == 2) && (t == 2);
          const int ntap = last ? 0 : 3 * g + t + 1;
          const int nch = last ? ch + 1 : ch;
          if (nch < chunks) issue_b((kstep + 1) & 1, nch, ntap);
        }
        if (more && g <= 1) {
          if (g == 0 && t == 0) issue_ab(ch + 1);
          raw[t] = load_piece(3 * g + t, csn);
        }
        // ---- MFMAs; the next fragments are requested as soon as the current ones have been issued ----
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
          for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) Mma2<T>::run(a[mi], b[ni], acc[mi][ni]);
          if (kk < 3) {
            const int piece = 2 * (kk + 1) + fhalf;
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
              a[mi] = *(const vec_t*)(sA + a_addr[mi] + ((piece ^ ((a_addr[mi] >> 8) & 7)) << 4));
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) b[ni] = *(const vec_t*)(sB + b_off0 + ni * 4096 + ((piece ^ b_sw0) << 4));
          }
        }
      }
    }
  }

  // ---------------- epilogue (as conv_igemm: per-wave slab -> 16-byte NHWC stores, bias, residual, GN partials) ----------------
  wait_vmcnt0();
  __syncthreads();
  constexpr int LDC = WTN + 4;
  float* slab = (float*)smem + wave * (32 * LDC);
  const int Cout = p.Cout;
  float st_s[VE], st_q[VE];  // GroupNorm partial statistics of this wave's 128 pixels (fused gn_partial)
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * fhalf;
        slab[row * LDC + ni * 32 + frow] = acc[mi][ni][r];
      }
    __syncthreads();
    const int y = y0 + wm * 4 + mi;                        // this fragment = image row y, pixels x0 .. x0+31
    const size_t mbase = ((size_t)img * p.H + y) * p.W + x0;
    const int nbase = n0 + wn * WTN;
    constexpr int LPR = WTN / VE, RPP = 64 / LPR;
    const int lr = lane / LPR, lc = (lane - lr * LPR) * VE;
    if (mi == 0) {
#pragma unroll
      for (int e = 0; e < VE; ++e) st_s[e] = st_q[e] = 0.f;
    }
#pragma unroll
    for (int ps = 0; ps < 32 / RPP; ++ps) {
      const int row = ps * RPP + lr;
      const size_t m = mbase + row;
      const int n = nbase + lc;
      if (n < Cout) {
        float v[VE];
#pragma unroll
        for (int e = 0; e < VE; e += 4) {
          const f32x4 t = *(const f32x4*)(slab + row * LDC + lc + e);
          v[e] = t[0]; v[e + 1] = t[1]; v[e + 2] = t[2]; v[e + 3] = t[3];
        }
        if (p.bias) {
#pragma unroll
          for (int e = 0; e < VE; ++e) v[e] += p.bias[n + e];
        }
        if (p.res_mode == 1) {
          float rv[VE];
          vec_to_f32<T>(*(const vec_t*)(p.res + (m * Cout + n) * sizeof(T)), rv);
#pragma unroll
          for (int e = 0; e < VE; ++e) v[e] += rv[e];
        } else if (p.res_mode == 2) {
          const size_t pix = ((size_t)img * (p.H >> 1) + (y >> 1)) * (p.W >> 1) + ((x0 + row) >> 1);
          float rv[VE];
          vec_to_f32<T>(*(const vec_t*)(p.res + (pix * Cout + n) * sizeof(T)), rv);
#pragma unroll
          for (int e = 0; e < VE; ++e) v[e] += rv[e];
        }
        const vec_t ov = f32_to_vec<T>(v);
        *(vec_t*)(p.out + (m * Cout + n) * sizeof(T)) = ov;
        if (p.stats) {
          float sv[VE];
          vec_to_f32<T>(ov, sv);
#pragma unroll
          for (int e = 0; e < VE; ++e) {
            st_s[e] += sv[e];
            st_q[e] += sv[e] * sv[e];
          }
        }
      }
    }
    if (p.stats && mi == MI - 1) {  // one partial per wave: 4 image rows x 32 pixels = a 128-pixel block
#pragma unroll
      for (int off = LPR; off < 64; off <<= 1) {
#pragma unroll
        for (int e = 0; e < VE; ++e) {
          st_s[e] += __shfl_xor(st_s[e], off);
          st_q[e] += __shfl_xor(st_q[e], off);
        }
      }
      const int n = nbase + lc;
      if (lr == 0 && n < Cout) {
        // block id inside the image: (pair of 4-row bands) x (32-pixel columns); any partition of the image works
        const size_t blk = (size_t)img * (p.H / 4) * p.tiles_x + (size_t)(ty * 2 + wm) * p.tiles_x + tx;
        float* sp = p.stats + (blk * Cout + n) * 2;
#pragma unroll
        for (int e = 0; e < VE; e += 2) *(f32x4*)(sp + e * 2) = f32x4{st_s[e], st_q[e], st_s[e + 1], st_q[e + 1]};
      }
    }
    __syncthreads();
  }
}

template <typename T> int launch_fused(const FusedArgs& a, hipStream_t stream) {
  auto kern = conv3x3_fused_kernel<T>;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    if (e != hipSuccess) return ivid_set_error("conv3x3_gn: hipFuncSetAttribute", e);
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3(a.ntiles_total), dim3(NT), LDS_BYTES, stream, a);
  return ivid_check_launch("conv3x3_gn");
}

}  // namespace

extern "C" int ivid_conv3x3_gn(int dtype, const void* src0, int C0, const void* src1, int C1, const float* ab, int up,
                               const void* weight, const float* bias, void* out, const void* res, int res_mode, int N,
                               int H, int W, int Cout, float* stats, void* stream) {
  const int esz = dtype == IVID_F32 ? 4 : 2;
  const int bke = 128 / esz, ve = 16 / esz;
  if (dtype != IVID_F32 && dtype != IVID_BF16) return ivid_set_error("conv3x3_gn: bad dtype", hipSuccess);
  if (C0 <= 0 || C0 % bke || C1 < 0 || C1 % bke) return ivid_set_error("conv3x3_gn: channels must be multiples of the K-step", hipSuccess);
  if (C1 > 0 && !src1) return ivid_set_error("conv3x3_gn: src1 missing", hipSuccess);
  if (W % TW || H % TH) return ivid_set_error("conv3x3_gn: needs W % 32 == 0 and H % 8 == 0 (use ivid_gn_apply + ivid_conv2d otherwise)", hipSuccess);
  if (Cout % ve) return ivid_set_error("conv3x3_gn: Cout must be a multiple of 16 bytes", hipSuccess);
  if (up && ((H | W) & 1)) return ivid_set_error("conv3x3_gn: upsample needs even H,W", hipSuccess);
  if (res_mode < 0 || res_mode > 2 || (res_mode && !res)) return ivid_set_error("conv3x3_gn: bad residual", hipSuccess);
  if (!ab) return ivid_set_error("conv3x3_gn: ab missing", hipSuccess);
  FusedArgs a;
  a.src0 = (const char*)src0; a.src1 = (const char*)src1; a.ab = ab; a.w = (const char*)weight; a.bias = bias;
  a.out = (char*)out; a.res = (const char*)res; a.zero = (const char*)ivid_zero_page(); a.stats = stats;
  if (!a.zero) return -1;
  a.C0 = C0; a.C1 = C1; a.N = N; a.H = H; a.W = W; a.Cout = Cout; a.up = up ? 1 : 0; a.res_mode = res_mode;
  a.tiles_x = W / TW; a.tiles_y = H / TH; a.ntiles_n = (Cout + BN - 1) / BN;
  a.ntiles_total = N * a.tiles_x * a.tiles_y * a.ntiles_n;
  if (dtype == IVID_BF16) return launch_fused<__bf16>(a, (hipStream_t)stream);
  return launch_fused<float>(a, (hipStream_t)stream);
}
