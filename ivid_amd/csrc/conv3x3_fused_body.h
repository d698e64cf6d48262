// Body of the fused  GroupNorm-apply (+FiLM) + SiLU [+ nearest x2 upsample] + 3x3 convolution  kernel: ONE source for both tile
// shapes (FusedShape<WIDE>), included by conv3x3_fused.hip (WIDE: Cout > 128, the C entry points) and conv3x3_fused128.hip
// (NARROW: Cout <= 128).  Read the header comment of conv3x3_fused.hip first.
#pragma once
#include <cstdlib>
#include <type_traits>
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
  // optional 1x1 skip convolution of the ResBlock input accumulated into the same tile (adm.py:190,222)
  const char* sk0;
  const char* sk1;
  const char* skw;     // [Cout][skC0+skC1]
  int skC0, skC1;
  // compensated 16-bit storage (precision mode fp16c, see conv_igemm.hip ConvArgs): optional lo planes of the output, of
  // the residual source and of the two convolution inputs (the halo transform then starts from hi + lo; the 1x1 skip
  // phase reads the hi planes: they are the MFMA operand); only the LO instantiation of the kernel looks at them
  char* out_lo;
  const char* res_lo;
  const char* src0_lo;
  const char* src1_lo;
  // split-precision skip phase (SKS instantiation, precision mode fp16s): lo planes of the skip sources and the lo part of
  // the skip weights; the phase then accumulates x_hi.w_hi + x_lo.w_hi + x_hi.w_lo
  const char* sk0_lo;
  const char* sk1_lo;
  const char* skw_lo;
  // O16 instantiation (bf16x3 island of the fp16s mode): the result ALSO (out != NULL) or ONLY (out == NULL) leaves as two
  // fp16 planes hi + lo -- the compensated storage form the 16-bit part of the network reads
  char* out16_hi;
  char* out16_lo;
#ifdef IVID_DEV_TIMELINE
  unsigned long long* dbg;             // [blocks][8] phase time stamps (scripts/dev/fused_timeline.py)
#endif
};

// Development build only (-DIVID_DEV_TIMELINE, never in the product library): thread 0 of every workgroup stamps the
// 100 MHz real-time counter at its phase boundaries.
#ifdef IVID_DEV_TIMELINE
#define TL_STAMP(k) do { if (p.dbg && threadIdx.x == 0) p.dbg[(size_t)blockIdx.x * 8 + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
unsigned long long* g_timeline = nullptr;
#else
#define TL_STAMP(k) do {} while (0)
#endif

// The two tile shapes of the kernel.  WIDE: 8 x 32 output pixels x 256 output channels, 128-byte channel chunks, 8 waves as
// 2 (m) x 4 (n).  NARROW (Cout <= 128: the first levels of the small / SR models): narrowing the wide tile to 128 channels would
// leave the halo transform (a cost per pixel and chunk) as long as the MFMA phase (a cost per pixel, chunk AND output channel),
// so the pixel tile doubles and the chunk halves instead: 16 x 32 pixels x 128 channels, 64-byte chunks, 8 waves as 4 x 2 -- a
// wave owns 4 image rows x 64 channels in both (MI = 4, NI = 2, 128 accumulator VGPRs), the halo image is 48,960 B in both.
template <bool WIDE> struct FusedShape {
  static constexpr int TH = WIDE ? 8 : 16, TW = 32;       // output tile (pixels)
  static constexpr int HW_ = TW + 2, HH_ = TH + 2;          // halo
  static constexpr int HROWS = HH_ * HW_;                   // 340 / 612 halo pixels
  static constexpr int BN = WIDE ? 256 : 128, NT = 512;
  static constexpr int CHB = WIDE ? 128 : 64;               // bytes of channels per chunk
  static constexpr int WN = WIDE ? 4 : 2;                   // waves along the output channels (64 each)
  // Halo image in LDS: one row per halo pixel = CHB bytes of channels + a 16-byte pad.  The odd 16-byte stride (9 / 5 slots)
  // spreads the 16 lanes of a ds_read_b128 group over all 16 bank slots for ANY row shift, so every fragment address is
  // ONE per-lane base + a compile-time offset (tap, fragment, k-piece) -- no swizzle arithmetic in the K loop.  The pads of
  // the first rows carry the GroupNorm coefficients of the image's channel chunk.
  static constexpr int AROW = CHB + 16;                     // 144 / 80
  static constexpr int A_BYTES = HROWS * AROW;              // 48,960 in both shapes
  static constexpr int B_BYTES = BN * CHB;                  // 32,768 / 8,192: the [BN cout][CHB] weight slab of (chunk, tap)
  static constexpr int CPP = CHB / 16;                      // 16-byte pieces per pixel and chunk
  static constexpr int PPP = NT / CPP;                      // halo pixels per pass of the 512 threads
  static constexpr int PIECES = (HROWS + PPP - 1) / PPP;    // halo pieces per thread: 6 / 5
  // weight slab in LDS: [row][CPP pieces], the piece index XOR-swizzled by (row >> SWZ_SH) & (CPP - 1) (conflict-free
  // ds_read_b128 of a 32-row fragment; applied to the DMA's per-lane SOURCE piece and again on the fragment read)
  static constexpr int SWZ_SH = WIDE ? 1 : 2, SWZ_MASK = CPP - 1;
  static constexpr int LDS_BYTES = 2 * A_BYTES + 2 * B_BYTES;  // 163,456 of the CU's 163,840 / 114,304
  static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
  static_assert((PPP >> SWZ_SH) % CPP == 0 && (32 >> SWZ_SH) % CPP == 0, "rows PPP / 32 apart share their swizzle");
};

// One output tile (8 x 32 pixels x 256 output channels) of the launch: the whole kernel body.  `tile` is the logical tile id.
// LO: output / residual lo planes in the epilogue.  LOIN: the halo transform reads lo planes of the inputs as well.
// SKS: the 1x1 skip phase runs in split precision (three MFMA passes per chunk).
template <typename T, bool WIDE, bool LO, bool LOIN, bool SKS, bool O16>
__device__ __forceinline__ void fused_tile(const FusedArgs& p, const int tile) {
  typedef typename Elem<T>::vec vec_t;
  typedef FusedShape<WIDE> S;
  constexpr int TH = S::TH, TW = S::TW, HW_ = S::HW_, HROWS = S::HROWS, BN = S::BN, NT = S::NT, CHB = S::CHB, AROW = S::AROW;
  constexpr int A_BYTES = S::A_BYTES, B_BYTES = S::B_BYTES, CPP = S::CPP, PPP = S::PPP, PIECES = S::PIECES;
  constexpr int VE = Elem<T>::VE;
  constexpr int BKE = CHB / (int)sizeof(T);
  constexpr int KK = CHB / 32;               // 32-byte k-pieces per K-step
  constexpr int NSB = CHB / 64;              // bf16x3: MFMA k-blocks (16 fp32 channels = 64 halo bytes) per K-step
  constexpr int MI = 4, NI = 2, WTN = 64;
  static_assert(PIECES >= 4 && PIECES <= 6, "the halo pipeline below moves one piece per tap: pieces 0..PIECES-1 at taps 0..PIECES-1");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const sA0 = smem;
  char* const sB0 = smem + 2 * A_BYTES;

  TL_STAMP(0);
  TL_STAMP(1);
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
  const int wm = wave / S::WN, wn = wave % S::WN;   // wave = 4 image rows x 64 channels
  const int grp = wave >> 2;                        // wave group of the ping-pong schedule (waves w and w + 4 share a SIMD)
  const int Ctot = p.C0 + p.C1;
  const int chunks = Ctot / BKE;
  const size_t Ktot = (size_t)9 * Ctot;
  const int Hs = p.up ? p.H >> 1 : p.H, Ws = p.up ? p.W >> 1 : p.W;

  // ---- halo staging: thread handles channel piece cpc = tid % CPP (16 bytes) of halo pixels hrow = PPP j + tid / CPP,
  //      j = 0..PIECES-1.  Per piece only the source pixel index is kept (PIECES VGPRs + one validity bit mask). ----
  const int cpc = tid & (CPP - 1);
  const int hrow0 = tid / CPP;
  int pix[PIECES];       // source pixel index INSIDE the image (0 when padded / idle: valid memory, zeroed later)
  unsigned okbits = 0;   // bit j: halo pixel of piece j lies inside the image
#pragma unroll
  for (int j = 0; j < PIECES; ++j) {
    const int hrow = j * PPP + hrow0;
    const int hy = hrow / HW_, hx = hrow - hy * HW_;
    const int y = y0 + hy - 1, x = x0 + hx - 1;
    const bool ok = hrow < HROWS && (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W;
    const int ys = p.up ? y >> 1 : y, xs = p.up ? x >> 1 : x;
    pix[j] = ok ? ys * Ws + xs : 0;
    okbits |= (ok ? 1u : 0u) << j;
  }
  // LDS byte of this thread's piece inside a halo row (piece j adds PPP j rows)
  const int st_lds = hrow0 * AROW + (IsSplit<T>::value ? (cpc >> 1) * 32 + (cpc & 1) * 8 : cpc * 16);
  const bool act5 = hrow0 < HROWS - (PIECES - 1) * PPP;     // the last piece exists for the first 20 / 100 halo rows only
  // wave-uniform description of where channel chunk ch lives (src0 or the skip tensor src1): addresses are a
  // wave-uniform 64-bit base + a 32-bit lane offset (no 64-bit VALU arithmetic, no address VGPR pairs)
  const size_t img_px = (size_t)img * Hs * Ws;
  const char* const src0_img = p.src0 + img_px * p.C0 * sizeof(T);
  const char* const src1_img = p.src1 + img_px * p.C1 * sizeof(T);
  // LO: every halo piece is fetched from the lo plane as well.  A source without one is given its own hi plane as a
  // stand-in with weight 0: the number of memory operations per issue window stays a compile-time constant (the counted
  // vmcnt waits of the main loop depend on it).
  const char* const lo0_img = LOIN ? (p.src0_lo ? p.src0_lo : p.src0) + img_px * p.C0 * sizeof(T) : nullptr;
  const char* const lo1_img = LOIN ? (p.src1_lo ? p.src1_lo : p.src1) + img_px * p.C1 * sizeof(T) : nullptr;
  struct ChunkSrc { const char* base; const char* lo; float lw; int cb; };  // cb = bytes per source pixel; lw: weight of the lo piece
  auto chunk_src = [&](int ch) -> ChunkSrc {
    const int cbase = ch * BKE;
    ChunkSrc c;
    c.lo = nullptr; c.lw = 0.f;
    if (cbase >= p.C0) {
      c.base = src1_img + (size_t)(cbase - p.C0) * sizeof(T); c.cb = p.C1 * (int)sizeof(T);
      if constexpr (LOIN) { c.lo = lo1_img + (size_t)(cbase - p.C0) * sizeof(T); c.lw = p.src1_lo ? 1.f : 0.f; }
    } else {
      c.base = src0_img + (size_t)cbase * sizeof(T); c.cb = p.C0 * (int)sizeof(T);
      if constexpr (LOIN) { c.lo = lo0_img + (size_t)cbase * sizeof(T); c.lw = p.src0_lo ? 1.f : 0.f; }
    }
    return c;
  };
  auto load_piece = [&](int j, const ChunkSrc& cs) -> vec_t {  // raw 16 bytes of halo piece j
    return *(const vec_t*)(cs.base + (size_t)(__umul24(pix[j], cs.cb) + cpc * 16));
  };
  auto load_piece_lo = [&](int j, const ChunkSrc& cs) -> vec_t {  // the same piece of the lo plane (LO only)
    return *(const vec_t*)(cs.lo + (size_t)(__umul24(pix[j], cs.cb) + cpc * 16));
  };

  // ---- GroupNorm coefficients of a chunk (BKE channels x (a,b) fp32): lanes 0..BKE/2-1 of wave 0 fetch 16 bytes =
  //      (a0,b0,a1,b1) each and park them, re-paired as (a0,a1,b0,b1) for packed math, in the pad of halo row `lane` of
  //      the image the chunk is transformed INTO ----
  const float* abn = p.ab + (size_t)img * Ctot * 2;
  auto ab_load = [&](int ch) -> f32x4 {
    f32x4 q = {0.f, 0.f, 0.f, 0.f};
    if (wave == 0 && lane < BKE / 2) q = *(const f32x4*)(abn + (size_t)ch * BKE * 2 + lane * 4);
    return q;
  };
  auto ab_store = [&](const f32x4& q, char* sAdst) {
    if (wave == 0 && lane < BKE / 2) *(f32x4*)(sAdst + lane * AROW + CHB) = f32x4{q[0], q[2], q[1], q[3]};
  };
  // store of one transformed halo piece (fp32 lanes f[VE]) into the halo image, zero outside the image.
  // bf16x3: the piece is 4 channels; its bf16 hi / lo halves go to 8-byte slots of the hi piece (2g) and the lo piece
  // (2g+1) of the 8-channel group g = cpc>>1 the MFMA fragments are read from.
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
  auto store_piece = [&](int j, const float* f, char* sAdst) {
    const unsigned keep = (okbits >> j) & 1 ? 0xffffffffu : 0u;
    if constexpr (IsSplit<T>::value) {
      bf16x4 h, l;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        h[e] = (__bf16)f[e];
        l[e] = (__bf16)(f[e] - (float)h[e]);
      }
      u32x2 hb = __builtin_bit_cast(u32x2, h), lb = __builtin_bit_cast(u32x2, l);
      hb &= keep;
      lb &= keep;
      if (j < PIECES - 1 || act5) {
        *(u32x2*)(sAdst + st_lds + j * PPP * AROW) = hb;
        *(u32x2*)(sAdst + st_lds + j * PPP * AROW + 16) = lb;
      }
    } else {
      u32x4 ob = __builtin_bit_cast(u32x4, f32_to_vec<T>(f));
      ob &= keep;
      if (j < PIECES - 1 || act5) *(u32x4*)(sAdst + st_lds + j * PPP * AROW) = ob;
    }
  };
  // y = silu(x*a + b) (exactly silu_f's operations, two channels per packed instruction)
  auto xform_store = [&](int j, const vec_t& raw, const vec_t& rawl, float lw, char* sAdst) {
    const char* cf = sAdst + CHB + cpc * (VE / 2) * AROW;
    float f[VE];
    vec_to_f32<T>(raw, f);
    if constexpr (LOIN) {   // x = hi + lo (lw = 0 for a source without a lo plane)
      float l[VE];
      vec_to_f32<T>(rawl, l);
#pragma unroll
      for (int e = 0; e < VE; ++e) f[e] = __builtin_fmaf(l[e], lw, f[e]);
    }
#pragma unroll
    for (int e = 0; e < VE; e += 2) {
      const f32x4 q = *(const f32x4*)(cf + (e / 2) * AROW);
      const f32x2 x = {f[e], f[e + 1]};
      const f32x2 v = x * f32x2{q[0], q[1]} + f32x2{q[2], q[3]};
      const f32x2 t = v * -1.4426950408889634f;
      f32x2 d = {__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1])};
      d = d + 1.0f;
      const f32x2 y = v * f32x2{__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
      f[e] = y[0];
      f[e + 1] = y[1];
    }
    store_piece(j, f, sAdst);
  };
  // The main loop runs the same transform in two halves (round 6): xform_lin reads the chunk's coefficients of this thread's
  // channel piece in ONE batch and leaves v = x*a + b -- the 16 coefficient registers are dead after VE/2 instructions and the
  // step's fragment reads are issued into them; the one-piece form above re-used a single register quadruple and paid four LDS
  // round trips in a row in front of its exp / rcp chains -- xform_act applies silu and stores.  Same operations, same bits.
  auto xform_lin = [&](const vec_t& raw, const vec_t& rawl, float lw, const char* sAdst, float* f) {
    const char* cf = sAdst + CHB + cpc * (VE / 2) * AROW;
    f32x4 q[VE / 2];
#pragma unroll
    for (int k = 0; k < VE / 2; ++k) q[k] = *(const f32x4*)(cf + k * AROW);
    vec_to_f32<T>(raw, f);
    if constexpr (LOIN) {
      float l[VE];
      vec_to_f32<T>(rawl, l);
#pragma unroll
      for (int e = 0; e < VE; ++e) f[e] = __builtin_fmaf(l[e], lw, f[e]);
    }
#pragma unroll
    for (int k = 0; k < VE / 2; ++k) {
      const f32x2 v = f32x2{f[2 * k], f[2 * k + 1]} * f32x2{q[k][0], q[k][1]} + f32x2{q[k][2], q[k][3]};
      f[2 * k] = v[0];
      f[2 * k + 1] = v[1];
    }
  };
  auto xform_act = [&](int j, float* f, char* sAdst) {
    f32x2 v[VE / 2], d[VE / 2];
#pragma unroll
    for (int k = 0; k < VE / 2; ++k) {
      v[k] = f32x2{f[2 * k], f[2 * k + 1]};
      const f32x2 t = v[k] * -1.4426950408889634f;
      d[k] = f32x2{__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1])};
    }
#pragma unroll
    for (int k = 0; k < VE / 2; ++k) {
      d[k] = d[k] + 1.0f;
      const f32x2 y = v[k] * f32x2{__builtin_amdgcn_rcpf(d[k][0]), __builtin_amdgcn_rcpf(d[k][1])};
      f[2 * k] = y[0];
      f[2 * k + 1] = y[1];
    }
    store_piece(j, f, sAdst);
  };

  // ---- weight staging (as conv_igemm): thread owns NBI pieces of the [BN][CHB] slab, rows PPP apart (same swizzle) ----
  constexpr int NBI = BN * CPP / NT;   // 4 / 1
  const int b_row = tid / CPP;
  unsigned b_voff[NBI];  // rows past Cout are clamped: they produce columns the epilogue never stores
#pragma unroll
  for (int i = 0; i < NBI; ++i) {
    const int row = min(n0 + b_row + PPP * i, p.Cout - 1);
    b_voff[i] = (unsigned)((size_t)row * Ktot * sizeof(T)) + (((tid & (CPP - 1)) ^ ((b_row >> S::SWZ_SH) & S::SWZ_MASK)) << 4);
  }
  auto issue_b = [&](int stage, int ch, int tap) {
    char* sB = sB0 + stage * B_BYTES;
    const char* wk = p.w + ((size_t)tap * Ctot + (size_t)ch * BKE) * sizeof(T);  // wave-uniform
#pragma unroll
    for (int i = 0; i < NBI; ++i) glds16_s(wk, b_voff[i], sB + (i * NT + wave * 64) * 16);
  };
  // The same slab staged by the LAGGING wave group alone (256 threads x 8 / 2 pieces, rows 32 / 64 apart): in the ping-pong loop
  // all weight DMA is issued from that group's phase 1, so the leading group's MFMA phase holds nothing but MFMAs and
  // fragment reads (measured: its issue window cost ~600 of 1800 cycles there).
  constexpr int G1I = BN * CPP / 256, G1R = 256 / CPP;   // pieces per thread of the lagging group, rows between them
  static_assert((G1R >> S::SWZ_SH) % CPP == 0, "rows G1R apart share their swizzle");
  const int g1_row = (tid & 255) / CPP;
  const unsigned g1_swz = ((tid & (CPP - 1)) ^ ((g1_row >> S::SWZ_SH) & S::SWZ_MASK)) << 4;
  const int krow_bytes = (int)(Ktot * sizeof(T));   // < 2^24 (checked on the host)
  auto issue_b_g1 = [&](int stage, int ch, int tap) {
    char* sB = sB0 + stage * B_BYTES;
    const char* wk = p.w + ((size_t)tap * Ctot + (size_t)ch * BKE) * sizeof(T);  // wave-uniform
#pragma unroll
    for (int i = 0; i < G1I; ++i) {
      const int row = min(n0 + g1_row + G1R * i, p.Cout - 1);
      glds16_s(wk, __umul24(row, krow_bytes) + g1_swz, sB + (i * 256 + (wave - 4) * 64) * 16);
    }
  };

  const int frow = lane & 31, fhalf = lane >> 5;
  // weight fragment (ni = 0, k-piece 0) inside a stage; fragment ni adds 32 rows = 32 CHB bytes (same swizzle), k-piece kk
  // flips address bits 5-6:  ((2kk + fhalf) ^ sw) << 4  ==  ((fhalf ^ sw) << 4) ^ (kk << 5)
  // (bf16x3: piece = 4s + 2*fhalf + l for MFMA s = 0/1 and l = 0 hi / 1 lo  ->  base uses 2*fhalf, s flips bit 6, l bit 4)
  constexpr int FH = IsSplit<T>::value ? 2 : 1;
  const int b_frow = wn * WTN + frow;
  const int b_addr0 = b_frow * CHB + (((FH * fhalf) ^ ((b_frow >> S::SWZ_SH) & S::SWZ_MASK)) << 4);
  // halo fragment base = (fragment 0, lane pixel, k-piece 0) for the TOP-LEFT tap; fragment mi adds mi halo rows of
  // pixels (HW_ each), tap (g, t) adds g*HW_ + t pixels, k-piece kk adds 32 B: all compile-time ds_read offsets
  // (bf16x3: halo row = [h0 l0 h1 l1 h2 l2 h3 l3] pieces of 8 channels; lane half picks 32 B, s adds 64 B, lo adds 16 B)
  const int a_base = ((wm * 4) * HW_ + frow) * AROW + fhalf * 16 * FH;

  // ---------------- prologue: everything of (chunk 0, tap 0) in ONE memory round trip ----------------
  {
    const f32x4 q0 = ab_load(0);
    issue_b(0, 0, 0);
    const ChunkSrc cs0 = chunk_src(0);
    vec_t rawp[PIECES];
    vec_t rawpl[LOIN ? PIECES : 1];
#pragma unroll
    for (int j = 0; j < PIECES; ++j) rawp[j] = load_piece(j, cs0);
    if constexpr (LOIN) {
#pragma unroll
      for (int j = 0; j < PIECES; ++j) rawpl[j] = load_piece_lo(j, cs0);
    }
    ab_store(q0, sA0);
    wait_vmcnt0();
    __syncthreads();  // coefficients of chunk 0 visible
    TL_STAMP(2);
    // all six pieces in lockstep: they share the channel piece, hence the coefficients (read once), and their 6 x VE/2
    // independent exp/rcp chains overlap instead of running one piece after the other (pipeline fill, no MFMA yet)
    {
      const char* cf = sA0 + CHB + cpc * (VE / 2) * AROW;
      f32x4 q[VE / 2];
#pragma unroll
      for (int k = 0; k < VE / 2; ++k) q[k] = *(const f32x4*)(cf + k * AROW);
      float f[PIECES][VE];
#pragma unroll
      for (int j = 0; j < PIECES; ++j) vec_to_f32<T>(rawp[j], f[j]);
      if constexpr (LOIN) {
#pragma unroll
        for (int j = 0; j < PIECES; ++j) {
          float l[VE];
          vec_to_f32<T>(rawpl[j], l);
#pragma unroll
          for (int e = 0; e < VE; ++e) f[j][e] = __builtin_fmaf(l[e], cs0.lw, f[j][e]);
        }
      }
#pragma unroll
      for (int k = 0; k < VE / 2; ++k) {
        f32x2 v[PIECES], d[PIECES];
#pragma unroll
        for (int j = 0; j < PIECES; ++j) {
          v[j] = f32x2{f[j][2 * k], f[j][2 * k + 1]} * f32x2{q[k][0], q[k][1]} + f32x2{q[k][2], q[k][3]};
          const f32x2 t = v[j] * -1.4426950408889634f;
          d[j] = f32x2{__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1])};
        }
#pragma unroll
        for (int j = 0; j < PIECES; ++j) {
          d[j] = d[j] + 1.0f;
          const f32x2 y = v[j] * f32x2{__builtin_amdgcn_rcpf(d[j][0]), __builtin_amdgcn_rcpf(d[j][1])};
          f[j][2 * k] = y[0];
          f[j][2 * k + 1] = y[1];
        }
      }
#pragma unroll
      for (int j = 0; j < PIECES; ++j) store_piece(j, f[j], sA0);
    }
  }

  f32x16 acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

  // ---------------- main loop ----------------
  // K-step = (chunk, tap), tap = 3g + t with (dy, dx) = (g-1, t-1); the 9 taps of a chunk are unrolled (straight-line
  // code, static register indices).  Halo pipeline of the NEXT chunk: at tap k slot k&1 of raw[] is consumed (piece k-2,
  // requested two taps ago: transformed and stored, taps 2..7) and refilled (piece k, taps 0..5); its coefficients are fetched at
  // tap 0 and parked in LDS at tap 1.  The two wave groups (wm = 0 / 1: waves w and w+4 share a SIMD) run the step's
  // two halves in OPPOSITE order: while one group transforms its halo piece (VALU + transcendental pipes) the other
  // owns the matrix pipe, then they swap; both meet at the next step's barrier.  Fragment registers rotate: a fragment
  // is re-requested for k-piece kk+1 right after its last MFMA of k-piece kk has been issued.
  auto chunk_body = [&](const int ch, auto more_c) {
    constexpr bool MORE = decltype(more_c)::value;
    const char* aptr = sA0 + (ch & 1) * A_BYTES + a_base;
    char* sAn = sA0 + ((ch + 1) & 1) * A_BYTES;
    const ChunkSrc csn = chunk_src(MORE ? ch + 1 : ch);
    vec_t raw[2];
    vec_t rawl[LOIN ? 2 : 1];
    f32x4 abq;
#pragma unroll
    for (int g = 0; g < 3; ++g) {
#pragma unroll
      for (int t = 0; t < 3; ++t) {
        const int tap = 3 * g + t;
        const bool do_store = MORE && tap >= 2 && tap <= PIECES + 1;   // piece tap-2, requested two taps ago
        const bool do_load = MORE && tap <= PIECES - 1;                // piece tap
        // ---------- phase 1 of the step ("other": issue, fragment fetch, halo transform) ----------
        // Counted waits: the raw halo piece of the latest issue window (always the newest VMEM operation of a wave,
        // issued AFTER the weights) may stay in flight -- it is consumed three steps later; everything older has landed.
        const bool prev_loaded = MORE && tap >= 1 && tap <= PIECES;
        if (prev_loaded) {   // LO: the window's two newest operations are the hi and the lo piece
          if constexpr (LOIN) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
          else asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
        } else wait_vmcnt0();
        __syncthreads();  // barrier X
        const int par = (ch + tap) & 1;  // parity of the running K-step index 9 ch + tap: selects the weight stage
        const int b_off = par * B_BYTES + b_addr0;
        // consume BEFORE issuing (the compiler counts only its own loads, not the asm LDS-DMA: a use placed after an
        // issue window would make it wait for that window's weights)
        vec_t cur, curl;
        if (do_store) cur = raw[tap & 1];
        asm volatile("" : "+v"(cur));  // pins the copy (and the compiler's vmcnt for it) here
        if constexpr (LOIN) {
          if (do_store) curl = rawl[tap & 1];
          asm volatile("" : "+v"(curl));
        }
        if (do_load && tap == 1) ab_store(abq, sAn);
        // issue window of the step (both groups in phase 1): the lagging group stages the whole weight slab of the NEXT
        // K-step (global time = the leading group's MFMA phase: the stage it overwrites was read until the last barrier),
        // then every wave requests one raw halo piece of the next chunk
        if (grp == 1 && (MORE || tap < 8)) issue_b_g1(par ^ 1, tap == 8 ? ch + 1 : ch, tap == 8 ? 0 : tap + 1);
        if (do_load) {
          if (tap == 0) abq = ab_load(ch + 1);
          raw[tap & 1] = load_piece(tap, csn);
          if constexpr (LOIN) rawl[tap & 1] = load_piece_lo(tap, csn);
        }
        const char* const ap = aptr + (g * HW_ + t) * AROW;   // fragment mi adds mi halo rows of pixels
        if constexpr (IsSplit<T>::value) {
          // ---- bf16x3: the K-step is 32 channels = 2 MFMA k-blocks s; per block three products hi*hi, hi*lo, lo*hi.
          //      Fragments rotate in place: a register is re-requested for its next use right after its last MFMA. ----
          auto lda = [&](int mi, int sl) { return *(const bf16x8*)(ap + mi * HW_ * AROW + sl); };          // sl = 64 s + 16 l
          auto ldb = [&](int ni, int sl) { return *(const bf16x8*)(sB0 + (b_off ^ sl) + ni * (32 * CHB)); };
          auto mm = [&](const bf16x8& a, const bf16x8& b, f32x16& c) { c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); };
          bf16x8 aH[MI], aL[MI], bH[NI], bL[NI];
#pragma unroll
          for (int mi = 0; mi < MI; ++mi) aH[mi] = lda(mi, 0);
#pragma unroll
          for (int ni = 0; ni < NI; ++ni) { bH[ni] = ldb(ni, 0); bL[ni] = ldb(ni, 16); }
          if (do_store) xform_store(tap - 2, cur, curl, csn.lw, sAn);
          // ---------- phase 2 ("mma") ----------
          if (do_load) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");   // (bf16x3 has no LO instantiation)
          else wait_vmcnt0();
          __syncthreads();  // barrier Y
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int sidx = 0; sidx < NSB; ++sidx) {
            const int so = sidx * 64, sn = 64;   // this block's / the next block's byte offset
            const bool nx = sidx + 1 < NSB;      // a next block exists inside this K-step
            // group 1: hi*hi (the lo fragments of A arrive under it)
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) aL[mi] = lda(mi, so + 16);
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
#pragma unroll
              for (int mi = 0; mi < MI; ++mi) mm(aH[mi], bH[ni], acc[mi][ni]);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
              __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
            __builtin_amdgcn_sched_barrier(0);
            // group 2: hi*lo; bL and aH are dead afterwards -> re-requested for the next block
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) mm(aH[mi], bL[0], acc[mi][0]);
            if (nx) bL[0] = ldb(0, sn + 16);
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
              mm(aH[mi], bL[1], acc[mi][1]);
              if (nx) aH[mi] = lda(mi, sn);
            }
            if (nx) bL[1] = ldb(1, sn + 16);
            if (nx) {
              __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
              __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
#pragma unroll
              for (int i = 0; i < 4; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
              }
              __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            // group 3: lo*hi; bH is dead afterwards
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) mm(aL[mi], bH[0], acc[mi][0]);
            if (nx) bH[0] = ldb(0, sn);
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) mm(aL[mi], bH[1], acc[mi][1]);
            if (nx) bH[1] = ldb(1, sn);
            if (nx) {
              __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
              __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
              __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
              __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
          }
        } else {
        // ---- halo transform, first half; fragments of k-piece 0; second half ----
        vec_t a[MI], b[NI];
        float xf[VE];
        if (do_store) {
          xform_lin(cur, curl, csn.lw, sAn, xf);
          __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) a[mi] = *(const vec_t*)(ap + mi * HW_ * AROW);
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) b[ni] = *(const vec_t*)(sB0 + b_off + ni * (32 * CHB));
        if (do_store) {
          __builtin_amdgcn_sched_barrier(0);
          xform_act(tap - 2, xf, sAn);
        }
        // ---------- phase 2 ("mma"): this group owns the matrix pipe, the other group is in its phase 1 ----------
        if (do_load) {
          if constexpr (LOIN) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
          else asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
        } else wait_vmcnt0();
        __syncthreads();  // barrier Y
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) {
          const int xo = (kk + 1) << 5;
          const bool pf = kk < KK - 1;
          auto mma = [&](int mi, int ni) { MmaT<T>::run(a[mi], b[ni], acc[mi][ni]); };
          auto a_next = [&](int mi) { a[mi] = *(const vec_t*)(ap + mi * HW_ * AROW + (kk + 1) * 32); };
          mma(0, 0); mma(1, 0); mma(2, 0); mma(3, 0);
          if (pf) b[0] = *(const vec_t*)(sB0 + (b_off ^ xo));
          mma(0, 1);
          if (pf) a_next(0);
          mma(1, 1);
          if (pf) a_next(1);
          mma(2, 1);
          if (pf) a_next(2);
          mma(3, 1);
          if (pf) {
            a_next(3);
            b[1] = *(const vec_t*)(sB0 + (b_off ^ xo) + 32 * CHB);
            // pin the rotation: 4 MFMA, read, then (MFMA, read) x 4 (the last one 2 reads)
            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
#pragma unroll
            for (int i = 0; i < 3; ++i) {
              __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
              __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
          }
          // k-pieces stay apart: otherwise the last weight fragment is re-requested INTO the register of the other one,
          // i.e. only after the next k-piece's first four MFMAs, with its LDS latency exposed
          __builtin_amdgcn_sched_barrier(0);
        }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  };
  // Ping-pong: wave group 1 (waves 4-7) runs ONE BARRIER behind group 0 (waves w and w+4 share a SIMD).  Every step has
  // two barriers (X, Y); while one group executes its MFMA phase the other fetches fragments, transforms its halo piece
  // and issues loads, so the matrix pipe always has a group whose fragments are already in registers.
  // Hazards under the skew (global barrier index n; group 0: phase 1 of step s in [2s, 2s+1], MFMAs in [2s+1, 2s+2];
  // group 1 one index later):  weight stage (s+1)&1 is read until 2s+1 (group 1's MFMAs of step s-1) and both groups
  // issue its refill inside [2s+1, 2s+2]; the refill is waited for (counted vmcnt) before barrier 2s+2, after which
  // group 0 reads it.  Halo image c+1 is written in phase 1 of taps 3..8 of chunk c and first read after two more
  // barriers; its previous content was last read three steps before the first write.
  wait_vmcnt0();
  TL_STAMP(3);
  if (grp == 1) __syncthreads();
  for (int ch = 0; ch + 1 < chunks; ++ch) chunk_body(ch, std::true_type{});
  chunk_body(chunks - 1, std::false_type{});
  if (grp == 0) __syncthreads();  // the two wave groups are aligned again
  TL_STAMP(4);

  // ---------------- optional skip phase: acc += x[tile pixels] . Wskip  (the ResBlock's 1x1 skip_connection on its raw
  // input x = cat(sk0, sk1)); a plain 2-stage LDS-DMA pipeline like conv_igemm with taps = 1: A stage = the tile's 256
  // pixels x 128 B (swizzled) inside the now idle halo region, B stage as before ----------------
  if (p.skC0 > 0) {
    const int sk_ctot = p.skC0 + p.skC1;
    const int sk_chunks = sk_ctot / BKE;
    constexpr int SA_BYTES = TH * TW * CHB;                   // A stage: the tile's pixels x CHB = 32,768 B in both shapes
    static_assert(2 * SA_BYTES <= 2 * A_BYTES && TH * TW * CPP / NT == 4, "skip stages live in the halo region, 4 pieces per thread");
    const int r0 = tid / CPP;                                 // stage row of piece i: PPP i + r0 (same swizzle for all i)
    const int swz = ((tid & (CPP - 1)) ^ ((r0 >> S::SWZ_SH) & S::SWZ_MASK)) << 4;
    const size_t sk_px = (size_t)img * p.H * p.W;
    const char* const sk0_img = p.sk0 + sk_px * p.skC0 * sizeof(T);
    const char* const sk1_img = p.sk1 + sk_px * p.skC1 * sizeof(T);
    int spix[4];
    unsigned sb_voff[NBI];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = PPP * i + r0;                           // tile pixel: image row y0 + row/32, column x0 + row%32
      spix[i] = (y0 + (row >> 5)) * p.W + x0 + (row & 31);
      if (i < NBI) {
        const int wrow = min(n0 + row, p.Cout - 1);
        sb_voff[i] = (unsigned)((size_t)wrow * sk_ctot * sizeof(T)) + swz;
      }
    }
    // One virtual step v of the phase.  Plain: v = chunk c, both operands of the chunk go to stage v&1.  SKS (the trunk itself
    // passes through this 1x1 convolution -- adm.py:190,222 -- so its operand roundings reach every later layer undamped): three
    // steps per chunk, s = 0: x_hi.w_hi, s = 1: x_lo.w_hi, s = 2: x_hi.w_lo.  The A and B stages are managed separately so that
    // only FOUR slabs are staged per chunk: step 1 keeps the w_hi slab of step 0, step 2 finds x_hi still in step 0's A stage
    // (step 1 staged x_lo into the other one).  A stage of (c, s) = (c + (s == 1)) & 1, B stage = (s == 2).
    const int nv = SKS ? 3 * sk_chunks : sk_chunks;
    auto stage_a = [&](int v) -> int { return SKS ? ((v / 3 + ((v % 3) == 1 ? 1 : 0)) & 1) : (v & 1); };
    auto stage_b = [&](int v) -> int { return SKS ? ((v % 3) == 2 ? 1 : 0) : (v & 1); };
    auto issue_skip = [&](int v) {
      const int c = SKS ? v / 3 : v, sub = SKS ? v % 3 : 0;
      const int cbase = c * BKE;
      const bool second = cbase >= p.skC0;
      if (!SKS || sub != 2) {
        const char* b0 = (SKS && sub == 1) ? p.sk0_lo + sk_px * p.skC0 * sizeof(T) : sk0_img;
        const char* b1 = (SKS && sub == 1) ? p.sk1_lo + sk_px * p.skC1 * sizeof(T) : sk1_img;
        const char* abase = second ? b1 + (size_t)(cbase - p.skC0) * sizeof(T) : b0 + (size_t)cbase * sizeof(T);
        const int cb = (second ? p.skC1 : p.skC0) * (int)sizeof(T);
#pragma unroll
        for (int i = 0; i < 4; ++i)
          glds16_s(abase, __umul24(spix[i], cb) + swz, sA0 + stage_a(v) * SA_BYTES + (i * NT + wave * 64) * 16);
      }
      if (!SKS || sub != 1) {
        const char* wbase = ((SKS && sub == 2) ? p.skw_lo : p.skw) + (size_t)cbase * sizeof(T);
#pragma unroll
        for (int i = 0; i < NBI; ++i) glds16_s(wbase, sb_voff[i], sB0 + stage_b(v) * B_BYTES + (i * NT + wave * 64) * 16);
      }
    };
    const int sa_row = wm * 128 + frow;                        // fragment mi adds 32 rows = 32 CHB bytes (same swizzle)
    const int sa_addr0 = sa_row * CHB + (((FH * fhalf) ^ ((sa_row >> S::SWZ_SH) & S::SWZ_MASK)) << 4);
    issue_skip(0);
    for (int c = 0; c < nv; ++c) {
      wait_vmcnt0();
      __syncthreads();  // the stages of step c landed for every wave; everyone finished reading the stages of step c-1
      // skewed issue (round 6, as in conv_igemm): the first wave group issues the next step's DMA in front of its MFMAs, its SIMD
      // partners behind their first k-piece
      if (grp == 0 && c + 1 < nv) issue_skip(c + 1);
      const int a_off = stage_a(c) * SA_BYTES + sa_addr0;
      const int b_off = stage_b(c) * B_BYTES + b_addr0;
      if constexpr (IsSplit<T>::value) {
        // raw fp32 block input: split into bf16 hi / lo in registers (as conv_igemm's bf16x3 path)
#pragma unroll
        for (int sidx = 0; sidx < NSB; ++sidx) {
          const int so = sidx * 64;
          if (sidx == NSB - 1 && grp == 1 && c + 1 < nv) issue_skip(c + 1);
          bf16x8 ah[MI], al[MI], bh[NI], bl[NI];
#pragma unroll
          for (int mi = 0; mi < MI; ++mi)
            split_bf16x8(*(const f32x4*)(sA0 + (a_off ^ so) + mi * (32 * CHB)), *(const f32x4*)(sA0 + (a_off ^ so ^ 16) + mi * (32 * CHB)),
                         ah[mi], al[mi]);
#pragma unroll
          for (int ni = 0; ni < NI; ++ni) {
            bh[ni] = *(const bf16x8*)(sB0 + (b_off ^ so) + ni * (32 * CHB));
            bl[ni] = *(const bf16x8*)(sB0 + (b_off ^ so ^ 16) + ni * (32 * CHB));
          }
#pragma unroll
          for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
              f32x16& cc = acc[mi][ni];
              cc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[mi], bh[ni], cc, 0, 0, 0);
              cc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[mi], bl[ni], cc, 0, 0, 0);
              cc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[mi], bh[ni], cc, 0, 0, 0);
            }
        }
      } else {
      vec_t a[MI], b[NI];
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) a[mi] = *(const vec_t*)(sA0 + a_off + mi * (32 * CHB));
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) b[ni] = *(const vec_t*)(sB0 + b_off + ni * (32 * CHB));
#pragma unroll
      for (int kk = 0; kk < KK; ++kk) {
        const int xo = (kk + 1) << 5;
        const bool pf = kk < KK - 1;
        if (kk == (KK > 1 ? 1 : 0) && grp == 1 && c + 1 < nv) issue_skip(c + 1);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) MmaT<T>::run(a[mi], b[0], acc[mi][0]);
        if (pf) b[0] = *(const vec_t*)(sB0 + (b_off ^ xo));
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
          MmaT<T>::run(a[mi], b[1], acc[mi][1]);
          if (pf) a[mi] = *(const vec_t*)(sA0 + (a_off ^ xo) + mi * (32 * CHB));
        }
        if (pf) b[1] = *(const vec_t*)(sB0 + (b_off ^ xo) + 32 * CHB);
      }
      }
    }
  }

  // ---------------- epilogue (as conv_igemm: per-wave slab -> 16-byte NHWC stores, bias, residual, GN partials) ----------------
  // Round 6: the passes of a fragment are ONE straight-line instantiation per (residual kind, full / partial channel range),
  // picked by a wave-uniform branch per fragment.  Carried as runtime branches inside the pass loop (residual mode, lo planes present or not, `channel < Cout`), the
  // same code made the compiler's wait-count pass give up at every join and wait vmcnt(0) -- for the residual loads AND, because
  // vmcnt retires in issue order and counts stores, for the acknowledgement of every output store issued before: 17 us per tile
  // with a hi + lo residual against 8.6 without one.  Residual loads now run ONE fragment ahead -- requested in front of the previous fragment's output stores -- with exact counted waits.
  TL_STAMP(5);
  constexpr int LDC = WTN + 4;
  constexpr int LPR = WTN / VE, RPP = 64 / LPR, NPS = 32 / RPP;   // lanes per slab row, rows per pass, passes per fragment
  const int Cout = p.Cout;
  const int nbase = n0 + wn * WTN;
  const int lr = lane / LPR, lc = (lane - lr * LPR) * VE;
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // LDS reads of the last K-step done
  // The accumulators leave through code every form shares (slab writes); only the four passes of a fragment -- slab reads,
  // bias, residual, rounding, stores, statistics: nothing that touches an accumulator -- exist once per form
  // (RES: 0 no residual, 1 same-size or nearest-x2 residual = res_mode 1 / 2, 3 the 2x2 average pool of a (2H, 2W) tensor;
  // FULL: all 64 output channels of this wave exist, no lane guard -- Cout % 64 == 0 is every layer of the models).  A dispatch
  // over whole epilogues, with 128 live accumulators flowing into six blocks, made the register allocator spill them.
  // (global address space said explicitly: with the pointers selected / offset as below the compiler otherwise falls back to FLAT
  // operations, which also count on lgkmcnt and serialise the slab reads behind them)
  auto gl = [](const char* q, size_t ub, unsigned vo) { return (const __attribute__((address_space(1))) char*)q + ub + vo; };
  auto gs = [](char* q, size_t ub, unsigned vo) { return (__attribute__((address_space(1))) char*)q + ub + vo; };
  typedef const __attribute__((address_space(1))) vec_t gvec_c;
  typedef __attribute__((address_space(1))) vec_t gvec;
  typedef __attribute__((address_space(1))) f16x4 gf16x4;
  const int n = nbase + lc;                    // this lane's first output channel: the same 16-byte piece in every pass
  const bool full = nbase + WTN <= Cout;       // wave-uniform
  const bool lane_on = n < Cout;
  const int rk = (p.res_mode == 1 || p.res_mode == 2) ? 1 : p.res_mode;
  // a missing lo plane of the residual is replaced by the hi plane with weight 0: the loads stay unconditional
  const bool res_has_lo = LO && p.res_lo != nullptr;
  const bool out_has_lo = LO && p.out_lo != nullptr;
  const float rlw = res_has_lo ? 1.f : 0.f, olw = out_has_lo ? 1.f : 0.f;
  // every address = a wave-uniform 64-bit base (scalar registers) + ONE of three 32-bit lane offsets: the unrolled passes then
  // need no address registers of their own
  const int nl = lane_on ? n : 0;
  const unsigned vo_same = (unsigned)((lr * Cout + nl) * (int)sizeof(T));          // pixel lr of a pass, same-size tensor
  const unsigned vo_half = (unsigned)(((lr >> 1) * Cout + nl) * (int)sizeof(T));   // ... of a half-size tensor (RPP is even)
  const unsigned vo_dbl = (unsigned)((2 * lr * Cout + nl) * (int)sizeof(T));       // ... of a double-size tensor
  const bool res_up = p.res_mode == 2;
  constexpr int HB = 2;                        // residual buffers: the fragment being consumed + the one in flight
  vec_t rres[HB][NPS];
  vec_t rres_lo[LO ? HB : 1][LO ? NPS : 1];
  auto load_res = [&](int mi) {                // residual (kind 1) of fragment mi
    const int y = y0 + wm * 4 + mi;
#pragma unroll
    for (int ps = 0; ps < NPS; ++ps) {
      const int xu = x0 + ps * RPP;            // wave-uniform
      const size_t pixu = res_up ? ((size_t)img * (p.H >> 1) + (y >> 1)) * (p.W >> 1) + (xu >> 1) : ((size_t)img * p.H + y) * p.W + xu;
      const size_t ub = pixu * Cout * sizeof(T);
      const unsigned vo = res_up ? vo_half : vo_same;
      rres[mi % HB][ps] = *(gvec_c*)gl(p.res, ub, vo);
      if constexpr (LO) {
        rres_lo[mi % HB][ps] = *(gvec_c*)gl(res_has_lo ? p.res_lo : p.res, ub, vo);
      }
    }
  };
  if (rk == 1) load_res(0);
  float bv[VE];              // this lane's bias values
#pragma unroll
  for (int e = 0; e < VE; ++e) bv[e] = (p.bias && lane_on) ? p.bias[n + e] : 0.f;
  // every LDS-DMA of the main loop / skip phase has landed long ago (the last stage was consumed); only the residual / bias
  // loads are in flight, and they stay in flight across this barrier
  __builtin_amdgcn_s_barrier();
  float* slab = (float*)smem + wave * (32 * LDC);
  float st_s[VE], st_q[VE];  // GroupNorm partial statistics of this wave's 128 pixels (fused gn_partial)
#pragma unroll
  for (int e = 0; e < VE; ++e) st_s[e] = st_q[e] = 0.f;
  auto passes = [&](const int mi, auto res_c, auto full_c) {
    constexpr int RES = decltype(res_c)::value;
    constexpr bool FULL = decltype(full_c)::value;
    const bool on = FULL || lane_on;
    const int y = y0 + wm * 4 + mi;                            // this fragment = image row y, pixels x0 .. x0+31
    const size_t mbase = ((size_t)img * p.H + y) * p.W + x0;   // wave-uniform
#pragma unroll
    for (int ps = 0; ps < NPS; ++ps) {
      const int row = ps * RPP + lr;
      const size_t ob = (mbase + ps * RPP) * Cout * sizeof(T);   // wave-uniform byte offset of the pass in a same-size tensor
      float v[VE];
#pragma unroll
      for (int e = 0; e < VE; e += 4) {
        const f32x4 q = *(const f32x4*)(slab + row * LDC + lc + e);
        v[e] = q[0]; v[e + 1] = q[1]; v[e + 2] = q[2]; v[e + 3] = q[3];
      }
#pragma unroll
      for (int e = 0; e < VE; ++e) v[e] += bv[e];
      if constexpr (RES == 1) {
        float rv[VE];
        vec_to_f32<T>(rres[mi % HB][ps], rv);
        if constexpr (LO) {
          float rl[VE];
          vec_to_f32<T>(rres_lo[mi % HB][ps], rl);
#pragma unroll
          for (int e = 0; e < VE; ++e) rv[e] = __builtin_fmaf(rl[e], rlw, rv[e]);   // (= rv + rl, or rv without a lo plane)
        }
#pragma unroll
        for (int e = 0; e < VE; ++e) v[e] += rv[e];
      } else if constexpr (RES == 3) {  // residual source is (2H, 2W): 2x2 average pool (Downsample2d on the skip path)
        float sacc[VE];
#pragma unroll
        for (int e = 0; e < VE; ++e) sacc[e] = 0.f;
#pragma unroll
        for (int d = 0; d < 4; ++d) {
          const size_t pixu = ((size_t)img * (p.H << 1) + 2 * y + (d >> 1)) * (p.W << 1) + 2 * (x0 + ps * RPP) + (d & 1);
          const size_t ub = pixu * Cout * sizeof(T);
          float rv[VE];
          { const vec_t rq = *(gvec_c*)gl(p.res, ub, vo_dbl); vec_to_f32<T>(rq, rv); }
#pragma unroll
          for (int e = 0; e < VE; ++e) sacc[e] += rv[e];
          if constexpr (LO) {
            if (res_has_lo) {
              { const vec_t rq = *(gvec_c*)gl(p.res_lo, ub, vo_dbl); vec_to_f32<T>(rq, rv); }
#pragma unroll
              for (int e = 0; e < VE; ++e) sacc[e] += rv[e];
            }
          }
        }
#pragma unroll
        for (int e = 0; e < VE; ++e) v[e] += 0.25f * sacc[e];
      }
      const vec_t ov = f32_to_vec<T>(v);
      float sv[VE];
      vec_to_f32<T>(ov, sv);
      if (on) {
        if constexpr (O16) {   // fp32 storage (VE = 4): the fp16 twin of the value, 8-byte stores
          f16x4 th, tl;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            th[e] = (_Float16)v[e];
            tl[e] = (_Float16)(v[e] - (float)th[e]);
          }
          *(gf16x4*)gs(p.out16_hi, ob / 2, vo_same / 2) = th;   // (fp32 storage: the fp16 planes are half the bytes)
          *(gf16x4*)gs(p.out16_lo, ob / 2, vo_same / 2) = tl;
          if (p.out) *(gvec*)gs(p.out, ob, vo_same) = ov;
        } else {
          *(gvec*)gs(p.out, ob, vo_same) = ov;
        }
      }
      if constexpr (LO) {   // lo plane: what the 16-bit rounding dropped; the statistics describe hi + lo
        // (computed unconditionally, weight 0 without an output lo plane: a branch around VALUE computations inside the unrolled
        // passes lets the compiler sink the statistics sums behind the last pass, with every pass's values spilled)
        float lv[VE];
#pragma unroll
        for (int e = 0; e < VE; ++e) lv[e] = v[e] - sv[e];
        const vec_t ol = f32_to_vec<T>(lv);
        if (out_has_lo && on) *(gvec*)gs(p.out_lo, ob, vo_same) = ol;
        vec_to_f32<T>(ol, lv);
#pragma unroll
        for (int e = 0; e < VE; ++e) sv[e] = __builtin_fmaf(lv[e], olw, sv[e]);
      }
#pragma unroll
      for (int e = 0; e < VE; ++e) {
        st_s[e] += sv[e];
        st_q[e] = __builtin_fmaf(sv[e], sv[e], st_q[e]);
      }
    }
  };
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * fhalf;
        slab[row * LDC + ni * 32 + frow] = acc[mi][ni][r];
      }
    // the next fragment's residual is requested as soon as this fragment's accumulators are dead -- and in front of this
    // fragment's output stores in the wave's in-order memory queue
    if (rk == 1 && mi + 1 < MI) load_res(mi + 1);
    wave_lds_sync();  // the slab is private to this wave
    if (full) {
      if (rk == 0) passes(mi, std::integral_constant<int, 0>{}, std::true_type{});
      else if (rk == 1) passes(mi, std::integral_constant<int, 1>{}, std::true_type{});
      else passes(mi, std::integral_constant<int, 3>{}, std::true_type{});
    } else {
      if (rk == 0) passes(mi, std::integral_constant<int, 0>{}, std::false_type{});
      else if (rk == 1) passes(mi, std::integral_constant<int, 1>{}, std::false_type{});
      else passes(mi, std::integral_constant<int, 3>{}, std::false_type{});
    }
    wave_lds_sync();  // the slab is private to this wave
  }
  if (p.stats) {  // one partial per wave: 4 image rows x 32 pixels = a 128-pixel block
#pragma unroll
    for (int off = LPR; off < 64; off <<= 1) {
#pragma unroll
      for (int e = 0; e < VE; ++e) {
        st_s[e] += __shfl_xor(st_s[e], off);
        st_q[e] += __shfl_xor(st_q[e], off);
      }
    }
    if (lr == 0 && lane_on) {
      // block id inside the image: (4-row band) x (32-pixel column strip), the same partition in both tile shapes
      const size_t blk = (size_t)img * (p.H / 4) * p.tiles_x + (size_t)(ty * (TH / 4) + wm) * p.tiles_x + tx;
      float* sp = p.stats + (blk * Cout + n) * 2;
#pragma unroll
      for (int e = 0; e < VE; e += 2) *(f32x4*)(sp + e * 2) = f32x4{st_s[e], st_q[e], st_s[e + 1], st_q[e + 1]};
    }
  }
#ifdef IVID_DEV_TIMELINE
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // stores of this wave retired
  TL_STAMP(6);
  if (p.dbg && threadIdx.x == 0)
    p.dbg[(size_t)blockIdx.x * 8 + 7] = (unsigned long long)__builtin_amdgcn_s_getreg(63492) |       // HW_ID
                                        ((unsigned long long)__builtin_amdgcn_s_getreg(63508) << 32); // XCC_ID
#endif
}

// (Measured and dropped, round 3: a PERSISTENT form -- one workgroup per CU looping over its share of the tiles, the XCD's
// range walked side by side -- is bit-identical and 1.5-3 % slower on every layer (128^2 256->256: 1011 vs 1027 TF/s, 512->256:
// 1199 vs 1218): the barrier between tiles and ~100 scalar spills of the hoisted launch constants cost more than the
// workgroup dispatch it saves.)
template <typename T, bool WIDE, bool LO = false, bool LOIN = false, bool SKS = false, bool O16 = false>
__global__ __launch_bounds__(512) void conv3x3_fused_kernel(const FusedArgs p) {
  fused_tile<T, WIDE, LO, LOIN, SKS, O16>(p, xcd_remap(blockIdx.x, p.ntiles_total));
}

template <typename T, bool WIDE, bool LO = false, bool LOIN = false, bool SKS = false, bool O16 = false>
int launch_fused(const FusedArgs& a, hipStream_t stream) {
  auto kern = conv3x3_fused_kernel<T, WIDE, LO, LOIN, SKS, O16>;
  constexpr int LDS_BYTES = FusedShape<WIDE>::LDS_BYTES;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    if (e != hipSuccess) return ivid_set_error("conv3x3_gn: hipFuncSetAttribute", e);
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3(a.ntiles_total), dim3(512), LDS_BYTES, stream, a);
  return ivid_check_launch("conv3x3_gn");
}

// host side: fill the tile geometry of a launch for one of the two shapes
template <bool WIDE> void fused_geometry(FusedArgs& a) {
  typedef FusedShape<WIDE> S;
  a.tiles_x = a.W / S::TW; a.tiles_y = a.H / S::TH; a.ntiles_n = (a.Cout + S::BN - 1) / S::BN;
  a.ntiles_total = a.N * a.tiles_x * a.tiles_y * a.ntiles_n;
}

}  // namespace
