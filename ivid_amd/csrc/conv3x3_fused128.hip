// Fused  GroupNorm-apply (+FiLM) + SiLU [+ nearest x2 upsample] + 3x3 convolution  for layers with Cout <= 128
// (the first two levels of the small / SR models: 49 % + 19.5 % of their FLOPs).
//
// Same design as conv3x3_fused.hip (read that file first: halo image with odd 16-byte row stride, ping-pong wave groups, asm
// LDS-DMA weights, wave-local epilogue), re-balanced for a 128-wide output: narrowing the 8x32x256 tile to 128 channels would
// leave the halo transform as long as the MFMA phase (the transform cost is per pixel and chunk, the MFMA work per pixel,
// chunk AND output channel).  Here the pixel tile doubles and the channel chunk halves instead:
//
//   tile      : 16 x 32 output pixels of one image (M = 512) x 128 output channels, 8 waves as 4 (m) x 2 (n); a wave owns 4
//               image rows x 64 channels exactly like in the wide kernel (MI = 4, NI = 2, 128 accumulator VGPRs)
//   chunk     : 64 bytes of channels (32 bf16): halo image = 18 x 34 = 612 rows of 64 + 16 pad bytes (odd 16-byte stride of 5),
//               48,960 B per image -- the same LDS budget as the wide kernel's 340 x 144; weights [128][64 B] per (chunk, tap)
//   K-step    : (chunk, tap) = 2 MFMA k-blocks, 16 MFMAs per wave; 5 halo pieces per thread and chunk (pipeline distance 2)
//   LDS       : 2 x 48,960 + 2 x 8,192 = 114,304 B
//   optional  : the ResBlock's 1x1 skip_connection accumulated into the same tile after the 3x3 K-steps (as in the wide kernel)
// 16-bit and fp32 storage types; the split-bf16 mode stays with gn_apply + conv_igemm.
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
  // compensated 16-bit storage (precision mode fp16c, see conv_igemm.hip ConvArgs): optional lo planes of the output / residual
  char* out_lo;
  const char* res_lo;
  const char* src0_lo;   // lo planes of the convolution inputs: the halo transform starts from hi + lo
  const char* src1_lo;
};

constexpr int TH = 16, TW = 32;            // output tile (pixels)
constexpr int HW_ = TW + 2, HH_ = TH + 2;  // halo
constexpr int HROWS = HH_ * HW_;           // 612 halo pixels
constexpr int BN = 128, NT = 512;
constexpr int CHB = 64;                    // bytes of channels per chunk
// Halo image in LDS: one row per halo pixel = 128 B of channels + a 16-byte pad.  The odd 16-byte stride (9 slots)
// spreads the 16 lanes of a ds_read_b128 group over all 16 bank slots for ANY row shift, so every fragment address is
// ONE per-lane base + a compile-time offset (tap, fragment, k-piece) -- no swizzle arithmetic in the K loop.  The pads of
// rows 0..31 carry the GroupNorm coefficients of the image's channel chunk.
constexpr int AROW = CHB + 16;            // 80: 5 slots of 16 B (odd)
constexpr int A_BYTES = HROWS * AROW;      // 48,960
constexpr int B_BYTES = BN * CHB;         // 8,192
constexpr int PIECES = (HROWS + 127) / 128; // halo pieces per thread (128 halo pixels x 4 pieces per pass of 512 threads)
constexpr int LDS_BYTES = 2 * A_BYTES + 2 * B_BYTES;  // 114,304
static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");

template <typename T, bool LO = false, bool LOIN = false>
__global__ __launch_bounds__(NT) void conv3x3_fused128_kernel(const FusedArgs p) {
  typedef typename Elem<T>::vec vec_t;
  constexpr int VE = Elem<T>::VE;
  constexpr int BKE = CHB / (int)sizeof(T);
  constexpr int KK = CHB / 32;              // MFMA k-blocks per K-step
  constexpr int MI = 4, NI = 2, WTN = 64;
  static_assert(PIECES == 5, "schedule below assumes 5 halo pieces per thread");
  static_assert(!IsSplit<T>::value, "bf16x3 layers with Cout <= 128 use conv_igemm");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const sA0 = smem;
  char* const sB0 = smem + 2 * A_BYTES;

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
  const int wm = wave >> 1, wn = wave & 1;   // 4 x 2: wave = 4 image rows x 64 channels; groups: wm < 2 / wm >= 2
  const int Ctot = p.C0 + p.C1;
  const int chunks = Ctot / BKE;
  const size_t Ktot = (size_t)9 * Ctot;
  const int Hs = p.up ? p.H >> 1 : p.H, Ws = p.up ? p.W >> 1 : p.W;

  // ---- halo staging: thread handles channel piece cpc = tid&3 (16 bytes of the 64-byte chunk) of halo pixels
  //      hrow = 128 j + (tid>>2), j = 0..4.  Per piece only the source pixel index is kept (5 VGPRs + one validity bit mask). ----
  const int cpc = tid & 3;
  const int hrow0 = tid >> 2;
  int pix[PIECES];       // source pixel index INSIDE the image (0 when padded / idle: valid memory, zeroed later)
  unsigned okbits = 0;   // bit j: halo pixel of piece j lies inside the image
#pragma unroll
  for (int j = 0; j < PIECES; ++j) {
    const int hrow = j * 128 + hrow0;
    const int hy = hrow / HW_, hx = hrow - hy * HW_;
    const int y = y0 + hy - 1, x = x0 + hx - 1;
    const bool ok = hrow < HROWS && (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W;
    const int ys = p.up ? y >> 1 : y, xs = p.up ? x >> 1 : x;
    pix[j] = ok ? ys * Ws + xs : 0;
    okbits |= (ok ? 1u : 0u) << j;
  }
  // LDS byte of this thread's piece inside a halo row (piece j adds 128 j rows)
  const int st_lds = hrow0 * AROW + cpc * 16;
  const bool act5 = hrow0 < HROWS - 4 * 128;    // the last piece (j = 4) exists for the first 100 halo rows only
  // wave-uniform description of where channel chunk ch lives (src0 or the skip tensor src1): addresses are a
  // wave-uniform 64-bit base + a 32-bit lane offset (no 64-bit VALU arithmetic, no address VGPR pairs)
  const size_t img_px = (size_t)img * Hs * Ws;
  const char* const src0_img = p.src0 + img_px * p.C0 * sizeof(T);
  const char* const src1_img = p.src1 + img_px * p.C1 * sizeof(T);
  // LO: every halo piece is fetched from the lo plane as well; a source without one gets its own hi plane as a stand-in
  // with weight 0, so that the number of memory operations per issue window is a compile-time constant (counted vmcnt waits)
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
  auto load_piece_lo = [&](int j, const ChunkSrc& cs) -> vec_t {
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
  // store of one transformed halo piece (fp32 lanes f[VE]) into the halo image, zero outside the image
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  auto store_piece = [&](int j, const float* f, char* sAdst) {
    const unsigned keep = (okbits >> j) & 1 ? 0xffffffffu : 0u;
    {
      u32x4 ob = __builtin_bit_cast(u32x4, f32_to_vec<T>(f));
      ob &= keep;
      if (j < PIECES - 1 || act5) *(u32x4*)(sAdst + st_lds + j * 128 * AROW) = ob;
    }
  };
  // y = silu(x*a + b) (exactly silu_f's operations, two channels per packed instruction)
  auto xform_store = [&](int j, const vec_t& raw, const vec_t& rawl, float lw, char* sAdst) {
    const char* cf = sAdst + CHB + cpc * (VE / 2) * AROW;
    float f[VE];
    vec_to_f32<T>(raw, f);
    if constexpr (LOIN) {
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

  // ---- weight staging: the [128 cout][64 B] slab of (chunk, tap) = 512 pieces.  LDS image [row][4 pieces], the piece
  //      index XOR-swizzled by (row>>2)&3: the 16 lanes of a ds_read_b128 group read 16 consecutive rows (64-byte stride
  //      = 4 slots of 16 B), rows r and r+4k would share a slot -- the swizzle spreads them over all 16.  The DMA writes
  //      lane-linear, so the swizzle is applied to the per-lane SOURCE piece (and again on the fragment read).
  //      Prologue: one piece per thread.  Loop: the LAGGING wave group alone stages the slab (256 threads x 2 pieces). ----
  const int krow_bytes = (int)(Ktot * sizeof(T));   // < 2^24 (checked on the host)
  const int b_row = tid >> 2;                        // 0..127
  const unsigned b_voff0 = __umul24(min(n0 + b_row, p.Cout - 1), krow_bytes) + (((tid & 3) ^ ((b_row >> 2) & 3)) << 4);
  auto issue_b = [&](int stage, int ch, int tap) {
    const char* wk = p.w + ((size_t)tap * Ctot + (size_t)ch * BKE) * sizeof(T);  // wave-uniform
    glds16_s(wk, b_voff0, sB0 + stage * B_BYTES + wave * 64 * 16);
  };
  const int g1_row = (tid & 255) >> 2;               // 0..63; piece i adds 64 rows (same swizzle)
  const unsigned g1_swz = ((tid & 3) ^ ((g1_row >> 2) & 3)) << 4;
  auto issue_b_g1 = [&](int stage, int ch, int tap) {
    char* sB = sB0 + stage * B_BYTES;
    const char* wk = p.w + ((size_t)tap * Ctot + (size_t)ch * BKE) * sizeof(T);  // wave-uniform
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int row = min(n0 + g1_row + 64 * i, p.Cout - 1);
      glds16_s(wk, __umul24(row, krow_bytes) + g1_swz, sB + (i * 256 + (wave - 4) * 64) * 16);
    }
  };

  const int frow = lane & 31, fhalf = lane >> 5;
  // weight fragment (ni = 0, k-block 0) inside a stage: row * 64 + ((piece ^ sw) << 4) with piece = 2 kk + fhalf;
  // fragment ni adds 32 rows = 2048 B (same swizzle: (row + 32) >> 2 differs by 8), k-block 1 flips address bit 5
  const int b_frow = wn * WTN + frow;
  const int b_addr0 = b_frow * CHB + ((fhalf ^ ((b_frow >> 2) & 3)) << 4);
  // halo fragment base = (fragment 0, lane pixel, k-block 0) for the TOP-LEFT tap; fragment mi adds mi halo rows of
  // pixels (HW_ each), tap (g, t) adds g*HW_ + t pixels, k-block kk adds 32 B: all compile-time ds_read offsets
  const int a_base = ((wm * 4) * HW_ + frow) * AROW + fhalf * 16;

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
    // all five pieces in lockstep: they share the channel piece, hence the coefficients (read once), and their 5 x VE/2
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
  // K-step = (chunk, tap), tap = 3g + t with (dy, dx) = (g-1, t-1); the 9 taps of a chunk are unrolled.  Halo pipeline of
  // the NEXT chunk: at tap k slot k&1 of raw[] is consumed (piece k-2, requested two taps ago: transformed and stored, taps
  // 2..6) and refilled (piece k, taps 0..4); its coefficients are fetched at tap 0 and parked in LDS at tap 1.  The two
  // wave groups (waves 0-3 / 4-7: waves w and w+4 share a SIMD) run ping-pong exactly as in conv3x3_fused.hip; the hazard
  // analysis there carries over (the halo image of chunk c+1 is written in phase 1 of taps 3..7 of chunk c).
  const int grp = wave >> 2;
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
        // consume BEFORE issuing (the compiler counts only its own loads, not the asm LDS-DMA)
        vec_t cur, curl;
        if (do_store) cur = raw[tap & 1];
        asm volatile("" : "+v"(cur));  // pins the copy (and the compiler's vmcnt for it) here
        if constexpr (LOIN) {
          if (do_store) curl = rawl[tap & 1];
          asm volatile("" : "+v"(curl));
        }
        if (do_load && tap == 1) ab_store(abq, sAn);
        // issue window: the lagging group stages the weight slab of the NEXT K-step, then every wave requests one raw halo
        // piece of the next chunk
        if (grp == 1 && (MORE || tap < 8)) issue_b_g1(par ^ 1, tap == 8 ? ch + 1 : ch, tap == 8 ? 0 : tap + 1);
        if (do_load) {
          if (tap == 0) abq = ab_load(ch + 1);
          raw[tap & 1] = load_piece(tap, csn);
          if constexpr (LOIN) rawl[tap & 1] = load_piece_lo(tap, csn);
        }
        const char* const ap = aptr + (g * HW_ + t) * AROW;   // fragment mi adds mi halo rows of pixels
        // ---- fragments of k-block 0 ----
        vec_t a[MI], b[NI];
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) a[mi] = *(const vec_t*)(ap + mi * HW_ * AROW);
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) b[ni] = *(const vec_t*)(sB0 + b_off + ni * (32 * CHB));
        if (do_store) xform_store(tap - 2, cur, curl, csn.lw, sAn);
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
          __builtin_amdgcn_sched_barrier(0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  };
  wait_vmcnt0();
  if (grp == 1) __syncthreads();
  for (int ch = 0; ch + 1 < chunks; ++ch) chunk_body(ch, std::true_type{});
  chunk_body(chunks - 1, std::false_type{});
  if (grp == 0) __syncthreads();  // the two wave groups are aligned again

  // ---------------- optional skip phase: acc += x[tile pixels] . Wskip  (the ResBlock's 1x1 skip_connection on its raw
  // input x = cat(sk0, sk1), adm.py:190,222): a plain 2-stage LDS-DMA pipeline with taps = 1.  A stage = the tile's 512
  // pixels x 64 B inside the now idle halo region (piece index XOR-swizzled by (row>>2)&3 like the weights), B stage as before ----
  if (p.skC0 > 0) {
    constexpr int SA_BYTES = TH * TW * CHB;                     // 32,768 per stage
    static_assert(2 * SA_BYTES <= 2 * A_BYTES, "skip stages live in the halo region");
    const int sk_ctot = p.skC0 + p.skC1;
    const int sk_chunks = sk_ctot / BKE;
    const int r0 = tid >> 2;                                    // stage row of piece i: 128 i + r0 (same swizzle for all i)
    const int swz = ((tid & 3) ^ ((r0 >> 2) & 3)) << 4;
    const size_t sk_px = (size_t)img * p.H * p.W;
    const char* const sk0_img = p.sk0 + sk_px * p.skC0 * sizeof(T);
    const char* const sk1_img = p.sk1 + sk_px * p.skC1 * sizeof(T);
    int spix[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = 128 * i + r0;                             // tile pixel: image row y0 + row/32, column x0 + row%32
      spix[i] = (y0 + (row >> 5)) * p.W + x0 + (row & 31);
    }
    const unsigned sb_voff = (unsigned)((size_t)min(n0 + r0, p.Cout - 1) * sk_ctot * sizeof(T)) + swz;
    auto issue_skip = [&](int stage, int c) {
      const int cbase = c * BKE;
      const bool second = cbase >= p.skC0;
      const char* abase = second ? sk1_img + (size_t)(cbase - p.skC0) * sizeof(T) : sk0_img + (size_t)cbase * sizeof(T);
      const int cb = (second ? p.skC1 : p.skC0) * (int)sizeof(T);
      const char* wbase = p.skw + (size_t)cbase * sizeof(T);
#pragma unroll
      for (int i = 0; i < 4; ++i)
        glds16_s(abase, __umul24(spix[i], cb) + swz, sA0 + stage * SA_BYTES + (i * NT + wave * 64) * 16);
      glds16_s(wbase, sb_voff, sB0 + stage * B_BYTES + wave * 64 * 16);
    };
    const int sa_row = wm * 128 + frow;                          // fragment mi adds 32 rows = 2048 B (same swizzle)
    const int sa_addr0 = sa_row * CHB + ((fhalf ^ ((sa_row >> 2) & 3)) << 4);
    issue_skip(0, 0);
    for (int c = 0; c < sk_chunks; ++c) {
      wait_vmcnt0();
      __syncthreads();  // stage c&1 landed for every wave; everyone finished reading the other stage
      if (c + 1 < sk_chunks) issue_skip((c + 1) & 1, c + 1);
      const int a_off = (c & 1) * SA_BYTES + sa_addr0;
      const int b_off = (c & 1) * B_BYTES + b_addr0;
      vec_t a[MI], b[NI];
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) a[mi] = *(const vec_t*)(sA0 + a_off + mi * (32 * CHB));
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) b[ni] = *(const vec_t*)(sB0 + b_off + ni * (32 * CHB));
#pragma unroll
      for (int kk = 0; kk < KK; ++kk) {
        const int xo = (kk + 1) << 5;
        const bool pf = kk < KK - 1;
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

  // ---------------- epilogue (as conv_igemm: per-wave slab -> 16-byte NHWC stores, bias, residual, GN partials) ----------------
  constexpr int LDC = WTN + 4;
  constexpr int LPR = WTN / VE, RPP = 64 / LPR, NPS = 32 / RPP;   // lanes per slab row, rows per pass, passes per fragment
  const int Cout = p.Cout;
  const int nbase = n0 + wn * WTN;
  const int lr = lane / LPR, lc = (lane - lr * LPR) * VE;
  // Residual (same size / nearest-x2 of a half-size tensor): ALL loads of the wave's four fragments are issued here, in
  // one batch, before the accumulators start moving -- one HBM latency for the whole epilogue instead of one per
  // fragment (the per-fragment form exposed it four times: a same-size residual cost ~20 % on the 128^2 256->256 layers).
  // The fragment registers of the main loop are dead by now, so the 8 pieces fit.
  // (fp32 storage: 8 pieces per lane and fragment -- batching four fragments would spill, so those modes prefetch per fragment)
  constexpr int HB = NPS <= 2 ? MI : 1;   // fragments whose residual loads are batched
  vec_t rres[HB][NPS];
  vec_t rres_lo[LO ? HB : 1][LO ? NPS : 1];
  const bool res_has_lo = LO && p.res_lo != nullptr;
  const bool out_has_lo = LO && p.out_lo != nullptr;
  auto load_res = [&](int mi) {
    const int y = y0 + wm * 4 + mi;
#pragma unroll
    for (int ps = 0; ps < NPS; ++ps) {
      const int xr = x0 + ps * RPP + lr;
      const size_t pix = p.res_mode == 1 ? ((size_t)img * p.H + y) * p.W + xr
                                         : ((size_t)img * (p.H >> 1) + (y >> 1)) * (p.W >> 1) + (xr >> 1);
      rres[mi % HB][ps] = *(const vec_t*)(p.res + (pix * Cout + nbase + lc) * sizeof(T));
      if constexpr (LO) {
        if (res_has_lo) rres_lo[mi % HB][ps] = *(const vec_t*)(p.res_lo + (pix * Cout + nbase + lc) * sizeof(T));
      }
    }
  };
  const bool res12 = (p.res_mode == 1 || p.res_mode == 2) && nbase + lc < Cout;
  if (HB == MI && res12) {
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) load_res(mi);
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // LDS reads of the last K-step done (DMA is waited below)
  {
    // every LDS-DMA of the main loop / skip phase has landed long ago (the last stage was consumed); only the residual
    // loads may be in flight, and they must stay in flight across this barrier
    __builtin_amdgcn_s_barrier();
  }
  float* slab = (float*)smem + wave * (32 * LDC);
  float st_s[VE], st_q[VE];  // GroupNorm partial statistics of this wave's 128 pixels (fused gn_partial)
  float bv[VE];              // this lane's bias values: the same 16-byte channel piece in every pass
  {
    const int nb = nbase + lc;
#pragma unroll
    for (int e = 0; e < VE; ++e) bv[e] = (p.bias && nb < Cout) ? p.bias[nb + e] : 0.f;
  }
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    const int y = y0 + wm * 4 + mi;                        // this fragment = image row y, pixels x0 .. x0+31
    const size_t mbase = ((size_t)img * p.H + y) * p.W + x0;
    if (HB == 1 && res12) load_res(mi);
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * fhalf;
        slab[row * LDC + ni * 32 + frow] = acc[mi][ni][r];
      }
    wave_lds_sync();  // the slab is private to this wave
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
#pragma unroll
        for (int e = 0; e < VE; ++e) v[e] += bv[e];
        if (p.res_mode == 1 || p.res_mode == 2) {
          float rv[VE];
          vec_to_f32<T>(rres[mi % HB][ps], rv);
          if constexpr (LO) {
            if (res_has_lo) {
              float rl[VE];
              vec_to_f32<T>(rres_lo[mi % HB][ps], rl);
#pragma unroll
              for (int e = 0; e < VE; ++e) rv[e] += rl[e];
            }
          }
#pragma unroll
          for (int e = 0; e < VE; ++e) v[e] += rv[e];
        } else if (p.res_mode == 3) {  // residual source is (2H, 2W): 2x2 average pool (Downsample2d on the skip path)
          float sacc[VE];
#pragma unroll
          for (int e = 0; e < VE; ++e) sacc[e] = 0.f;
#pragma unroll
          for (int d = 0; d < 4; ++d) {
            const size_t pix = ((size_t)img * (p.H << 1) + 2 * y + (d >> 1)) * (p.W << 1) + 2 * (x0 + row) + (d & 1);
            float rv[VE];
            vec_to_f32<T>(*(const vec_t*)(p.res + (pix * Cout + n) * sizeof(T)), rv);
#pragma unroll
            for (int e = 0; e < VE; ++e) sacc[e] += rv[e];
            if constexpr (LO) {
              if (res_has_lo) {
                vec_to_f32<T>(*(const vec_t*)(p.res_lo + (pix * Cout + n) * sizeof(T)), rv);
#pragma unroll
                for (int e = 0; e < VE; ++e) sacc[e] += rv[e];
              }
            }
          }
#pragma unroll
          for (int e = 0; e < VE; ++e) v[e] += 0.25f * sacc[e];
        }
        const vec_t ov = f32_to_vec<T>(v);
        *(vec_t*)(p.out + (m * Cout + n) * sizeof(T)) = ov;
        float sv[VE];
        vec_to_f32<T>(ov, sv);
        if constexpr (LO) {
          if (out_has_lo) {   // lo plane: what the 16-bit rounding dropped; the statistics describe hi + lo
            float lv[VE];
#pragma unroll
            for (int e = 0; e < VE; ++e) lv[e] = v[e] - sv[e];
            const vec_t ol = f32_to_vec<T>(lv);
            *(vec_t*)(p.out_lo + (m * Cout + n) * sizeof(T)) = ol;
            vec_to_f32<T>(ol, lv);
#pragma unroll
            for (int e = 0; e < VE; ++e) sv[e] += lv[e];
          }
        }
        if (p.stats) {
#pragma unroll
          for (int e = 0; e < VE; ++e) {
            st_s[e] += sv[e];
            st_q[e] = __builtin_fmaf(sv[e], sv[e], st_q[e]);
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
        // block id inside the image: (4-row band) x (32-pixel column strip), the same partition as the wide kernel writes
        const size_t blk = (size_t)img * (p.H / 4) * p.tiles_x + (size_t)(ty * 4 + wm) * p.tiles_x + tx;
        float* sp = p.stats + (blk * Cout + n) * 2;
#pragma unroll
        for (int e = 0; e < VE; e += 2) *(f32x4*)(sp + e * 2) = f32x4{st_s[e], st_q[e], st_s[e + 1], st_q[e + 1]};
      }
    }
    wave_lds_sync();  // the slab is private to this wave
  }
}

template <typename T, bool LO = false, bool LOIN = false> int launch_fused128(const FusedArgs& a, hipStream_t stream) {
  auto kern = conv3x3_fused128_kernel<T, LO, LOIN>;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    if (e != hipSuccess) return ivid_set_error("conv3x3_gn (128-wide): hipFuncSetAttribute", e);
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3(a.ntiles_total), dim3(NT), LDS_BYTES, stream, a);
  return ivid_check_launch("conv3x3_gn (128-wide)");
}

}  // namespace

// Can this kernel take the layer?  (Cout <= 128, 16 x 32 pixel tiles, 64-byte channel chunks, not bf16x3)
bool ivid_fused128_supports(int dtype, int C0, int C1, int H, int W, int Cout, int skipC0, int skipC1) {
  const int esz = ivid_esz(dtype);
  if (!esz || dtype == IVID_BF16X3) return false;
  const int bke = CHB / esz;
  return Cout <= BN && H % TH == 0 && W % TW == 0 && C0 > 0 && C0 % bke == 0 && C1 >= 0 && C1 % bke == 0 &&
         skipC0 >= 0 && skipC0 % bke == 0 && skipC1 >= 0 && skipC1 % bke == 0;
}

// Arguments already validated by ivid_conv3x3_gn_skip (csrc/conv3x3_fused.hip), which dispatches here.
int ivid_fused128_launch(int dtype, const void* src0, int C0, const void* src1, int C1, const float* ab, int up,
                         const void* weight, const float* bias, void* out, const void* res, int res_mode, int N, int H, int W,
                         int Cout, float* stats, const void* skip0, int skipC0, const void* skip1, int skipC1,
                         const void* skip_weight, void* stream, void* out_lo, const void* res_lo, const void* src0_lo,
                         const void* src1_lo) {
  FusedArgs a;
  a.src0 = (const char*)src0; a.src1 = (const char*)src1; a.ab = ab; a.w = (const char*)weight; a.bias = bias;
  a.out = (char*)out; a.res = (const char*)res; a.zero = (const char*)ivid_zero_page(); a.stats = stats;
  if (!a.zero) return -1;
  a.C0 = C0; a.C1 = C1; a.N = N; a.H = H; a.W = W; a.Cout = Cout; a.up = up ? 1 : 0; a.res_mode = res_mode;
  a.tiles_x = W / TW; a.tiles_y = H / TH; a.ntiles_n = (Cout + BN - 1) / BN;
  a.ntiles_total = N * a.tiles_x * a.tiles_y * a.ntiles_n;
  a.sk0 = (const char*)skip0; a.sk1 = (const char*)skip1; a.skw = (const char*)skip_weight; a.skC0 = skipC0; a.skC1 = skipC1;
  a.out_lo = (char*)out_lo; a.res_lo = (const char*)res_lo; a.src0_lo = (const char*)src0_lo; a.src1_lo = (const char*)src1_lo;
  if (src0_lo || src1_lo) {
    if (dtype == IVID_BF16) return launch_fused128<__bf16, true, true>(a, (hipStream_t)stream);
    return launch_fused128<_Float16, true, true>(a, (hipStream_t)stream);
  }
  if (out_lo || res_lo) {
    if (dtype == IVID_BF16) return launch_fused128<__bf16, true>(a, (hipStream_t)stream);
    return launch_fused128<_Float16, true>(a, (hipStream_t)stream);
  }
  if (dtype == IVID_BF16) return launch_fused128<__bf16>(a, (hipStream_t)stream);
  if (dtype == IVID_F16) return launch_fused128<_Float16>(a, (hipStream_t)stream);
  return launch_fused128<float>(a, (hipStream_t)stream);
}
