// Implicit-GEMM convolution (3x3 pad 1 / 1x1 / Linear) on the CDNA4 matrix cores.
//
// Replaces every nn.Conv2d / nn.Conv1d(k=1) / nn.Linear call of the reference UNet
// (/root/reference/diffusion/backbones/adm.py:160,182,190,275,278,359,361,176,369,486).
//
//   out[m][co] = bias[co] + sum_{tap,c} act[pixel(m)+tap][c] * w[co][tap][c]   (+ residual)
//
// GEMM view: M = N*H*W output pixels, N = Cout, K = taps*(C0+C1).  Both operands are
// "K-contiguous rows": an A row is a pixel's channel vector (NHWC), a B row is one output
// channel's [tap][cin] weights.  A K-step is 128 bytes of K (64 bf16 / 32 fp32) of ONE tap,
// so the im2col gather is a per-lane source address into the NHWC tensor (or into a zero page
// for padding) fed to global_load_lds (HBM/L2 -> LDS without touching VGPRs).
//
// LDS image per operand and stage: [rows][128 B], 16-byte pieces XOR-swizzled by
// ((row>>1)&7) so that the ds_read_b128 of an MFMA fragment (32 rows x one piece column)
// is bank-conflict free.  global_load_lds writes lane-linear, so the swizzle is applied to
// the per-lane SOURCE address and again on the fragment read (same involution).
//
// MFMA: perf mode  v_mfma_f32_32x32x16_bf16 (bf16 in, fp32 accumulate),
//       parity mode v_mfma_f32_32x32x2_f32  (exact fp32 == fmaf chain).
// C/D layout (both): col = lane&31 (cout), row = (reg&3)+8*(reg>>2)+4*(lane>>5) (pixel).
//
// Epilogue: accumulators -> per-wave LDS slab -> 16-byte coalesced NHWC stores with fused
// bias, residual add (same / nearest-up x2 / avg-pool-down x2 of the residual source, i.e.
// ResBlock2d's x_upd skip path, adm.py:203-208,222) or fp32 NCHW output for the final conv.
#include "common.h"
#include "internal.h"

namespace {

struct ConvArgs {
  const char* src0;
  const char* src1;
  const char* w;
  const float* bias;
  char* out;
  const char* res;
  const char* zero;
  int C0, C1;
  int N, H, W;
  int Cout;
  int taps;      // 1 or 9; 4 = phase-decomposed "nearest x2 upsample + conv 3x3" (ivid_conv3x3_up, see there)
  int nt_phase;  // taps == 4: cout tiles per output phase (ntiles_n = 4 * nt_phase), else 0
  int res_mode;  // 0 none, 1 same size, 2 residual is (H/2,W/2) nearest-up, 3 residual is (2H,2W) avg-pool
  int out_mode;  // 0 NHWC T, 1 NCHW fp32
  int M;
  int ntiles_n;
  int ntiles_total;
  float* stats;  // optional: per (32-pixel row block, cout) sum / sum of squares of the STORED output, [M/32][Cout][2]
  // Compensated 16-bit storage (precision mode fp16c): a tensor may carry a second plane `lo` with the part of the fp32
  // value the 16-bit rounding dropped, lo = T(v - float(T(v))), so that hi + lo keeps 22 mantissa bits.  MFMA operands
  // read the hi plane alone (it IS the rounded operand); residual adds and GroupNorm kernels read hi + lo.
  char* out_lo;         // optional lo plane of the NHWC output
  const char* res_lo;   // optional lo plane of the residual source
  // IVID_BF16X3 only (fp32 storage): optional fp16 twin of the NHWC output, hi = fp16(v), lo = fp16(v - hi) -- the form the
  // 16-bit part of the network reads (the stem of the fp16s mode's split-precision island, ivid_conv2d_o16)
  char* out16_hi;
  char* out16_lo;
};

// UP4: the phase-decomposed "nearest x2 upsample + conv 3x3" form (taps == 4, ivid_conv3x3_up) -- a separate
// instantiation, so that the plain convolution's issue path and epilogue carry none of its branches / index arithmetic
// (a runtime switch cost the plain launches 3-8 %)
template <typename T, int BM, int BN, int WAVES_M, int WAVES_N, bool UP4>
__global__ __launch_bounds__(WAVES_M* WAVES_N * 64) void conv_igemm_kernel(const ConvArgs p) {
  typedef typename Elem<T>::vec vec_t;
  constexpr int NT = WAVES_M * WAVES_N * 64;
  constexpr int VE = Elem<T>::VE;
  constexpr int BKE = 128 / (int)sizeof(T);  // elements of K per step
  constexpr int WTM = BM / WAVES_M, WTN = BN / WAVES_N;
  constexpr int MI = WTM / 32, NI = WTN / 32;
  constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;
  constexpr int A_IT = BM * 8 / NT, B_IT = BN * 8 / NT;
  static_assert(BM * 8 % NT == 0 && BN * 8 % NT == 0, "tile/thread mismatch");

  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tile = xcd_remap(blockIdx.x, p.ntiles_total);
  const int tm = tile / p.ntiles_n;
  int tn = tile - tm * p.ntiles_n;
  // taps == 4: the cout tiles of the four output phases (py, px) of one pixel tile are neighbours (shared A in L2)
  int phase = 0;
  if constexpr (UP4) {
    phase = tn / p.nt_phase;
    tn -= phase * p.nt_phase;
  }
  const int py = phase >> 1, px = phase & 1;
  const int m0 = tm * BM, n0 = tn * BN;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // provably wave-uniform: LDS-DMA bases stay in SGPRs
  const int wm = wave / WAVES_N, wn = wave - wm * WAVES_N;
  const int HW = p.H * p.W;
  const int Ctot = p.C0 + p.C1;
  const int chunks = Ctot / BKE;  // K-steps per tap
  const int nk = p.taps * chunks;
  const size_t Ktot = (size_t)p.taps * Ctot;

  // ---- per-thread gather descriptors (K-step invariant) ----
  // A piece i of this thread: tile row (pixel) and 16-byte column; its source address for K-step (tap, chunk) is
  //   base{0,1}[i] + ((dy*W + dx)*Cs + chunk_offset) * sizeof(T)      (64-bit base + one wave-uniform 32-bit delta)
  // or the zero page when the shifted pixel falls outside the image / the row is beyond M.  No branches, no 64-bit
  // multiplies in the loop: everything the loop adds is uniform across the wave.
  const char* a_base0[A_IT];
  const char* a_base1[A_IT];
  int a_y[A_IT], a_x[A_IT];
#pragma unroll
  for (int i = 0; i < A_IT; ++i) {
    const int q = i * NT + tid;
    const int row = q >> 3, pos = q & 7;
    const int cpiece = (pos ^ ((row >> 1) & 7)) * 16;  // byte offset of the (swizzled) piece inside the 128-byte slab
    const int m = m0 + row;
    const bool ok = m < p.M;
    const int mm = ok ? m : 0;
    const int n = mm / HW, rem = mm - n * HW;
    const int y = rem / p.W;
    a_y[i] = ok ? y : -0x40000000;  // out-of-range rows fail every bounds test
    a_x[i] = rem - y * p.W;
    a_base0[i] = p.src0 + (size_t)mm * p.C0 * sizeof(T) + cpiece;
    a_base1[i] = p.src1 ? p.src1 + (size_t)mm * p.C1 * sizeof(T) + cpiece : p.zero;
  }
  const char* b_ptr[B_IT];
#pragma unroll
  for (int i = 0; i < B_IT; ++i) {
    const int q = i * NT + tid;
    const int row = q >> 3, pos = q & 7;
    const int c = (pos ^ ((row >> 1) & 7)) * VE;
    const int co = n0 + row;
    b_ptr[i] = (co < p.Cout) ? p.w + (((size_t)phase * p.Cout + co) * Ktot + c) * sizeof(T) : nullptr;
  }

  // K-loop order: channel-chunk-major, taps inner — the 9 shifted re-reads of one 128-byte channel slab are back to
  // back, so they hit in the XCD's L2 (tile working set ~50 KB) instead of re-streaming the whole channel extent of
  // the tile once per tap.  (The order must be a compile-time property: a runtime switch here made hipcc place an
  // s_waitcnt vmcnt(0) between the A and the B batch of global_load_lds, serialising the load latency.)
  int ld_tap = 0, ld_ch = 0;  // K-step that the next issue() loads
  auto issue = [&](int stage) {
    char* sA = smem + stage * STAGE;
    char* sB = sA + A_BYTES;
    const int cbase = ld_ch * BKE;
    const bool second = cbase >= p.C0;
    const int Cs = second ? p.C1 : p.C0;
    const int coff = second ? cbase - p.C0 : cbase;
    int dy = 0, dx = 0;
    if constexpr (UP4) {  // tap (a, b) of output phase (py, px) reads the source pixel at (py - 1 + a, px - 1 + b)
      dy = py - 1 + (ld_tap >> 1);
      dx = px - 1 + (ld_tap & 1);
    } else if (p.taps == 9) {
      dy = ld_tap / 3 - 1;
      dx = ld_tap - (dy + 1) * 3 - 1;
    }
    const int delta = ((dy * p.W + dx) * Cs + coff) * (int)sizeof(T);  // wave-uniform, |delta| < 2^31
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
      const bool inb = (unsigned)(a_y[i] + dy) < (unsigned)p.H && (unsigned)(a_x[i] + dx) < (unsigned)p.W;
      const char* g = (second ? a_base1[i] : a_base0[i]) + delta;
      g = inb ? g : p.zero;
      glds16(g, sA + (i * NT + wave * 64) * 16);
    }
    const size_t koff = ((size_t)ld_tap * Ctot + cbase) * sizeof(T);
#pragma unroll
    for (int i = 0; i < B_IT; ++i) {
      const char* g = b_ptr[i] ? b_ptr[i] + koff : p.zero;
      glds16(g, sB + (i * NT + wave * 64) * 16);
    }
    if (++ld_tap == p.taps) {
      ld_tap = 0;
      ++ld_ch;
    }
  };
  // (Measured and dropped, round 2: spreading the eight LDS-DMA pieces of the next step between the MFMA groups of the
  // current one, with a single rotating fragment set -- +4..5 % kernel time on every layer: a DMA issue among MFMAs costs
  // more than it hides, and the two waves of a SIMD already overlap each other's issue block.)

  f32x16 acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

  // fragment read offsets: row (lane&31) of a 32-row block, piece 2*kk + (lane>>5)
  const int frow = lane & 31, fhalf = lane >> 5;
  int a_off[MI], b_off[NI], a_sw[MI], b_sw[NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    const int row = wm * WTM + mi * 32 + frow;
    a_off[mi] = row * 128;
    a_sw[mi] = (row >> 1) & 7;
  }
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    const int row = wn * WTN + ni * 32 + frow;
    b_off[ni] = row * 128;
    b_sw[ni] = (row >> 1) & 7;
  }

  auto load_frags = [&](const char* sA, const char* sB, int kk, vec_t* a, vec_t* b) {
    const int piece = 2 * kk + fhalf;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) a[mi] = *(const vec_t*)(sA + a_off[mi] + ((piece ^ a_sw[mi]) << 4));
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) b[ni] = *(const vec_t*)(sB + b_off[ni] + ((piece ^ b_sw[ni]) << 4));
  };

  // Skewed issue (round 6, 8-wave tiles): the waves of the workgroup's first half issue the next stage's LDS-DMA in front of
  // their MFMAs, their SIMD partners (waves w + 4) behind their first k-piece -- one wave of every SIMD feeds the matrix pipe
  // while the other one is in its issue block (8 DMA pieces = several hundred cycles of an in-order wave).  In lockstep both
  // waves of a SIMD issued first and shared the pipe afterwards: 16^2 768 -> 768 0.377 -> 0.342 ms, 1792 -> 768 0.820 -> 0.764,
  // 1024 -> 1024 0.553 -> 0.534 (same box; behind the second k-piece: -3 %, behind the third: -1 %, four-way stagger: 0).
  // The stage being filled is free for the whole step (its last readers passed this step's barrier), so the position is a pure
  // scheduling choice; the data is waited for in front of the next barrier as before.
  const bool late = NT >= 512 && wave >= NT / 128;
  issue(0);
  for (int kt = 0; kt < nk; ++kt) {
    wait_vmcnt0();
    __syncthreads();  // stage kt&1 landed for every wave; everyone finished reading the other stage
    const char* sA = smem + (kt & 1) * STAGE;
    const char* sB = sA + A_BYTES;
    // the first fragments are requested BEFORE the next stage's 8 LDS-DMA loads are issued, so their LDS latency
    // hides behind the address arithmetic instead of sitting in front of the first MFMA of the K-step
    if constexpr (IsSplit<T>::value) {
      // bf16x3: the K-step is 32 channels.  A stage = fp32 activations (split into bf16 hi/lo in registers, ~20 VALU
      // ops per fragment against 3 x NI MFMAs of 8 passes each); B stage = host-split weights, pieces 2j / 2j+1 =
      // hi / lo of channels 8j..8j+7.  MFMA s takes channels 16s + 8*(lane>>5) .. +7 per lane half.
      auto load_split = [&](int sidx, bf16x8* ah, bf16x8* al, bf16x8* bh, bf16x8* bl) {
        const int piece = 4 * sidx + 2 * fhalf;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
          const f32x4 x0 = *(const f32x4*)(sA + a_off[mi] + ((piece ^ a_sw[mi]) << 4));
          const f32x4 x1 = *(const f32x4*)(sA + a_off[mi] + (((piece + 1) ^ a_sw[mi]) << 4));
          split_bf16x8(x0, x1, ah[mi], al[mi]);
        }
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
          bh[ni] = *(const bf16x8*)(sB + b_off[ni] + ((piece ^ b_sw[ni]) << 4));
          bl[ni] = *(const bf16x8*)(sB + b_off[ni] + (((piece + 1) ^ b_sw[ni]) << 4));
        }
      };
      bf16x8 ah[2][MI], al[2][MI], bh[2][NI], bl[2][NI];
      load_split(0, ah[0], al[0], bh[0], bl[0]);
      if (!late && kt + 1 < nk) issue((kt + 1) & 1);
#pragma unroll
      for (int sidx = 0; sidx < 2; ++sidx) {
        if (sidx == 0) load_split(1, ah[1], al[1], bh[1], bl[1]);
        if (sidx == 1 && late && kt + 1 < nk) issue((kt + 1) & 1);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
          for (int ni = 0; ni < NI; ++ni) {
            f32x16& c = acc[mi][ni];
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[sidx][mi], bh[sidx][ni], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[sidx][mi], bl[sidx][ni], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[sidx][mi], bh[sidx][ni], c, 0, 0, 0);
          }
      }
    } else {
    vec_t a[2][MI], b[2][NI];
    load_frags(sA, sB, 0, a[0], b[0]);
    if (!late && kt + 1 < nk) issue((kt + 1) & 1);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      if (kk < 3) load_frags(sA, sB, kk + 1, a[(kk + 1) & 1], b[(kk + 1) & 1]);
      if (kk == 1 && late && kt + 1 < nk) issue((kt + 1) & 1);
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) MmaT<T>::run(a[kk & 1][mi], b[kk & 1][ni], acc[mi][ni]);
    }
    }
  }

  // ---------------- epilogue ----------------
  __syncthreads();  // all waves done with the operand stages (every DMA was waited for in the loop); LDS is reused as per-wave slabs
  constexpr int LDC = WTN + 4;  // floats per slab row (pad keeps the two half-waves on different banks)
  float* slab = (float*)smem + wave * (32 * LDC);
  const int Cout = p.Cout;
  // GroupNorm partial statistics of this wave's pixels (fused gn_partial).  The fp32 summation ORDER of a 64-pixel block must
  // not depend on the tile a launch happens to use (the choice depends on the batch size, and a sample's result has to
  // be bit-identical in any batch / on any rank): canonical order = the one of the 64-channel-wide wave tiles -- rows of
  // equal (row mod VE) are summed sequentially in increasing row order, the VE partial sums are combined by a binary
  // tree over the bits of (row mod VE), lowest bit first.  A 128-wide wave tile covers fewer rows per pass (RPP = VE/2):
  // it keeps NA = 2 accumulators per lane (row mod VE = lr and lr + RPP), runs the tree over the lane bits first and
  // adds the two accumulators last -- the same association.
  // (A 96-wide wave tile -- the 128 x 384 tile -- needs 12 lanes per slab row: it allocates 16, the last 4 idle, which
  // gives it the 128-wide tile's row schedule and therefore the same two-accumulator order.)
  constexpr int LPRN = WTN / VE;                                           // lanes a slab row needs
  constexpr int LPR = LPRN <= 4 ? 4 : LPRN <= 8 ? 8 : LPRN <= 16 ? 16 : 32;   // lanes per slab row (power of two)
  constexpr int RPP = 64 / LPR;   // rows per pass
  constexpr int NPS = 32 / RPP;   // passes per fragment
  constexpr int NA = RPP < VE ? VE / RPP : 1;
  float st_s[NA][VE], st_q[NA][VE];
  const int nbase = n0 + wn * WTN;
  const int lr = lane / LPR;
  const bool lane_on = (lane - lr * LPR) < LPRN;                            // idle lanes of a padded row do nothing
  const int lc = lane_on ? (lane - lr * LPR) * VE : (1 << 28);              // ... because their channel fails every n < Cout test
  float bv[VE];              // this lane's bias values (NHWC path): the same 16-byte channel piece in every pass
  {
    const int nb = nbase + lc;
#pragma unroll
    for (int e = 0; e < VE; ++e) bv[e] = (p.bias && nb < Cout) ? p.bias[nb + e] : 0.f;
  }
  // same-size residual: the loads of ALL fragments of the wave are issued up front, in one batch -- one memory latency
  // for the whole epilogue instead of one per fragment; their latency hides behind the first transpose
  // (fp32 storage: a fragment's residual is 8 pieces per lane -- batching all fragments would spill, so those modes keep
  // the per-fragment prefetch)
  constexpr int HB = NPS <= 2 ? MI : 1;   // fragments whose residual loads are batched
  constexpr bool LO = sizeof(T) == 2 && !UP4;   // compensated storage exists for the 16-bit types only
  vec_t rres[HB][NPS];
  vec_t rres_lo[LO ? HB : 1][LO ? NPS : 1];
  const bool res_has_lo = LO && p.res_lo != nullptr;
  const bool out_has_lo = LO && p.out_lo != nullptr;
  auto load_res = [&](int mi) {
#pragma unroll
    for (int ps = 0; ps < NPS; ++ps) {
      const int m = m0 + wm * WTM + mi * 32 + ps * RPP + lr, n = nbase + lc;
      if (m < p.M && n < Cout) {
        rres[mi % HB][ps] = *(const vec_t*)(p.res + ((size_t)m * Cout + n) * sizeof(T));
        if constexpr (LO) {
          if (res_has_lo) rres_lo[mi % HB][ps] = *(const vec_t*)(p.res_lo + ((size_t)m * Cout + n) * sizeof(T));
        }
      }
    }
  };
  if (HB == MI && p.out_mode == 0 && p.res_mode == 1) {
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) load_res(mi);
  }
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    const int mbase = m0 + wm * WTM + mi * 32;
    if (HB == 1 && p.out_mode == 0 && p.res_mode == 1) load_res(mi);
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * fhalf;
        slab[row * LDC + ni * 32 + frow] = acc[mi][ni][r];
      }
    wave_lds_sync();  // the slab is private to this wave
    if (p.out_mode == 0) {
      // statistics blocks are ALWAYS 64 consecutive pixels (32 for the narrow tile), whatever tile the launch uses:
      // the fp32 summation order is then independent of the batch-size-dependent tile choice, so a sample's result
      // is bit-identical in any batch (in bf16 mode a 1e-7 change of a GroupNorm coefficient decorrelates rounding
      // decisions downstream up to the bf16 noise floor)
      constexpr int FL = MI >= 2 ? 2 : 1;  // fragments per statistics block
      if (mi % FL == 0) {
#pragma unroll
        for (int j = 0; j < NA; ++j)
#pragma unroll
          for (int e = 0; e < VE; ++e) st_s[j][e] = st_q[j][e] = 0.f;
      }
#pragma unroll
      for (int ps = 0; ps < 32 / RPP; ++ps) {
        const int row = ps * RPP + lr;
        const int m = mbase + row, n = nbase + lc;
        if (m < p.M && n < Cout) {
          float v[VE];
#pragma unroll
          for (int e = 0; e < VE; e += 4) {
            const f32x4 t = *(const f32x4*)(slab + row * LDC + lc + e);
            v[e] = t[0]; v[e + 1] = t[1]; v[e + 2] = t[2]; v[e + 3] = t[3];
          }
#pragma unroll
          for (int e = 0; e < VE; ++e) v[e] += bv[e];
          if (p.res_mode == 1) {
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
          } else if (p.res_mode != 0) {
            const int img = m / HW, rem = m - img * HW;
            const int y = rem / p.W, x = rem - y * p.W;
            if (p.res_mode == 2) {  // residual source is (H/2, W/2): nearest-neighbour x2
              const int Hs = p.H >> 1, Ws = p.W >> 1;
              const size_t pix = ((size_t)img * Hs + (y >> 1)) * Ws + (x >> 1);
              float rv[VE];
              vec_to_f32<T>(*(const vec_t*)(p.res + (pix * Cout + n) * sizeof(T)), rv);
              if constexpr (LO) {
                if (res_has_lo) {
                  float rl[VE];
                  vec_to_f32<T>(*(const vec_t*)(p.res_lo + (pix * Cout + n) * sizeof(T)), rl);
#pragma unroll
                  for (int e = 0; e < VE; ++e) rv[e] += rl[e];
                }
              }
#pragma unroll
              for (int e = 0; e < VE; ++e) v[e] += rv[e];
            } else {  // residual source is (2H, 2W): 2x2 average pool
              const int Hs = p.H << 1, Ws = p.W << 1;
              float s[VE];
#pragma unroll
              for (int e = 0; e < VE; ++e) s[e] = 0.f;
#pragma unroll
              for (int d = 0; d < 4; ++d) {
                const size_t pix = ((size_t)img * Hs + 2 * y + (d >> 1)) * Ws + 2 * x + (d & 1);
                float rv[VE];
                vec_to_f32<T>(*(const vec_t*)(p.res + (pix * Cout + n) * sizeof(T)), rv);
#pragma unroll
                for (int e = 0; e < VE; ++e) s[e] += rv[e];
                if constexpr (LO) {
                  if (res_has_lo) {
                    vec_to_f32<T>(*(const vec_t*)(p.res_lo + (pix * Cout + n) * sizeof(T)), rv);
#pragma unroll
                    for (int e = 0; e < VE; ++e) s[e] += rv[e];
                  }
                }
              }
#pragma unroll
              for (int e = 0; e < VE; ++e) v[e] += 0.25f * s[e];
            }
          }
          const vec_t ov = f32_to_vec<T>(v);
          size_t mo = (size_t)m;
          if constexpr (UP4) {  // source pixel (img, y, x) of phase (py, px) -> output pixel (2y + py, 2x + px) of the 2H x 2W image
            const int img = m / HW, rem = m - img * HW;
            const int y = rem / p.W, x = rem - y * p.W;
            mo = ((size_t)img * (2 * p.H) + 2 * y + py) * (2 * p.W) + 2 * x + px;
          }
          *(vec_t*)(p.out + (mo * Cout + n) * sizeof(T)) = ov;
          if constexpr (IsSplit<T>::value) {
            if (p.out16_hi) {   // fp32 storage (VE = 4): 8-byte stores of the fp16 twin
              f16x4 th, tl;
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                th[e] = (_Float16)v[e];
                tl[e] = (_Float16)(v[e] - (float)th[e]);
              }
              *(f16x4*)(p.out16_hi + (mo * Cout + n) * 2) = th;
              *(f16x4*)(p.out16_lo + (mo * Cout + n) * 2) = tl;
            }
          }
          float sv[VE];
          vec_to_f32<T>(ov, sv);  // statistics of the values the consumer will actually read
          if constexpr (LO) {
            if (out_has_lo) {   // lo plane: what the rounding dropped (exact difference, rounded once); statistics of hi + lo
              float lv[VE];
#pragma unroll
              for (int e = 0; e < VE; ++e) lv[e] = v[e] - sv[e];
              const vec_t ol = f32_to_vec<T>(lv);
              *(vec_t*)(p.out_lo + (mo * Cout + n) * sizeof(T)) = ol;
              vec_to_f32<T>(ol, lv);
#pragma unroll
              for (int e = 0; e < VE; ++e) sv[e] += lv[e];
            }
          }
          if (p.stats) {
#pragma unroll
            for (int e = 0; e < VE; ++e) {
              st_s[ps % NA][e] += sv[e];
              st_q[ps % NA][e] = __builtin_fmaf(sv[e], sv[e], st_q[ps % NA][e]);   // explicit: contraction must not vary per instantiation
            }
          }
        }
      }
      if (p.stats && mi % FL == FL - 1) {
#pragma unroll
        for (int off = LPR; off < 64; off <<= 1) {
#pragma unroll
          for (int e = 0; e < VE; ++e) {
#pragma unroll
            for (int j = 0; j < NA; ++j) {
              st_s[j][e] += __shfl_xor(st_s[j][e], off);
              st_q[j][e] += __shfl_xor(st_q[j][e], off);
            }
          }
        }
        const int n = nbase + lc;
        const int wbase = m0 + wm * WTM + (mi / FL) * (FL * 32);  // first pixel of this block
        if (lr == 0 && wbase < p.M && n < Cout) {
          size_t slot = (size_t)(wbase / (FL * 32));
          if constexpr (UP4) {  // blocks of the 2H x 2W output image: [image][phase][source block] (any partition of an image works)
            const int bpi = HW / (FL * 32), img = (int)slot / bpi;
            slot = (size_t)img * (4 * bpi) + (size_t)phase * bpi + (slot - (size_t)img * bpi);
          }
          float* sp = p.stats + (slot * Cout + n) * 2;
#pragma unroll
          for (int e = 0; e < VE; ++e)
#pragma unroll
            for (int j = 1; j < NA; ++j) { st_s[0][e] += st_s[j][e]; st_q[0][e] += st_q[j][e]; }
#pragma unroll
          for (int e = 0; e < VE; e += 2) *(f32x4*)(sp + e * 2) = f32x4{st_s[0][e], st_q[0][e], st_s[0][e + 1], st_q[0][e + 1]};
        }
      }
    } else {  // fp32 NCHW output (small Cout): lanes run along pixels for coalescing
      const int ncols = min(WTN, Cout - nbase);
      for (int idx = lane; idx < 32 * ncols; idx += 64) {
        const int row = idx & 31, col = idx >> 5;
        const int m = mbase + row;
        if (m < p.M) {
          const int img = m / HW, rem = m - img * HW;
          float v = slab[row * LDC + col];
          if (p.bias) v += p.bias[nbase + col];
          ((float*)p.out)[((size_t)img * Cout + nbase + col) * HW + rem] = v;
        }
      }
    }
    wave_lds_sync();  // the slab is private to this wave
  }
}

template <typename T, int BM, int BN, int WAVES_M, int WAVES_N, bool UP4 = false>
int launch_conv(const ConvArgs& a0, hipStream_t stream) {
  ConvArgs a = a0;
  constexpr int NT = WAVES_M * WAVES_N * 64;
  constexpr int STAGE = (BM + BN) * 128;
  constexpr int EPI = WAVES_M * WAVES_N * 32 * (BN / WAVES_N + 4) * 4;
  constexpr int SMEM = (2 * STAGE > EPI) ? 2 * STAGE : EPI;
  static_assert(SMEM <= 160 * 1024, "LDS budget");
  const int mt = (a.M + BM - 1) / BM, nt = (a.Cout + BN - 1) / BN;
  a.nt_phase = UP4 ? nt : 0;
  a.ntiles_n = UP4 ? 4 * nt : nt;
  a.ntiles_total = mt * a.ntiles_n;
  auto kern = conv_igemm_kernel<T, BM, BN, WAVES_M, WAVES_N, UP4>;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
    if (e != hipSuccess) return ivid_set_error("conv: hipFuncSetAttribute", e);
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3(a.ntiles_total), dim3(NT), SMEM, stream, a);
  return ivid_check_launch("conv_igemm");
}

}  // namespace

// Tile of a launch (tile_cfg 0 = auto).  Per-wave tiles and what they cost in LDS reads per MFMA (the limiter of this
// kernel: 128 B/clk/CU of ds_read against 4 SIMDs of MFMA):
//   2: 256 x 256, 8 waves 2x4, wave 128 x 64  (6 reads / 8 MFMA)   wide layers whenever >= 1.5 rounds of workgroups exist
//   4: 512 x 128, 8 waves 8x1, wave  64 x 128 (6 reads / 8 MFMA)   Cout <= 128 (the small / SR models' first levels): the same
//                                                                   read:MFMA ratio as tile 2 instead of tile 1's 4 / 4
//   1: 128 x 128, 4 waves 2x2, wave  64 x 64  (4 reads / 4 MFMA)   everything else; two workgroups per CU
//   5:  64 x 128, 2 waves 1x2, wave  64 x 64  (4 reads / 4 MFMA)   tiny problems (8^2 levels at small batch): twice the
//                                                                   workgroups of tile 1 when that one leaves CUs idle
//   6: 128 x 384, 8 waves 2x4, wave  64 x 96  (5 reads / 6 MFMA)   Cout a multiple of 384 when tile 2 would leave a fractional
//                                                                   last round (768 channels at 16^2, batch 128: 384 tiles of
//                                                                   256 x 256 = 1.5 rounds of 256 CUs, 512 of these = 2 rounds)
//   3: 128 x 32,  4 waves 4x1                                       Cout <= 32 (the 4-channel output conv)
static int ivid_conv_pick_tile(long long M, int Cout, int tile_cfg, int nmult = 1) {   // nmult: cout-tile sets per pixel tile (4 in phase mode)
  int cfg = tile_cfg & 7;
  if (cfg != 0) return cfg;
  if (Cout <= 32) return 3;
  const long long big = ((M + 255) / 256) * ((Cout + 255) / 256) * nmult;
  if (Cout % 384 == 0 && big >= 256) {   // occupancy of the last round: tile 2 vs tile 6
    const long long t6 = ((M + 127) / 128) * (Cout / 384) * nmult;
    const double e2 = (double)big / (double)(((big + 255) / 256) * 256), e6 = (double)t6 / (double)(((t6 + 255) / 256) * 256);
    if (e2 < 0.8 && e6 > e2 + 0.1) return 6;
  }
  if (Cout >= 256 && big >= 384) return 2;
  if (Cout <= 128 && Cout > 64 && (M + 511) / 512 * nmult >= 256) return 4;
  const long long mid = ((M + 127) / 128) * ((Cout + 127) / 128) * nmult;
  if (mid < 256 && Cout >= 64) return 5;
  return 1;
}

static int conv2d_any(int dtype, const void* src0, int C0, const void* src1, int C1, const void* weight,
                      const float* bias, void* out, const void* res, int res_mode, int out_mode, int N, int H,
                      int W, int Cout, int taps, int tile_cfg, float* stats, void* stream, void* out_lo = nullptr,
                      const void* res_lo = nullptr, void* out16_hi = nullptr, void* out16_lo = nullptr) {
  const int esz = ivid_esz(dtype);
  if (!esz) return ivid_set_error("conv: bad dtype", hipSuccess);
  const int bke = 128 / esz, ve = 16 / esz;
  if (taps != 1 && taps != 9 && taps != 4) return ivid_set_error("conv: taps must be 1 or 9", hipSuccess);
  if (taps == 4 && (res_mode != 0 || out_mode != 0 || Cout <= 32 || (H * W) % 64 || (long long)N * H * W * 4 >= (1ll << 31)))
    return ivid_set_error("conv3x3_up: needs NHWC output without residual, Cout > 32, H*W % 64 == 0", hipSuccess);
  if (C0 <= 0 || C0 % bke || C1 < 0 || C1 % bke) return ivid_set_error("conv: channels must be multiples of the K-step", hipSuccess);
  if (C1 > 0 && !src1) return ivid_set_error("conv: src1 missing", hipSuccess);
  if (out_mode == 0 && Cout % ve) return ivid_set_error("conv: Cout must be a multiple of 16 bytes for NHWC output", hipSuccess);
  if (res_mode < 0 || res_mode > 3 || (res_mode && !res)) return ivid_set_error("conv: bad residual", hipSuccess);
  if (res_mode == 2 && ((H | W) & 1)) return ivid_set_error("conv: up-residual needs even H,W", hipSuccess);
  if ((long long)N * H * W >= (1ll << 31)) return ivid_set_error("conv: M too large", hipSuccess);
  if ((out_lo || res_lo) && (esz != 2 || taps == 4 || out_mode != 0))
    return ivid_set_error("conv: lo planes need a 16-bit dtype, NHWC output and taps 1 / 9", hipSuccess);
  if (res_lo && !res_mode) return ivid_set_error("conv: res_lo without a residual", hipSuccess);
  ConvArgs a;
  a.src0 = (const char*)src0; a.src1 = (const char*)src1; a.w = (const char*)weight; a.bias = bias;
  a.out = (char*)out; a.res = (const char*)res; a.zero = (const char*)ivid_zero_page();
  a.out_lo = (char*)out_lo; a.res_lo = (const char*)res_lo;
  a.out16_hi = (char*)out16_hi; a.out16_lo = (char*)out16_lo;
  if (out16_hi && (dtype != IVID_BF16X3 || !out16_lo || out_mode != 0 || taps == 4))
    return ivid_set_error("conv2d_o16: IVID_BF16X3, NHWC output, both fp16 planes", hipSuccess);
  if (!a.zero) return -1;
  a.C0 = C0; a.C1 = C1; a.N = N; a.H = H; a.W = W; a.Cout = Cout; a.taps = taps;
  a.res_mode = res_mode; a.out_mode = out_mode; a.M = N * H * W; a.ntiles_n = 0; a.ntiles_total = 0; a.nt_phase = 0;
  a.stats = stats;
  hipStream_t s = (hipStream_t)stream;
  tile_cfg = ivid_conv_pick_tile(a.M, Cout, tile_cfg, taps == 4 ? 4 : 1);
  if (taps == 4 && tile_cfg == 3) return ivid_set_error("conv3x3_up: the 128x32 tile is not available here", hipSuccess);
  if (tile_cfg < 1 || tile_cfg > 6) return ivid_set_error("conv: tile_cfg must be 0 (auto) .. 6", hipSuccess);
  if (stats) {  // a statistics block must not straddle two images
    const int gran = tile_cfg == 3 ? 32 : 64;
    if (out_mode != 0 || (H * W) % gran) return ivid_set_error("conv: stats need NHWC output and H*W % block == 0", hipSuccess);
  }
#define IVID_CONV_DISPATCH(TT)                                             \
  do {                                                                     \
    if (taps == 4) {                                                       \
      if (tile_cfg == 2) return launch_conv<TT, 256, 256, 2, 4, true>(a, s); \
      if (tile_cfg == 4) return launch_conv<TT, 512, 128, 8, 1, true>(a, s); \
      if (tile_cfg == 5) return launch_conv<TT, 64, 128, 1, 2, true>(a, s);  \
      if (tile_cfg == 6) return launch_conv<TT, 128, 384, 2, 4, true>(a, s); \
      return launch_conv<TT, 128, 128, 2, 2, true>(a, s);                  \
    }                                                                      \
    if (tile_cfg == 2) return launch_conv<TT, 256, 256, 2, 4>(a, s);       \
    if (tile_cfg == 3) return launch_conv<TT, 128, 32, 4, 1>(a, s);        \
    if (tile_cfg == 4) return launch_conv<TT, 512, 128, 8, 1>(a, s);       \
    if (tile_cfg == 5) return launch_conv<TT, 64, 128, 1, 2>(a, s);        \
    if (tile_cfg == 6) return launch_conv<TT, 128, 384, 2, 4>(a, s);       \
    return launch_conv<TT, 128, 128, 2, 2>(a, s);                          \
  } while (0)
  if (dtype == IVID_BF16) IVID_CONV_DISPATCH(__bf16);
  if (dtype == IVID_F16) IVID_CONV_DISPATCH(_Float16);
  if (dtype == IVID_BF16X3) IVID_CONV_DISPATCH(bf16x3_t);
  IVID_CONV_DISPATCH(float);
#undef IVID_CONV_DISPATCH
}

extern "C" int ivid_conv2d(int dtype, const void* src0, int C0, const void* src1, int C1, const void* weight,
                           const float* bias, void* out, const void* res, int res_mode, int out_mode, int N, int H,
                           int W, int Cout, int taps, int tile_cfg, float* stats, void* stream) {
  if (taps != 1 && taps != 9) return ivid_set_error("conv: taps must be 1 or 9", hipSuccess);
  return conv2d_any(dtype, src0, C0, src1, C1, weight, bias, out, res, res_mode, out_mode, N, H, W, Cout, taps, tile_cfg, stats,
                    stream);
}

// ivid_conv2d with compensated 16-bit storage (precision mode fp16c): `out_lo` / `res_lo` are the optional lo planes of the
// output and of the residual source (same NHWC shape as the hi planes; NULL = that tensor has none).
extern "C" int ivid_conv2d_c(int dtype, const void* src0, int C0, const void* src1, int C1, const void* weight,
                             const float* bias, void* out, void* out_lo, const void* res, const void* res_lo, int res_mode,
                             int out_mode, int N, int H, int W, int Cout, int taps, int tile_cfg, float* stats, void* stream) {
  if (taps != 1 && taps != 9) return ivid_set_error("conv: taps must be 1 or 9", hipSuccess);
  return conv2d_any(dtype, src0, C0, src1, C1, weight, bias, out, res, res_mode, out_mode, N, H, W, Cout, taps, tile_cfg, stats,
                    stream, out_lo, res_lo);
}

// ivid_conv2d for IVID_BF16X3 (fp32 storage, split-bf16 MFMA) whose NHWC result ALSO leaves as two fp16 planes hi + lo: the stem
// of the fp16s mode's split-precision island (adm.py:369) feeds the island in fp32 and the decoder's last level, through the
// skip stash, in the compensated 16-bit form -- without a conversion pass.
extern "C" int ivid_conv2d_o16(const void* src0, int C0, const void* src1, int C1, const void* weight, const float* bias, void* out,
                               void* out16_hi, void* out16_lo, const void* res, int res_mode, int N, int H, int W, int Cout, int taps,
                               int tile_cfg, float* stats, void* stream) {
  if (taps != 1 && taps != 9) return ivid_set_error("conv: taps must be 1 or 9", hipSuccess);
  if (!out || !out16_hi || !out16_lo) return ivid_set_error("conv2d_o16: the fp32 output and both fp16 planes are required", hipSuccess);
  return conv2d_any(IVID_BF16X3, src0, C0, src1, C1, weight, bias, out, res, res_mode, 0, N, H, W, Cout, taps, tile_cfg, stats, stream,
                    nullptr, nullptr, out16_hi, out16_lo);
}

// Upsample2d (nearest x2, adm.py:70-83 inside an `up` ResBlock's h_upd, adm.py:203-206) followed by the block's Conv2d 3x3,
// WITHOUT the upsampled tensor and with 4/9 of the multiplications: of the 3 x 3 taps an output pixel (2y + py, 2x + px)
// sees, those that land on the same source pixel are added up on the host, which leaves a 2 x 2 convolution of the SOURCE
// image per output phase (py, px):  rows {y-1 | y, y} for py = 0 and {y, y | y+1} for py = 1, columns likewise.  The zero
// padding of the upsampled image coincides with zero padding of the source.  Implicit GEMM with M = N*Hs*Ws source pixels,
// four cout-tile sets (one per phase) and K = 4 * C; the epilogue scatters every phase to its output pixels.
extern "C" int ivid_conv3x3_up(int dtype, const void* src0, int C0, const void* src1, int C1, const void* weight4,
                               const float* bias, void* out, int N, int Hs, int Ws, int Cout, int tile_cfg, float* stats,
                               void* stream) {
  return conv2d_any(dtype, src0, C0, src1, C1, weight4, bias, out, nullptr, 0, 0, N, Hs, Ws, Cout, 4, tile_cfg, stats, stream);
}

// Pixels per GroupNorm-statistics block that ivid_conv2d will write for this problem (depends on the tile it picks).
extern "C" int ivid_conv2d_stats_block(int N, int H, int W, int Cout, int tile_cfg) {
  return ivid_conv_pick_tile((long long)N * H * W, Cout, tile_cfg) == 3 ? 32 : 64;
}
