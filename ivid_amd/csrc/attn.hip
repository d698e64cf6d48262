// QKVAttention (adm.py:233-253) for the legacy [head][q|k|v][64] channel interleave, fused:
//   S = (q*d^-1/4)(k*d^-1/4)^T  ->  softmax in fp32 (adm.py:251)  ->  a = P.V
// Flash-style: no [T,T] score matrix ever reaches HBM (the reference materialises it twice).
//
// One workgroup = one (image n, head h, block of 32*NW queries); each wave owns 32 queries.
// Per 64-row K/V tile:
//   S^T[kv][q] = K . Q^T   (A operand = K rows from LDS, B operand = Q rows held in registers)
//       -> in the 32x32 C/D layout every lane owns ONE query column (q = lane&31) and 16 of the
//          32 kv rows, lane^32 owns the other 16: row max / row sum are 15 in-lane ops + one
//          cross-half shuffle, and the running (m, l) and the O rescale are lane-local.
//   O^T[d][q] += V^T . P^T (A operand = V^T fragments, B operand = P straight from the S^T
//          accumulators: the MFMA k index is (lane half, element), the same kv permutation is
//          applied to the V^T fragment, so no cross-lane movement of P is needed.)
// 16-bit modes: v_mfma_f32_32x32x16_{bf16,f16}.  K and V both arrive by LDS-DMA as [kv][d] rows (no register staging, no
//           VALU); the V^T fragments of the P.V MFMA come out of gfx950's transpose read: ds_read_b64_tr_b16 hands lane i of
//           a 16-lane group COLUMN i of the 4 x 16 halfword block whose row pieces the 16 lanes address (lane i: row i/4,
//           columns 4 (i%4) .. +3; measured with scripts/micro/tr_probe.hip) -- i.e. four consecutive kv of one d, exactly
//           half an MFMA fragment.  The V image swizzles its 16-byte pieces by ((row>>1)&1)<<2, which spreads the 32 lanes
//           of a transpose read over all 64 banks.  (Round 1 transposed V with 16 ds_write_b16 per thread and tile and paid
//           24 v_mov per tile to reassemble the fragments.)
// fp32 mode: v_mfma_f32_32x32x2_f32 (exact); V stays [kv][d] (one float per lane per MFMA).
#include "common.h"
#include "internal.h"

namespace {

constexpr int HD = 64;   // head dim (num_head_channels = 64 in every config)
constexpr int KVB = 64;  // K/V rows per tile

template <typename T> struct AttnCfg;
template <> struct AttnCfg<__bf16> {
  static constexpr int RB = 128;  // bytes per K row
  static __device__ __forceinline__ int swz(int row) { return (row >> 1) & 7; }
};
template <> struct AttnCfg<_Float16> {
  static constexpr int RB = 128;
  static __device__ __forceinline__ int swz(int row) { return (row >> 1) & 7; }
};
template <> struct AttnCfg<float> {
  static constexpr int RB = 256;
  static __device__ __forceinline__ int swz(int row) { return row & 15; }
};

template <typename T, int NW, int WPE = 2>
__global__ __launch_bounds__(NW * 64, WPE) void attn_kernel(const char* __restrict__ qkv, char* __restrict__ out, int T_,
                                                       int heads) {
  typedef typename Elem<T>::vec vec_t;
  constexpr int VE = Elem<T>::VE;
  constexpr int NT = NW * 64;
  constexpr int RB = AttnCfg<T>::RB;
  constexpr int PPR = RB / 16;              // 16-byte pieces per row
  constexpr int TILE_BYTES = KVB * RB;      // one K (or V) tile
  constexpr int KPIECES = KVB * PPR;
  constexpr int KK = HD * (int)sizeof(T) / 32;  // piece pairs along d: 4 (bf16) / 8 (fp32)
  constexpr bool IS_BF16 = sizeof(T) == 2;   // either 16-bit type: 32x32x16 MFMA, V transposed in LDS
  typedef T t16x4 __attribute__((ext_vector_type(4)));
  __shared__ __attribute__((aligned(16))) char smem[2 * TILE_BYTES];
  char* sK = smem;
  char* sV = smem + TILE_BYTES;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int qblk = blockIdx.x, h = blockIdx.y, n = blockIdx.z;
  const int C = heads * HD;
  const size_t rowstride = (size_t)3 * C * sizeof(T);  // bytes between tokens in qkv
  const char* base = qkv + ((size_t)n * T_) * rowstride + (size_t)h * 3 * HD * sizeof(T);
  const int fq = lane & 31, hf = lane >> 5;
  const int q = qblk * (32 * NW) + wave * 32 + fq;

  // Q fragments: B operand, row q, piece 2*kk+hf
  vec_t qf[KK];
#pragma unroll
  for (int kk = 0; kk < KK; ++kk) qf[kk] = *(const vec_t*)(base + (size_t)q * rowstride + (2 * kk + hf) * 16);

  // transpose-read base of this lane inside the V tile (16-bit modes): row 4 hf + i/4, piece (2 g + (i%4)/2) ^ swizzle,
  // 8-byte half (i%4)%2, with i = lane&15, g = (lane>>4)&1; the row's swizzle bit (row>>1)&1 = (i>>3)&1 is lane-constant
  // because fragments add multiples of 8 rows; the d >= 32 half (dt = 1) is piece + 4, i.e. byte ^ 64
  typedef short s16x4 __attribute__((ext_vector_type(4)));
  typedef __attribute__((address_space(3))) s16x4* lds_s16x4_t;
  const int tr_i = lane & 15;
  const int vt_base0 = (4 * hf + (tr_i >> 2)) * RB + (((2 * ((lane >> 4) & 1) + ((tr_i & 3) >> 1)) ^ (((tr_i >> 3) & 1) << 2)) << 4) +
                       (tr_i & 1) * 8;
  const int vt_base1 = vt_base0 ^ 64;

  f32x16 o[2];
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
  float m_run = -1e30f, l_run = 0.f;
  const float sc = 0.125f * 1.4426950408889634f;  // 64^-1/2 * log2(e): scores in the log2 domain

  // lane offsets of this thread's K / V pieces inside a 64-row tile (swizzled source piece, lane-linear LDS image)
  unsigned k_voff[KPIECES / NT], v_voff[KPIECES / NT];
#pragma unroll
  for (int i = 0; i < KPIECES / NT; ++i) {
    const int p = i * NT + tid;
    const int row = p / PPR, pos = p - row * PPR;
    k_voff[i] = (unsigned)((size_t)row * rowstride + HD * sizeof(T) + (pos ^ AttnCfg<T>::swz(row)) * 16);
    v_voff[i] = (unsigned)((size_t)row * rowstride + 2 * HD * sizeof(T) + (IS_BF16 ? pos ^ (((row >> 1) & 1) << 2) : pos) * 16);
  }
  for (int kv0 = 0; kv0 < T_; kv0 += KVB) {
    __syncthreads();  // previous tile fully consumed
    // ---- stage K: swizzled source, lane-linear LDS image ----
    // (wave-uniform 64-bit tile base + the lane offsets computed once before the loop: no per-tile address arithmetic)
    const char* const tile_base = base + (size_t)kv0 * rowstride;
#pragma unroll
    for (int i = 0; i < KPIECES / NT; ++i) glds16_s(tile_base, k_voff[i], sK + (i * NT + wave * 64) * 16);
    // ---- stage V as [kv][d]: fp32 plain; 16-bit with the pieces of row r at position pos ^ (((r>>1)&1)<<2) ----
#pragma unroll
    for (int i = 0; i < KPIECES / NT; ++i) glds16_s(tile_base, v_voff[i], sV + (i * NT + wave * 64) * 16);
    wait_vmcnt0();
    __syncthreads();

    // ---- S^T = K.Q^T for the two 32-row halves of the tile ----
    f32x16 s[2];
    if constexpr (IS_BF16) {
      // all eight K fragments are requested before the first MFMA (one LDS latency for the tile instead of one per MFMA)
      vec_t kf[2][KK];
#pragma unroll
      for (int kvh = 0; kvh < 2; ++kvh) {
        const int row = kvh * 32 + fq;
        const int sw = AttnCfg<T>::swz(row);
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) kf[kvh][kk] = *(const vec_t*)(sK + row * RB + (((2 * kk + hf) ^ sw) << 4));
      }
#pragma unroll
      for (int kvh = 0; kvh < 2; ++kvh) {
#pragma unroll
        for (int r = 0; r < 16; ++r) s[kvh][r] = 0.f;
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) MmaT<T>::run(kf[kvh][kk], qf[kk], s[kvh]);
      }
    } else {
#pragma unroll
    for (int kvh = 0; kvh < 2; ++kvh) {
#pragma unroll
      for (int r = 0; r < 16; ++r) s[kvh][r] = 0.f;
      const int row = kvh * 32 + fq;
      const int sw = AttnCfg<T>::swz(row);
#pragma unroll
      for (int kk = 0; kk < KK; ++kk) {
        const vec_t kf = *(const vec_t*)(sK + row * RB + (((2 * kk + hf) ^ sw) << 4));
#pragma unroll
        for (int j = 0; j < 4; ++j) s[kvh] = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[j], qf[kk][j], s[kvh], 0, 0, 0);
      }
    }
    }
    // ---- online softmax (per lane: one query, 32 of the 64 kv; lane^32 has the rest) ----
    float mt = -1e30f;
#pragma unroll
    for (int kvh = 0; kvh < 2; ++kvh)
#pragma unroll
      for (int r = 0; r < 16; ++r) mt = fmaxf(mt, s[kvh][r]);
    mt = fmaxf(mt, __shfl_xor(mt, 32)) * sc;  // sc > 0: the max of the scaled scores (log2 domain)
    const float m_new = fmaxf(m_run, mt);
    const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
    // two scores per packed instruction (v_pk_fma_f32 for the exponent argument, v_pk_add_f32 for the row sum): the kernel is
    // bound by VALU issue, not by the matrix pipe
    f32x2 ps2 = {0.f, 0.f};
    const f32x2 sc2 = {sc, sc}, mn2 = {-m_new, -m_new};
#pragma unroll
    for (int kvh = 0; kvh < 2; ++kvh)
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        const f32x2 t = f32x2{s[kvh][r], s[kvh][r + 1]} * sc2 + mn2;  // scale folded into the exponent
        const f32x2 e = {__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1])};
        s[kvh][r] = e[0];
        s[kvh][r + 1] = e[1];
        ps2 = ps2 + e;
      }
    float ps = ps2[0] + ps2[1];
    ps += __shfl_xor(ps, 32);
    l_run = l_run * alpha + ps;
    m_run = m_new;
    if (__any(alpha != 1.0f)) {  // wave-uniform: after the first tiles the running maxima rarely move
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;
    }

    // ---- O^T += V^T . P^T ----
    if constexpr (IS_BF16) {
#pragma unroll
      for (int kvh = 0; kvh < 2; ++kvh)
#pragma unroll
        for (int mm = 0; mm < 2; ++mm) {
          vec_t pf;
#pragma unroll
          for (int e = 0; e < 8; ++e) pf[e] = (T)s[kvh][8 * mm + e];
          // element e of lane-half hf is kv = kvh*32 + 16*mm + 8*(e>>2) + 4*hf + (e&3): two transpose reads, each four
          // consecutive kv of this lane's d = dt*32 + fq (rows kvb + 4 hf + 0..3 and + 8)
          const int kvb = kvh * 32 + 16 * mm;
#pragma unroll
          for (int dt = 0; dt < 2; ++dt) {
            const char* vb = sV + (dt ? vt_base1 : vt_base0) + kvb * RB;
            const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t)(vb));
            const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t)(vb + 8 * RB));
            const vec_t vf = __builtin_bit_cast(vec_t, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
            MmaT<T>::run(vf, pf, o[dt]);
          }
        }
    } else {
#pragma unroll
      for (int kvh = 0; kvh < 2; ++kvh)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int kv = kvh * 32 + (r & 3) + 8 * (r >> 2) + 4 * hf;
#pragma unroll
          for (int dt = 0; dt < 2; ++dt) {
            const float vv = *(const float*)(sV + kv * 256 + (dt * 32 + fq) * 4);
            o[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(vv, s[kvh][r], o[dt], 0, 0, 0);
          }
        }
    }
  }

  // ---- normalise and store: lane owns query q, d = dt*32 + (r&3) + 8*(r>>2) + 4*hf ----
  const float inv = 1.0f / l_run;
  char* orow = out + (((size_t)n * T_ + q) * C + (size_t)h * HD) * sizeof(T);
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int d = dt * 32 + 8 * g + 4 * hf;
      if constexpr (IS_BF16) {
        t16x4 w;
#pragma unroll
        for (int e = 0; e < 4; ++e) w[e] = (T)(o[dt][4 * g + e] * inv);
        *(t16x4*)(orow + d * 2) = w;
      } else {
        f32x4 w;
#pragma unroll
        for (int e = 0; e < 4; ++e) w[e] = o[dt][4 * g + e] * inv;
        *(f32x4*)(orow + d * 4) = w;
      }
    }
}

}  // namespace

extern "C" int ivid_attention(int dtype, const void* qkv, void* out, int N, int T, int heads, void* stream) {
  if (!ivid_esz(dtype)) return ivid_set_error("attention: bad dtype", hipSuccess);
  if (T <= 0 || T % 64) return ivid_set_error("attention: T must be a multiple of 64", hipSuccess);
  hipStream_t s = (hipStream_t)stream;
  const bool four = (T % 128) == 0;
  dim3 grid(T / (four ? 128 : 64), heads, N);
#define LAUNCH(TT, NW) \
  hipLaunchKernelGGL((attn_kernel<TT, NW>), grid, dim3(NW * 64), 0, s, (const char*)qkv, (char*)out, T, heads)
  if (dtype == IVID_BF16) {
    if (four) LAUNCH(__bf16, 4); else LAUNCH(__bf16, 2);
  } else if (dtype == IVID_F16) {
    if (four) LAUNCH(_Float16, 4); else LAUNCH(_Float16, 2);
  } else {   // IVID_F32 and IVID_BF16X3 (fp32 storage): exact fp32 MFMA
    if (four) LAUNCH(float, 4); else LAUNCH(float, 2);
  }
#undef LAUNCH
  return ivid_check_launch("attention");
}
