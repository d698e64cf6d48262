/* ivid_hip.h — C ABI of libivid_hip.so, the MI355X (gfx950) kernels behind ivid's sampling hot path.
 *
 * The reference (JeffreyXiang/ivid) has no C/FFI boundary: its plug-in boundary is a set of Python
 * call signatures.  This header is the boundary a maintainer binds from Python
 * (ctypes): every entry point below replaces the torch/cuDNN/OpenGL call sites
 * cited next to it.  Conventions:
 *   - plain pointers and sizes only; all pointers are DEVICE pointers unless named host_*;
 *   - the caller (torch) owns every buffer; the library allocates nothing that crosses the ABI
 *     (it owns one 256-byte zero page and hipGraph handles);
 *   - every kernel is enqueued on the caller's hipStream_t (passed as void*); nothing synchronises;
 *   - return value 0 = ok, nonzero = error; ivid_last_error() gives the message; never aborts;
 *   - activations inside the UNet are NHWC ("pixel-major") of dtype IVID_F32 (parity mode, exact
 *     fp32 MFMA), IVID_BF16 / IVID_F16 (16-bit storage and MFMA operands, fp32 accumulate) or
 *     IVID_BF16X3 (fp32 storage, split-bf16 MFMA); the model boundary is fp32 NCHW exactly like the
 *     reference (adm.py:557,565-566).  Precision mode "fp16c" of the Python surface is IVID_F16 plus the `*_c` / `*_split`
 *     entry points below: tensors of the residual stream carry a second 16-bit plane (compensated storage, see
 *     ivid_conv2d_c), which keeps one forward within 1e-3 of the reference's fp32 path at the fp16 MFMA rate.
 */
#ifndef IVID_HIP_H
#define IVID_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define IVID_F32 0     /* fp32 storage, exact fp32 MFMA (v_mfma_f32_32x32x2_f32): parity mode, ~1e-6 vs the reference */
#define IVID_BF16 1    /* bf16 storage + bf16 MFMA, fp32 accumulate: perf mode (~8e-3 on the large model) */
#define IVID_F16 2     /* fp16 storage + fp16 MFMA, fp32 accumulate: the reference's `use_fp16` torso (backbones/utils.py:6-13) */
#define IVID_BF16X3 3  /* fp32 storage; MFMA operands split into bf16 hi + lo, 3 bf16 MFMAs per product (a_hi*b_hi +
                        * a_hi*b_lo + a_lo*b_hi), fp32 accumulate: <= 1e-3 parity at MFMA speed.  Conv weights are
                        * pre-split by the host: per 8 input channels 16 bytes of hi followed by 16 bytes of lo, i.e. a
                        * weight row has the byte size of an fp32 row (see ivid_conv2d) */

/* ---- runtime ---- */
const char* ivid_last_error(void);
int ivid_version(void);
/* hipGraph capture of a launch sequence on `stream` (replaces ~300 eager launches per UNet forward). */
int ivid_graph_begin(void* stream);
int ivid_graph_end(void* stream, void** graph_exec_out);
int ivid_graph_launch(void* graph_exec, void* stream);
int ivid_graph_destroy(void* graph_exec);
/* Timing of a stream region with HIP events (used by bench.py; torch.cuda.Event only sees torch's stream). */
int ivid_event_create(void** ev_out);
int ivid_event_record(void* ev, void* stream);
int ivid_event_elapsed_ms(void* ev_start, void* ev_stop, float* ms_out); /* synchronises on ev_stop */
int ivid_event_destroy(void* ev);

/* ---- launch program: the C-side owner of one planned UNet forward (the `ivid_unet_create / ivid_unet_forward` handle of the
 *      boundary; replaces AdmUnet2d.forward's ~650-700 torch launches, adm.py:526-566) ----
 * The host plans the forward once (which entry points below, in which order, on which caller-owned buffers) and appends the
 * launches; afterwards a forward is ONE call.  `args` of ivid_program_add: one 8-byte slot per argument of the entry point,
 * WITHOUT the trailing stream, in declaration order: integers and device pointers as int64, floats as double. */
#define IVID_OP_CONV2D 1          /* ivid_conv2d */
#define IVID_OP_CONV3X3_GN 2      /* ivid_conv3x3_gn */
#define IVID_OP_CONV3X3_GN_SKIP 3 /* ivid_conv3x3_gn_skip */
#define IVID_OP_CONV3X3_GN_OUT 4  /* ivid_conv3x3_gn_out */
#define IVID_OP_GN_PARTIAL 5      /* ivid_gn_partial */
#define IVID_OP_GN_FINALIZE 6     /* ivid_gn_finalize */
#define IVID_OP_GN_FINALIZE2 7    /* ivid_gn_finalize2 */
#define IVID_OP_GN_APPLY 8        /* ivid_gn_apply */
#define IVID_OP_ATTENTION 9       /* ivid_attention */
#define IVID_OP_EMBED_INPUTS 10   /* ivid_embed_inputs */
#define IVID_OP_SILU_F32 11       /* ivid_silu_f32 */
#define IVID_OP_STEM_IM2COL 12    /* ivid_stem_im2col */
#define IVID_OP_CONV3X3_UP 13     /* ivid_conv3x3_up */
#define IVID_OP_COPY 14           /* ivid_copy */
#define IVID_OP_CONV2D_C 15       /* ivid_conv2d_c */
#define IVID_OP_CONV3X3_GN_SKIP_C 16 /* ivid_conv3x3_gn_skip_c */
#define IVID_OP_GN_APPLY_C 17     /* ivid_gn_apply_c */
#define IVID_OP_CONV3X3_GN_OUT_C 18 /* ivid_conv3x3_gn_out_c */
#define IVID_OP_STEM_IM2COL_SPLIT 19 /* ivid_stem_im2col_split */
#define IVID_OP_CONV3X3_GN_SKIP_S 20 /* ivid_conv3x3_gn_skip_s */
#define IVID_OP_F32_TO_HILO 21    /* ivid_f32_to_hilo */
#define IVID_OP_GN_APPLY_P 22     /* ivid_gn_apply_p */
#define IVID_OP_CONV3X3_GN_O16 23 /* ivid_conv3x3_gn_o16 */
#define IVID_OP_GN_PARTIAL_C 24   /* ivid_gn_partial_c */
#define IVID_OP_CONV2D_O16 25     /* ivid_conv2d_o16 */
#define IVID_OP_LAST 25
/* Version of the op-code table + the argument lists behind it: bumped whenever an entry point that can appear in a launch
 * program changes its signature or a code is added.  Engine files carry it; ivid_unet_load refuses any other value. */
#define IVID_ENGINE_ABI 5
int ivid_program_create(void** handle_out);
/* refuses an unknown op code and an `nargs` that is not the argument count of the op's entry point */
int ivid_program_add(void* handle, int op, const void* args, int nargs);
/* number of arguments (without the stream) of the entry point behind an op code; -1 for an unknown code */
int ivid_program_op_arity(int op);
int ivid_program_num_ops(void* handle);
/* Replay on `stream`: use_graph = 0 always eager; else run 1 eager (sets kernel attributes), run 2 captures a hipGraph,
 * later runs are one hipGraphLaunch (the stream must not be the legacy default stream). */
int ivid_program_launch(void* handle, int use_graph, void* stream);
int ivid_program_has_graph(void* handle);
int ivid_program_destroy(void* handle);
/* Model boundary of a UNet program: its static input buffers (x fp32 NCHW of x_bytes, times / classes int64 [batch];
 * c_in NULL for a model without class embedding) and its output buffer (fp32 NCHW of out_bytes). */
int ivid_unet_bind(void* handle, void* x_in, long long x_bytes, void* t_in, void* c_in, int batch, void* out,
                   long long out_bytes);
/* AdmUnet2d.forward(x, times, classes) (adm.py:526-566) as one call: device pointers; classes NULL = the null class for every
 * row; out NULL = leave the result in the program's own output buffer.  Everything is enqueued on `stream`. */
int ivid_unet_forward(void* handle, const void* x, const void* times, const void* classes, void* out, int use_graph,
                      void* stream);
/* A planned forward as a FILE: `AdmUnet2d.plan(...).export_engine()` (ivid_amd/diffusion/backbones/engine.py, layout documented
 * there) freezes one plan -- model weights repacked for one precision mode, one (batch, stacked-CFG) shape, the launch list with
 * relocatable pointer arguments.  ivid_unet_load builds the program from that file held in HOST memory: one device allocation for
 * every buffer, constants uploaded, pointers relocated, boundary bound.  What a serving host without Python does in place of the
 * reference's `AdmUnet2d(**args)` + `load_state_dict` (adm.py:385-524, inference/sample.py:35-56): load, then ivid_unet_forward
 * per call, ivid_program_destroy at the end (frees the allocation).  Malformed files are rejected with an error, never trusted. */
int ivid_unet_load(const void* blob, long long nbytes, void** handle_out);
/* Boundary of a bound / loaded program: rows of x, whether `classes` is read, bytes of x and of the output, dims[4] = output rows
 * (batch, or 2 x batch for a stacked classifier-free-guidance plan: conditional rows first), in channels, out channels, image size
 * (zeros for a program bound by hand).  Any pointer may be NULL. */
int ivid_unet_info(void* handle, int* batch, int* has_classes, long long* x_bytes, long long* out_bytes, int* dims);

/* ---- convolution / linear: nn.Conv2d 3x3 pad1 (adm.py:160,182,369,486), nn.Conv2d 1x1 skip (adm.py:190),
 *      nn.Conv1d k=1 qkv/proj_out (adm.py:275,278), nn.Linear (adm.py:176,359,361) ----
 * out[n,y,x,co] = bias[co] + sum_{tap,c} cat(src0,src1)[n,y+dy,x+dx,c] * weight[co][tap][c]  (+ residual)
 *   src0/src1 : NHWC [N,H,W,C0] / [N,H,W,C1] (C1 = 0, src1 = NULL when there is no skip concat,
 *               adm.py:563 torch.cat is never materialised); C0, C1 multiples of 64 (bf16) / 32 (f32)
 *   weight    : [Cout][taps][C0+C1], dtype as activations (repacked from [Cout,Cin,3,3] by the host); IVID_BF16X3: the same
 *               row order with every 8 consecutive K values stored as 8 x bf16 hi then 8 x bf16 lo (hi = bf16(w) RNE,
 *               lo = bf16(w - hi)) — 32 bytes per 8 values, i.e. the byte size of the fp32 row; C0, C1 multiples of 32
 *   bias      : fp32 [Cout] or NULL
 *   res_mode  : 0 none; 1 add res[N,H,W,Cout]; 2 add nearest-x2-upsampled res[N,H/2,W/2,Cout];
 *               3 add 2x2-avg-pooled res[N,2H,2W,Cout]  (ResBlock2d skip through x_upd, adm.py:203-208,222)
 *   out_mode  : 0 NHWC dtype [N,H,W,Cout]; 1 fp32 NCHW [N,Cout,H,W] (final conv, adm.py:566)
 *   tile_cfg  : 0 auto; 1 = 128x128 tile / 4 waves; 2 = 256x256 / 8 waves (wide layers); 3 = 128x32 (Cout <= 32: the
 *               4-channel output conv); 4 = 512x128 / 8 waves (Cout <= 128: the small / SR models' first levels at the
 *               256x256 tile's LDS-read : MFMA ratio); 5 = 64x128 / 2 waves (tiny problems that leave CUs idle with tile 1); 6 = 128x384 / 8 waves (Cout % 384 == 0
 *               when the 256x256 tile would leave a fractional last round of workgroups).  The fp32 summation
 *               order of `stats` is the same for every tile
 *   stats     : NULL, or fp32 [N*H*W/blk][Cout][2]: per block of blk consecutive pixels and output channel, sum and sum
 *               of squares of the stored output — the GroupNorm partial statistics of the NEXT layer, fused into this
 *               epilogue (same layout as ivid_gn_partial with H*W/blk chunks per image); blk =
 *               ivid_conv2d_stats_block(...) = 64 (32 for the narrow tile) independent of the tile, so the summation
 *               order — and with it every sample's result — does not depend on the batch size; H*W % blk must be 0 */
int ivid_conv2d(int dtype, const void* src0, int C0, const void* src1, int C1, const void* weight, const float* bias,
                void* out, const void* res, int res_mode, int out_mode, int N, int H, int W, int Cout, int taps,
                int tile_cfg, float* stats, void* stream);

int ivid_conv2d_stats_block(int N, int H, int W, int Cout, int tile_cfg);

/* ---- compensated 16-bit storage (precision mode "fp16c": IVID_F16 kernels + lo planes) ----
 * The reference's arithmetic on its benchmark config is fp32 (configs/rgbd_imagenet_adm_128_large_cfg.json:18 `use_fp16: false`,
 * adm.py:557 `h = x.type(self.dtype)`); a plain fp16 torso ends 1.07e-3 from it, a third of which is the ROUNDING OF THE
 * RESIDUAL TRUNK at every block (`return self.skip_connection(x) + h`, adm.py:222; `(x + h)`, adm.py:286).  In this mode a trunk
 * tensor is stored as TWO 16-bit NHWC planes: hi = T(v) and lo = T(v - float(hi)) of the fp32 value v (22 mantissa bits
 * together).  MFMA operands read the hi plane alone -- it is exactly the rounded operand an fp16 MFMA would be fed anyway --
 * while residual adds (below), GroupNorm-apply (ivid_gn_apply_c) and the output head (ivid_conv3x3_gn_out_c) read hi + lo.
 * GroupNorm partial statistics of a tensor with a lo plane describe hi + lo.
 *   out_lo : NULL or the lo plane of `out` (same shape);  res_lo : NULL or the lo plane of `res` (any res_mode).
 * 16-bit dtypes, out_mode 0 only; everything else as ivid_conv2d. */
int ivid_conv2d_c(int dtype, const void* src0, int C0, const void* src1, int C1, const void* weight, const float* bias,
                  void* out, void* out_lo, const void* res, const void* res_lo, int res_mode, int out_mode, int N, int H, int W,
                  int Cout, int taps, int tile_cfg, float* stats, void* stream);

/* ---- Upsample2d (nearest x2) + Conv2d 3x3 of an `up` ResBlock's in_layers (adm.py:70-83 `F.interpolate(..., mode="nearest")`,
 *      adm.py:203-206 `h = in_rest(x); h = self.h_upd(h); h = in_conv(h)`) without the upsampled tensor ----
 * out[n, 2y+py, 2x+px, co] = bias[co] + sum_{a,b in {0,1}} sum_c cat(src0,src1)[n, y+py-1+a, x+px-1+b, c] * weight4[py*2+px][co][a*2+b][c]
 * which equals conv3x3(pad0(nearest_x2(src))) when the host adds up the 3x3 taps that land on the same source pixel:
 *   weight4[py*2+px][co][a*2+b][c] = sum_{ky in R(py,a)} sum_{kx in R(px,b)} w[co][c][ky][kx],
 *   R(0,0) = {0}, R(0,1) = {1,2}, R(1,0) = {0,1}, R(1,1) = {2}
 * (4/9 of the multiplications of the direct form; the sums are formed in fp32 before the conversion to `dtype`).
 *   src0/src1 : the ACTIVATED source (ivid_gn_apply output), NHWC [N,Hs,Ws,C0/C1]; out NHWC [N,2Hs,2Ws,Cout]
 *   weight4   : [4][Cout][4][C0+C1] in `dtype` (IVID_BF16X3: hi/lo split rows as for ivid_conv2d); Cout > 32
 *   stats     : as ivid_conv2d, 64-pixel blocks of the OUTPUT image, 4*Hs*Ws/64 per image (Hs*Ws % 64 == 0);
 *               block order inside an image: [phase][source block] */
int ivid_conv3x3_up(int dtype, const void* src0, int C0, const void* src1, int C1, const void* weight4, const float* bias,
                    void* out, int N, int Hs, int Ws, int Cout, int tile_cfg, float* stats, void* stream);

/* ---- fused GroupNorm-apply (+FiLM) + SiLU [+ nearest x2 upsample] + Conv2d 3x3 (ResBlock2d in_layers / out_layers,
 *      adm.py:157-161,177-183,203-208,214-219) for W % 32 == 0, H % 8 == 0 ----
 * out = bias + conv3x3( pad0( up?( silu( cat(src0,src1) * a + b ) ) ) ) (+ residual), with a,b = ab[n][c][0..1] from
 * ivid_gn_finalize{,2}.  The activated tensor never exists in HBM: the (8+2)x(32+2) pixel halo of each tile is
 * transformed once while it is staged into LDS.  up = 1: src is [N,H/2,W/2,C] (Upsample2d inside an `up` ResBlock).
 * res_mode: 0 none, 1 same size, 2 nearest-x2-upsampled residual [N,H/2,W/2,Cout], 3 2x2-average-pooled residual
 *           [N,2H,2W,Cout] (the identity skip of a `down` ResBlock seen through Downsample2d, adm.py:115-117,203-208).
 * stats: as ivid_conv2d with 128-pixel blocks (4 image rows x 32 columns; H*W/128 blocks per image). */
int ivid_conv3x3_gn(int dtype, const void* src0, int C0, const void* src1, int C1, const float* ab, int up,
                    const void* weight, const float* bias, void* out, const void* res, int res_mode, int N, int H, int W,
                    int Cout, float* stats, void* stream);
/* Same, plus the ResBlock's 1x1 `skip_connection` (adm.py:190, `self.skip_connection(x) + h`, adm.py:222) accumulated
 * into the same output tile: out = conv3x3(silu(gn(src))) + conv1x1(cat(skip0, skip1)) + bias (+ res).
 *   skip0/skip1  NHWC [N,H,W,skipC0 / skipC1] (the block's raw input, e.g. the decoder's cat([h, hs.pop()])); skip_weight
 *   [Cout][skipC0+skipC1] in the compute dtype; `bias` must already hold conv bias + skip bias.  skipC0 = 0 disables it. */
int ivid_conv3x3_gn_skip(int dtype, const void* src0, int C0, const void* src1, int C1, const float* ab, int up,
                         const void* weight, const float* bias, void* out, const void* res, int res_mode, int N, int H,
                         int W, int Cout, float* stats, const void* skip0, int skipC0, const void* skip1, int skipC1,
                         const void* skip_weight, void* stream);

/* ivid_conv3x3_gn_skip with lo planes (see ivid_conv2d_c): of the output and the residual source, and of the two convolution
 * INPUTS -- the halo transform then starts from hi + lo (the block input `x` of `in_layers`, adm.py:203, and `h` of `out_layers`,
 * adm.py:214-219, enter GroupNorm unrounded); the 1x1 skip phase reads the hi planes (they are the MFMA operand).  NULL = none. */
int ivid_conv3x3_gn_skip_c(int dtype, const void* src0, const void* src0_lo, int C0, const void* src1, const void* src1_lo, int C1,
                           const float* ab, int up, const void* weight, const float* bias, void* out, void* out_lo, const void* res,
                           const void* res_lo, int res_mode, int N, int H, int W, int Cout, float* stats, const void* skip0,
                           int skipC0, const void* skip1, int skipC1, const void* skip_weight, void* stream);

/* ivid_conv3x3_gn_skip_c with the 1x1 skip phase in SPLIT precision (precision mode fp16s).  `self.skip_connection(x) + h`
 * (adm.py:190,222) sends the residual trunk itself through an MFMA: every operand rounding of that 1x1 convolution reaches
 * all later layers undamped (measured: 40 % of the fp16cx deviation on clean inputs).  With skip_weight_lo != NULL the phase
 * accumulates x_hi.w_hi + x_lo.w_hi + x_hi.w_lo (x_lo: the lo planes skip0_lo / skip1_lo of the block input, one per source;
 * w_lo = fp16(w - fp16(w)) in the layout of skip_weight): three MFMA passes over the 1x1 K range, four staged slabs per
 * chunk.  IVID_F16 and Cout > 128 only.  skip_weight_lo == NULL: identical to ivid_conv3x3_gn_skip_c. */
int ivid_conv3x3_gn_skip_s(int dtype, const void* src0, const void* src0_lo, int C0, const void* src1, const void* src1_lo, int C1,
                           const float* ab, int up, const void* weight, const float* bias, void* out, void* out_lo, const void* res,
                           const void* res_lo, int res_mode, int N, int H, int W, int Cout, float* stats, const void* skip0,
                           int skipC0, const void* skip1, int skipC1, const void* skip_weight, const void* skip0_lo,
                           const void* skip1_lo, const void* skip_weight_lo, void* stream);

/* ivid_conv3x3_gn for IVID_BF16X3 (fp32 storage, split-bf16 MFMA; no upsample, no skip phase) whose result leaves as two
 * fp16 planes out16_hi = fp16(v), out16_lo = fp16(v - hi) -- the compensated storage form of the 16-bit modes -- in addition
 * to (out != NULL) or instead of (out == NULL) the fp32 tensor: the last layers of the fp16s mode's split-precision island
 * (stem + first encoder level, adm.py:373-402) hand their tensors to the 16-bit part of the network without a conversion
 * pass.  GroupNorm partials as ivid_conv3x3_gn. */
int ivid_conv3x3_gn_o16(const void* src0, int C0, const void* src1, int C1, const float* ab, const void* weight, const float* bias,
                        void* out, void* out16_hi, void* out16_lo, const void* res, int res_mode, int N, int H, int W, int Cout,
                        float* stats, void* stream);

/* ivid_conv2d for IVID_BF16X3 whose NHWC result also leaves as two fp16 planes (see ivid_conv3x3_gn_o16): the stem of the island
 * (`input_blocks[0]`, adm.py:369) feeds the island in fp32 and the decoder's last level in the compensated 16-bit form. */
int ivid_conv2d_o16(const void* src0, int C0, const void* src1, int C1, const void* weight, const float* bias, void* out,
                    void* out16_hi, void* out16_lo, const void* res, int res_mode, int N, int H, int W, int Cout, int taps,
                    int tile_cfg, float* stats, void* stream);

/* The UNet's output head in one kernel (adm.py:483-487 `self.out`: GroupNorm32 -> SiLU -> zero_module(Conv2d 3x3 to
 * out_channels), adm.py:565-566): out = conv3x3(silu(src*a + b)) + bias, written as fp32 NCHW [N,Cout,H,W].
 *   src NHWC [N,H,W,C] in `dtype`; ab fp32 [N][C][2]; weight [Cout][9][C] in `dtype`; 1 <= Cout <= 16; W % 32 == 0, H % 8 == 0.
 *   IVID_BF16X3: runs the exact-fp32 kernel (src is fp32 anyway, the layer is 0.05 % of the FLOPs): weight is PLAIN fp32.
 * Reads the 1 GiB input once instead of three times (gn_apply round trip + nine shifted re-reads). */
int ivid_conv3x3_gn_out(int dtype, const void* src, int C, const float* ab, const void* weight, const float* bias, float* out,
                        int N, int H, int W, int Cout, void* stream);

/* The head in split form (16-bit dtypes): weight_lo = T(w - float(T(w))) in the layout of `weight`; the activated values are
 * split into hi + lo parts while the halo is staged and every product is three MFMAs (a_hi*w_hi + a_hi*w_lo + a_lo*w_hi), i.e.
 * the layer is evaluated to ~2^-21: its operand roundings reach the model output unaveraged (7 % of the fp16 error budget,
 * tests/tools/error_budget.py) and it has 0.05 % of the FLOPs.  src_lo: NULL or the lo plane of src (see ivid_conv2d_c).
 * weight_lo NULL = ivid_conv3x3_gn_out. */
int ivid_conv3x3_gn_out_c(int dtype, const void* src, const void* src_lo, int C, const float* ab, const void* weight,
                          const void* weight_lo, const float* bias, float* out, int N, int H, int W, int Cout, void* stream);

/* ---- GroupNorm32 + SiLU + FiLM (adm.py:36-41,159,175-180,214-218) ----
 * Step 1: per-(n, pixel-chunk, channel) partial sums of x and x^2 over cat(src0,src1) (NHWC).
 *   partial: fp32 [N][nchunks][C0+C1][2]; nchunks = ivid_gn_num_chunks(H*W). */
int ivid_gn_num_chunks(int HW);
int ivid_gn_partial(int dtype, const void* src0, int C0, const void* src1, int C1, int N, int HW, float* partial,
                    void* stream);
/* Same with the lo planes of the sources (compensated 16-bit storage, NULL = none): the statistics describe hi + lo, exactly like
 * the partials the convolution epilogues write for such a tensor. */
int ivid_gn_partial_c(int dtype, const void* src0, const void* src0_lo, int C0, const void* src1, const void* src1_lo, int C1, int N,
                      int HW, float* partial, void* stream);
/* Step 2: fold statistics, affine and FiLM into one per-(n,c) scale/offset:
 *   y = x*a + b,  a = rstd*gamma*(1+scale), b = (beta - mean*rstd*gamma)*(1+scale) + shift
 *   film: fp32 rows [N][film_stride]; scale = film[n][film_off + c], shift = film[n][film_off + C + c]
 *   (torch.chunk(emb_out,2), adm.py:216) or NULL.  ab: fp32 [N][C][2]. */
int ivid_gn_finalize(const float* partial, int nchunks, int N, int C, int HW, int groups, float eps,
                     const float* gamma, const float* beta, const float* film, int film_stride, int film_off,
                     float* ab, void* stream);
/* Same, with the statistics of a skip concat kept as two separate partial buffers (one per source tensor, as the
 * producing convolutions' epilogues wrote them): partial0 [N][nchunks0][C0][2], partial1 [N][nchunks1][C1][2]. */
int ivid_gn_finalize2(const float* partial0, int C0, int nchunks0, const float* partial1, int C1, int nchunks1, int N, int HW,
                      int groups, float eps, const float* gamma, const float* beta, const float* film,
                      int film_stride, int film_off, float* ab, void* stream);
/* Step 3: out = act(x*a+b) with optional resampling folded in (adm.py:203-208):
 *   resample 0: same size; 1: nearest x2 upsample (out is 2H x 2W); 2: 2x2 average pool of the ACTIVATED
 *   values (out is H/2 x W/2).  act: 0 identity (AttentionBlock.norm, adm.py:283), 1 SiLU.
 *   H, W are the SOURCE spatial dims.  out: NHWC dtype [N,Ho,Wo,C0+C1] (concat materialised here). */
int ivid_gn_apply(int dtype, const void* src0, int C0, const void* src1, int C1, const float* ab, void* out, int N,
                  int H, int W, int resample, int act, void* stream);

/* Same with the lo planes of the two sources (see ivid_conv2d_c; NULL = none): x = hi + lo.  `out` is a plain tensor. */
int ivid_gn_apply_c(int dtype, const void* src0, const void* src0_lo, int C0, const void* src1, const void* src1_lo, int C1,
                    const float* ab, void* out, int N, int H, int W, int resample, int act, void* stream);

/* ivid_gn_apply_c with resample 2 that also writes the 2x2 average of the RAW sources (hi + lo read, summed in fp32):
 * pool_hi [N,H/2,W/2,C0+C1] and, if pool_lo != NULL, the part its 16-bit rounding dropped.  That is `x_upd(x)` of a `down`
 * ResBlock (adm.py:205-208: `x = self.x_upd(x)` feeds `self.skip_connection(x) + h`): the block's second convolution then adds
 * a same-size residual (res_mode 1) instead of averaging four full-size pixels per output in its epilogue (res_mode 3).
 * pool_hi == NULL: identical to ivid_gn_apply_c.  16-bit dtypes. */
int ivid_gn_apply_p(int dtype, const void* src0, const void* src0_lo, int C0, const void* src1, const void* src1_lo, int C1,
                    const float* ab, void* out, void* pool_hi, void* pool_lo, int N, int H, int W, int resample, int act,
                    void* stream);

/* ---- QKVAttention (adm.py:233-253), legacy per-head [q|k|v] channel interleave ----
 * qkv: NHWC [N,T,3*C] with channel = head*192 + {0..63 q, 64..127 k, 128..191 v}; out: [N,T,C], channel = head*64+d.
 * softmax in fp32 (adm.py:251); scores scaled by 64^-1/2 overall (q*64^-1/4 . k*64^-1/4, adm.py:247-250). */
int ivid_attention(int dtype, const void* qkv, void* out, int N, int T, int heads, void* stream);

/* ---- embeddings: PosEncoding + label_emb with null-class mask (adm.py:30-33,545-555) ----
 * pos[n][0..half) = cos(t[n]*freqs[k]), pos[n][half..2half) = sin(...)   (cos first, adm.py:32)
 * cls[n][:] = classes? label_emb[max(c,0)] * (c>=0) : 0
 * times/classes are int64 device arrays of length Bsrc, broadcast to N rows by n % Bsrc;
 * null_from: rows n >= null_from use the null class (stacked classifier-free-guidance batch). */
int ivid_embed_inputs(const int64_t* times, const int64_t* classes, int Bsrc, int N, int null_from,
                      const float* freqs, int half, const float* label_emb, int emb_dim, float* pos, float* cls,
                      void* stream);
/* y = silu(x) elementwise on fp32 (emb_layers[0] / time_embed[2], adm.py:175,360). */
int ivid_silu_f32(const float* x, float* y, long long n, void* stream);
/* Device copy as a kernel (bytes % 16 == 0, 16-byte aligned pointers).  The stacked classifier-free-guidance forward
 * (classifier_free_guidance.py:39-42 = two backbone calls on the same x, t) shares what both halves of the batch have in
 * common: until the first FiLM (adm.py:214-218) nothing depends on the class, so the first ResBlock's in_layers convolution
 * is computed for one half and duplicated with this call. */
int ivid_copy(void* dst, const void* src, long long bytes, void* stream);

/* fp32 tensor of n elements (n % 8 == 0) -> two planes in the 16-bit `dtype`: hi = T(x), lo = T(x - hi): hands a tensor of
 * the split-precision island of the fp16s mode (fp32 storage: the stem and the first encoder level, adm.py:373-402) to the
 * 16-bit part of the network in its compensated storage form. */
int ivid_f32_to_hilo(int dtype, const float* src, void* hi, void* lo, long long n, void* stream);

/* ---- model boundary layout changes ----
 * fp32 NCHW [Bsrc,Cin,H,W] -> NHWC dtype [N,H,W,Cpad] (zero padded channels, batch replicated n % Bsrc). */
int ivid_nchw_to_nhwc(int dtype, const float* x, int Bsrc, int N, int Cin, int H, int W, int Cpad, void* out,
                      void* stream);
/* Stem convolution input (adm.py:369 `input_blocks[0]`, a 3x3 conv on the 4..10 model input channels): instead of a
 * channel-padded NHWC copy, lay out every pixel's 3x3 patch as one K row  k = tap*Cin + c  (zero-padded taps, zeros for
 * k >= 9*Cin), so the stem runs as ivid_conv2d(taps = 1, C0 = Kpad) with weights [Cout][Kpad] in the same k order.
 *   x fp32 NCHW [Bsrc,Cin,H,W]; row n reads source n % Bsrc (the stacked CFG batch);  out [N,H,W,Kpad] in `dtype`. */
int ivid_stem_im2col(int dtype, const float* x, int Bsrc, int N, int Cin, int H, int W, int Kpad, void* out,
                     void* stream);

/* The stem in split form (16-bit dtypes, precision mode fp16c): the K row holds three segments of 9*Cin values
 * [x_hi | x_lo | x_hi] (x_hi = T(x), x_lo = T(x - x_hi)), zeros behind; with weight rows [w_hi | w_hi | w_lo] in the same k order
 * ivid_conv2d(taps = 1, C0 = Kpad) accumulates x_hi*w_hi + x_lo*w_hi + x_hi*w_lo: the stem to ~2^-21 (its two operand
 * roundings were 9 % of the fp16 error budget) for two more K-steps of the model's cheapest layer.  Kpad >= 27*Cin. */
int ivid_stem_im2col_split(int dtype, const float* x, int Bsrc, int N, int Cin, int H, int W, int Kpad, void* out,
                           void* stream);

/* ---- samplers (fp32 NCHW [B,4,H,W]) ----
 * eps = (1+s)*eps_c - s*eps_u (classifier_free_guidance.py:39-42); eps_u may be NULL (s ignored). */
typedef struct {
  float sqrt_recip_ac;   /* sqrt(1/alpha_bar[t-1])           ddim.py:36 */
  float sqrt_recipm1_ac; /* sqrt(1/alpha_bar[t-1]-1)         ddim.py:37 */
  float sqrt_ac_prev;    /* sqrt(alpha_bar_prev[t_prev])     ddim.py:99 */
  float dir_coef;        /* sqrt(1-alpha_bar_prev-sigma^2)   ddim.py:99 */
  float sigma;           /* eta*...                          ddim.py:98 */
  float nonzero;         /* t_prev != 0                      ddim.py:83 */
  float cfg_strength;
  float replace_rgb_w;   /* <0: disabled                     ddim.py:86-89 */
  float replace_depth_w; /* <0: disabled                     ddim.py:90-92 */
  float constrain_w;     /* <0: disabled                     ddim.py:93-95 */
  int clip_denoised;
} ivid_ddim_coef;
/* One DDIM update (ddim.py:81-102).  rgb [B,3,H,W], rgb_mask [B,1,H,W], depth [B,1,H,W], depth_mask [B,1,H,W],
 * convex [B,1,H,W] may be NULL when the matching weight is < 0; noise may be NULL when sigma == 0. */
int ivid_ddim_step(const float* x_t, const float* eps_c, const float* eps_u, const ivid_ddim_coef* host_coef,
                   const float* rgb, const float* rgb_mask, const float* depth, const float* depth_mask,
                   const float* convex, const float* noise, float* x_prev, float* x0, int B, int HW, void* stream);
typedef struct {
  float sqrt_recip_ac, sqrt_recipm1_ac; /* ddpm.py:33-34 */
  float coef1, coef2;                   /* posterior mean coefficients, ddpm.py:40-41 */
  float std;                            /* exp(0.5*posterior_log_variance_clipped[t]) * (t != 0), ddpm.py:129-130 */
  float cfg_strength;
  int clip_denoised;
} ivid_ddpm_coef;
int ivid_ddpm_step(const float* x_t, const float* eps_c, const float* eps_u, const ivid_ddpm_coef* host_coef,
                   const float* noise, float* x_prev, float* x0, int B, int HW, void* stream);
/* InpaintCFG.make_cond_inputs (inpaint_cfg.py:24-49): out [B,10 or 9,H,W] =
 * cat[x(4), mask_rgb(1, if given), y_rgb*m_rgb + n_rgb*(1-m_rgb) (3), y_d*m + n_d*(1-m) (1), mask(1)]. */
int ivid_inpaint_cond(const float* x, const float* y, const float* mask, const float* mask_rgb, const float* noise_rgb,
                      const float* noise_depth, float* out, int B, int HW, void* stream);
/* SuperResCFG.make_cond_inputs (sr_cfg.py:23-36): out[B,Cx+Cy,S,S] = cat[x[B,Cx,S,S], bilinear x(S/s) upsample of
 * y[B,Cy,s,s] with align_corners=False] (fp32 NCHW), torch's upsample_bilinear2d arithmetic. */
int ivid_sr_cond(const float* x, const float* y, float* out, int B, int Cx, int Cy, int S, int s, void* stream);

/* ---- counter-based Gaussian noise on the device (what `torch.randn` / `randn_like` are to the reference's samplers, ddim.py:101,
 * ddpm.py:128, inpaint_cfg.py:36-45): out[i] ~ N(0,1), a pure function of (seed, stream_id, i) -- Philox4x32-10 blocks of four
 * values, 24-bit uniforms, Box-Muller -- so that a chain's noise needs no buffers and does not depend on launch shapes, ranks
 * or call order.  out: device fp32, 16-byte aligned. */
int ivid_randn(unsigned long long seed, unsigned long long stream_id, float* out, long long n, void* stream);

/* ---- the whole sampling loop as ONE call (SURVEY.md 8(b) `ivid_sample(handle, plan*, ...)`) ----
 * DdimSampler.sample / DdpmSampler.sample (diffusion/samplers/ddim.py:150-163, ddpm.py:172-185) with the frameworks'
 * model_inference inside (classifier_free_guidance.py:23-42; inpaint_cfg.py:24-49,51-83): per step [make_cond_inputs ->] one
 * UNet program (hipGraph replay) -> the fused step kernel.  Everything is enqueued on `stream`; nothing synchronises and no
 * host code runs between the steps, so a C host samples without Python and N ranks do not compete for host cores.
 *   engines        : UNet programs with a bound boundary (ivid_unet_load, or a plan's program), all with the same boundary:
 *                    B rows of x.  Output B rows: eps as is; 2B rows (a stacked classifier-free-guidance plan): rows [0,B) the
 *                    conditional and [B,2B) the null-class branch, combined inside the step kernel with the coefficients'
 *                    cfg_strength.  engine_of_step picks the program of step i (the precision tier of its timestep).
 *   plan           : host tables, one entry per step in sampling order: the timestep the model is fed (ddim.py:74 / ddpm.py:118
 *                    `t - 1`) and the step's coefficients (what DdimSampler._coef / DdpmSampler._coef compute, fp32 like the
 *                    reference); hw = pixels of an image
 *   classes        : device int64 [B] or NULL
 *   cond           : NULL, or the tensors that stay constant over the loop (device fp32):
 *                    y [B,4,HW], mask [B,1,HW], mask_rgb [B,1,HW] or NULL, hole_noise [n_steps][B,4,HW] (per step the rgb
 *                    noise [B,3,HW] followed by the depth noise [B,1,HW], inpaint_cfg.py:36-45) -- y == NULL: the model is fed x_t;
 *                    rgb [B,3,HW], rgb_mask, depth, depth_mask, convex [B,1,HW]: replace_rgb / replace_depth / constrain_depth of
 *                    DdimSampler.sample_once (ddim.py:86-95), needed when a step's weight is >= 0;
 *                    sr_y [B,sr_channels,sr_size,sr_size]: SuperResCFG.make_cond_inputs per step (sr_cfg.py:23-36: the model is fed
 *                    cat[x_t, bilinear upsample of sr_y]); not together with y
 *   x              : device fp32 [B,4,HW]: x_T on entry, the sample on return (when the stream has drained)
 *   step_noise     : device fp32 [n_steps][B,4,HW]; NULL when no step draws noise (DDIM with eta = 0) or with plan->generate_noise
 *   x0             : NULL or device fp32 [B,4,HW]: pred_x_0 of the last step
 *   scratch        : device memory of ivid_sample_scratch_bytes(...) bytes, owned by the caller, free for reuse when the
 *                    stream has drained
 *   stream         : as for ivid_unet_forward with use_graph = 1 -- the programs capture their hipGraph on it the second time
 *                    they run, so it must be a created stream, not the legacy default stream
 * The whole plan is validated before anything is enqueued.  Results are bit-identical to driving the same programs and step
 * kernels from the host (tests/test_sample_loop_gpu.py). */
#define IVID_SAMPLE_DDIM 0
#define IVID_SAMPLE_DDPM 1
typedef struct {
  int kind;                  /* IVID_SAMPLE_DDIM: coef = ivid_ddim_coef[n_steps]; IVID_SAMPLE_DDPM: ivid_ddpm_coef[n_steps] */
  int n_steps;
  int hw;
  const long long* t_model;  /* host [n_steps] */
  const void* coef;          /* host [n_steps] */
  const int* engine_of_step; /* host [n_steps] or NULL (= engines[0] for every step) */
  int generate_noise;        /* != 0: noise the caller did not pass (step_noise / cond->hole_noise NULL) is drawn on the device:    */
  int first_step;            /*       ivid_randn(noise_seed, stream_id = 2 * (first_step + i) [+ 1 for the hole noise]) at step i; */
  unsigned long long noise_seed; /*   first_step = index of this call's first step in the whole chain (a loop cut into calls)    */
} ivid_sample_plan;
typedef struct {
  const float* y; const float* mask; const float* mask_rgb; const float* hole_noise;
  const float* rgb; const float* rgb_mask; const float* depth; const float* depth_mask; const float* convex;
  const float* sr_y; int sr_channels; int sr_size;
} ivid_sample_cond;
long long ivid_sample_scratch_bytes(void* const* engines, int n_engines, const ivid_sample_plan* plan, const ivid_sample_cond* cond);
int ivid_sample(void* const* engines, int n_engines, const ivid_sample_plan* plan, const long long* classes,
                const ivid_sample_cond* cond, float* x, const float* step_noise, float* x0, void* scratch,
                long long scratch_bytes, void* stream);

/* ---- RGBD depth-warp conditioning (replaces rgbd_3d + the moderngl/OpenGL renderer) ----
 * Step 1, per generated view: depth -> textured grid mesh with frustum skirt (rgbd_3d/utils.py:144-260
 * depth_to_mesh(padding='frustum', cal_normal=True), linearize_depth :38-58, unproject :89-110,
 * triangulate :113-134, mask_discontinuity :137-141, cal_depth_normal :263-274), batched over B samples.
 *   rgbd          : fp32 [B,4,S,S]; input_mode 0: network output in [-1,1] (RGB + z-buffer depth encoded with nearv/farv,
 *                   inference/sample.py:83,126); input_mode 1: a stored scene — RGB in [0,1] and METRIC depth
 *                   (load_scene, inference/utils.py:103-113)
 *   padding       : -1: padding='frustum' (the sampling driver, sample.py:129-133); >= 0: numeric padding in pixels
 *                   (load_scene uses 32, forward_backward_warp the image size; utils.py:201-205); -2: padding=None
 *                   (forward_backward_warp's second mesh, utils.py:391-398): the ring of the (S+2)^2 grid then holds copies
 *                   of the border vertices -- degenerate triangles that are never rasterised -- and no padding flag
 *   atol, rtol    : discontinuity test (diff > atol AND inverse-diff > rtol); +inf/+inf = the reference's atol=rtol=None
 *                   (no flagging at all, utils.py:223)
 *   inv_modelview : fp32 [B][16] row-major camera->world matrix (glm.inverse(modelview), utils.py:232-237)
 *   verts         : out fp32 [B][(S+2)^2][9] = world position(3), world normal(3), uv(2), flag(1) with
 *                   flag = 1*discontinuity + 2*padding + 4*eroded (utils.py:249) — the reference's VBO layout
 *                   (moderngl_renderer.py:284-290)
 *   diag          : out u8 [B][(S+1)^2], 1 = quad split along its 00-11 diagonal (the index buffer is implicit)
 *   colors        : out fp32 [B][S][S][3] RGB in [0,1] (the NEAREST texture, moderngl_renderer.py:293)
 *   scratch_depth : fp32 [B][(S+2)^2]; scratch_flags: int32 [B][(S+2)^2] */
int ivid_mesh_build(const float* rgbd, int B, int S, const float* inv_modelview, float fov_deg, float nearv,
                    float farv, float atol, float rtol, int erode, float padding, int input_mode, float* verts,
                    unsigned char* diag, float* colors, float* scratch_depth, int* scratch_flags, void* stream);
/* Step 2, per target view: z-buffered rasterisation of every source-view mesh ALONE (depth test '<', no cull,
 * moderngl_renderer.py:198-202,307-312; aggregation.vsh/.fsh) + weighted aggregation across source views
 * (clear.csh, aggregation.csh) + the read-back math of AggregationRenderer.render (:317-331), image row 0 = top.
 *   verts/diag/colors : [NV][B][...] as produced by ivid_mesh_build for source views 0..NV-1
 *   campos  : fp32 [NV][B][3] source camera positions (u_sample_camera = inverse(mv_i)[3], :310-311)
 *   mvp     : fp32 [B][16] row-major projection * modelview of the TARGET view
 *   R       : render size (S * ssaa);  rnear/rfar: the renderer's own planes (0.01 / 200, :161)
 *   zbuf    : scratch u64 [NV][B][R*R]
 *   outputs : color8 u8 [B][R][R][3] (to8b of the resolved colour), depth_lin fp32 [B][R][R] (metric depth),
 *             mask_color / mask_depth u8 [B][R][R]
 *   work / work_cap   : caller-owned queue of the LARGE triangles (skirt / discontinuity sheets seen from another
 *                       camera; every triangle of a noisy depth map): int32 [2 + work_cap], one id per entry; they are
 *                       rasterised by one wave each in a second pass (8x8-pixel tiles, one tile test per lane) instead
 *                       of by their own thread.  NV*B*2*(S+1)^2 entries hold every triangle; a full queue or
 *                       work_cap = 0 only costs speed (the triangle's own thread walks its box), the result is the same.
 *   color_f32         : NULL, or fp32 [B][R][R][3]: the colour BEFORE to8b, i.e. `color` of the edict that
 *                       AggregationRenderer.render returns (moderngl_renderer.py:317-319,333-338)
 * Pixel centres exactly on a triangle edge follow the top-left fill rule. */
int ivid_warp_render(const float* verts, const unsigned char* diag, const float* colors, const float* campos, int NV,
                     int B, int S, const float* mvp, int R, float rnear, float rfar, unsigned long long* zbuf,
                     unsigned char* color8, float* depth_lin, unsigned char* mask_color, unsigned char* mask_depth,
                     int* work, int work_cap, float* color_f32, void* stream);
/* Step 3: aggregate_conditions' SSAA resolve (rgbd_3d/utils.py:450-467): Pillow-exact 8-bit LANCZOS R->S
 * (two integer passes; bounds int32 [S][2], coeffs int32 [S][ksize] with 22 fractional bits computed by the host),
 * centre-sample depth + project_depth (:61-67), masks > 75 % of the ssaa^2 sub-pixels, depth_edge (:311-332),
 * cv2.erode((2*erode-1)^2) of the depth mask into the colour mask, masked colour/depth.
 *   lut255 : fp32 [256] = float32(i / 255.0);   tmp_h u8 [B][R][S][3]; tmp_small u8 [B][S][S][3];
 *   tmp_dproj fp32 [B][S][S]; tmp_masks u8 [3][B][S][S]
 *   outputs (all in [0,1], fp32): color [B,3,S,S], depth [B,1,S,S], mask [B,1,S,S], mask_rgb [B,1,S,S], convex [B,1,S,S] */
int ivid_warp_resolve(const unsigned char* color8, const float* depth_lin, const unsigned char* mask_color,
                      const unsigned char* mask_depth, int B, int S, int ssaa, const int* bounds, const int* coeffs,
                      int ksize, const float* lut255, float nearv, float farv, float atol, float rtol, int erode,
                      unsigned char* tmp_h, unsigned char* tmp_small, float* tmp_dproj, unsigned char* tmp_masks,
                      float* color, float* depth, float* mask, float* mask_rgb, float* convex, void* stream);

/* ---- SimpleRenderer.render (moderngl_renderer.py:96-148; shaders/simple.vsh, simple.fsh): the training-time warp renderer
 *      of forward_backward_warp (rgbd_3d/utils.py:335-417, datasets/base.py:219,238) on the same z-buffer kernels: ONE mesh
 *      per sample (verts/diag/colors [B][...] as ivid_mesh_build writes them), one depth-tested draw, no discard.
 *   outputs: color_f32 fp32 [B][R][R][3] (NEAREST texel of front faces, else 0), depth_lin fp32 [B][R][R] (linearised
 *            window depth; `rfar` where nothing was drawn), mask u8 [B][R][R] (alpha > 0.5: front face and not a
 *            discontinuity edge).  zbuf: scratch u64 [B][R*R]; work as in ivid_warp_render. */
int ivid_simple_render(const float* verts, const unsigned char* diag, const float* colors, int B, int S, const float* mvp,
                       int R, float rnear, float rfar, unsigned long long* zbuf, int* work, int work_cap, float* color_f32,
                       float* depth_lin, unsigned char* mask, void* stream);
/* to8b + Pillow's 8-bit LANCZOS R -> S (`np.array(Image.fromarray(to8b(res.color)).resize(..., LANCZOS))`, utils.py:387,
 * 401,454) of a float colour buffer [B][R][R][3]: out8 u8 [B][S][S][3]; tmp_hi8 u8 [B][R][R][3], tmp_h u8 [B][R][S][3];
 * bounds/coeffs/ksize = Pillow's coefficient tables (ivid_amd/rgbd_3d/resample.py). */
int ivid_resample8_lanczos(const float* color_f32, int B, int R, int S, const int* bounds, const int* coeffs, int ksize,
                           unsigned char* tmp_hi8, unsigned char* tmp_h, unsigned char* out8, void* stream);

#ifdef __cplusplus
}
#endif
#endif
