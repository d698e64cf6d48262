// Launch program: the C-side owner of one UNet forward (SURVEY.md §8b "ivid_unet_create / ivid_unet_forward").
//
// The host plans a forward ONCE (which kernels, in which order, on which arena buffers: ivid_amd/diffusion/backbones/
// plan.py) and hands the resulting launch list to this object; from then on a forward is ONE C call: copy the caller's
// inputs into the program's static input buffers, replay the list (first call: eagerly, which also sets kernel
// attributes; second call: captured into a hipGraph; afterwards: hipGraphLaunch), copy the result out.  No Python runs
// per forward, and a non-Python host can drive the same object through include/ivid_hip.h.
#include <cstring>
#include <string>
#include <vector>
#include "internal.h"

namespace {

union Slot { long long i; double f; };
struct Op { int code = 0; int nargs = 0; Slot a[32] = {}; };

// Arguments (without the trailing stream) of the entry point behind every op code, index = IVID_OP_*: run_op reads fixed slots,
// so a launch record whose nargs disagrees -- a truncated engine file, or one written against other signatures -- is refused
// before anything is dispatched (ivid_program_add, ivid_unet_load).  tests/test_host_logic.py holds this table against the
// ctypes signatures of ivid_amd/_lib.py.
constexpr int kArity[IVID_OP_LAST + 1] = {
    -1,
    /* 1 CONV2D */ 18, /* 2 CONV3X3_GN */ 17, /* 3 CONV3X3_GN_SKIP */ 22, /* 4 CONV3X3_GN_OUT */ 11, /* 5 GN_PARTIAL */ 8,
    /* 6 GN_FINALIZE */ 13, /* 7 GN_FINALIZE2 */ 16, /* 8 GN_APPLY */ 12, /* 9 ATTENTION */ 6, /* 10 EMBED_INPUTS */ 11,
    /* 11 SILU_F32 */ 3, /* 12 STEM_IM2COL */ 9, /* 13 CONV3X3_UP */ 14, /* 14 COPY */ 3, /* 15 CONV2D_C */ 20,
    /* 16 CONV3X3_GN_SKIP_C */ 26, /* 17 GN_APPLY_C */ 14, /* 18 CONV3X3_GN_OUT_C */ 13, /* 19 STEM_IM2COL_SPLIT */ 9,
    /* 20 CONV3X3_GN_SKIP_S */ 29, /* 21 F32_TO_HILO */ 5, /* 22 GN_APPLY_P */ 16, /* 23 CONV3X3_GN_O16 */ 17,
    /* 24 GN_PARTIAL_C */ 10, /* 25 CONV2D_O16 */ 18};

struct Program {
  std::vector<Op> ops;
  hipGraphExec_t graph = nullptr;
  int runs = 0;
  // model boundary (optional): static input / output buffers of the plan
  float* x_in = nullptr; long long x_bytes = 0;
  long long* t_in = nullptr; long long* c_in = nullptr; int batch = 0;
  float* out = nullptr; long long out_bytes = 0;
  void* slab = nullptr; long long slab_bytes = 0;   // ivid_unet_load: the program owns every buffer its launches touch
  int dims[4] = {0, 0, 0, 0};                        // engine files: output rows, in / out channels, image size
};

#define P(k) ((void*)(intptr_t)o.a[k].i)
#define CP(k) ((const void*)(intptr_t)o.a[k].i)
#define FP(k) ((float*)(intptr_t)o.a[k].i)
#define CFP(k) ((const float*)(intptr_t)o.a[k].i)
#define I(k) ((int)o.a[k].i)
#define F(k) ((float)o.a[k].f)

int run_op(const Op& o, void* s) {
  switch (o.code) {
    case IVID_OP_CONV2D:
      return ivid_conv2d(I(0), CP(1), I(2), CP(3), I(4), CP(5), CFP(6), P(7), CP(8), I(9), I(10), I(11), I(12), I(13), I(14), I(15),
                         I(16), FP(17), s);
    case IVID_OP_CONV3X3_UP:
      return ivid_conv3x3_up(I(0), CP(1), I(2), CP(3), I(4), CP(5), CFP(6), P(7), I(8), I(9), I(10), I(11), I(12), FP(13), s);
    case IVID_OP_CONV3X3_GN:
      return ivid_conv3x3_gn(I(0), CP(1), I(2), CP(3), I(4), CFP(5), I(6), CP(7), CFP(8), P(9), CP(10), I(11), I(12), I(13), I(14),
                             I(15), FP(16), s);
    case IVID_OP_CONV3X3_GN_SKIP:
      return ivid_conv3x3_gn_skip(I(0), CP(1), I(2), CP(3), I(4), CFP(5), I(6), CP(7), CFP(8), P(9), CP(10), I(11), I(12), I(13),
                                  I(14), I(15), FP(16), CP(17), I(18), CP(19), I(20), CP(21), s);
    case IVID_OP_CONV3X3_GN_OUT:
      return ivid_conv3x3_gn_out(I(0), CP(1), I(2), CFP(3), CP(4), CFP(5), FP(6), I(7), I(8), I(9), I(10), s);
    case IVID_OP_GN_PARTIAL:
      return ivid_gn_partial(I(0), CP(1), I(2), CP(3), I(4), I(5), I(6), FP(7), s);
    case IVID_OP_GN_PARTIAL_C:
      return ivid_gn_partial_c(I(0), CP(1), CP(2), I(3), CP(4), CP(5), I(6), I(7), I(8), FP(9), s);
    case IVID_OP_GN_FINALIZE:
      return ivid_gn_finalize(CFP(0), I(1), I(2), I(3), I(4), I(5), F(6), CFP(7), CFP(8), CFP(9), I(10), I(11), FP(12), s);
    case IVID_OP_GN_FINALIZE2:
      return ivid_gn_finalize2(CFP(0), I(1), I(2), CFP(3), I(4), I(5), I(6), I(7), I(8), F(9), CFP(10), CFP(11), CFP(12), I(13),
                               I(14), FP(15), s);
    case IVID_OP_GN_APPLY:
      return ivid_gn_apply(I(0), CP(1), I(2), CP(3), I(4), CFP(5), P(6), I(7), I(8), I(9), I(10), I(11), s);
    case IVID_OP_ATTENTION:
      return ivid_attention(I(0), CP(1), P(2), I(3), I(4), I(5), s);
    case IVID_OP_EMBED_INPUTS:
      return ivid_embed_inputs((const int64_t*)CP(0), (const int64_t*)CP(1), I(2), I(3), I(4), CFP(5), I(6), CFP(7), I(8), FP(9),
                               FP(10), s);
    case IVID_OP_SILU_F32:
      return ivid_silu_f32(CFP(0), FP(1), o.a[2].i, s);
    case IVID_OP_COPY:
      return ivid_copy(P(0), CP(1), o.a[2].i, s);
    case IVID_OP_STEM_IM2COL:
      return ivid_stem_im2col(I(0), CFP(1), I(2), I(3), I(4), I(5), I(6), I(7), P(8), s);
    case IVID_OP_CONV2D_C:
      return ivid_conv2d_c(I(0), CP(1), I(2), CP(3), I(4), CP(5), CFP(6), P(7), P(8), CP(9), CP(10), I(11), I(12), I(13), I(14),
                           I(15), I(16), I(17), I(18), FP(19), s);
    case IVID_OP_CONV3X3_GN_SKIP_C:
      return ivid_conv3x3_gn_skip_c(I(0), CP(1), CP(2), I(3), CP(4), CP(5), I(6), CFP(7), I(8), CP(9), CFP(10), P(11), P(12), CP(13),
                                    CP(14), I(15), I(16), I(17), I(18), I(19), FP(20), CP(21), I(22), CP(23), I(24), CP(25), s);
    case IVID_OP_CONV3X3_GN_SKIP_S:
      return ivid_conv3x3_gn_skip_s(I(0), CP(1), CP(2), I(3), CP(4), CP(5), I(6), CFP(7), I(8), CP(9), CFP(10), P(11), P(12), CP(13),
                                    CP(14), I(15), I(16), I(17), I(18), I(19), FP(20), CP(21), I(22), CP(23), I(24), CP(25), CP(26),
                                    CP(27), CP(28), s);
    case IVID_OP_F32_TO_HILO:
      return ivid_f32_to_hilo(I(0), CFP(1), P(2), P(3), o.a[4].i, s);
    case IVID_OP_CONV2D_O16:
      return ivid_conv2d_o16(CP(0), I(1), CP(2), I(3), CP(4), CFP(5), P(6), P(7), P(8), CP(9), I(10), I(11), I(12), I(13), I(14), I(15),
                             I(16), FP(17), s);
    case IVID_OP_CONV3X3_GN_O16:
      return ivid_conv3x3_gn_o16(CP(0), I(1), CP(2), I(3), CFP(4), CP(5), CFP(6), P(7), P(8), P(9), CP(10), I(11), I(12), I(13), I(14),
                                 I(15), FP(16), s);
    case IVID_OP_GN_APPLY_P:
      return ivid_gn_apply_p(I(0), CP(1), CP(2), I(3), CP(4), CP(5), I(6), CFP(7), P(8), P(9), P(10), I(11), I(12), I(13), I(14), I(15), s);
    case IVID_OP_GN_APPLY_C:
      return ivid_gn_apply_c(I(0), CP(1), CP(2), I(3), CP(4), CP(5), I(6), CFP(7), P(8), I(9), I(10), I(11), I(12), I(13), s);
    case IVID_OP_CONV3X3_GN_OUT_C:
      return ivid_conv3x3_gn_out_c(I(0), CP(1), CP(2), I(3), CFP(4), CP(5), CP(6), CFP(7), FP(8), I(9), I(10), I(11), I(12), s);
    case IVID_OP_STEM_IM2COL_SPLIT:
      return ivid_stem_im2col_split(I(0), CFP(1), I(2), I(3), I(4), I(5), I(6), I(7), P(8), s);
    default:
      return ivid_set_error("program: unknown op code", hipSuccess);
  }
}

}  // namespace

extern "C" int ivid_program_create(void** handle_out) {
  if (!handle_out) return ivid_set_error("program_create: null", hipSuccess);
  *handle_out = new Program();
  return 0;
}

// args: nargs slots of 8 bytes each; integers and device pointers as int64, floats as double (see ivid_hip.h)
extern "C" int ivid_program_op_arity(int op) { return (op >= 1 && op <= IVID_OP_LAST) ? kArity[op] : -1; }

extern "C" int ivid_program_add(void* handle, int op, const void* args, int nargs) {
  Program* p = (Program*)handle;
  if (!p || nargs < 0 || nargs > 32 || (nargs && !args)) return ivid_set_error("program_add: bad arguments", hipSuccess);
  if (op < 1 || op > IVID_OP_LAST) return ivid_set_error("program_add: unknown op code", hipSuccess);
  if (nargs != kArity[op]) return ivid_set_error("program_add: argument count does not match the entry point of this op code", hipSuccess);
  if (p->graph) return ivid_set_error("program_add: program already captured", hipSuccess);
  Op o;
  o.code = op;
  o.nargs = nargs;
  memcpy(o.a, args, (size_t)nargs * sizeof(Slot));
  p->ops.push_back(o);
  return 0;
}

extern "C" int ivid_program_num_ops(void* handle) { return handle ? (int)((Program*)handle)->ops.size() : -1; }

static int enqueue(Program* p, void* stream) {
  for (size_t k = 0; k < p->ops.size(); ++k) {
    const int st = run_op(p->ops[k], stream);
    if (st != 0) return st;
  }
  return 0;
}

// Replay the launch list on `stream`.  use_graph = 0: always eager.  Otherwise run 1 is eager (kernel attributes, zero
// page), run 2 captures the list into a hipGraph, later runs are one hipGraphLaunch.
extern "C" int ivid_program_launch(void* handle, int use_graph, void* stream) {
  Program* p = (Program*)handle;
  if (!p) return ivid_set_error("program_launch: null handle", hipSuccess);
  hipStream_t s = (hipStream_t)stream;
  if (!use_graph || p->runs == 0) {
    p->runs++;
    return enqueue(p, stream);
  }
  if (!p->graph) {
    hipError_t e = hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
    if (e != hipSuccess) return ivid_set_error("program: hipStreamBeginCapture", e);
    const int st = enqueue(p, stream);
    hipGraph_t g = nullptr;
    e = hipStreamEndCapture(s, &g);
    if (st != 0) { if (g) hipGraphDestroy(g); return st; }
    if (e != hipSuccess) return ivid_set_error("program: hipStreamEndCapture", e);
    e = hipGraphInstantiate(&p->graph, g, nullptr, nullptr, 0);
    hipGraphDestroy(g);
    if (e != hipSuccess) return ivid_set_error("program: hipGraphInstantiate", e);
  }
  p->runs++;
  hipError_t e = hipGraphLaunch(p->graph, s);
  return e == hipSuccess ? 0 : ivid_set_error("program: hipGraphLaunch", e);
}

extern "C" int ivid_program_has_graph(void* handle) { return handle && ((Program*)handle)->graph ? 1 : 0; }

// Declare the model boundary of a UNet program: its static input buffers (x fp32 [batch,Cin,S,S] = x_bytes, times /
// classes int64 [batch], classes NULL for models without class embedding) and its output buffer (fp32 NCHW, out_bytes).
extern "C" int ivid_unet_bind(void* handle, void* x_in, long long x_bytes, void* t_in, void* c_in, int batch, void* out,
                              long long out_bytes) {
  Program* p = (Program*)handle;
  if (!p || !x_in || !t_in || !out || batch <= 0) return ivid_set_error("unet_bind: bad arguments", hipSuccess);
  p->x_in = (float*)x_in; p->x_bytes = x_bytes; p->t_in = (long long*)t_in; p->c_in = (long long*)c_in; p->batch = batch;
  p->out = (float*)out; p->out_bytes = out_bytes;
  return 0;
}

// AdmUnet2d.forward (adm.py:526-566) as ONE call: x fp32 NCHW, times int64 [batch], classes int64 [batch] or NULL (= null
// class for every row), out fp32 NCHW (NULL: leave the result in the program's own output buffer).  All device pointers,
// everything enqueued on `stream`, nothing synchronises.
extern "C" int ivid_unet_forward(void* handle, const void* x, const void* times, const void* classes, void* out, int use_graph,
                                 void* stream) {
  Program* p = (Program*)handle;
  if (!p || !p->x_in) return ivid_set_error("unet_forward: program has no bound boundary (ivid_unet_bind)", hipSuccess);
  hipStream_t s = (hipStream_t)stream;
  hipError_t e = hipMemcpyAsync(p->x_in, x, (size_t)p->x_bytes, hipMemcpyDeviceToDevice, s);
  if (e == hipSuccess) e = hipMemcpyAsync(p->t_in, times, (size_t)p->batch * 8, hipMemcpyDeviceToDevice, s);
  if (e == hipSuccess && p->c_in) {
    if (classes) e = hipMemcpyAsync(p->c_in, classes, (size_t)p->batch * 8, hipMemcpyDeviceToDevice, s);
    else e = hipMemsetAsync(p->c_in, 0xff, (size_t)p->batch * 8, s);   // int64 -1 = null class (adm.py:550-552)
  }
  if (e != hipSuccess) return ivid_set_error("unet_forward: input copy", e);
  const int st = ivid_program_launch(handle, use_graph, stream);
  if (st != 0) return st;
  if (out && out != (void*)p->out) {
    e = hipMemcpyAsync(out, p->out, (size_t)p->out_bytes, hipMemcpyDeviceToDevice, s);
    if (e != hipSuccess) return ivid_set_error("unet_forward: output copy", e);
  }
  return 0;
}

// ---- engine files: a planned forward frozen by ivid_amd/diffusion/backbones/engine.py (layout documented there) ----
namespace {
struct Reader {
  const unsigned char* b; size_t n; size_t p = 0; bool ok = true;
  template <class T> T get() {
    T v{};
    if (!ok || n - p < sizeof(T) || p > n) { ok = false; return v; }
    memcpy(&v, b + p, sizeof(T)); p += sizeof(T);
    return v;
  }
};
struct EngBuf { int kind; unsigned long long nbytes, src, dev; };
}  // namespace

// Build a UNet program from an engine file held in HOST memory: one device allocation holds every buffer (arena scratch,
// repacked weights, the static inputs and the output), constants are uploaded, pointer arguments are relocated, the model
// boundary is bound.  The handle is what ivid_unet_forward / ivid_program_destroy take; destroy frees the allocation.
extern "C" int ivid_unet_load(const void* blob, long long nbytes, void** handle_out) {
  if (!blob || nbytes < 64 || !handle_out) return ivid_set_error("unet_load: bad arguments", hipSuccess);
  Reader r{(const unsigned char*)blob, (size_t)nbytes};
  if (memcmp(r.b, "IVIDENG", 7) != 0) return ivid_set_error("unet_load: not an ivid engine file (magic)", hipSuccess);
  if (r.b[7] != '2') return ivid_set_error("unet_load: engine file format version not supported by this library (re-export it)", hipSuccess);
  r.p = 8;
  // the launch list is a list of calls INTO this library: a file written against other entry-point signatures must not run
  if (r.get<unsigned>() != (unsigned)IVID_ENGINE_ABI)
    return ivid_set_error("unet_load: engine file was exported for another IVID_ENGINE_ABI (entry-point signatures changed): re-export it", hipSuccess);
  const unsigned batch = r.get<unsigned>(), has_cls = r.get<unsigned>();
  const unsigned rows = r.get<unsigned>(), cin = r.get<unsigned>(), cout = r.get<unsigned>(), size = r.get<unsigned>();
  const unsigned long long x_bytes = r.get<unsigned long long>(), out_bytes = r.get<unsigned long long>();
  const unsigned ix = r.get<unsigned>(), it = r.get<unsigned>(), ic = r.get<unsigned>(), io = r.get<unsigned>();
  const unsigned nb = r.get<unsigned>();
  if (!r.ok || nb == 0 || nb > (1u << 20) || batch == 0) return ivid_set_error("unet_load: truncated or malformed header", hipSuccess);
  std::vector<EngBuf> bufs(nb);
  unsigned long long total = 0;
  for (unsigned k = 0; k < nb; ++k) {
    EngBuf& e = bufs[k];
    e.kind = r.get<unsigned char>(); e.nbytes = r.get<unsigned long long>(); e.src = r.get<unsigned long long>();
    if (!r.ok || e.kind > 1 || e.nbytes > (1ull << 40)) return ivid_set_error("unet_load: malformed buffer table", hipSuccess);
    if (e.kind == 1 && (e.src > (unsigned long long)nbytes || e.nbytes > (unsigned long long)nbytes - e.src))
      return ivid_set_error("unet_load: constant data outside the file", hipSuccess);
    e.dev = total;
    total += (e.nbytes + 255) / 256 * 256 + 256;   // every buffer starts on a 256-byte boundary, as the arena's do
  }
  if (ix >= nb || it >= nb || io >= nb || (has_cls && ic >= nb)) return ivid_set_error("unet_load: boundary buffer index out of range", hipSuccess);
  if (x_bytes != 4ull * batch * cin * size * size || out_bytes != 4ull * rows * cout * size * size || (rows != batch && rows != 2 * batch))
    return ivid_set_error("unet_load: boundary sizes disagree with the stated shape", hipSuccess);
  if (bufs[ix].nbytes < x_bytes || bufs[io].nbytes < out_bytes || bufs[it].nbytes < 8ull * batch || (has_cls && bufs[ic].nbytes < 8ull * batch))
    return ivid_set_error("unet_load: boundary buffer smaller than the boundary", hipSuccess);
  const unsigned nops = r.get<unsigned>();
  if (!r.ok || nops > (1u << 20)) return ivid_set_error("unet_load: malformed op count", hipSuccess);
  Program* p = new Program();
  p->ops.reserve(nops);
  const char* bad = nullptr;
  // pass 1: decode with pointer slots holding (buffer, offset) -> relocate after the allocation
  std::vector<std::pair<size_t, int>> relocs;   // (op, arg)
  for (unsigned k = 0; k < nops && !bad; ++k) {
    Op o;
    o.code = (int)r.get<unsigned>(); o.nargs = (int)r.get<unsigned>();
    if (!r.ok || o.nargs < 0 || o.nargs > 32 || o.code < 1 || o.code > IVID_OP_LAST) { bad = "unet_load: malformed op"; break; }
    if (o.nargs != kArity[o.code]) { bad = "unet_load: op argument count does not match its entry point"; break; }
    for (int a = 0; a < o.nargs; ++a) {
      const unsigned char tag = r.get<unsigned char>();
      if (tag == 0) o.a[a].i = r.get<long long>();
      else if (tag == 1) o.a[a].f = r.get<double>();
      else if (tag == 3) { r.get<long long>(); o.a[a].i = 0; }
      else if (tag == 2) {
        const unsigned bi = r.get<unsigned>(); r.get<unsigned>();
        const unsigned long long off = r.get<unsigned long long>();
        if (!r.ok || bi >= nb || off > bufs[bi].nbytes) { bad = "unet_load: pointer argument outside its buffer"; break; }
        o.a[a].i = (long long)(bufs[bi].dev + off);
        relocs.push_back({p->ops.size(), a});
      } else { bad = "unet_load: unknown argument tag"; break; }
      if (!r.ok) { bad = "unet_load: truncated op list"; break; }
    }
    if (!bad) p->ops.push_back(o);
  }
  if (bad) { delete p; return ivid_set_error(bad, hipSuccess); }
  hipError_t e = hipMalloc(&p->slab, (size_t)total);
  if (e != hipSuccess) { p->slab = nullptr; ivid_program_destroy(p); return ivid_set_error("unet_load: hipMalloc of the engine's buffers", e); }
  p->slab_bytes = (long long)total;
  char* base = (char*)p->slab;
  for (unsigned k = 0; k < nb && e == hipSuccess; ++k)
    if (bufs[k].kind == 1 && bufs[k].nbytes) e = hipMemcpy(base + bufs[k].dev, r.b + bufs[k].src, (size_t)bufs[k].nbytes, hipMemcpyHostToDevice);
  if (e != hipSuccess) { ivid_program_destroy(p); return ivid_set_error("unet_load: constant upload", e); }
  for (auto& rc : relocs) p->ops[rc.first].a[rc.second].i += (long long)(intptr_t)base;
  p->x_in = (float*)(base + bufs[ix].dev); p->x_bytes = (long long)x_bytes;
  p->t_in = (long long*)(base + bufs[it].dev);
  p->c_in = has_cls ? (long long*)(base + bufs[ic].dev) : nullptr;
  p->batch = (int)batch;
  p->out = (float*)(base + bufs[io].dev); p->out_bytes = (long long)out_bytes;
  p->dims[0] = (int)rows; p->dims[1] = (int)cin; p->dims[2] = (int)cout; p->dims[3] = (int)size;
  *handle_out = p;
  return 0;
}

// The boundary of a bound / loaded UNet program: rows of x, whether it takes classes, bytes of x and of the output;
// dims[4] = output rows, in channels, out channels, image size (zeros for a program bound by hand).
extern "C" int ivid_unet_info(void* handle, int* batch, int* has_classes, long long* x_bytes, long long* out_bytes, int* dims) {
  Program* p = (Program*)handle;
  if (!p || !p->x_in) return ivid_set_error("unet_info: program has no bound boundary", hipSuccess);
  if (batch) *batch = p->batch;
  if (has_classes) *has_classes = p->c_in ? 1 : 0;
  if (x_bytes) *x_bytes = p->x_bytes;
  if (out_bytes) *out_bytes = p->out_bytes;
  if (dims) memcpy(dims, p->dims, sizeof(p->dims));
  return 0;
}

// ---- the whole sampling loop as one call (include/ivid_hip.h: ivid_sample) ----
namespace {
__global__ void fill_i64_kernel(long long* dst, long long v, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = v;
}
struct SampleGeom { int B, HW, cin, rows; bool stacked; };
// every program of the loop must have the same boundary: B rows of x with cin channels of HW pixels, B (plain) or 2B (both
// guidance branches stacked) rows of 4 output channels
int sample_geom(void* const* engines, int n_engines, int HW, SampleGeom* g) {
  if (!engines || n_engines <= 0 || HW <= 0) return ivid_set_error("sample: no programs / bad image size", hipSuccess);
  for (int k = 0; k < n_engines; ++k) {
    const Program* p = (const Program*)engines[k];
    if (!p || !p->x_in || !p->out || p->batch <= 0) return ivid_set_error("sample: program without a bound boundary", hipSuccess);
    const long long px = (long long)p->batch * HW * 4;
    if (p->x_bytes % px || p->out_bytes % (4 * px)) return ivid_set_error("sample: program boundary does not match the image size", hipSuccess);
    const int cin = (int)(p->x_bytes / px), mult = (int)(p->out_bytes / (4 * px));
    if (mult != 1 && mult != 2) return ivid_set_error("sample: program output must hold B or 2B four-channel rows", hipSuccess);
    if (k == 0) *g = SampleGeom{p->batch, HW, cin, mult * p->batch, mult == 2};
    else if (p->batch != g->B || cin != g->cin || (mult == 2) != g->stacked)
      return ivid_set_error("sample: the programs of a loop must share one boundary", hipSuccess);
  }
  return 0;
}
inline long long align256(long long v) { return (v + 255) / 256 * 256; }
}  // namespace

extern "C" long long ivid_sample_scratch_bytes(void* const* engines, int n_engines, const ivid_sample_plan* plan,
                                               const ivid_sample_cond* cond) {
  SampleGeom g;
  if (!plan || plan->n_steps <= 0) { ivid_set_error("sample: empty plan", hipSuccess); return -1; }
  if (sample_geom(engines, n_engines, plan->hw, &g) != 0) return -1;
  const long long img = (long long)g.B * 4 * g.HW * 4;
  return align256((long long)plan->n_steps * g.B * 8) + (plan->generate_noise ? 4 : 2) * align256(img) +
         ((cond && (cond->y || cond->sr_y)) ? align256((long long)g.B * g.cin * g.HW * 4) : 0);
}

// DdimSampler.sample / DdpmSampler.sample (ddim.py:150-163, ddpm.py:172-185) with the frameworks' model_inference
// (classifier_free_guidance.py:23-42, inpaint_cfg.py:51-83) as ONE call: per step [make_cond_inputs ->] UNet program (hipGraph) ->
// fused step kernel, everything enqueued on `stream`, nothing synchronises and no host code runs between the steps.
extern "C" int ivid_sample(void* const* engines, int n_engines, const ivid_sample_plan* plan, const long long* classes,
                           const ivid_sample_cond* cond, float* x, const float* step_noise, float* x0, void* scratch,
                           long long scratch_bytes, void* stream) {
  SampleGeom g;
  if (!plan || plan->n_steps <= 0 || !plan->t_model || !plan->coef || !x) return ivid_set_error("sample: bad arguments", hipSuccess);
  if (plan->kind != IVID_SAMPLE_DDIM && plan->kind != IVID_SAMPLE_DDPM) return ivid_set_error("sample: plan kind must be DDIM or DDPM", hipSuccess);
  int st = sample_geom(engines, n_engines, plan->hw, &g);
  if (st != 0) return st;
  const bool inpaint = cond && cond->y;
  const bool superres = cond && cond->sr_y;
  if (inpaint && superres) return ivid_set_error("sample: y and sr_y are two different frameworks' conditioning", hipSuccess);
  const bool gen = plan->generate_noise != 0;
  if (inpaint && (!cond->mask || !(cond->hole_noise || gen))) return ivid_set_error("sample: inpainting needs y, mask and the hole noise", hipSuccess);
  if (gen && plan->first_step < 0) return ivid_set_error("sample: first_step < 0", hipSuccess);
  int S = 0;
  if (superres) {
    while ((long long)S * S < g.HW) ++S;
    if (S * S != g.HW || cond->sr_channels <= 0 || cond->sr_size <= 0 || cond->sr_size > S)
      return ivid_set_error("sample: super-resolution needs a square image and sr_channels / sr_size", hipSuccess);
  }
  if (g.cin != (inpaint ? (cond->mask_rgb ? 10 : 9) : superres ? 4 + cond->sr_channels : 4))
    return ivid_set_error("sample: the programs' input channels do not match the conditioning", hipSuccess);
  const long long need = ivid_sample_scratch_bytes(engines, n_engines, plan, cond);
  if (!scratch || scratch_bytes < need) return ivid_set_error("sample: scratch too small (ivid_sample_scratch_bytes)", hipSuccess);
  const ivid_ddim_coef* kd = plan->kind == IVID_SAMPLE_DDIM ? (const ivid_ddim_coef*)plan->coef : nullptr;
  const ivid_ddpm_coef* kp = plan->kind == IVID_SAMPLE_DDPM ? (const ivid_ddpm_coef*)plan->coef : nullptr;
  for (int i = 0; i < plan->n_steps; ++i) {   // refuse the whole loop before anything is enqueued
    const int e = plan->engine_of_step ? plan->engine_of_step[i] : 0;
    if (e < 0 || e >= n_engines) return ivid_set_error("sample: engine_of_step out of range", hipSuccess);
    const bool noisy = kd ? kd[i].sigma != 0.f : kp[i].std != 0.f;
    if (noisy && !step_noise && !gen) return ivid_set_error("sample: a step draws noise but step_noise is NULL", hipSuccess);
    if (kd && ((kd[i].replace_rgb_w >= 0.f && !(cond && cond->rgb && cond->rgb_mask)) ||
               (kd[i].replace_depth_w >= 0.f && !(cond && cond->depth && cond->depth_mask)) ||
               (kd[i].constrain_w >= 0.f && !(cond && cond->convex))))
      return ivid_set_error("sample: a step replaces / constrains but the tensors are missing", hipSuccess);
  }
  hipStream_t s = (hipStream_t)stream;
  const long long img_elems = (long long)g.B * 4 * g.HW;
  char* sp = (char*)scratch;
  long long* t_dev = (long long*)sp;  sp += align256((long long)plan->n_steps * g.B * 8);
  float* x_alt = (float*)sp;          sp += align256(img_elems * 4);
  float* x0_own = (float*)sp;         sp += align256(img_elems * 4);
  float* gen_step = gen ? (float*)sp : nullptr;   sp += gen ? align256(img_elems * 4) : 0;
  float* gen_hole = gen ? (float*)sp : nullptr;   sp += gen ? align256(img_elems * 4) : 0;
  float* cond_in = (inpaint || superres) ? (float*)sp : nullptr;
  float* cur = x;
  float* nxt = x_alt;
  float* x0w = x0 ? x0 : x0_own;
  for (int i = 0; i < plan->n_steps; ++i) {
    Program* p = (Program*)engines[plan->engine_of_step ? plan->engine_of_step[i] : 0];
    long long* ti = t_dev + (long long)i * g.B;
    hipLaunchKernelGGL(fill_i64_kernel, dim3((g.B + 255) / 256), dim3(256), 0, s, ti, plan->t_model[i], g.B);
    const float* model_in = cur;
    if (inpaint) {
      const float* hn = cond->hole_noise ? cond->hole_noise + (long long)i * img_elems : gen_hole;   // rgb noise [B,3,HW], then depth noise [B,1,HW]
      if (!cond->hole_noise) {
        st = ivid_randn(plan->noise_seed, 2ull * (unsigned long long)(plan->first_step + i) + 1, gen_hole, img_elems, stream);
        if (st != 0) return st;
      }
      st = ivid_inpaint_cond(cur, cond->y, cond->mask, cond->mask_rgb, hn, hn + (long long)g.B * 3 * g.HW, cond_in, g.B, g.HW, stream);
      if (st != 0) return st;
      model_in = cond_in;
    } else if (superres) {
      st = ivid_sr_cond(cur, cond->sr_y, cond_in, g.B, 4, cond->sr_channels, S, cond->sr_size, stream);
      if (st != 0) return st;
      model_in = cond_in;
    }
    st = ivid_unet_forward(p, model_in, ti, classes, nullptr, 1, stream);
    if (st != 0) return st;
    const float* eps_c = p->out;
    const float* eps_u = g.stacked ? p->out + img_elems : nullptr;
    const float* nz = step_noise ? step_noise + (long long)i * img_elems : nullptr;
    if (!step_noise && gen && (kd ? kd[i].sigma != 0.f : kp[i].std != 0.f)) {
      st = ivid_randn(plan->noise_seed, 2ull * (unsigned long long)(plan->first_step + i), gen_step, img_elems, stream);
      if (st != 0) return st;
      nz = gen_step;
    }
    if (kd) {
      st = ivid_ddim_step(cur, eps_c, eps_u, &kd[i], cond ? cond->rgb : nullptr, cond ? cond->rgb_mask : nullptr,
                          cond ? cond->depth : nullptr, cond ? cond->depth_mask : nullptr, cond ? cond->convex : nullptr,
                          kd[i].sigma != 0.f ? nz : nullptr, nxt, x0w, g.B, g.HW, stream);
    } else {
      st = ivid_ddpm_step(cur, eps_c, eps_u, &kp[i], kp[i].std != 0.f ? nz : nullptr, nxt, x0w, g.B, g.HW, stream);
    }
    if (st != 0) return st;
    float* tmp = cur; cur = nxt; nxt = tmp;
  }
  if (cur != x) {
    const hipError_t e = hipMemcpyAsync(x, cur, (size_t)img_elems * 4, hipMemcpyDeviceToDevice, s);
    if (e != hipSuccess) return ivid_set_error("sample: result copy", e);
  }
  return ivid_check_launch("sample");
}

extern "C" int ivid_program_destroy(void* handle) {
  Program* p = (Program*)handle;
  if (!p) return 0;
  if (p->graph) hipGraphExecDestroy(p->graph);
  if (p->slab) hipFree(p->slab);
  delete p;
  return 0;
}
