// Runtime glue of libivid_hip.so: error reporting, zero page, hipGraph capture, HIP-event timing.
#include <mutex>
#include <string>
#include <cstdio>
#include "internal.h"

namespace {
thread_local std::string g_err;
std::mutex g_mu;
void* g_zero[64] = {nullptr};
}  // namespace

int ivid_set_error(const char* what, hipError_t e) {
  g_err = what;
  if (e != hipSuccess) {
    g_err += ": ";
    g_err += hipGetErrorString(e);
  }
  return e != hipSuccess ? (int)e : -1;
}

int ivid_check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return ivid_set_error(what, e);
  return 0;
}

const void* ivid_zero_page() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) {
    ivid_set_error("zero page: no device", hipSuccess);
    return nullptr;
  }
  std::lock_guard<std::mutex> lk(g_mu);
  if (!g_zero[dev]) {
    void* p = nullptr;
    hipError_t e = hipMalloc(&p, 256);
    if (e != hipSuccess) { ivid_set_error("zero page: hipMalloc", e); return nullptr; }
    e = hipMemset(p, 0, 256);
    if (e != hipSuccess) { ivid_set_error("zero page: hipMemset", e); return nullptr; }
    g_zero[dev] = p;
  }
  return g_zero[dev];
}

extern "C" const char* ivid_last_error(void) { return g_err.c_str(); }
extern "C" int ivid_version(void) { return 1; }

extern "C" int ivid_graph_begin(void* stream) {
  hipError_t e = hipStreamBeginCapture((hipStream_t)stream, hipStreamCaptureModeThreadLocal);
  return e == hipSuccess ? 0 : ivid_set_error("hipStreamBeginCapture", e);
}
extern "C" int ivid_graph_end(void* stream, void** graph_exec_out) {
  hipGraph_t g = nullptr;
  hipError_t e = hipStreamEndCapture((hipStream_t)stream, &g);
  if (e != hipSuccess) return ivid_set_error("hipStreamEndCapture", e);
  hipGraphExec_t ge = nullptr;
  e = hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
  hipGraphDestroy(g);
  if (e != hipSuccess) return ivid_set_error("hipGraphInstantiate", e);
  *graph_exec_out = (void*)ge;
  return 0;
}
extern "C" int ivid_graph_launch(void* graph_exec, void* stream) {
  hipError_t e = hipGraphLaunch((hipGraphExec_t)graph_exec, (hipStream_t)stream);
  return e == hipSuccess ? 0 : ivid_set_error("hipGraphLaunch", e);
}
extern "C" int ivid_graph_destroy(void* graph_exec) {
  hipError_t e = hipGraphExecDestroy((hipGraphExec_t)graph_exec);
  return e == hipSuccess ? 0 : ivid_set_error("hipGraphExecDestroy", e);
}

extern "C" int ivid_event_create(void** ev_out) {
  hipEvent_t ev;
  hipError_t e = hipEventCreate(&ev);
  if (e != hipSuccess) return ivid_set_error("hipEventCreate", e);
  *ev_out = (void*)ev;
  return 0;
}
extern "C" int ivid_event_record(void* ev, void* stream) {
  hipError_t e = hipEventRecord((hipEvent_t)ev, (hipStream_t)stream);
  return e == hipSuccess ? 0 : ivid_set_error("hipEventRecord", e);
}
extern "C" int ivid_event_elapsed_ms(void* ev_start, void* ev_stop, float* ms_out) {
  hipError_t e = hipEventSynchronize((hipEvent_t)ev_stop);
  if (e != hipSuccess) return ivid_set_error("hipEventSynchronize", e);
  e = hipEventElapsedTime(ms_out, (hipEvent_t)ev_start, (hipEvent_t)ev_stop);
  return e == hipSuccess ? 0 : ivid_set_error("hipEventElapsedTime", e);
}
extern "C" int ivid_event_destroy(void* ev) {
  hipError_t e = hipEventDestroy((hipEvent_t)ev);
  return e == hipSuccess ? 0 : ivid_set_error("hipEventDestroy", e);
}
