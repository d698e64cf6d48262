/* A serving host without Python: run a planned ADM-UNet forward from an engine file through the C ABI alone.
 *
 *   unet_engine_host <engine file> <inputs.bin> <output.bin> [repeats]
 *
 * inputs.bin = x (fp32 NCHW, x_bytes) | times (int64 [batch]) | classes (int64 [batch], only for class-conditional models);
 * output.bin = the forward's fp32 NCHW result (both guidance branches for a stacked plan).  With repeats > 1 the forward is
 * replayed (second run captures the hipGraph) and the mean time of the graph launches is printed.
 * The engine file is what `AdmUnet2d.export_engine(batch, stacked)` wrote (ivid_amd/diffusion/backbones/engine.py); this
 * program links libivid_hip.so and the HIP runtime and nothing else.  Build: see __graft_entry__.build().               */
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include <stdio.h>
#include <stdlib.h>
#include "../include/ivid_hip.h"

static void* slurp(const char* path, long long* n) {
  FILE* f = fopen(path, "rb");
  if (!f) { fprintf(stderr, "cannot open %s\n", path); exit(2); }
  fseek(f, 0, SEEK_END);
  *n = ftell(f);
  fseek(f, 0, SEEK_SET);
  void* p = malloc((size_t)*n ? (size_t)*n : 1);
  if (fread(p, 1, (size_t)*n, f) != (size_t)*n) { fprintf(stderr, "short read on %s\n", path); exit(2); }
  fclose(f);
  return p;
}

#define HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 3; } } while (0)
#define IVID(x) do { if ((x) != 0) { fprintf(stderr, "%s: %s\n", #x, ivid_last_error()); return 4; } } while (0)

int main(int argc, char** argv) {
  if (argc < 4) { fprintf(stderr, "usage: %s engine inputs.bin output.bin [repeats]\n", argv[0]); return 1; }
  const int repeats = argc > 4 ? atoi(argv[4]) : 1;
  long long nblob = 0, nin = 0;
  void* blob = slurp(argv[1], &nblob);
  void* unet = NULL;
  IVID(ivid_unet_load(blob, nblob, &unet));
  free(blob);
  int batch = 0, has_classes = 0;
  long long x_bytes = 0, out_bytes = 0;
  IVID(ivid_unet_info(unet, &batch, &has_classes, &x_bytes, &out_bytes, NULL));
  char* in = (char*)slurp(argv[2], &nin);
  const long long want = x_bytes + 8LL * batch * (has_classes ? 2 : 1);
  if (nin != want) { fprintf(stderr, "inputs.bin holds %lld bytes, the engine wants %lld\n", nin, want); return 2; }
  void *dx, *dt, *dc = NULL, *dout;
  hipStream_t s;
  HIP(hipStreamCreate(&s));
  HIP(hipMalloc(&dx, (size_t)x_bytes));
  HIP(hipMalloc(&dt, 8 * (size_t)batch));
  HIP(hipMalloc(&dout, (size_t)out_bytes));
  HIP(hipMemcpy(dx, in, (size_t)x_bytes, hipMemcpyHostToDevice));
  HIP(hipMemcpy(dt, in + x_bytes, 8 * (size_t)batch, hipMemcpyHostToDevice));
  if (has_classes) {
    HIP(hipMalloc(&dc, 8 * (size_t)batch));
    HIP(hipMemcpy(dc, in + x_bytes + 8LL * batch, 8 * (size_t)batch, hipMemcpyHostToDevice));
  }
  hipEvent_t e0, e1;
  HIP(hipEventCreate(&e0));
  HIP(hipEventCreate(&e1));
  for (int r = 0; r < repeats; ++r) {
    if (r == 2) HIP(hipEventRecord(e0, s));
    IVID(ivid_unet_forward(unet, dx, dt, dc, dout, repeats > 1, s));
  }
  HIP(hipEventRecord(e1, s));
  HIP(hipStreamSynchronize(s));
  if (repeats > 2) {
    float ms = 0.f;
    HIP(hipEventElapsedTime(&ms, e0, e1));
    printf("batch %d: %.3f ms per forward over %d hipGraph launches\n", batch, ms / (repeats - 2), repeats - 2);
  }
  void* out = malloc((size_t)out_bytes);
  HIP(hipMemcpy(out, dout, (size_t)out_bytes, hipMemcpyDeviceToHost));
  FILE* f = fopen(argv[3], "wb");
  if (!f || fwrite(out, 1, (size_t)out_bytes, f) != (size_t)out_bytes) { fprintf(stderr, "cannot write %s\n", argv[3]); return 2; }
  fclose(f);
  IVID(ivid_program_destroy(unet));
  return 0;
}
