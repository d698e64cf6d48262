/* A sampling host without Python: the whole DDIM / DDPM loop (classifier-free guidance, one UNet program per precision tier)
 * from engine files through ONE call of the C ABI, `ivid_sample`.
 *
 *   sample_loop_host <plan file> <x_T.bin> <samples.bin> <engine file> [<engine file> ...]
 *
 * plan file (little endian; written by ivid_amd.diffusion.samplers.device_loop.write_plan_file):
 *   int32 magic 0x50535649 ("IVSP"), kind (0 DDIM, 1 DDPM), n_steps, hw, batch, has_classes, noise (0 none, 1 in the file,
 *   2 drawn on the device with ivid_randn), n_engines | uint64 noise_seed
 *   int64 t_model[n_steps] | int32 engine_of_step[n_steps] | coef[n_steps] (ivid_ddim_coef / ivid_ddpm_coef) |
 *   int64 classes[batch] (if has_classes) | float step_noise[n_steps][batch*4*hw] (if has_noise)
 * x_T.bin = fp32 [batch,4,hw]; samples.bin = the chain's result in the same layout.  The engine files are what
 * `AdmUnet2d.export_engine(batch, stacked, high_t=tier)` wrote.  Links libivid_hip.so and the HIP runtime, nothing else.   */
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
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
#define MAX_ENGINES 8

int main(int argc, char** argv) {
  if (argc < 5 || argc > 4 + MAX_ENGINES) { fprintf(stderr, "usage: %s plan x_T.bin samples.bin engine [engine ...]\n", argv[0]); return 1; }
  long long nplan = 0, nx = 0;
  char* pf = (char*)slurp(argv[1], &nplan);
  if (nplan < 40) { fprintf(stderr, "plan file too short\n"); return 2; }
  int hdr[8];
  memcpy(hdr, pf, sizeof(hdr));
  const int kind = hdr[1], n_steps = hdr[2], hw = hdr[3], batch = hdr[4], has_classes = hdr[5], has_noise = hdr[6] == 1, n_engines = hdr[7];
  const int gen_noise = hdr[6] == 2;
  unsigned long long seed = 0;
  memcpy(&seed, pf + 32, 8);
  if (hdr[0] != 0x50535649 || (kind != IVID_SAMPLE_DDIM && kind != IVID_SAMPLE_DDPM) || n_steps <= 0 || hw <= 0 || batch <= 0 ||
      n_engines != argc - 4) {
    fprintf(stderr, "bad plan file (or its engine count %d does not match the %d engine files given)\n", n_engines, argc - 4);
    return 2;
  }
  const long long csz = kind == IVID_SAMPLE_DDIM ? (long long)sizeof(ivid_ddim_coef) : (long long)sizeof(ivid_ddpm_coef);
  const long long img = 4LL * batch * hw * (long long)sizeof(float);
  const long long want = 40 + 8LL * n_steps + 4LL * n_steps + csz * n_steps + (has_classes ? 8LL * batch : 0) + (has_noise ? img * n_steps : 0);
  if (nplan != want) { fprintf(stderr, "plan file holds %lld bytes, its header describes %lld\n", nplan, want); return 2; }
  /* the tables are copied out of the file image: its int64 / struct sections are not aligned */
  long long* t_model = (long long*)malloc(8 * (size_t)n_steps);
  int* eng_of = (int*)malloc(4 * (size_t)n_steps);
  void* coef = malloc((size_t)(csz * n_steps));
  const char* q = pf + 40;
  memcpy(t_model, q, 8 * (size_t)n_steps); q += 8LL * n_steps;
  memcpy(eng_of, q, 4 * (size_t)n_steps);  q += 4LL * n_steps;
  memcpy(coef, q, (size_t)(csz * n_steps)); q += csz * n_steps;
  const char* classes_h = has_classes ? q : NULL;  q += has_classes ? 8LL * batch : 0;
  const char* noise_h = has_noise ? q : NULL;

  void* engines[MAX_ENGINES];
  for (int k = 0; k < n_engines; ++k) {
    long long nb = 0;
    void* blob = slurp(argv[4 + k], &nb);
    IVID(ivid_unet_load(blob, nb, &engines[k]));
    free(blob);
  }
  char* xT = (char*)slurp(argv[2], &nx);
  if (nx != img) { fprintf(stderr, "x_T.bin holds %lld bytes, the plan wants %lld\n", nx, img); return 2; }

  hipStream_t s;
  HIP(hipStreamCreate(&s));
  void *dx, *dcls = NULL, *dnoise = NULL, *dscratch;
  HIP(hipMalloc(&dx, (size_t)img));
  HIP(hipMemcpy(dx, xT, (size_t)img, hipMemcpyHostToDevice));
  if (has_classes) { HIP(hipMalloc(&dcls, 8 * (size_t)batch)); HIP(hipMemcpy(dcls, classes_h, 8 * (size_t)batch, hipMemcpyHostToDevice)); }
  if (has_noise) { HIP(hipMalloc(&dnoise, (size_t)(img * n_steps))); HIP(hipMemcpy(dnoise, noise_h, (size_t)(img * n_steps), hipMemcpyHostToDevice)); }
  ivid_sample_plan plan = {kind, n_steps, hw, t_model, coef, eng_of, gen_noise, 0, seed};
  const long long sb = ivid_sample_scratch_bytes(engines, n_engines, &plan, NULL);
  if (sb < 0) { fprintf(stderr, "ivid_sample_scratch_bytes: %s\n", ivid_last_error()); return 4; }
  HIP(hipMalloc(&dscratch, (size_t)sb));

  hipEvent_t e0, e1;
  HIP(hipEventCreate(&e0));
  HIP(hipEventCreate(&e1));
  HIP(hipEventRecord(e0, s));
  IVID(ivid_sample(engines, n_engines, &plan, (const long long*)dcls, NULL, (float*)dx, (const float*)dnoise, NULL, dscratch, sb, s));
  HIP(hipEventRecord(e1, s));
  HIP(hipStreamSynchronize(s));
  float ms = 0.f;
  HIP(hipEventElapsedTime(&ms, e0, e1));
  printf("batch %d: %d steps over %d engine(s) in %.3f ms, one call\n", batch, n_steps, n_engines, ms);

  void* out = malloc((size_t)img);
  HIP(hipMemcpy(out, dx, (size_t)img, hipMemcpyDeviceToHost));
  FILE* f = fopen(argv[3], "wb");
  if (!f || fwrite(out, 1, (size_t)img, f) != (size_t)img) { fprintf(stderr, "cannot write %s\n", argv[3]); return 2; }
  fclose(f);
  for (int k = 0; k < n_engines; ++k) IVID(ivid_program_destroy(engines[k]));
  return 0;
}
