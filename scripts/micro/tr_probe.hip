// Probe of ds_read_b64_tr_b16 (gfx950): which LDS elements does lane l receive, as a function of the per-lane addresses?
// LDS is filled with element index (u16 at byte 2i holds i).  Test 0: dense addresses (lane l -> byte 8 l).  Test 1: a
// [row][64 halfwords] image (row stride 128 B), lane l of a 16-lane group -> row (l&15)/4, column quad (l&15)%4, +32 cols per
// upper group.  Prints per lane the four u16 it received.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void probe(uint16_t* out, int mode) {
  __shared__ __attribute__((aligned(16))) uint16_t lds[4096];
  const int l = threadIdx.x;
  for (int i = l; i < 4096; i += 64) lds[i] = (uint16_t)i;
  __syncthreads();
  unsigned addr;
  if (mode == 0) addr = 8 * l;
  else {
    const int i = l & 15, g = (l >> 4) & 1, hf = l >> 5;
    addr = ((4 * hf + i / 4) * 64 + 16 * g + 4 * (i % 4)) * 2;      // halfword (row, col) -> byte
  }
  addr += (unsigned)(uintptr_t)lds;   // LDS pointers are 32-bit offsets in the low word
  typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
  u32x2 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
  out[l * 4 + 0] = v[0] & 0xffff; out[l * 4 + 1] = v[0] >> 16; out[l * 4 + 2] = v[1] & 0xffff; out[l * 4 + 3] = v[1] >> 16;
}
int main() {
  uint16_t* d; hipMalloc(&d, 64 * 4 * 2);
  uint16_t h[256];
  for (int mode = 0; mode < 2; ++mode) {
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, mode);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("mode %d\n", mode);
    for (int l = 0; l < 64; ++l) {
      if (mode == 0) printf("lane %2d: %4d %4d %4d %4d\n", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3]);
      else printf("lane %2d: (r%2d,c%2d) (r%2d,c%2d) (r%2d,c%2d) (r%2d,c%2d)\n", l, h[l*4]/64, h[l*4]%64, h[l*4+1]/64, h[l*4+1]%64, h[l*4+2]/64, h[l*4+2]%64, h[l*4+3]/64, h[l*4+3]%64);
    }
  }
  return 0;
}
