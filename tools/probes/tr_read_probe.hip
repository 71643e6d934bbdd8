// Probe of gfx950's ds_read_b64_tr_b16: which four 16-bit LDS elements does lane l receive for a given set of per-lane
// addresses?  LDS holds u16 value = element index; four address patterns; prints, per pattern, the 4 element indices
// every lane got.   hipcc --offload-arch=gfx950 -O2 -o tr_read_probe tr_read_probe.hip && ./tr_read_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

__global__ void probe(int pattern, uint16_t *out) {
  __shared__ __attribute__((aligned(16))) uint16_t lds[8192];
  for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = static_cast<uint16_t>(i);
  __syncthreads();
  const int l = threadIdx.x;
  unsigned addr = 0;
  if (pattern == 0) addr = l * 8;                                                     // lane l -> elements 4l .. 4l+3
  if (pattern == 1) addr = (l & 3) * 8 + ((l >> 2) & 3) * 64 + (l >> 4) * 256;       // 4 chunks along a row, 4 rows of 32 elements
  if (pattern == 2) addr = (l & 15) * 64 + (l >> 4) * 8;                              // lane (l & 15) = row of 32 elements, group = chunk
  if (pattern == 3) addr = 0;
  if (pattern == 4) addr = (l & 15) * 2 * 2 + (l >> 4) * 128;                         // the guide's formula in elements -> bytes (unaligned)
  const unsigned base = static_cast<unsigned>(reinterpret_cast<uintptr_t>(lds));     // LDS aperture offset
  uint2 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(base + addr) : "memory");
  out[l * 4 + 0] = v.x & 0xffff; out[l * 4 + 1] = v.x >> 16; out[l * 4 + 2] = v.y & 0xffff; out[l * 4 + 3] = v.y >> 16;
}

int main() {
  uint16_t *d, h[256];
  hipMalloc(&d, sizeof(h));
  for (int p = 0; p < 5; ++p) {
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, p, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("pattern %d\n", p);
    for (int l = 0; l < 64; ++l) printf("  lane %2d: %5d %5d %5d %5d%s", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3], (l & 3) == 3 ? "\n" : "");
  }
  return 0;
}
