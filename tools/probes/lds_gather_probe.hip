// Probe (round 6): what would a sampling kernel gain if the feature-map tile its taps fall into were resident in LDS?
// The north star names "feature maps staged through LDS tiles"; two LDS-staged forms of the SCA kernel lost in rounds 1-2
// (DESIGN K1-LDS) to their address arithmetic and occupancy, and gather_path_probe.hip showed that a tap line reaches the
// lanes at 60.5 B / clk / CU from the vector L1 whatever the landing path.  This probe measures the OTHER side: the rate at
// which 8-lane groups pull 128-byte lines (one bilinear tap of one head) at data-dependent addresses out of an LDS-resident
// region with ds_read_b128 — the ceiling of any design that re-uses staged lines — next to the same gather from the L1, and
// what is left of it when the region has to be re-staged (LDS-DMA) every `reuse` taps per staged line.
//   R  region of REGION_KB in LDS, staged once per workgroup, then U x ITERS taps per 8-lane group from it
//   S  the same, the region re-staged so that every staged line serves `reuse` taps (TSA at a 16 x 8 tile: ~4, 16 x 16: ~6)
//   V  the taps from global memory (L1-resident 16 KB buffer): the reference path of gather_path_probe.hip
//   hipcc --offload-arch=gfx950 -O3 -o lds_gather_probe tools/probes/lds_gather_probe.hip && ./lds_gather_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int U = 16;          // taps in flight per wavefront
constexpr int ITERS = 64;      // batches of U per wavefront

__device__ __forceinline__ uint32_t next_line(uint32_t &s, uint32_t mask) {
  s = s * 1664525u + 1013904223u;
  return (s >> 8) & mask;
}

__device__ __forceinline__ void dma16(const void *src, uint32_t lds_byte) {
  const unsigned d = __builtin_amdgcn_readfirstlane(lds_byte);
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(src), "s"(d) : "memory", "m0");
}

// MODE 0: R / S (restage_every = 0: staged once), MODE 1: V
template <int MODE, int NTHREADS>
__global__ void __launch_bounds__(NTHREADS) probe(const float *buf, uint32_t region_lines, int restage_every, float *out) {
  extern __shared__ __attribute__((aligned(16))) float lds[];          // region_lines x 128 bytes
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int grp = threadIdx.x >> 3, j = threadIdx.x & 7;
  constexpr int NW = NTHREADS / 64;
  uint32_t s = (blockIdx.x * (NTHREADS / 8) + grp) * 2654435761u + 12345u;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  const char *base = reinterpret_cast<const char *>(buf);
  const uint32_t lds0 = static_cast<uint32_t>(reinterpret_cast<size_t>((__attribute__((address_space(3))) float *)lds));
  const uint32_t mask = region_lines - 1;
  auto stage = [&]() {                       // the region, 1 KB (8 lines) per wavefront instruction
    for (uint32_t l = wave * 8; l < region_lines; l += NW * 8)
      dma16(base + (static_cast<size_t>(l + (lane >> 3)) << 7) + j * 16, lds0 + l * 128);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  };
  if (MODE == 0) stage();
  for (int it = 0; it < ITERS; ++it) {
    if (MODE == 0 && restage_every > 0 && it > 0 && it % restage_every == 0) {
      __syncthreads();
      stage();
    }
    f32x4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const uint32_t line = next_line(s, mask);
      if (MODE == 0) v[u] = *reinterpret_cast<const f32x4 *>(lds + line * 32 + j * 4);
      else v[u] = *reinterpret_cast<const f32x4 *>(base + (static_cast<size_t>(line) << 7) + j * 16);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) acc += v[u];
  }
  if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) out[threadIdx.x] = acc[0];
}

template <int MODE, int NTHREADS>
static float run(int blocks, size_t lds_bytes, const float *buf, uint32_t lines, int restage, float *out) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e30f;
  hipFuncSetAttribute(reinterpret_cast<const void *>(probe<MODE, NTHREADS>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  for (int rep = 0; rep < 5; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((probe<MODE, NTHREADS>), dim3(blocks), dim3(NTHREADS), lds_bytes, 0, buf, lines, restage, out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (rep > 0 && ms < best) best = ms;
  }
  return best;
}

int main() {
  float *buf, *out;
  hipMalloc(&buf, 1u << 20);
  hipMalloc(&out, 4096);
  hipMemset(buf, 0, 1u << 20);
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, 0);
  const int cus = prop.multiProcessorCount;
  const double clk = prop.clockRate * 1e3;
  printf("CUs %d, peak clock %.0f MHz; %d taps of 128 bytes per 8-lane group, 16 in flight per wavefront\n", cus, clk / 1e6, U * ITERS);
  auto report = [&](const char *what, float ms, int blocks, int nthreads) {
    const double bytes = static_cast<double>(blocks) * (nthreads / 8) * U * ITERS * 128;
    printf("%-78s %8.1f us  %7.2f TB/s  %6.1f B/clk/CU\n", what, ms * 1e3, bytes / (ms * 1e-3) / 1e12, bytes / (ms * 1e-3) / clk / cus);
  };
  // V: from the L1 (16 KB buffer), 4 x 256 threads per CU
  report("V  taps from the vector L1 (16 KB buffer), 4 x 256 threads per CU", run<1, 256>(cus * 4 * 8, 0, buf, 128, 0, out), cus * 4 * 8, 256);
  // R: region in LDS, staged once
  report("R  64 KB region in LDS, staged once, 2 x 256 threads per CU", run<0, 256>(cus * 2 * 8, 64 << 10, buf, 512, 0, out), cus * 2 * 8, 256);
  report("R  64 KB region in LDS, staged once, 2 x 512 threads per CU", run<0, 512>(cus * 2 * 8, 64 << 10, buf, 512, 0, out), cus * 2 * 8, 512);
  report("R  128 KB region in LDS, staged once, 1 x 512 threads per CU", run<0, 512>(cus * 8, 128 << 10, buf, 1024, 0, out), cus * 8, 512);
  report("R  128 KB region in LDS, staged once, 1 x 1024 threads per CU", run<0, 1024>(cus * 8, 128 << 10, buf, 1024, 0, out), cus * 8, 1024);
  // S: re-staged: a 64 KB region = 512 lines; a workgroup of 512 threads pulls 64 groups x 16 = 1,024 taps per iteration, so
  // restaging every k iterations gives every staged line 2 k taps
  for (int k : {1, 2, 3, 4, 8}) {
    char what[128];
    snprintf(what, sizeof(what), "S  64 KB region re-staged every %d iterations (%d taps per staged line), 2 x 512 threads per CU", k, 2 * k);
    report(what, run<0, 512>(cus * 2 * 8, 64 << 10, buf, 512, k, out), cus * 2 * 8, 512);
  }
  if (hipGetLastError() != hipSuccess) { printf("HIP error\n"); return 1; }
  return 0;
}
