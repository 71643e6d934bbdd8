// Probe (round 6): is the 64 B / clk / CU that bounds the sampling kernels a property of the vector L1's RETURN path to the
// register file only, or of the L1 itself?  The sampling kernels gather 128-byte lines (one bilinear tap of one head: 8 lanes
// x 16 bytes) at data-dependent addresses; the SCA forward runs at ~73 % of 64 B / clk / CU whether the lines hit in the L1
// or bypass it (sc1: +3 %, profiles/r6/r6m_tap_aux_ab.txt).  gfx950 can also land a load in LDS (global_load_lds_dwordx4: lane
// i's 16 bytes go to LDS[M0 + 16 i]) from where ds_read_b128 moves 256 B / clk.  Three kernels over the same pseudo-random
// line addresses, same occupancy (256 threads, 4 workgroups per CU), 16 loads in flight per wavefront:
//   V  every tap through a VGPR  (buffer of `lines` 128-byte lines: 128 = L1-resident, 16 Ki = L2-resident, 4 Mi = HBM)
//   L  every tap through LDS-DMA, then one ds_read_b128 per lane
//   H  alternating: half the taps by either path (do the two paths add up?)
//   hipcc --offload-arch=gfx950 -O3 -o gather_path_probe tools/probes/gather_path_probe.hip && ./gather_path_probe
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

template <int MODE>   // 0 = V, 1 = L, 2 = H
__global__ void __launch_bounds__(256) probe(const float *buf, uint32_t mask, float *out) {
  extern __shared__ __attribute__((aligned(16))) float lds[];          // 4 wavefronts x U x 1 KB
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int grp = threadIdx.x >> 3, j = threadIdx.x & 7;
  uint32_t s = (blockIdx.x * 32u + grp) * 2654435761u + 12345u;        // one stream per 8-lane group
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  const char *base = reinterpret_cast<const char *>(buf);
  float *my = lds + wave * (U * 256);
  const uint32_t my_byte = static_cast<uint32_t>(reinterpret_cast<size_t>((__attribute__((address_space(3))) float *)my));
  for (int it = 0; it < ITERS; ++it) {
    f32x4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const uint32_t line = next_line(s, mask);
      const char *p = base + (static_cast<size_t>(line) << 7) + j * 16;
      const bool via_lds = MODE == 1 || (MODE == 2 && (u & 1));
      if (via_lds) dma16(p, my_byte + u * 1024);
      else v[u] = *reinterpret_cast<const f32x4 *>(p);
    }
    if (MODE != 0) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
      for (int u = 0; u < U; ++u)
        if (MODE == 1 || (u & 1)) v[u] = *reinterpret_cast<const f32x4 *>(my + u * 256 + lane * 4);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) acc += v[u];
  }
  if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) out[threadIdx.x] = acc[0];
}

int main() {
  const size_t max_lines = 4u << 20;
  float *buf, *out;
  hipMalloc(&buf, max_lines * 128);
  hipMalloc(&out, 4096);
  hipMemset(buf, 0, max_lines * 128);
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, 0);
  const int cus = prop.multiProcessorCount;
  const double clk = prop.clockRate * 1e3;   // Hz (peak engine clock)
  const int blocks = cus * 4 * 8;            // 8 rounds of 4 workgroups per CU
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const size_t lds_bytes = 4 * U * 1024;
  printf("CUs %d, peak clock %.0f MHz, %d workgroups x 256 threads, %d taps of 128 bytes per 8-lane group\n", cus, clk / 1e6, blocks, U * ITERS);
  const uint32_t sizes[3] = {128, 16u << 10, 4u << 20};
  const char *names[3] = {"16 KB (L1-resident)", "2 MB (L2-resident)", "512 MB (HBM)"};
  for (int si = 0; si < 3; ++si) {
    for (int mode = 0; mode < 3; ++mode) {
      float best = 1e30f;
      for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0);
        if (mode == 0) hipLaunchKernelGGL(probe<0>, dim3(blocks), dim3(256), lds_bytes, 0, buf, sizes[si] - 1, out);
        if (mode == 1) hipLaunchKernelGGL(probe<1>, dim3(blocks), dim3(256), lds_bytes, 0, buf, sizes[si] - 1, out);
        if (mode == 2) hipLaunchKernelGGL(probe<2>, dim3(blocks), dim3(256), lds_bytes, 0, buf, sizes[si] - 1, out);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (rep > 0 && ms < best) best = ms;
      }
      const double bytes = static_cast<double>(blocks) * 32 * U * ITERS * 128;
      printf("%-22s %s: %8.1f us  %7.2f TB/s  %5.1f B/clk/CU at the peak clock\n", names[si],
             mode == 0 ? "V (VGPR)   " : (mode == 1 ? "L (LDS-DMA)" : "H (half)   "), best * 1e3, bytes / (best * 1e-3) / 1e12,
             bytes / (best * 1e-3) / clk / cus);
    }
  }
  if (hipGetLastError() != hipSuccess) { printf("HIP error\n"); return 1; }
  return 0;
}
