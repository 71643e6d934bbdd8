// Probe: fp32 atomic-add throughput on MI355X by memory scope and access shape.
// Build: hipcc --offload-arch=gfx950 -O3 -o atomic_probe atomic_probe.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

template <int SCOPE>
__device__ __forceinline__ void add(float *p, float v) {
  if (SCOPE == 0) unsafeAtomicAdd(p, v);  // device scope, no return
  else if (SCOPE == 1) __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  else if (SCOPE == 2) __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else if (SCOPE == 3) __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
  else *p += v;  // plain RMW (racy) as an upper bound
}

__device__ __forceinline__ uint32_t hash(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x;
}

// each 8-lane group picks a pseudo-random 128-byte line; lane j adds to floats 4j..4j+3
template <int SCOPE, bool PRIVATE>
__global__ void __launch_bounds__(256) k(float *buf, size_t lines_per_copy, int iters) {
  uint32_t xcc = 0;
  if (PRIVATE) { asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc)); xcc &= 7; }
  float *base = buf + (size_t)xcc * lines_per_copy * 32;
  const uint32_t gid = (blockIdx.x * 256 + threadIdx.x) >> 3;
  const int lig = threadIdx.x & 7;
  for (int i = 0; i < iters; ++i) {
    const size_t line = hash(gid * 977u + i) % lines_per_copy;
    float *p = base + line * 32 + lig * 4;
#pragma unroll
    for (int c = 0; c < 4; ++c) add<SCOPE>(p + c, 1.0f);
  }
}

// shape B: 32 consecutive lanes cover one 128-byte line (lane j -> float j), 4 lines per lane
template <int SCOPE, bool PRIVATE>
__global__ void __launch_bounds__(256) k32(float *buf, size_t lines_per_copy, int iters) {
  uint32_t xcc = 0;
  if (PRIVATE) { asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc)); xcc &= 7; }
  float *base = buf + (size_t)xcc * lines_per_copy * 32;
  const uint32_t gid = (blockIdx.x * 256 + threadIdx.x) >> 5;
  const int lig = threadIdx.x & 31;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const size_t line = hash(gid * 977u + i * 4 + c) % lines_per_copy;
      add<SCOPE>(base + line * 32 + lig, 1.0f);
    }
  }
}

template <int SCOPE, bool PRIVATE>
void run32(const char *name, float *buf, size_t lines, int blocks, int iters) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  k32<SCOPE, PRIVATE><<<blocks, 256>>>(buf, lines, 2);
  hipDeviceSynchronize();
  hipEventRecord(a);
  k32<SCOPE, PRIVATE><<<blocks, 256>>>(buf, lines, iters);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  const double n = (double)blocks * 256 * iters * 4;
  printf("%-34s lines=%9zu  %8.3f ms  %8.1f G atomics/s\n", name, lines, ms, n / ms / 1e6);
}

template <int SCOPE, bool PRIVATE>
void run(const char *name, float *buf, size_t lines, int blocks, int iters) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  k<SCOPE, PRIVATE><<<blocks, 256>>>(buf, lines, 2);
  hipDeviceSynchronize();
  hipEventRecord(a);
  k<SCOPE, PRIVATE><<<blocks, 256>>>(buf, lines, iters);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  const double n = (double)blocks * 256 * iters * 4;
  printf("%-34s lines=%9zu  %8.3f ms  %8.1f G atomics/s\n", name, lines, ms, n / ms / 1e6);
}

int main() {
  const size_t lines = (189u << 20) / 128;  // one 189 MB gradient buffer
  float *buf; hipMalloc(&buf, lines * 128 * 8);
  hipMemset(buf, 0, lines * 128 * 8);
  const int blocks = 256 * 16, iters = 64;
  run<0, false>("device unsafeAtomicAdd", buf, lines, blocks, iters);
  run<2, false>("agent scope fetch_add", buf, lines, blocks, iters);
  run<1, false>("workgroup scope, shared buf (!)", buf, lines, blocks, iters);
  run<1, true>("workgroup scope, per-XCD copy", buf, lines, blocks, iters);
  run<3, true>("wavefront scope, per-XCD copy", buf, lines, blocks, iters);
  run<4, true>("plain RMW (racy bound)", buf, lines, blocks, iters);
  run<0, false>("device, 3 MB hot set", buf, (3u << 20) / 128, blocks, iters);
  run<1, true>("workgroup per-XCD, 3 MB hot set", buf, (3u << 20) / 128, blocks, iters);
  run32<0, false>("lane=dword shape, device", buf, lines, blocks, iters);
  run32<1, true>("lane=dword shape, per-XCD copy", buf, lines, blocks, iters);
  run32<4, true>("lane=dword shape, plain RMW", buf, lines, blocks, iters);
  // correctness of the per-XCD scheme: total must equal the number of adds
  hipMemset(buf, 0, lines * 128 * 8);
  k<1, true><<<blocks, 256>>>(buf, (3u << 20) / 128, 8);
  hipDeviceSynchronize();
  std::vector<float> h((3u << 20) / 4 * 8);
  double tot = 0;
  for (int x = 0; x < 8; ++x) {
    hipMemcpy(h.data(), buf + (size_t)x * lines * 32, (3u << 20), hipMemcpyDeviceToHost);
    for (size_t i = 0; i < (3u << 20) / 4; ++i) tot += h[i];
  }
  printf("per-XCD sum check: %.0f expected %.0f\n", tot, (double)blocks * 256 * 8 * 4);
  return 0;
}
