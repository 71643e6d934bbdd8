// Probe (round 6, after gather_path_probe): the SCA sampling kernel runs at 68 % of what a bare gather of random 128-byte
// lines reaches (60.5 B / clk / CU).  Is the rest the ADDRESS PATTERN of bilinear taps or the kernel around the loads?
// Same skeleton as gather_path_probe (256 threads, 4 workgroups per CU, an 8-lane group per (row, head), 16 tap loads in
// flight per wavefront, nothing but the loads and one add per tap), three address generators over a 4-level pyramid of one
// camera (200 x 116, 100 x 58, 50 x 29, 25 x 15 pixels, 8 heads x 32 fp32 channels per pixel):
//   R  random lines of the pyramid                                            (the reference point: what gather_path_probe measured)
//   P  the four taps of a point, value laid out as the reference has it, (pixel, head, channel): x-neighbours 1 KB apart
//   H  the same taps with the value laid out head-major, (head, pixel, channel): x-neighbours are adjacent 128-byte lines
// each with the points of a workgroup scattered over the whole map ("far") or within +-8 pixels of a per-workgroup centre
// ("near": what image-ordered rows look like to the L1).
//   hipcc --offload-arch=gfx950 -O3 -o tap_pattern_probe tools/probes/tap_pattern_probe.hip && ./tap_pattern_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int PTS = 4;         // points per batch: 16 taps in flight per wavefront instruction stream
constexpr int ITERS = 64;
constexpr int kS = 30825;

__device__ __forceinline__ uint32_t rnd(uint32_t &s) {
  s = s * 1664525u + 1013904223u;
  return s >> 8;
}
// uniform in [0, n) from a 24-bit draw: one v_mul_hi_u32 (an integer modulo would make the probe VALU-bound)
__device__ __forceinline__ uint32_t below(uint32_t &s, uint32_t n) { return __umulhi(rnd(s) << 8, n); }

template <int MODE, bool NEAR>   // 0 = R, 1 = P, 2 = H
__global__ void __launch_bounds__(256) probe(const float *buf, float *out) {
  const int grp = threadIdx.x >> 3, j = threadIdx.x & 7;
  const uint32_t m = grp & 7;
  uint32_t s = (blockIdx.x * 32u + grp) * 2654435761u + 12345u;
  uint32_t sw = blockIdx.x * 747796405u + 2891336453u;                 // per-workgroup stream: the centre of its points
  const float cx = (rnd(sw) & 0xffff) * (1.f / 65536.f), cy = (rnd(sw) & 0xffff) * (1.f / 65536.f);
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  const char *base = reinterpret_cast<const char *>(buf);
  for (int it = 0; it < ITERS; ++it) {
    f32x4 v[PTS][4];
    const int l = it & 3;
    const int W = 200 >> l, H = (116 >> l) + (l == 3);                  // 200 x 116, 100 x 58, 50 x 29, 25 x 15
    const int start = l == 0 ? 0 : (l == 1 ? 23200 : (l == 2 ? 29000 : 30450));
#pragma unroll
    for (int p = 0; p < PTS; ++p) {
      uint32_t line[4];
      if (MODE == 0) {
#pragma unroll
        for (int t = 0; t < 4; ++t) line[t] = below(s, kS * 8u);
      } else {
        int x0, y0;
        if (NEAR) {
          x0 = static_cast<int>(cx * (W - 18)) + static_cast<int>(rnd(s) & 15);
          y0 = static_cast<int>(cy * (H - 18 > 0 ? H - 18 : 0)) + static_cast<int>(below(s, H < 17 ? H - 1 : 16));
        } else {
          x0 = static_cast<int>(below(s, W - 1));
          y0 = static_cast<int>(below(s, H - 1));
        }
        const uint32_t pix = start + y0 * W + x0;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const uint32_t q = pix + (t & 1) + (t >> 1) * W;
          line[t] = MODE == 1 ? q * 8u + m : m * kS + q;
        }
      }
#pragma unroll
      for (int t = 0; t < 4; ++t) v[p][t] = *reinterpret_cast<const f32x4 *>(base + (static_cast<size_t>(line[t]) << 7) + j * 16);
    }
#pragma unroll
    for (int p = 0; p < PTS; ++p)
#pragma unroll
      for (int t = 0; t < 4; ++t) acc += v[p][t];
  }
  if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) out[threadIdx.x] = acc[0];
}

int main() {
  float *buf, *out;
  const size_t bytes_buf = static_cast<size_t>(kS) * 8 * 128;
  hipMalloc(&buf, bytes_buf);
  hipMalloc(&out, 4096);
  hipMemset(buf, 0, bytes_buf);
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, 0);
  const int cus = prop.multiProcessorCount;
  const double clk = prop.clockRate * 1e3;
  const int blocks = cus * 4 * 8;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  printf("CUs %d, peak clock %.0f MHz, %d workgroups x 256 threads, %d taps per 8-lane group; one camera's pyramid = %.1f MB\n", cus, clk / 1e6,
         blocks, PTS * 4 * ITERS, bytes_buf / 1e6);
  const char *names[6] = {"R random lines          ", "P (pixel, head) far     ", "P (pixel, head) near    ", "H (head, pixel) far     ",
                          "H (head, pixel) near    ", "R random lines (again)  "};
  for (int mode = 0; mode < 6; ++mode) {
    float best = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {
      hipEventRecord(e0);
      switch (mode) {
        case 0: case 5: hipLaunchKernelGGL((probe<0, false>), dim3(blocks), dim3(256), 0, 0, buf, out); break;
        case 1: hipLaunchKernelGGL((probe<1, false>), dim3(blocks), dim3(256), 0, 0, buf, out); break;
        case 2: hipLaunchKernelGGL((probe<1, true>), dim3(blocks), dim3(256), 0, 0, buf, out); break;
        case 3: hipLaunchKernelGGL((probe<2, false>), dim3(blocks), dim3(256), 0, 0, buf, out); break;
        case 4: hipLaunchKernelGGL((probe<2, true>), dim3(blocks), dim3(256), 0, 0, buf, out); break;
      }
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      if (rep > 0 && ms < best) best = ms;
    }
    const double bytes = static_cast<double>(blocks) * 32 * PTS * 4 * ITERS * 128;
    printf("%s: %8.1f us  %7.2f TB/s  %5.1f B/clk/CU at the peak clock\n", names[mode], best * 1e3, bytes / (best * 1e-3) / 1e12,
           bytes / (best * 1e-3) / clk / cus);
  }
  if (hipGetLastError() != hipSuccess) { printf("HIP error\n"); return 1; }
  return 0;
}
