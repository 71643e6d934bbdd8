// Probe (round 5): how fast does the chip take the OUTPUT STREAM of the hoisted value projection (184,950 rows x 1536 fp32
// columns, written as six (M, 256) matrices) in three lane -> address patterns, nothing else going on?
//   A  the row-panel kernel's epilogue today (transposed MFMA tile): lane l owns row l & 31 and 4 consecutive columns
//      4 (l >> 5) + 8 g per store g = 0..3 — an instruction touches 32 rows x 2 pieces of 16 bytes; a 128-byte line is
//      completed by FOUR instructions;
//   B  whole lines: 8 consecutive lanes cover one 128-byte line (row (l >> 3) + 8 g, columns 4 (l & 7)), 8 lines per
//      instruction — what a transpose of the tile through LDS would give;
//   C  the untransposed MFMA tile: lane l owns column l & 31, rows (r & 3) + 8 (r >> 2) + 4 (l >> 5): 16 dword stores, each
//      writing two whole 128-byte lines.
// Same workgroup geometry as linear_panel_kernel<3, 4, 1, 8>: 512 threads own 128 rows, wavefront w the column tiles w, w + 8 ..
//   hipcc --offload-arch=gfx950 -O3 -o store_pattern_probe tools/probes/store_pattern_probe.hip && ./store_pattern_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int PATTERN>
__global__ void __launch_bounds__(512) probe(float *y, long M, int N, int gcols) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long m0 = static_cast<long>(blockIdx.x) * 128;
  const int nct = N / 32;
  float v = static_cast<float>(threadIdx.x);
  for (int ct = wave; ct < nct; ct += 8) {
    const int n0 = ct * 32;
    const int grp = n0 / gcols;
    float *yg = y + static_cast<long>(grp) * M * gcols - grp * gcols;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (PATTERN == 0) {
        const long m = m0 + i * 32 + (lane & 31);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int n = n0 + 4 * (lane >> 5) + 8 * g;
          if (m < M) *reinterpret_cast<float4 *>(yg + m * gcols + n) = make_float4(v, v + 1.f, v + 2.f, v + 3.f);
        }
      } else if (PATTERN == 1) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const long m = m0 + i * 32 + (lane >> 3) + 8 * g;
          const int n = n0 + 4 * (lane & 7);
          if (m < M) *reinterpret_cast<float4 *>(yg + m * gcols + n) = make_float4(v, v + 1.f, v + 2.f, v + 3.f);
        }
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const long m = m0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
          const int n = n0 + (lane & 31);
          if (m < M) yg[m * gcols + n] = v + r;
        }
      }
      v += 1.f;
    }
  }
}

int main() {
  const long M = 184950;
  const int N = 1536, gcols = 256;
  float *y;
  hipMalloc(&y, sizeof(float) * M * N);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const dim3 grid((M + 127) / 128);
  const char *names[3] = {"A transposed-tile pieces (today)", "B whole 128-byte lines", "C dword rows (untransposed tile)"};
  for (int rep = 0; rep < 3; ++rep)
    for (int p = 0; p < 3; ++p) {
      float ms = 0.f, best = 1e9f;
      for (int it = 0; it < 6; ++it) {
        hipEventRecord(e0);
        if (p == 0) hipLaunchKernelGGL(probe<0>, grid, dim3(512), 0, 0, y, M, N, gcols);
        else if (p == 1) hipLaunchKernelGGL(probe<1>, grid, dim3(512), 0, 0, y, M, N, gcols);
        else hipLaunchKernelGGL(probe<2>, grid, dim3(512), 0, 0, y, M, N, gcols);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        if (it > 0 && ms < best) best = ms;
      }
      printf("%-36s %8.1f us  %6.2f TB/s\n", names[p], best * 1e3, 4.0 * M * N / (best * 1e-3) / 1e12);
    }
  return 0;
}
