// tools/gemm_diag: phase clocks of the row-chain kernel (csrc/linear_chain.h)
#define BEVMSDA_CHAIN_PROF 1
#include "../../include/bevmsda.h"
#include "../../bevformer_amd/csrc/linear_chain.h"
#include "../experimental/linear_rowreg.h"   // (retired from the library: includes "linear_chain.h" from csrc via -I)

extern "C" int diag_chain(const float *rows, long ld_rows, const int32_t *idx, const float *scale, const uint16_t *w0, const float *b0,
                          const float *res, const float *g0, const float *be0, const uint16_t *w1, const float *b1, const uint16_t *w2,
                          const float *b2, const float *g1, const float *be1, long M, float *y, unsigned long long *prof, int shape, void *stream) {
  bevmsda::ChainArgs a{};
  a.rows = rows; a.ld_rows = ld_rows; a.gidx = idx; a.gscale = scale; a.w0 = w0; a.w1 = w1; a.w2 = w2; a.b0 = b0; a.b1 = b1; a.b2 = b2;
  a.res = res; a.ld_res = 256; a.gamma0 = g0; a.beta0 = be0; a.gamma1 = g1; a.beta1 = be1; a.eps0 = a.eps1 = 1e-5f;
  a.y = y; a.ld_y = 256; a.M = M; a.prof = prof;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (shape >= 30) {                // rows resident in registers (weights: the rowreg images); shape = 30 + stop phase
    a.ld_y2 = shape - 30;
    const dim3 grid(static_cast<unsigned>((M + 127) / 128)), block(256);
    if (idx) hipLaunchKernelGGL((bevmsda::linear_rowreg_chain_kernel<3, 2, 0>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((bevmsda::linear_rowreg_chain_kernel<3, 0, 0>), grid, block, 0, st, a);
  } else if (shape == 1) {
    const dim3 grid(static_cast<unsigned>((M + 63) / 64)), block(512);
    if (idx) hipLaunchKernelGGL((bevmsda::linear_chain_kernel<3, 2, 0, 2, 1, 8>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((bevmsda::linear_chain_kernel<3, 0, 0, 2, 1, 8>), grid, block, 0, st, a);
  } else {
    const dim3 grid(static_cast<unsigned>((M + 31) / 32)), block(256);
    if (idx) hipLaunchKernelGGL((bevmsda::linear_chain_kernel<3, 2, 0, 1, 2, 4>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((bevmsda::linear_chain_kernel<3, 0, 0, 1, 2, 4>), grid, block, 0, st, a);
  }
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

extern "C" void diag_rowreg_pack(const float *w, long ldw, int N, int K, int kmajor, uint16_t *out, void *stream) {
  const long total = static_cast<long>(N / 32) * (K / 16) * 64;
  hipLaunchKernelGGL(bevmsda::lin_rowreg_pack_weight_kernel, dim3(static_cast<unsigned>((total + 255) / 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), w, ldw, N, K, kmajor, out);
}
