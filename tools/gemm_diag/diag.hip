// Diagnostic build of the projection kernel (tools only, never part of libbevmsda.so): the production
// instantiation with parts of its loop switched off at run time, to see which resource bounds it.
#define BEVMSDA_LIN_DIAG 1
#include "../../bevformer_amd/csrc/linear_mfma.h"

extern "C" int diag_linear(const float *x, long ldx, const uint16_t *wpack, const float *bias, float *y, long ldy,
                           long M, int N, int K, int group_cols, int nprod, int diag, void *stream) {
  bevmsda::LinArgs a{};
  a.x0 = x; a.ldx0 = ldx; a.wpack = wpack; a.bias = bias; a.y = y; a.ldy = ldy; a.M = M; a.N = N; a.K0 = K; a.K1 = 0;
  a.group_cols = group_cols; a.diag = diag;
  const long long nbm = (M + 127) / 128, nbn = (N + 127) / 128;
  a.nblk_m = static_cast<int>(nbm); a.nblk_n = static_cast<int>(nbn);
  const dim3 g(static_cast<unsigned>(((nbm + 7) / 8) * 8 * nbn)), b(256);
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (nprod == 3) hipLaunchKernelGGL((bevmsda::linear_splitbf16_kernel<3, false, 32, true, 3, 128>), g, b, 0, st, a);
  else hipLaunchKernelGGL((bevmsda::linear_splitbf16_kernel<1, false, 32, true, 3, 128>), g, b, 0, st, a);
  return hipGetLastError() == hipSuccess ? 0 : 1;
}
