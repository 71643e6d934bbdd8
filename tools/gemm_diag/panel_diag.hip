// tools/gemm_diag: the row-panel projection kernel (csrc/linear_panel.h) with parts switched off at run time
#define BEVMSDA_PANEL_DIAG 1
#include "../../include/bevmsda.h"
#include "../../bevformer_amd/csrc/linear_panel.h"

template <int WD>
static void launch_wd(const bevmsda::PanelArgs &a, int nprod, int shape, dim3 grid, hipStream_t st) {
  if (nprod == 3) {
    if (shape == 1) hipLaunchKernelGGL((bevmsda::linear_panel_kernel<3, 2, 2, 4, false, 0, 0, 0, false, WD>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((bevmsda::linear_panel_kernel<3, 4, 1, 8, false, 0, 0, 0, false, WD>), grid, dim3(512), 0, st, a);
  } else {
    if (shape == 1) hipLaunchKernelGGL((bevmsda::linear_panel_kernel<1, 2, 2, 4, false, 0, 0, 0, false, WD>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((bevmsda::linear_panel_kernel<1, 4, 1, 8, false, 0, 0, 0, false, WD>), grid, dim3(512), 0, st, a);
  }
}

template <int ST, int LD, bool DRIP = false>
static void launch(const bevmsda::PanelArgs &a, int nprod, int shape, dim3 grid, hipStream_t st) {
  if (shape == 3) {
    if (nprod == 3) hipLaunchKernelGGL((bevmsda::linear_panel_kernel<3, 2, 1, 4, false, 0, ST, LD, DRIP>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((bevmsda::linear_panel_kernel<1, 2, 1, 4, false, 0, ST, LD, DRIP>), grid, dim3(256), 0, st, a);
    return;
  }
  if (nprod == 3) {
    if (shape == 1) hipLaunchKernelGGL((bevmsda::linear_panel_kernel<3, 2, 2, 4, false, 0, ST, LD, DRIP>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((bevmsda::linear_panel_kernel<3, 4, 1, 8, false, 0, ST, LD, DRIP>), grid, dim3(512), 0, st, a);
  } else {
    if (shape == 1) hipLaunchKernelGGL((bevmsda::linear_panel_kernel<1, 2, 2, 4, false, 0, ST, LD, DRIP>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((bevmsda::linear_panel_kernel<1, 4, 1, 8, false, 0, ST, LD, DRIP>), grid, dim3(512), 0, st, a);
  }
}

// policy: store bits + 100 * panel-fetch bits
extern "C" int diag_panel_policy(const float *x, long ldx, const uint16_t *wp, unsigned wp_bytes, const float *bias, float *y, long ldy,
                                 long M, int N, int K, int group_cols, int nprod, int shape, int mask, int policy, void *stream) {
  bevmsda::PanelArgs a{};
  a.x0 = x; a.ldx0 = ldx; a.wp = wp; a.wp_bytes = wp_bytes; a.bias = bias; a.y = y; a.ldy = ldy; a.M = M; a.N = N;
  a.K0 = K; a.K1 = 0; a.group_cols = group_cols; a.diag = mask;
  const int bm = shape == 2 ? 128 : 64;
  const dim3 grid(static_cast<unsigned>((M + bm - 1) / bm));
  hipStream_t st = static_cast<hipStream_t>(stream);
  switch (policy) {
    case 0: launch<0, 0>(a, nprod, shape, grid, st); break;
    case 2: launch<2, 0>(a, nprod, shape, grid, st); break;
    case 16: launch<16, 0>(a, nprod, shape, grid, st); break;
    case 18: launch<18, 0>(a, nprod, shape, grid, st); break;
    case 200: launch<0, 2>(a, nprod, shape, grid, st); break;
    case 216: launch<16, 2>(a, nprod, shape, grid, st); break;
    case 202: launch<2, 2>(a, nprod, shape, grid, st); break;
    case 3003: launch_wd<3>(a, nprod, shape, grid, st); break;
    case 3004: launch_wd<4>(a, nprod, shape, grid, st); break;
    case 3006: launch_wd<6>(a, nprod, shape, grid, st); break;
    case 1000: launch<0, 0, true>(a, nprod, shape, grid, st); break;
    case 1200: launch<0, 2, true>(a, nprod, shape, grid, st); break;
    default: return -2;
  }
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

extern "C" int diag_panel(const float *x, long ldx, const uint16_t *wp, unsigned wp_bytes, const float *bias, float *y, long ldy,
                          long M, int N, int K, int group_cols, int nprod, int shape, int mask, void *stream) {
  bevmsda::PanelArgs a{};
  a.x0 = x; a.ldx0 = ldx; a.wp = wp; a.wp_bytes = wp_bytes; a.bias = bias; a.y = y; a.ldy = ldy; a.M = M; a.N = N;
  a.K0 = K; a.K1 = 0; a.group_cols = group_cols; a.diag = mask;
  const int bm = shape == 2 ? 128 : 64;
  const dim3 grid(static_cast<unsigned>((M + bm - 1) / bm));
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (nprod == 3) {
    if (shape == 1) hipLaunchKernelGGL((bevmsda::linear_panel_kernel<3, 2, 2, 4, false, 0>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((bevmsda::linear_panel_kernel<3, 4, 1, 8, false, 0>), grid, dim3(512), 0, st, a);
  } else {
    if (shape == 1) hipLaunchKernelGGL((bevmsda::linear_panel_kernel<1, 2, 2, 4, false, 0>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((bevmsda::linear_panel_kernel<1, 4, 1, 8, false, 0>), grid, dim3(512), 0, st, a);
  }
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

// per-phase clocks of the production instantiation (PANEL_CLK): prof = 8 zeroed 64-bit counters on the device
extern "C" int diag_panel_phases(const float *x, long ldx, const uint16_t *wp, unsigned wp_bytes, const float *bias, float *y, long ldy,
                                 long M, int N, int K, int group_cols, int nprod, int shape, unsigned long long *prof, void *stream) {
  bevmsda::PanelArgs a{};
  a.x0 = x; a.ldx0 = ldx; a.wp = wp; a.wp_bytes = wp_bytes; a.bias = bias; a.y = y; a.ldy = ldy; a.M = M; a.N = N;
  a.K0 = K; a.K1 = 0; a.group_cols = group_cols; a.diag = 0; a.prof = prof;
  const int bm = shape == 2 ? 128 : 64;
  const dim3 grid(static_cast<unsigned>((M + bm - 1) / bm));
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (nprod == 3) {
    if (shape == 1) hipLaunchKernelGGL((bevmsda::linear_panel_kernel<3, 2, 2, 4, false, 0>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((bevmsda::linear_panel_kernel<3, 4, 1, 8, false, 0>), grid, dim3(512), 0, st, a);
  } else {
    if (shape == 1) hipLaunchKernelGGL((bevmsda::linear_panel_kernel<1, 2, 2, 4, false, 0>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((bevmsda::linear_panel_kernel<1, 4, 1, 8, false, 0>), grid, dim3(512), 0, st, a);
  }
  return hipGetLastError() == hipSuccess ? 0 : -1;
}
