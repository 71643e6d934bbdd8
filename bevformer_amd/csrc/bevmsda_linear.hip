// C ABI of libbevmsda.so, dense projections (declared in include/bevmsda.h): argument checks and
// launches of the MFMA projection kernels (linear_mfma.h, linear_pipe.h, linear_panel.h, wgrad_mfma.h).  No torch, no allocation, no global state.
#include "../../include/bevmsda.h"
#include "linear_mfma.h"
#include "linear_pipe.h"
#include "linear_panel.h"
#include "linear_chain.h"
#include "wgrad_mfma.h"
#include "wgrad_tr.h"

namespace {
constexpr bool kLinearPipeDefault = false;       // linear_pipe.h (software-pipelined) as the default where it applies
constexpr long long kLinearPipeMaxRows = 8192;   // ... and always for M <= this
// linear_chain.h workgroup shape by measurement (tools/chain_small_m.py, profiles/r3): 32-row panels up to this many
// rows (a rank's tile of a BEV-tiled frame: 5,000 rows 20 vs 28 us), 64-row panels above (40,000 rows: 103 vs 103-110 us,
// half the weight traffic from L2)
constexpr long long kChainSmallRows = 8192;
inline bool misaligned(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) != 0; }
}  // namespace

extern "C" {

static int linear_launch(const float *x0, const float *a0, const float *x1, const float *a1, const float *w,
                         const uint16_t *wpack, const float *bias, const bevmsda_linear_desc *d, float *y,
                         void *stream, const int32_t *gidx = nullptr, const float *gscale = nullptr,
                         const float *mask = nullptr, long ldmask = 0, float mask_scale = 1.0f) {
  if (!d) return BEVMSDA_ERR_NULL_POINTER;
  if (d->M < 0 || d->N < 0 || d->K0 < 0 || d->K1 < 0) return BEVMSDA_ERR_BAD_SHAPE;
  if (d->precision != 0 && d->precision != 1) return BEVMSDA_ERR_BAD_OPTION;
  if (d->M == 0 || d->N == 0) return BEVMSDA_OK;
  if (d->K0 == 0 || d->K0 % bevmsda::kLinKGran != 0 || d->K1 % bevmsda::kLinKGran != 0) return BEVMSDA_ERR_UNSUPPORTED;
  if (!x0 || (!w && !wpack) || !y || (d->K1 > 0 && !x1)) return BEVMSDA_ERR_NULL_POINTER;
  if (d->ldx0 % 4 != 0 || (w && d->ldw % 4 != 0) || (a0 && d->lda0 % 4 != 0) ||
      (d->K1 > 0 && (d->ldx1 % 4 != 0 || (a1 && d->lda1 % 4 != 0))))
    return BEVMSDA_ERR_UNSUPPORTED;
  const int gcols = d->group_cols;
  if (gcols < 0) return BEVMSDA_ERR_BAD_SHAPE;
  if (gcols > 0 && (gcols % bevmsda::kLinBN != 0 || d->N % gcols != 0)) return BEVMSDA_ERR_UNSUPPORTED;
  if (d->ldx0 < d->K0 || (w && d->ldw < d->K0 + d->K1) || d->ldy < (gcols > 0 ? gcols : d->N) ||
      (d->K1 > 0 && d->ldx1 < d->K1))
    return BEVMSDA_ERR_BAD_SHAPE;
  if (misaligned(x0) || (w && misaligned(w)) || (wpack && misaligned(wpack)) ||
      (reinterpret_cast<uintptr_t>(y) & 3u) != 0 || (a0 && misaligned(a0)) ||
      (d->K1 > 0 && (misaligned(x1) || (a1 && misaligned(a1)))))
    return BEVMSDA_ERR_MISALIGNED;
  bevmsda::LinArgs a;
  a.x0 = x0; a.a0 = a0; a.x1 = d->K1 > 0 ? x1 : nullptr; a.a1 = d->K1 > 0 ? a1 : nullptr;
  a.ldx0 = d->ldx0; a.lda0 = d->lda0; a.ldx1 = d->ldx1; a.lda1 = d->lda1;
  a.w = w; a.ldw = d->ldw; a.wpack = wpack; a.bias = bias; a.y = y; a.ldy = d->ldy;
  a.gidx = gidx; a.gscale = gscale;
  a.M = d->M; a.N = d->N; a.K0 = d->K0; a.K1 = d->K1; a.relu = d->relu ? 1 : 0;
  a.group_cols = gcols;
  a.out_bf16 = d->out_bf16 ? 1 : 0;
  // desc->reserved[0] = 1: y += result (first kernel, float4 epilogue: fp32 y, N and ldy multiples of 4, aligned y / bias)
  a.accum = d->reserved[0] == 1 ? 1 : 0;
  if (d->reserved[0] != 0 && d->reserved[0] != 1) return BEVMSDA_ERR_BAD_OPTION;
  a.mask = mask; a.ldmask = ldmask; a.mask_scale = mask_scale;
  if (mask && (gcols != 0 || d->out_bf16 || d->N % 4 != 0 || d->ldy % 4 != 0 || ldmask % 4 != 0 || ldmask < d->N ||
               misaligned(y) || misaligned(mask) || (bias && misaligned(bias)) || d->variant == 131))
    return BEVMSDA_ERR_UNSUPPORTED;
  if (a.accum && (d->out_bf16 || d->N % 4 != 0 || d->ldy % 4 != 0 || gcols % 4 != 0 || misaligned(y) || (bias && misaligned(bias)) ||
                  d->variant == 131))
    return BEVMSDA_ERR_UNSUPPORTED;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const bool add = a.a0 != nullptr || a.a1 != nullptr || gidx != nullptr;
  // desc->variant: 0 = library default; 1 = the first kernel over the fp32 weight matrix, 13 = over the packed weight
  // image (LDS-DMA into a single W area); 131 = the software-pipelined kernel (BEVMSDA_ERR_UNSUPPORTED when it does not
  // cover the call).  desc->reserved[1] = 1 keeps the default off the software-pipelined kernel.
  if (d->variant != 0 && d->variant != 1 && d->variant != 13 && d->variant != 131 && d->variant != 17) return BEVMSDA_ERR_BAD_OPTION;
  // desc->variant = 17 (round 6): 64-row x 256-column tiles of the first kernel over the packed weight image — for
  // 128 < N <= 256 every input row is staged ONCE (the 128 x 128 default stages it once per column tile) and the
  // 64-row workgroups (3 per CU) cover a 40,000-row call in one round
  if (d->variant == 17) {
    if (!wpack || a.accum || mask || d->N <= 128 || d->N % 4 != 0 || d->ldy % 4 != 0 || misaligned(y) || (bias && misaligned(bias)))
      return BEVMSDA_ERR_UNSUPPORTED;
    const long long nbm = (d->M + 63) / 64, nbn = (d->N + 255) / 256;
    const long long grid = ((nbm + 7) / 8) * 8 * nbn;
    if (grid >= (1LL << 31) || nbm >= (1LL << 28)) return BEVMSDA_ERR_TOO_LARGE;
    a.nblk_m = static_cast<int>(nbm);
    a.nblk_n = static_cast<int>(nbn);
    const dim3 g(static_cast<unsigned>(grid)), b(256);
    if (d->precision == 0) {
      if (add) hipLaunchKernelGGL((bevmsda::linear_splitbf16_kernel<3, true, 32, true, 3, 256, false, 64>), g, b, 0, st, a);
      else hipLaunchKernelGGL((bevmsda::linear_splitbf16_kernel<3, false, 32, true, 3, 256, false, 64>), g, b, 0, st, a);
    } else {
      if (add) hipLaunchKernelGGL((bevmsda::linear_splitbf16_kernel<1, true, 32, true, 3, 256, false, 64>), g, b, 0, st, a);
      else hipLaunchKernelGGL((bevmsda::linear_splitbf16_kernel<1, false, 32, true, 3, 256, false, 64>), g, b, 0, st, a);
    }
    return hipGetLastError() == hipSuccess ? BEVMSDA_OK : BEVMSDA_ERR_LAUNCH;
  }
  // software-pipelined kernel (linear_pipe.h)
  {
    const int nch = (d->K0 + d->K1) / 32;
    const bool covered = wpack && !add && !a.accum && !mask && (d->N % 4) == 0 && (d->ldy % 4) == 0 && (gcols % 4) == 0 && !misaligned(y) &&
                         (!bias || !misaligned(bias)) && (!d->out_bf16 || (reinterpret_cast<uintptr_t>(y) & 7u) == 0) &&
                         (nch == 8 || nch == 16);
    if (d->variant == 131 && !covered) return BEVMSDA_ERR_UNSUPPORTED;
    // default for tile-sized row counts (a rank's share of a BEV-tiled frame: one workgroup per CU, where its two-chunk
    // lead is worth 8-13 %: profiles/r2/r2_gemm_small_m.txt); at base size the first kernel's 4 workgroups per CU win
    if (covered && (d->variant == 131 || (d->variant == 0 && d->reserved[1] == 0 && (kLinearPipeDefault || d->M <= kLinearPipeMaxRows)))) {
      const long long nbm = (d->M + 127) / 128, nbn = (d->N + 127) / 128;
      const long long grid = ((nbm + 7) / 8) * 8 * nbn;
      if (grid >= (1LL << 31) || nbm >= (1LL << 28)) return BEVMSDA_ERR_TOO_LARGE;
      a.nblk_m = static_cast<int>(nbm);
      a.nblk_n = static_cast<int>(nbn);
#define BEVMSDA_PIPE(NP_, NCH_)                                                                                       \
  do {                                                                                                                \
    auto kern = bevmsda::linear_pipe_kernel<NP_, NCH_>;                                                               \
    const int dyn = 2 * ((NP_) == 3 ? 2 : 1) * bevmsda::kPipePlane * 2;                                               \
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, dyn) != \
        hipSuccess) return BEVMSDA_ERR_LAUNCH;                                                                        \
    hipLaunchKernelGGL(kern, dim3(static_cast<unsigned>(grid)), dim3(256), dyn, st, a);                               \
  } while (0)
      if (d->precision == 0) { if (nch == 8) BEVMSDA_PIPE(3, 8); else BEVMSDA_PIPE(3, 16); }
      else { if (nch == 8) BEVMSDA_PIPE(1, 8); else BEVMSDA_PIPE(1, 16); }
#undef BEVMSDA_PIPE
      return hipGetLastError() == hipSuccess ? BEVMSDA_OK : BEVMSDA_ERR_LAUNCH;
    }
  }
  if ((d->variant == 1 && wpack) || (d->variant == 13 && !wpack)) return BEVMSDA_ERR_BAD_OPTION;
  const long long nbm = (d->M + bevmsda::kLinBM - 1) / bevmsda::kLinBM;
  const long long nbn = (d->N + bevmsda::kLinBN - 1) / bevmsda::kLinBN;
  const long long grid = ((nbm + 7) / 8) * 8 * nbn;
  if (grid >= (1LL << 31) || nbm >= (1LL << 28)) return BEVMSDA_ERR_TOO_LARGE;
  a.nblk_m = static_cast<int>(nbm);
  a.nblk_n = static_cast<int>(nbn);
  const dim3 g(static_cast<unsigned>(grid)), b(256);
  // bf16 output: the transposed-tile epilogue packs 4 consecutive columns per lane
  if (d->out_bf16 && (d->N % 4 != 0 || d->ldy % 4 != 0 || (gcols % 4) != 0 ||
                      (reinterpret_cast<uintptr_t>(y) & 7u) != 0 || (bias && misaligned(bias))))
    return BEVMSDA_ERR_UNSUPPORTED;
#define BEVMSDA_LIN(NP_, WM_)                                                                                     \
  do {                                                                                                            \
    if (add) hipLaunchKernelGGL((bevmsda::linear_splitbf16_kernel<NP_, true, 32, true, WM_>), g, b, 0, st, a);    \
    else hipLaunchKernelGGL((bevmsda::linear_splitbf16_kernel<NP_, false, 32, true, WM_>), g, b, 0, st, a);       \
  } while (0)
  if (d->precision == 0) { if (wpack) BEVMSDA_LIN(3, 3); else BEVMSDA_LIN(3, 0); }
  else { if (wpack) BEVMSDA_LIN(1, 3); else BEVMSDA_LIN(1, 0); }
#undef BEVMSDA_LIN
  return hipGetLastError() == hipSuccess ? BEVMSDA_OK : BEVMSDA_ERR_LAUNCH;
}

int bevmsda_linear_f32(const float *x0, const float *a0, const float *x1, const float *a1, const float *w,
                       const float *bias, const bevmsda_linear_desc *d, float *y, void *stream) {
  if (!w) return BEVMSDA_ERR_NULL_POINTER;
  return linear_launch(x0, a0, x1, a1, w, nullptr, bias, d, y, stream);
}

int bevmsda_linear_packed_f32(const float *x0, const float *a0, const float *x1, const float *a1,
                              const uint16_t *wpack, const float *bias, const bevmsda_linear_desc *d, float *y,
                              void *stream) {
  if (!wpack) return BEVMSDA_ERR_NULL_POINTER;
  return linear_launch(x0, a0, x1, a1, nullptr, wpack, bias, d, y, stream);
}

int bevmsda_linear_relu_backward_packed_f32(const float *g, const uint16_t *wpack, const float *act, int64_t ld_act,
                                            float scale, const bevmsda_linear_desc *d, float *y, void *stream) {
  if (!wpack || !act) return BEVMSDA_ERR_NULL_POINTER;
  if (d && (d->K1 != 0 || d->relu)) return BEVMSDA_ERR_BAD_SHAPE;
  return linear_launch(g, nullptr, nullptr, nullptr, nullptr, wpack, nullptr, d, y, stream, nullptr, nullptr, act,
                       static_cast<long>(ld_act), scale);
}

int bevmsda_linear_gather_packed_f32(const float *rows, int64_t ld_rows, const int32_t *idx, const float *scale,
                                     const uint16_t *wpack, const float *bias, const bevmsda_linear_desc *d,
                                     float *y, void *stream) {
  if (!d) return BEVMSDA_ERR_NULL_POINTER;
  if (!wpack || !idx || !scale) return BEVMSDA_ERR_NULL_POINTER;
  if (d->K1 != 0) return BEVMSDA_ERR_BAD_SHAPE;
  bevmsda_linear_desc dd = *d;
  dd.ldx0 = ld_rows;
  return linear_launch(rows, nullptr, nullptr, nullptr, nullptr, wpack, bias, &dd, y, stream, idx, scale);
}

int bevmsda_linear_wgrad_f32(const float *g, int64_t ldg, const float *x, int64_t ldx, int64_t M, int N, int K,
                             float *grad_w, int64_t ldgw, float *grad_b, int precision, void *stream) {
  if (M < 0 || N <= 0 || K <= 0) return BEVMSDA_ERR_BAD_SHAPE;
  if (precision != 0 && precision != 1) return BEVMSDA_ERR_BAD_OPTION;
  if (N % 4 != 0 || K % 4 != 0 || ldg % 4 != 0 || ldx % 4 != 0) return BEVMSDA_ERR_UNSUPPORTED;
  if (ldg < N || ldx < K || ldgw < K) return BEVMSDA_ERR_BAD_SHAPE;
  if (M == 0) return BEVMSDA_OK;
  if (!g || !x || !grad_w) return BEVMSDA_ERR_NULL_POINTER;
  if (misaligned(g) || misaligned(x) || (reinterpret_cast<uintptr_t>(grad_w) & 3u) != 0) return BEVMSDA_ERR_MISALIGNED;
  bevmsda::WgradArgs a;
  a.g = g; a.x = x; a.ldg = ldg; a.ldx = ldx; a.gw = grad_w; a.ldgw = ldgw; a.gb = grad_b; a.M = M; a.N = N; a.K = K;
  a.tiles_n = (N + 127) / 128;
  a.tiles_k = (K + 127) / 128;
  // row slices: as many workgroups as are resident at once (2 per CU), at least 128 rows each
  const long long tiles = 1LL * a.tiles_n * a.tiles_k;
  // measured (tools/wgrad_ref.py): one round of workgroups (2 per CU -> 512) beats 1.5 rounds by 6-15 % for the
  // shapes with <= 8 output tiles; the 12-tile shape (768 x 256) is 20 % faster with 768 workgroups
  long long slices = tiles >= 12 ? (768 + tiles - 1) / tiles : 512 / tiles;
  long long rows = (M + slices - 1) / slices;
  rows = ((rows + 31) / 32) * 32;
  if (rows < 128) rows = 128;
  a.rows_per_block = static_cast<int>(rows);
  slices = (M + rows - 1) / rows;
  const long long slices8 = ((slices + 7) / 8) * 8;        // the kernel deals slices to XCDs: slice = 8 i + xcd
  if (tiles * slices8 >= (1LL << 31)) return BEVMSDA_ERR_TOO_LARGE;
  const dim3 grid(static_cast<unsigned>(tiles * slices8)), block(256);
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (precision == 0) hipLaunchKernelGGL((bevmsda::wgrad_splitbf16_kernel<3>), grid, block, 0, st, a);
  else hipLaunchKernelGGL((bevmsda::wgrad_splitbf16_kernel<1>), grid, block, 0, st, a);
  return hipGetLastError() == hipSuccess ? BEVMSDA_OK : BEVMSDA_ERR_LAUNCH;
}

int bevmsda_linear_wgrad_multi_f32(const bevmsda_wgrad_problem *probs, int nprob, int64_t M, int precision, int workgroups,
                                   int variant, void *stream) {
  if (variant != 0 && variant != 1 && variant != 2) return BEVMSDA_ERR_BAD_OPTION;
  if (nprob < 0 || nprob > bevmsda::kWgMaxProblems || M < 0) return BEVMSDA_ERR_BAD_SHAPE;
  if (precision != 0 && precision != 1) return BEVMSDA_ERR_BAD_OPTION;
  if (nprob == 0 || M == 0) return BEVMSDA_OK;
  if (!probs) return BEVMSDA_ERR_NULL_POINTER;
  if (variant == 2 && workgroups == 0) {
    // the 256 x 128 shape runs ONE 512-thread workgroup per CU, slices dealt to the XCDs in eights: it is taken when
    // tiles x 8 k workgroups fill at least 208 of the 256 CUs in one round (the grouped value projections, 12 tiles ->
    // 192 workgroups, stay on the 128 x 128 shape: 553 vs 538 us)
    long long t2 = 0;
    for (int i = 0; i < nprob; ++i) t2 += 1LL * ((probs[i].N + 255) / 256) * ((probs[i].K + 127) / 128);
    const long long s2 = t2 > 0 ? (256 / t2 / 8) * 8 : 0;
    if (s2 < 8 || t2 * s2 < 208) variant = 0;
  }
  bevmsda::WgradMultiArgs a{};
  long long tiles = 0;
  for (int i = 0; i < nprob; ++i) {
    const bevmsda_wgrad_problem &q = probs[i];
    if (q.N <= 0 || q.K <= 0) return BEVMSDA_ERR_BAD_SHAPE;
    if (q.N % 4 != 0 || q.K % 4 != 0 || q.ldg % 4 != 0 || q.ldx % 4 != 0) return BEVMSDA_ERR_UNSUPPORTED;
    if (q.ldg < q.N || q.ldx < q.K || q.ldgw < q.K) return BEVMSDA_ERR_BAD_SHAPE;
    if (!q.g || !q.x || !q.grad_w) return BEVMSDA_ERR_NULL_POINTER;
    if (misaligned(q.g) || misaligned(q.x) || (reinterpret_cast<uintptr_t>(q.grad_w) & 3u) != 0) return BEVMSDA_ERR_MISALIGNED;
    bevmsda::WgradArgs &w = a.p[i];
    w.g = q.g; w.x = q.x; w.ldg = q.ldg; w.ldx = q.ldx; w.gw = q.grad_w; w.ldgw = q.ldgw; w.gb = q.grad_b; w.M = M;
    // variant 2 (wgrad_tr.h, third shape): 256 x 128 output tiles
    w.N = q.N; w.K = q.K; w.tiles_n = (q.N + (variant == 2 ? 255 : 127)) / (variant == 2 ? 256 : 128); w.tiles_k = (q.K + 127) / 128;
    a.tile0[i] = static_cast<int>(tiles);
    tiles += 1LL * w.tiles_n * w.tiles_k;
  }
  if (tiles >= (1LL << 20)) return BEVMSDA_ERR_TOO_LARGE;
  a.tile0[nprob] = static_cast<int>(tiles);
  a.nprob = nprob;
  // one round of workgroups (2 per CU) over all tiles; 1.5 rounds from 12 tiles on (the single-problem policy);
  // `workgroups` > 0: the caller's target instead (benchmark sweeps)
  if (workgroups < 0 || workgroups > (1 << 20)) return BEVMSDA_ERR_BAD_OPTION;
  long long slices = workgroups > 0 ? workgroups / tiles : (tiles >= 12 ? (768 + tiles - 1) / tiles : 512 / tiles);
  if (variant == 2 && workgroups == 0) {
    // one 512-thread workgroup per CU and slices dealt to the XCDs in eights: the largest grid of tiles x 8 k workgroups
    // that still fits ONE round of 256 (a 257th workgroup costs a whole second round: 2.85 ms per step at 192 workgroups
    // against 3.81 at "256" = 288 launched, profiles/r6/r6e_wgrad_256x128_sweep.txt)
    slices = (256 / tiles / 8) * 8;
    if (slices < 8) slices = 8;
  }
  if (slices < 1) slices = 1;
  long long rows = (M + slices - 1) / slices;
  rows = ((rows + 31) / 32) * 32;
  if (rows < 128) rows = 128;
  slices = (M + rows - 1) / rows;
  for (int i = 0; i < nprob; ++i) a.p[i].rows_per_block = static_cast<int>(rows);
  const long long slices8 = ((slices + 7) / 8) * 8;
  if (tiles * slices8 >= (1LL << 31)) return BEVMSDA_ERR_TOO_LARGE;
  const dim3 grid(static_cast<unsigned>(tiles * slices8)), block(256);
  hipStream_t st = static_cast<hipStream_t>(stream);
  // variant 0: bf16 planes + transposing LDS reads (wgrad_tr.h); 1: fp32 tiles + gathered fragments (wgrad_mfma.h);
  // 2: 256 x 128 tiles, eight wavefronts, two LDS stages (wgrad_tr.h)
  if (variant == 2) {
    if (precision == 0) hipLaunchKernelGGL((bevmsda::wgrad_tr2_multi_kernel<3>), grid, dim3(512), 0, st, a);
    else hipLaunchKernelGGL((bevmsda::wgrad_tr2_multi_kernel<1>), grid, dim3(512), 0, st, a);
  } else if (variant == 0) {
    if (precision == 0) hipLaunchKernelGGL((bevmsda::wgrad_tr_multi_kernel<3>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((bevmsda::wgrad_tr_multi_kernel<1>), grid, block, 0, st, a);
  } else {
    if (precision == 0) hipLaunchKernelGGL((bevmsda::wgrad_multi_kernel<3>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((bevmsda::wgrad_multi_kernel<1>), grid, block, 0, st, a);
  }
  return hipGetLastError() == hipSuccess ? BEVMSDA_OK : BEVMSDA_ERR_LAUNCH;
}

int64_t bevmsda_linear_packed_bytes(int N, int K) {
  if (N <= 0 || K <= 0 || K % 32 != 0) return 0;
  return static_cast<int64_t>((N + 127) / 128) * (K / 32) * 2 * 128 * 40 * 2;
}

int bevmsda_linear_pack_weight_f32(const float *w, int64_t ldw, int N, int K, uint16_t *blob, void *stream) {
  if (N <= 0 || K <= 0) return BEVMSDA_ERR_BAD_SHAPE;
  if (K % 32 != 0 || ldw % 4 != 0) return BEVMSDA_ERR_UNSUPPORTED;
  if (ldw < K) return BEVMSDA_ERR_BAD_SHAPE;
  if (!w || !blob) return BEVMSDA_ERR_NULL_POINTER;
  if (misaligned(w) || misaligned(blob)) return BEVMSDA_ERR_MISALIGNED;
  const long long threads = static_cast<long long>((N + 127) / 128) * 128 * (K / 8);
  const long long nb = (threads + 255) / 256;
  if (nb >= (1LL << 31)) return BEVMSDA_ERR_TOO_LARGE;
  hipLaunchKernelGGL(bevmsda::lin_pack_weight_kernel<false>, dim3(static_cast<unsigned>(nb)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), w, static_cast<long>(ldw), N, K, blob);
  return hipGetLastError() == hipSuccess ? BEVMSDA_OK : BEVMSDA_ERR_LAUNCH;
}

int bevmsda_linear_pack_weight_t_f32(const float *wt, int64_t ldwt, int N, int K, uint16_t *blob, void *stream) {
  if (N <= 0 || K <= 0) return BEVMSDA_ERR_BAD_SHAPE;
  if (K % 32 != 0) return BEVMSDA_ERR_UNSUPPORTED;
  if (ldwt < N) return BEVMSDA_ERR_BAD_SHAPE;
  if (!wt || !blob) return BEVMSDA_ERR_NULL_POINTER;
  if (misaligned(blob) || (reinterpret_cast<uintptr_t>(wt) & 3u)) return BEVMSDA_ERR_MISALIGNED;
  const long long threads = static_cast<long long>((N + 127) / 128) * 128 * (K / 8);
  const long long nb = (threads + 255) / 256;
  if (nb >= (1LL << 31)) return BEVMSDA_ERR_TOO_LARGE;
  hipLaunchKernelGGL(bevmsda::lin_pack_weight_kernel<true>, dim3(static_cast<unsigned>(nb)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), wt, static_cast<long>(ldwt), N, K, blob);
  return hipGetLastError() == hipSuccess ? BEVMSDA_OK : BEVMSDA_ERR_LAUNCH;
}

// ---- row-panel projection (linear_panel.h)
int64_t bevmsda_linear_panel_packed_bytes(int N, int K) {
  if (N <= 0 || K <= 0 || K % bevmsda::kPanelK != 0) return 0;
  return static_cast<int64_t>((N + 63) / 64) * 2 * (K / 16) * 2 * 1024;
}

int bevmsda_linear_panel_pack_weight_f32(const float *w, int64_t ldw, int N, int K, uint16_t *blob, void *stream) {
  if (N <= 0 || K <= 0) return BEVMSDA_ERR_BAD_SHAPE;
  if (K % bevmsda::kPanelK != 0 || ldw % 4 != 0) return BEVMSDA_ERR_UNSUPPORTED;
  if (ldw < K) return BEVMSDA_ERR_BAD_SHAPE;
  if (!w || !blob) return BEVMSDA_ERR_NULL_POINTER;
  if (misaligned(w) || misaligned(blob)) return BEVMSDA_ERR_MISALIGNED;
  const int tiles32 = ((N + 63) / 64) * 2;
  const long long threads = 1LL * tiles32 * (K / 16) * 64;
  const long long nb = (threads + 255) / 256;
  if (nb >= (1LL << 31)) return BEVMSDA_ERR_TOO_LARGE;
  hipLaunchKernelGGL(bevmsda::lin_panel_pack_weight_kernel<false>, dim3(static_cast<unsigned>(nb)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), w, static_cast<long>(ldw), N, K, tiles32, blob);
  return hipGetLastError() == hipSuccess ? BEVMSDA_OK : BEVMSDA_ERR_LAUNCH;
}

int bevmsda_linear_panel_pack_weight_t_f32(const float *wt, int64_t ldwt, int N, int K, uint16_t *blob, void *stream) {
  if (N <= 0 || K <= 0) return BEVMSDA_ERR_BAD_SHAPE;
  if (K % bevmsda::kPanelK != 0) return BEVMSDA_ERR_UNSUPPORTED;
  if (ldwt < N) return BEVMSDA_ERR_BAD_SHAPE;
  if (!wt || !blob) return BEVMSDA_ERR_NULL_POINTER;
  if (misaligned(blob) || (reinterpret_cast<uintptr_t>(wt) & 3u)) return BEVMSDA_ERR_MISALIGNED;
  const int tiles32 = ((N + 63) / 64) * 2;
  const long long threads = 1LL * tiles32 * (K / 16) * 64;
  const long long nb = (threads + 255) / 256;
  if (nb >= (1LL << 31)) return BEVMSDA_ERR_TOO_LARGE;
  hipLaunchKernelGGL(bevmsda::lin_panel_pack_weight_kernel<true>, dim3(static_cast<unsigned>(nb)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), wt, static_cast<long>(ldwt), N, K, tiles32, blob);
  return hipGetLastError() == hipSuccess ? BEVMSDA_OK : BEVMSDA_ERR_LAUNCH;
}

// Many weight images in one launch (linear_panel.h: lin_pack_weights_multi_kernel).  `jobs` is a DEVICE array of `njobs`
// bevmsda_pack_job (the caller keeps it alive and unchanged while launches that read it — graph replays included — can
// run); `blocks` = the total number of 256-thread blocks = first_block of a job past the last (bevmsda_linear_pack_job_blocks
// gives a job's block count).  Jobs are validated by the caller against the single-image entry points' rules.
int64_t bevmsda_linear_pack_job_blocks(int N, int K, int kind) {
  if (N <= 0 || K <= 0 || kind < 0 || kind > 3) return 0;
  if (kind & 2) {
    if (K % bevmsda::kPanelK != 0) return 0;
    const long long threads = 1LL * ((N + 63) / 64) * 2 * (K / 16) * 64;
    return (threads + 255) / 256;
  }
  if (K % 32 != 0) return 0;
  const long long threads = static_cast<long long>((N + 127) / 128) * 128 * (K / 8);
  return (threads + 255) / 256;
}

int bevmsda_linear_pack_weights_multi_f32(const bevmsda_pack_job *jobs, int njobs, int64_t blocks, void *stream) {
  static_assert(sizeof(bevmsda_pack_job) == sizeof(bevmsda::PackJob), "bevmsda_pack_job layout");
  if (njobs < 0 || blocks < 0) return BEVMSDA_ERR_BAD_SHAPE;
  if (njobs == 0 || blocks == 0) return BEVMSDA_OK;
  if (!jobs) return BEVMSDA_ERR_NULL_POINTER;
  if (misaligned(jobs)) return BEVMSDA_ERR_MISALIGNED;
  if (blocks >= (1LL << 31)) return BEVMSDA_ERR_TOO_LARGE;
  hipLaunchKernelGGL(bevmsda::lin_pack_weights_multi_kernel, dim3(static_cast<unsigned>(blocks)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), reinterpret_cast<const bevmsda::PackJob *>(jobs), njobs);
  return hipGetLastError() == hipSuccess ? BEVMSDA_OK : BEVMSDA_ERR_LAUNCH;
}

// phase skew of the plain row-panel projections, in units of 1024 clocks (0 = none; tools/gemm_epilogue_ab.py)
static constexpr int kPanelSkewDefault = 0;

static int panel_launch(const float *x0, const float *a0, const float *x1, const float *a1, const int32_t *idx,
                        const float *scale, const uint16_t *wpanel, const float *bias,
                        const bevmsda_linear_desc *d, const bevmsda_layernorm_desc *ln, float *y, void *stream,
                        const int32_t *seg_start, int64_t seg_len, const int64_t *level_shapes, int num_levels,
                        const float *xb = nullptr, int64_t m_split = 0) {
  if (!d) return BEVMSDA_ERR_NULL_POINTER;
  if (d->M < 0 || d->N < 0 || d->K0 < 0 || d->K1 < 0 || d->group_cols < 0) return BEVMSDA_ERR_BAD_SHAPE;
  if (xb && (m_split < 0 || m_split > d->M)) return BEVMSDA_ERR_BAD_SHAPE;
  if (xb && (d->K1 != 0 || a0 || idx || ln || seg_start)) return BEVMSDA_ERR_UNSUPPORTED;
  if (xb && misaligned(xb)) return BEVMSDA_ERR_MISALIGNED;
  if (d->precision != 0 && d->precision != 1) return BEVMSDA_ERR_BAD_OPTION;
  if (d->M == 0 || d->N == 0) return BEVMSDA_OK;
  const int K = d->K0 + d->K1;
  if ((K != 256 && K != 512) || (d->K0 != 256 && d->K0 != 512) || (d->K1 != 0 && d->K1 != 256)) return BEVMSDA_ERR_UNSUPPORTED;
  if (!x0 || !wpanel || !y || (d->K1 > 0 && !x1)) return BEVMSDA_ERR_NULL_POINTER;
  if ((idx == nullptr) != (scale == nullptr)) return BEVMSDA_ERR_NULL_POINTER;
  if (idx && (d->K1 != 0 || a0)) return BEVMSDA_ERR_BAD_SHAPE;
  const int gcols = d->group_cols;
  if (d->N % 4 != 0 || d->ldy % 4 != 0 || d->ldx0 % 4 != 0 || (a0 && d->lda0 % 4 != 0) ||
      (d->K1 > 0 && (d->ldx1 % 4 != 0 || (a1 && d->lda1 % 4 != 0))) || (gcols > 0 && (gcols % 64 != 0 || d->N % gcols != 0)))
    return BEVMSDA_ERR_UNSUPPORTED;
  if (d->ldx0 < d->K0 || d->ldy < (gcols > 0 ? gcols : d->N) || (d->K1 > 0 && d->ldx1 < d->K1)) return BEVMSDA_ERR_BAD_SHAPE;
  if (misaligned(x0) || misaligned(wpanel) || misaligned(y) || (bias && misaligned(bias)) || (a0 && misaligned(a0)) ||
      (d->K1 > 0 && (misaligned(x1) || (a1 && misaligned(a1)))))
    return BEVMSDA_ERR_MISALIGNED;
  const int64_t wbytes = bevmsda_linear_panel_packed_bytes(d->N, K);
  if (wbytes >= (1LL << 31)) return BEVMSDA_ERR_TOO_LARGE;
  // (the epilogue's raw buffer covers ONE row panel of one output group — 64-bit base, 32-bit record count and offsets
  // inside it: at most 128 rows x ldy elements of 2 or 4 bytes have to fit)
  if (!ln && 128LL * d->ldy * (d->out_bf16 ? 2 : 4) >= (1LL << 31)) return BEVMSDA_ERR_TOO_LARGE;
  bevmsda::PanelArgs a;
  a.x0 = x0; a.a0 = a0; a.x1 = d->K1 > 0 ? x1 : nullptr; a.a1 = d->K1 > 0 ? a1 : nullptr;
  a.ldx0 = d->ldx0; a.lda0 = d->lda0; a.ldx1 = d->ldx1; a.lda1 = d->lda1;
  a.gidx = idx; a.gscale = scale;
  a.wp = wpanel; a.wp_bytes = static_cast<unsigned>(wbytes);
  a.bias = bias; a.y = y; a.ldy = d->ldy; a.M = d->M; a.N = d->N; a.K0 = d->K0; a.K1 = d->K1;
  a.relu = d->relu ? 1 : 0; a.group_cols = gcols; a.out_bf16 = d->out_bf16 ? 1 : 0;
  a.res = nullptr; a.ldres = 0; a.gamma = a.beta = nullptr; a.eps = 0.f;
  a.xb = xb; a.m_split = m_split;
  a.seg_start = seg_start; a.seg_len = seg_len; a.level_shapes = level_shapes; a.num_levels = level_shapes ? num_levels : 0;
  if (seg_start && (seg_len <= 0 || num_levels < 0)) return BEVMSDA_ERR_BAD_SHAPE;
  if (ln) {
    if (d->N != 256 || gcols != 0 || d->relu || d->out_bf16) return BEVMSDA_ERR_UNSUPPORTED;
    if (!ln->gamma || !ln->beta) return BEVMSDA_ERR_NULL_POINTER;
    if (misaligned(ln->gamma) || misaligned(ln->beta) || (ln->res && (misaligned(ln->res) || ln->ldres % 4 != 0 || ln->ldres < d->N)))
      return BEVMSDA_ERR_MISALIGNED;
    a.res = ln->res; a.ldres = ln->ldres; a.gamma = ln->gamma; a.beta = ln->beta; a.eps = ln->eps;
  }
  // workgroup shape (desc->reserved[2]): 1 = 64-row panels, 4 wavefronts of 64 x 64 tiles, two workgroups per CU, weight
  // fragments 2 k16 steps ahead (desc->reserved[3] = 6: six, a benchmark knob — no gain, tools/gemm_ab.py);
  // 2 = 128-row panels, 8 wavefronts of 128 x 32 tiles, one workgroup per CU (half the weight traffic per MFMA);
  // 0 = by shape.  Two panel passes (K = 512) and the LayerNorm epilogue need one column tile per wavefront: N <= 256
  int shape = d->reserved[2];
  if (shape < 0 || shape > 2) return BEVMSDA_ERR_BAD_OPTION;
  if ((K == 512 || ln) && d->N > 256) return BEVMSDA_ERR_UNSUPPORTED;
  if (shape == 0) shape = d->N >= 1024 && d->M >= 65536 ? 2 : 1;
  const int bm = shape == 1 ? 64 : 128;
  const long long nb = (d->M + bm - 1) / bm;
  if (nb >= (1LL << 31)) return BEVMSDA_ERR_TOO_LARGE;
  hipStream_t st = static_cast<hipStream_t>(stream);
  // one workgroup per row panel (a persistent grid was measured in round 5 and changed nothing: linear_panel.h)
  const dim3 grid(static_cast<unsigned>(nb));
  // desc->reserved[3]: weight fragments in flight for shape 1 (0 = default, 2 or 6 k16 steps ahead); 32 + bits: epilogue /
  // prefetch variants of the plain projection (no prepass, no LayerNorm; tools/gemm_ab.py): bit 0 the finished tile's
  // pieces stored one per k16 step of the next tile (DRIP), bit 1 weight fragments 4 steps ahead, bit 2 the round-4
  // epilogue (bias loaded per piece: 16 waited store round trips per tile — the A/B record of its removal)
  // 64 + n (n = 0 .. 31): phase skew of the column sweep, n x 1024 clocks (linear_panel.h; plain projections only);
  // 0 (the default) = kPanelSkewDefault for the plain projections
  int ev = d->reserved[3] >= 32 && d->reserved[3] < 64 ? d->reserved[3] - 32 : -1;
  const bool plain = !idx && !a.a0 && !a.a1 && !ln;
  a.skew = plain && d->reserved[3] == 0 ? kPanelSkewDefault : 0;
  if (d->reserved[3] == 97 || d->reserved[3] == 98) {
    // 97 / 98: the one-wavefront-per-SIMD dripping form (below)
  } else if (d->reserved[3] >= 64) {
    if (d->reserved[3] > 98 || !plain) return BEVMSDA_ERR_BAD_OPTION;
    a.skew = d->reserved[3] - 64;
  } else if (ev >= 0) {
    if (ev > 4 || idx || a.a0 || a.a1 || ln) return BEVMSDA_ERR_BAD_OPTION;
  } else if (d->reserved[3] != 0 && d->reserved[3] != 2 && d->reserved[3] != 6) {
    return BEVMSDA_ERR_BAD_OPTION;
  }
  const bool deep = d->reserved[3] == 6;       // (measured in one process: 616 vs 612 us, 272 vs 274 us — no default)
  if (d->reserved[3] == 97 || d->reserved[3] == 98) {
    // 97 / 98: the dripping-store form on 128-row panels, 4 wavefronts of 128 x 64 tiles, one wavefront per SIMD (weight
    // fragments 2 / 4 steps ahead): plain projections only
    if (!plain) return BEVMSDA_ERR_BAD_OPTION;
    const dim3 g128(static_cast<unsigned>((d->M + 127) / 128));
    if (d->precision == 0) {
      if (d->reserved[3] == 97) hipLaunchKernelGGL((bevmsda::linear_panel_kernel<3, 4, 2, 4, false, 0, 0, 0, true, 2>), g128, dim3(256), 0, st, a);
      else hipLaunchKernelGGL((bevmsda::linear_panel_kernel<3, 4, 2, 4, false, 0, 0, 0, true, 4>), g128, dim3(256), 0, st, a);
    } else {
      if (d->reserved[3] == 97) hipLaunchKernelGGL((bevmsda::linear_panel_kernel<1, 4, 2, 4, false, 0, 0, 0, true, 2>), g128, dim3(256), 0, st, a);
      else hipLaunchKernelGGL((bevmsda::linear_panel_kernel<1, 4, 2, 4, false, 0, 0, 0, true, 4>), g128, dim3(256), 0, st, a);
    }
    return hipGetLastError() == hipSuccess ? BEVMSDA_OK : BEVMSDA_ERR_LAUNCH;
  }
  if (ev > 0) {
#define BEVMSDA_PANEL_EV(NP_)                                                                                                  \
    do {                                                                                                                       \
      if (shape == 1) {                                                                                                        \
        if (ev == 1) hipLaunchKernelGGL((bevmsda::linear_panel_kernel<NP_, 2, 2, 4, false, 0, 0, 0, true, 2>), grid, dim3(256), 0, st, a);  \
        else if (ev == 2) hipLaunchKernelGGL((bevmsda::linear_panel_kernel<NP_, 2, 2, 4, false, 0, 0, 0, false, 4>), grid, dim3(256), 0, st, a); \
        else if (ev == 3) hipLaunchKernelGGL((bevmsda::linear_panel_kernel<NP_, 2, 2, 4, false, 0, 0, 0, true, 4>), grid, dim3(256), 0, st, a);  \
        else hipLaunchKernelGGL((bevmsda::linear_panel_kernel<NP_, 2, 2, 4, false, 0, 0, 0, false, 2, true>), grid, dim3(256), 0, st, a);    \
      } else {                                                                                                                 \
        if (ev == 1) hipLaunchKernelGGL((bevmsda::linear_panel_kernel<NP_, 4, 1, 8, false, 0, 0, 0, true, 2>), grid, dim3(512), 0, st, a);  \
        else if (ev == 2) hipLaunchKernelGGL((bevmsda::linear_panel_kernel<NP_, 4, 1, 8, false, 0, 0, 0, false, 4>), grid, dim3(512), 0, st, a); \
        else if (ev == 3) hipLaunchKernelGGL((bevmsda::linear_panel_kernel<NP_, 4, 1, 8, false, 0, 0, 0, true, 4>), grid, dim3(512), 0, st, a);  \
        else hipLaunchKernelGGL((bevmsda::linear_panel_kernel<NP_, 4, 1, 8, false, 0, 0, 0, false, 2, true>), grid, dim3(512), 0, st, a);    \
      }                                                                                                                        \
    } while (0)
    if (d->precision == 0) BEVMSDA_PANEL_EV(3); else BEVMSDA_PANEL_EV(1);
#undef BEVMSDA_PANEL_EV
    return hipGetLastError() == hipSuccess ? BEVMSDA_OK : BEVMSDA_ERR_LAUNCH;
  }
#define BEVMSDA_PANEL2(NP_, LN_, PRE_)                                                                                   \
  do {                                                                                                                   \
    if (shape == 1 && deep && (PRE_) == 0)                                                                               \
      hipLaunchKernelGGL((bevmsda::linear_panel_kernel<NP_, 2, 2, 4, LN_, PRE_, 0, 0, false, 6>), grid, dim3(256), 0, st, a); \
    else if (shape == 1) hipLaunchKernelGGL((bevmsda::linear_panel_kernel<NP_, 2, 2, 4, LN_, PRE_>), grid, dim3(256), 0, st, a); \
    else hipLaunchKernelGGL((bevmsda::linear_panel_kernel<NP_, 4, 1, 8, LN_, PRE_>), grid, dim3(512), 0, st, a);           \
  } while (0)
#define BEVMSDA_PANEL(NP_, LN_)                              \
  do {                                                       \
    if (idx) BEVMSDA_PANEL2(NP_, LN_, 2);                    \
    else if (a.a0 || a.a1) BEVMSDA_PANEL2(NP_, LN_, 1);      \
    else BEVMSDA_PANEL2(NP_, LN_, 0);                        \
  } while (0)
  if (d->precision == 0) { if (ln) BEVMSDA_PANEL(3, true); else BEVMSDA_PANEL(3, false); }
  else { if (ln) BEVMSDA_PANEL(1, true); else BEVMSDA_PANEL(1, false); }
#undef BEVMSDA_PANEL2
#undef BEVMSDA_PANEL
  return hipGetLastError() == hipSuccess ? BEVMSDA_OK : BEVMSDA_ERR_LAUNCH;
}

int bevmsda_linear_panel_f32(const float *x0, const float *a0, const float *x1, const float *a1, const int32_t *idx,
                             const float *scale, const uint16_t *wpanel, const float *bias,
                             const bevmsda_linear_desc *d, const bevmsda_layernorm_desc *ln, float *y, void *stream) {
  return panel_launch(x0, a0, x1, a1, idx, scale, wpanel, bias, d, ln, y, stream, nullptr, 0, nullptr, 0);
}

int bevmsda_linear_panel_segments_f32(const float *x0, const uint16_t *wpanel, const float *bias, const bevmsda_linear_desc *d,
                                      const int32_t *seg_start, int64_t seg_len, const int64_t *level_shapes,
                                      int num_levels, float *y, void *stream) {
  if (!seg_start) return BEVMSDA_ERR_NULL_POINTER;
  return panel_launch(x0, nullptr, nullptr, nullptr, nullptr, nullptr, wpanel, bias, d, nullptr, y, stream, seg_start, seg_len,
                      level_shapes, num_levels);
}

int bevmsda_linear_panel_rows2_f32(const float *x_lo, const float *x_hi, int64_t m_split, const uint16_t *wpanel, const float *bias,
                                   const bevmsda_linear_desc *d, float *y, void *stream) {
  if (!x_hi) return BEVMSDA_ERR_NULL_POINTER;
  return panel_launch(x_lo, nullptr, nullptr, nullptr, nullptr, nullptr, wpanel, bias, d, nullptr, y, stream, nullptr, 0, nullptr, 0,
                      x_hi, m_split);
}

// ---- row-local tail of an encoder layer in one kernel (linear_chain.h)
// the next layer's [first | y + pos] projection behind y (linear_chain.h TP; bevmsda_proj_ffn_chain_tail_f32)
struct ChainTail {
  const float *first; long long ld_first;
  const float *pos; long long ld_pos;
  const uint16_t *w3p; const float *b3;
  float *proj; long long ld_proj;
  int n3;
};

static int ffn_chain_launch(const float *rows, const int32_t *idx, const float *scale, const uint16_t *w0p, const float *b0,
                            const float *res, const float *gamma0, const float *beta0, const uint16_t *w1p, const float *b1,
                            const uint16_t *w2p, const float *b2, const float *gamma1, const float *beta1,
                            const bevmsda_chain_desc *d, float *y, void *stream, bool save, float *sv_z0, float *sv_x,
                            float *sv_h, float *sv_z1, const float *dk0 = nullptr, const float *dkh = nullptr,
                            const float *dk1 = nullptr, const ChainTail *tp = nullptr) {
  if (!d) return BEVMSDA_ERR_NULL_POINTER;
  const bool drop = dk0 || dkh || dk1;
  if (tp) {
    if (save || drop) return BEVMSDA_ERR_BAD_OPTION;
    if (tp->n3 <= 0 || tp->n3 > 256 || tp->n3 % 64 != 0) return BEVMSDA_ERR_UNSUPPORTED;      // (whole tiles of both workgroup shapes)
    if (d->M > 0 && (!tp->first || !tp->w3p || !tp->proj)) return BEVMSDA_ERR_NULL_POINTER;
    if (tp->ld_first % 4 != 0 || tp->ld_proj % 4 != 0 || (tp->pos && tp->ld_pos % 4 != 0)) return BEVMSDA_ERR_UNSUPPORTED;
    if (tp->ld_first < 256 || tp->ld_proj < tp->n3 || (tp->pos && tp->ld_pos < 256)) return BEVMSDA_ERR_BAD_SHAPE;
    if (misaligned(tp->first) || misaligned(tp->w3p) || misaligned(tp->proj) || (tp->pos && misaligned(tp->pos)) ||
        (tp->b3 && misaligned(tp->b3)))
      return BEVMSDA_ERR_MISALIGNED;
  }
  if (drop && !save) return BEVMSDA_ERR_BAD_OPTION;
  if ((dk0 && misaligned(dk0)) || (dkh && misaligned(dkh)) || (dk1 && misaligned(dk1))) return BEVMSDA_ERR_MISALIGNED;
  if (save && d->M > 0 && (!sv_z0 || !sv_x || !sv_h || !sv_z1)) return BEVMSDA_ERR_NULL_POINTER;
  if (save && (misaligned(sv_z0) || misaligned(sv_x) || misaligned(sv_h) || misaligned(sv_z1))) return BEVMSDA_ERR_MISALIGNED;
  if (d->M < 0) return BEVMSDA_ERR_BAD_SHAPE;
  if (d->precision != 0 && d->precision != 1) return BEVMSDA_ERR_BAD_OPTION;
  if (d->C != bevmsda::kChainC || d->F != bevmsda::kChainF) return BEVMSDA_ERR_UNSUPPORTED;
  if (d->M == 0) return BEVMSDA_OK;
  if (!rows || !w0p || !w1p || !w2p || !gamma0 || !beta0 || !gamma1 || !beta1 || !y) return BEVMSDA_ERR_NULL_POINTER;
  if ((idx == nullptr) != (scale == nullptr)) return BEVMSDA_ERR_NULL_POINTER;
  if (d->ld_rows % 4 != 0 || d->ld_y % 4 != 0 || (res && d->ld_res % 4 != 0)) return BEVMSDA_ERR_UNSUPPORTED;
  if (d->ld_rows < d->C || d->ld_y < d->C || (res && d->ld_res < d->C)) return BEVMSDA_ERR_BAD_SHAPE;
  if (misaligned(rows) || misaligned(w0p) || misaligned(w1p) || misaligned(w2p) || misaligned(y) || (res && misaligned(res)) ||
      (b0 && misaligned(b0)) || (b1 && misaligned(b1)) || (b2 && misaligned(b2)) || misaligned(gamma0) || misaligned(beta0) ||
      misaligned(gamma1) || misaligned(beta1))
    return BEVMSDA_ERR_MISALIGNED;
  // workgroup shape (desc->reserved[1]): 1 = 64-row panels (8 wavefronts, one workgroup per CU), 2 = 32-row panels
  // (4 wavefronts, two workgroups per CU); 0 = default
  int shape = d->reserved[1];
  if (shape < 0 || shape > 3) return BEVMSDA_ERR_BAD_OPTION;
  // desc->reserved[2]: columns of idx (0 = 2; J > 2: rows idx[m, 2..] >= 0 are added to the first row's — inference launches only)
  const int gst = d->reserved[2] == 0 ? 2 : d->reserved[2];
  if (gst < 2 || gst > 64 || (gst > 2 && (!idx || save))) return BEVMSDA_ERR_BAD_OPTION;
  // default: 32-row panels up to kChainSmallRows rows, mixed from one whole round of 64-row panels on (measured at 40,000
  // rows, interleaved runs on one box: 103.2-103.6 vs 106.6-107.2 us; profiles/r4/r4f_chain_mixed_shape_ab.txt)
  // (with the next layer's projection behind it the 32-row shape wins at every row count: two workgroups per CU cover each
  // other's fetch / split / barrier phases of the extra stage — 127 vs 140 us at 40,000 rows, base frame 4.09 vs 4.20 ms;
  // profiles/r6/r6u_seam_ab.txt)
  // (round 6, last day — at FRAME level, in a replayed graph, the inference launches are fastest as ONE launch of 32-row panels at
  // every row count: the mixed shape's isolated 3 us per launch do not pay for its second launch and the round of one workgroup
  // per CU — base frame 4.00 against 4.03 ms, bf16 2.78 against 2.81, first frame 3.94 against 4.03, small4 1.29 against 1.32,
  // queue 17.02 against 17.27; the training step too, by less: small4 bf16 4.96-4.97 against 4.98-5.00 ms, base 18.65-18.73 against
  // 18.70-18.76; profiles/r6z/r6zz_chain_shape_frame_ab.txt)
  // (between 8,192 and 16,384 rows — a tile of four ranks — one partial round of 64-row panels stays ahead: rank 1 of 4 1.61
  // against 1.66 ms)
  if (shape == 0) shape = (tp || d->M <= kChainSmallRows) ? 2 : (d->M >= 256LL * 64 ? 2 : 1);
  if (shape == 3) {
    // mixed (round 4): whole rounds of the 64-row shape (one workgroup per CU: 256 x 64 rows per round)
    // and the remainder — a partial round that would leave most CUs idle behind a few 64-row workgroups — on the 32-row
    // shape at two workgroups per CU, as a second launch over the tail rows
    const long long round64 = 256LL * 64;
    const long long head = (d->M / round64) * round64;
    const long long tail = d->M - head;
    if (head == 0 || tail == 0 || tail > 256LL * 2 * 32) {
      bevmsda_chain_desc one = *d;
      one.reserved[1] = (head == 0) ? 2 : 1;
      return ffn_chain_launch(rows, idx, scale, w0p, b0, res, gamma0, beta0, w1p, b1, w2p, b2, gamma1, beta1, &one, y, stream,
                              save, sv_z0, sv_x, sv_h, sv_z1, dk0, dkh, dk1, tp);
    }
    bevmsda_chain_desc dh = *d, dt = *d;
    dh.M = head; dh.reserved[1] = 1;
    dt.M = tail; dt.reserved[1] = 2;
    int rc = ffn_chain_launch(rows, idx, scale, w0p, b0, res, gamma0, beta0, w1p, b1, w2p, b2, gamma1, beta1, &dh, y, stream,
                              save, sv_z0, sv_x, sv_h, sv_z1, dk0, dkh, dk1, tp);
    if (rc != BEVMSDA_OK) return rc;
    const long long o = head;
    ChainTail tt{};
    if (tp) {
      tt = *tp;
      tt.first += o * tp->ld_first;
      if (tt.pos) tt.pos += o * tp->ld_pos;
      tt.proj += o * tp->ld_proj;
    }
    return ffn_chain_launch(idx ? rows : rows + o * d->ld_rows, idx ? idx + o * gst : nullptr, scale ? scale + o : nullptr, w0p, b0,
                            res ? res + o * d->ld_res : nullptr, gamma0, beta0, w1p, b1, w2p, b2, gamma1, beta1, &dt,
                            y + o * d->ld_y, stream, save, save ? sv_z0 + o * 256 : nullptr, save ? sv_x + o * 256 : nullptr,
                            save ? sv_h + o * 512 : nullptr, save ? sv_z1 + o * 256 : nullptr, dk0 ? dk0 + o * 256 : nullptr,
                            dkh ? dkh + o * 512 : nullptr, dk1 ? dk1 + o * 256 : nullptr, tp ? &tt : nullptr);
  }
  const int bm = shape == 1 ? 64 : 32;
  const long long nb = (d->M + bm - 1) / bm;
  if (nb >= (1LL << 31)) return BEVMSDA_ERR_TOO_LARGE;
  bevmsda::ChainArgs a{};
  a.rows = rows; a.ld_rows = d->ld_rows; a.gidx = idx; a.gscale = scale; a.gstride = gst;
  a.w0 = w0p; a.w1 = w1p; a.w2 = w2p; a.b0 = b0; a.b1 = b1; a.b2 = b2;
  a.res = res; a.ld_res = d->ld_res; a.gamma0 = gamma0; a.beta0 = beta0; a.gamma1 = gamma1; a.beta1 = beta1;
  a.eps0 = d->eps0; a.eps1 = d->eps1; a.y = y; a.ld_y = d->ld_y; a.M = d->M;
  a.sv_z0 = sv_z0; a.sv_x = sv_x; a.sv_h = sv_h; a.sv_z1 = sv_z1;
  a.dk0 = dk0; a.dkh = dkh; a.dk1 = dk1;
  if (tp) {
    a.tp_first = tp->first; a.ld_tp_first = tp->ld_first; a.tp_pos = tp->pos; a.ld_tp_pos = tp->ld_pos;
    a.w3 = tp->w3p; a.b3 = tp->b3; a.y3 = tp->proj; a.ld_y3 = tp->ld_proj; a.N3 = tp->n3;
  }
  hipStream_t st = static_cast<hipStream_t>(stream);
  const dim3 grid(static_cast<unsigned>(nb));
#define BEVMSDA_CHAIN_TP(NP_, PRE_)                                                                                               \
  do {                                                                                                                            \
    if (shape == 1) hipLaunchKernelGGL((bevmsda::linear_chain_kernel<NP_, PRE_, 0, 2, 1, 8, false, false, true>), grid, dim3(512), 0, st, a); \
    else hipLaunchKernelGGL((bevmsda::linear_chain_kernel<NP_, PRE_, 0, 1, 2, 4, false, false, true>), grid, dim3(256), 0, st, a);            \
  } while (0)
  if (tp) {
    if (d->precision == 0) { if (idx) BEVMSDA_CHAIN_TP(3, 2); else BEVMSDA_CHAIN_TP(3, 0); }
    else { if (idx) BEVMSDA_CHAIN_TP(1, 2); else BEVMSDA_CHAIN_TP(1, 0); }
    return hipGetLastError() == hipSuccess ? BEVMSDA_OK : BEVMSDA_ERR_LAUNCH;
  }
#undef BEVMSDA_CHAIN_TP
#define BEVMSDA_CHAIN(NP_, PRE_, SV_, DR_)                                                                                        \
  do {                                                                                                                            \
    if (shape == 1) hipLaunchKernelGGL((bevmsda::linear_chain_kernel<NP_, PRE_, 0, 2, 1, 8, SV_, DR_>), grid, dim3(512), 0, st, a); \
    else hipLaunchKernelGGL((bevmsda::linear_chain_kernel<NP_, PRE_, 0, 1, 2, 4, SV_, DR_>), grid, dim3(256), 0, st, a);            \
  } while (0)
  if (drop) {                                  // (train() mode with active dropout: the gather form only)
    if (!idx) return BEVMSDA_ERR_UNSUPPORTED;
    if (d->precision == 0) BEVMSDA_CHAIN(3, 2, true, true); else BEVMSDA_CHAIN(1, 2, true, true);
  } else if (save) {
    if (d->precision == 0) { if (idx) BEVMSDA_CHAIN(3, 2, true, false); else BEVMSDA_CHAIN(3, 0, true, false); }
    else { if (idx) BEVMSDA_CHAIN(1, 2, true, false); else BEVMSDA_CHAIN(1, 0, true, false); }
  } else {
    if (d->precision == 0) { if (idx) BEVMSDA_CHAIN(3, 2, false, false); else BEVMSDA_CHAIN(3, 0, false, false); }
    else { if (idx) BEVMSDA_CHAIN(1, 2, false, false); else BEVMSDA_CHAIN(1, 0, false, false); }
  }
#undef BEVMSDA_CHAIN
  return hipGetLastError() == hipSuccess ? BEVMSDA_OK : BEVMSDA_ERR_LAUNCH;
}

int bevmsda_proj_ffn_chain_f32(const float *rows, const int32_t *idx, const float *scale, const uint16_t *w0p, const float *b0,
                               const float *res, const float *gamma0, const float *beta0, const uint16_t *w1p, const float *b1,
                               const uint16_t *w2p, const float *b2, const float *gamma1, const float *beta1,
                               const bevmsda_chain_desc *d, float *y, void *stream) {
  return ffn_chain_launch(rows, idx, scale, w0p, b0, res, gamma0, beta0, w1p, b1, w2p, b2, gamma1, beta1, d, y, stream, false,
                          nullptr, nullptr, nullptr, nullptr);
}

int bevmsda_proj_ffn_chain_tail_f32(const float *rows, const int32_t *idx, const float *scale, const uint16_t *w0p, const float *b0,
                                    const float *res, const float *gamma0, const float *beta0, const uint16_t *w1p, const float *b1,
                                    const uint16_t *w2p, const float *b2, const float *gamma1, const float *beta1,
                                    const bevmsda_chain_desc *d, float *y, const float *first, int64_t ld_first,
                                    const float *pos, int64_t ld_pos, const uint16_t *w3p, const float *b3, int n3,
                                    float *proj_out, int64_t ld_proj, void *stream) {
  const ChainTail tp{first, ld_first, pos, ld_pos, w3p, b3, proj_out, ld_proj, n3};
  return ffn_chain_launch(rows, idx, scale, w0p, b0, res, gamma0, beta0, w1p, b1, w2p, b2, gamma1, beta1, d, y, stream, false,
                          nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, &tp);
}

int bevmsda_proj_ffn_chain_train_f32(const float *rows, const int32_t *idx, const float *scale, const uint16_t *w0p, const float *b0,
                                     const float *res, const float *gamma0, const float *beta0, const uint16_t *w1p,
                                     const float *b1, const uint16_t *w2p, const float *b2, const float *gamma1,
                                     const float *beta1, const bevmsda_chain_desc *d, float *y, float *save_z0, float *save_x,
                                     float *save_h, float *save_z1, const float *drop0, const float *droph,
                                     const float *drop1, void *stream) {
  return ffn_chain_launch(rows, idx, scale, w0p, b0, res, gamma0, beta0, w1p, b1, w2p, b2, gamma1, beta1, d, y, stream, true,
                          save_z0, save_x, save_h, save_z1, drop0, droph, drop1);
}

// ---- backward of the projection + FFN chain (linear_chain.h, MODE 2)
int bevmsda_proj_ffn_chain_backward_f32(const float *grad_y, int64_t ld_grad_y, const float *save_z0, const float *save_h,
                                        const float *save_z1, const float *gamma0, const float *gamma1, const uint16_t *w0t_p,
                                        const uint16_t *w1t_p, const uint16_t *w2t_p, const bevmsda_chain_desc *d, float *grad_z1,
                                        float *grad_h, float *grad_z0, float *grad_in, float *grad_gamma_beta1,
                                        float *grad_gamma_beta0, const float *drop0, const float *drop1, float hidden_scale,
                                        float *grad_zp, void *stream) {
  if (!d) return BEVMSDA_ERR_NULL_POINTER;
  if (d->M < 0) return BEVMSDA_ERR_BAD_SHAPE;
  if (d->precision != 0 && d->precision != 1) return BEVMSDA_ERR_BAD_OPTION;
  if (d->C != bevmsda::kChainC || d->F != bevmsda::kChainF) return BEVMSDA_ERR_UNSUPPORTED;
  if (d->M == 0) return BEVMSDA_OK;
  if (!grad_y || !save_z0 || !save_h || !save_z1 || !gamma0 || !gamma1 || !w0t_p || !w1t_p || !w2t_p || !grad_z1 || !grad_h ||
      !grad_z0 || !grad_in || !grad_gamma_beta1 || !grad_gamma_beta0)
    return BEVMSDA_ERR_NULL_POINTER;
  if (ld_grad_y % 4 != 0) return BEVMSDA_ERR_UNSUPPORTED;
  if (ld_grad_y < d->C) return BEVMSDA_ERR_BAD_SHAPE;
  if (misaligned(grad_y) || misaligned(save_z0) || misaligned(save_h) || misaligned(save_z1) || misaligned(gamma0) ||
      misaligned(gamma1) || misaligned(w0t_p) || misaligned(w1t_p) || misaligned(w2t_p) || misaligned(grad_z1) ||
      misaligned(grad_h) || misaligned(grad_z0) || misaligned(grad_in) || (reinterpret_cast<uintptr_t>(grad_gamma_beta1) & 3u) ||
      (reinterpret_cast<uintptr_t>(grad_gamma_beta0) & 3u))
    return BEVMSDA_ERR_MISALIGNED;
  const long long nb = (d->M + 31) / 32;
  if (nb >= (1LL << 31)) return BEVMSDA_ERR_TOO_LARGE;
  bevmsda::ChainArgs a{};
  // the three GEMMs of the backward read the images of the TRANSPOSED weights in the slots of the forward's: stage "0" =
  // gz0 W0 (256 x 256), stage "1" = gz1 W2 (512 columns, K = 256), stage "2" = gh W1 (256 columns, K = 512)
  a.w0 = w0t_p; a.w1 = w2t_p; a.w2 = w1t_p;
  a.gamma0 = gamma0; a.gamma1 = gamma1; a.eps0 = d->eps0; a.eps1 = d->eps1; a.M = d->M;
  a.bw_gy = grad_y; a.bw_ld_gy = ld_grad_y; a.bw_z1 = save_z1; a.bw_h = save_h; a.bw_z0 = save_z0;
  a.bw_gz1 = grad_z1; a.bw_gh = grad_h; a.bw_gz0 = grad_z0; a.bw_din = grad_in;
  a.bw_dgb1 = grad_gamma_beta1; a.bw_dgb0 = grad_gamma_beta0;
  if (drop0 && !grad_zp) return BEVMSDA_ERR_NULL_POINTER;
  if ((drop0 && misaligned(drop0)) || (drop1 && misaligned(drop1)) || (grad_zp && misaligned(grad_zp))) return BEVMSDA_ERR_MISALIGNED;
  a.dk0 = drop0; a.dk1 = drop1; a.bw_hscale = hidden_scale; a.bw_gzp = grad_zp;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const dim3 grid(static_cast<unsigned>(nb));
  if (d->precision == 0) hipLaunchKernelGGL((bevmsda::linear_chain_kernel<3, 0, 2, 1, 2, 4, false, false>), grid, dim3(256), 0, st, a);
  else hipLaunchKernelGGL((bevmsda::linear_chain_kernel<1, 0, 2, 1, 2, 4, false, false>), grid, dim3(256), 0, st, a);
  return hipGetLastError() == hipSuccess ? BEVMSDA_OK : BEVMSDA_ERR_LAUNCH;
}

// ---- backward of the projection + LayerNorm + projection chain (linear_chain.h, MODE 3)
int bevmsda_proj_ln_proj_chain_backward_f32(const float *grad_proj, int64_t ld_grad_proj, const float *grad_x, const float *save_z0,
                                            const float *gamma0, const uint16_t *w0t_p, const uint16_t *w1t_p,
                                            const bevmsda_chain_desc *d, float *grad_z0, float *grad_in, float *grad_gamma_beta0,
                                            const float *drop0, float *grad_zp, void *stream) {
  if (!d) return BEVMSDA_ERR_NULL_POINTER;
  if (d->M < 0) return BEVMSDA_ERR_BAD_SHAPE;
  if (d->precision != 0 && d->precision != 1) return BEVMSDA_ERR_BAD_OPTION;
  const int N2 = d->reserved[0];
  if (d->C != bevmsda::kChainC || N2 <= 0 || N2 % 256 != 0 || N2 > bevmsda::kChainMaxN2) return BEVMSDA_ERR_UNSUPPORTED;
  if (d->M == 0) return BEVMSDA_OK;
  if (!grad_proj || !save_z0 || !gamma0 || !w0t_p || !w1t_p || !grad_z0 || !grad_in || !grad_gamma_beta0) return BEVMSDA_ERR_NULL_POINTER;
  if (ld_grad_proj % 4 != 0) return BEVMSDA_ERR_UNSUPPORTED;
  if (ld_grad_proj < N2) return BEVMSDA_ERR_BAD_SHAPE;
  if (misaligned(grad_proj) || (grad_x && misaligned(grad_x)) || misaligned(save_z0) || misaligned(gamma0) || misaligned(w0t_p) ||
      misaligned(w1t_p) || misaligned(grad_z0) || misaligned(grad_in) || (reinterpret_cast<uintptr_t>(grad_gamma_beta0) & 3u))
    return BEVMSDA_ERR_MISALIGNED;
  const long long nb = (d->M + 31) / 32;
  if (nb >= (1LL << 31)) return BEVMSDA_ERR_TOO_LARGE;
  bevmsda::ChainArgs a{};
  a.w0 = w0t_p; a.w1 = w1t_p; a.N2 = N2;
  a.gamma0 = gamma0; a.eps0 = d->eps0; a.M = d->M;
  a.bw_gy = grad_proj; a.bw_ld_gy = ld_grad_proj; a.bw_z1 = grad_x; a.bw_z0 = save_z0;
  a.bw_gz0 = grad_z0; a.bw_din = grad_in; a.bw_dgb0 = grad_gamma_beta0;
  if (drop0 && !grad_zp) return BEVMSDA_ERR_NULL_POINTER;
  if ((drop0 && misaligned(drop0)) || (grad_zp && misaligned(grad_zp))) return BEVMSDA_ERR_MISALIGNED;
  a.dk0 = drop0; a.bw_gzp = grad_zp; a.bw_hscale = 1.f;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const dim3 grid(static_cast<unsigned>(nb));
  if (d->precision == 0) hipLaunchKernelGGL((bevmsda::linear_chain_kernel<3, 0, 3, 1, 2, 4, false, false>), grid, dim3(256), 0, st, a);
  else hipLaunchKernelGGL((bevmsda::linear_chain_kernel<1, 0, 3, 1, 2, 4, false, false>), grid, dim3(256), 0, st, a);
  return hipGetLastError() == hipSuccess ? BEVMSDA_OK : BEVMSDA_ERR_LAUNCH;
}

static int ln_proj_chain_launch(const float *rows, const int32_t *idx, const float *scale, const uint16_t *w0p, const float *b0,
                                const float *res, const float *gamma0, const float *beta0, const uint16_t *w1p, const float *b1,
                                const bevmsda_chain_desc *d, float *x_out, float *proj_out, void *stream, bool save,
                                float *sv_z0, const float *dk0 = nullptr) {
  if (!d) return BEVMSDA_ERR_NULL_POINTER;
  if (dk0 && (!save || misaligned(dk0))) return BEVMSDA_ERR_BAD_OPTION;
  if (save && d->M > 0 && !sv_z0) return BEVMSDA_ERR_NULL_POINTER;
  if (save && misaligned(sv_z0)) return BEVMSDA_ERR_MISALIGNED;
  if (d->M < 0) return BEVMSDA_ERR_BAD_SHAPE;
  if (d->precision != 0 && d->precision != 1) return BEVMSDA_ERR_BAD_OPTION;
  const long ld_y2 = d->reserved[0];
  if (d->C != bevmsda::kChainC || d->F <= 0 || d->F % 32 != 0 || d->F > bevmsda::kChainMaxN2) return BEVMSDA_ERR_UNSUPPORTED;
  if (d->M == 0) return BEVMSDA_OK;
  if (!rows || !w0p || !w1p || !gamma0 || !beta0 || !x_out || !proj_out) return BEVMSDA_ERR_NULL_POINTER;
  if ((idx == nullptr) != (scale == nullptr)) return BEVMSDA_ERR_NULL_POINTER;
  if (d->ld_rows % 4 != 0 || d->ld_y % 4 != 0 || ld_y2 % 4 != 0 || (res && d->ld_res % 4 != 0)) return BEVMSDA_ERR_UNSUPPORTED;
  if (d->ld_rows < d->C || d->ld_y < d->C || ld_y2 < d->F || (res && d->ld_res < d->C)) return BEVMSDA_ERR_BAD_SHAPE;
  if (misaligned(rows) || misaligned(w0p) || misaligned(w1p) || misaligned(x_out) || misaligned(proj_out) || (res && misaligned(res)) ||
      (b0 && misaligned(b0)) || (b1 && misaligned(b1)) || misaligned(gamma0) || misaligned(beta0))
    return BEVMSDA_ERR_MISALIGNED;
  int shape = d->reserved[1];
  if (shape < 0 || shape > 3) return BEVMSDA_ERR_BAD_OPTION;
  // (isolated: mixed 100.4-101.5 vs 103.8-105.7 us at 40,000 rows; in the frame one launch of 32-row panels wins — see ffn_chain_launch)
  if (shape == 0) shape = d->M <= kChainSmallRows ? 2 : (d->M >= 256LL * 64 ? 2 : 1);
  if (d->F % 64 != 0) shape = 1;               // the 32-row shape walks 64-column tiles
  if (shape == 3) {                            // mixed: whole rounds on the 64-row shape, the tail on the 32-row shape
    const long long round64 = 256LL * 64;
    const long long head = (d->M / round64) * round64;
    const long long tail = d->M - head;
    bevmsda_chain_desc dh = *d, dt = *d;
    if (head == 0 || tail == 0 || tail > 256LL * 2 * 32) {
      dh.reserved[1] = (head == 0) ? 2 : 1;
      return ln_proj_chain_launch(rows, idx, scale, w0p, b0, res, gamma0, beta0, w1p, b1, &dh, x_out, proj_out, stream, save, sv_z0,
                                  dk0);
    }
    dh.M = head; dh.reserved[1] = 1;
    dt.M = tail; dt.reserved[1] = 2;
    int rc = ln_proj_chain_launch(rows, idx, scale, w0p, b0, res, gamma0, beta0, w1p, b1, &dh, x_out, proj_out, stream, save, sv_z0,
                                  dk0);
    if (rc != BEVMSDA_OK) return rc;
    const long long o = head;
    return ln_proj_chain_launch(idx ? rows : rows + o * d->ld_rows, idx ? idx + o * 2 : nullptr, scale ? scale + o : nullptr, w0p,
                                b0, res ? res + o * d->ld_res : nullptr, gamma0, beta0, w1p, b1, &dt, x_out + o * d->ld_y,
                                proj_out + o * ld_y2, stream, save, save ? sv_z0 + o * 256 : nullptr,
                                dk0 ? dk0 + o * 256 : nullptr);
  }
  const int bm = shape == 1 ? 64 : 32;
  const long long nb = (d->M + bm - 1) / bm;
  if (nb >= (1LL << 31)) return BEVMSDA_ERR_TOO_LARGE;
  bevmsda::ChainArgs a{};
  a.rows = rows; a.ld_rows = d->ld_rows; a.gidx = idx; a.gscale = scale;
  a.w0 = w0p; a.w1 = w1p; a.w2 = nullptr; a.b0 = b0; a.b1 = b1; a.b2 = nullptr;
  a.res = res; a.ld_res = d->ld_res; a.gamma0 = gamma0; a.beta0 = beta0; a.gamma1 = a.beta1 = nullptr;
  a.eps0 = d->eps0; a.eps1 = 0.f; a.y = x_out; a.ld_y = d->ld_y; a.M = d->M; a.y2 = proj_out; a.ld_y2 = ld_y2; a.N2 = d->F;
  a.sv_z0 = sv_z0;
  a.dk0 = dk0;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const dim3 grid(static_cast<unsigned>(nb));
#define BEVMSDA_CHAIN(NP_, PRE_, SV_, DR_)                                                                                        \
  do {                                                                                                                            \
    if (shape == 1) hipLaunchKernelGGL((bevmsda::linear_chain_kernel<NP_, PRE_, 1, 2, 1, 8, SV_, DR_>), grid, dim3(512), 0, st, a); \
    else hipLaunchKernelGGL((bevmsda::linear_chain_kernel<NP_, PRE_, 1, 1, 2, 4, SV_, DR_>), grid, dim3(256), 0, st, a);            \
  } while (0)
  if (save) {                                  // (the autograd path's forward has no gather prepass: plain rows only)
    if (idx) return BEVMSDA_ERR_UNSUPPORTED;
    if (dk0) { if (d->precision == 0) BEVMSDA_CHAIN(3, 0, true, true); else BEVMSDA_CHAIN(1, 0, true, true); }
    else { if (d->precision == 0) BEVMSDA_CHAIN(3, 0, true, false); else BEVMSDA_CHAIN(1, 0, true, false); }
  } else {
    if (d->precision == 0) { if (idx) BEVMSDA_CHAIN(3, 2, false, false); else BEVMSDA_CHAIN(3, 0, false, false); }
    else { if (idx) BEVMSDA_CHAIN(1, 2, false, false); else BEVMSDA_CHAIN(1, 0, false, false); }
  }
#undef BEVMSDA_CHAIN
  return hipGetLastError() == hipSuccess ? BEVMSDA_OK : BEVMSDA_ERR_LAUNCH;
}

int bevmsda_proj_ln_proj_chain_f32(const float *rows, const int32_t *idx, const float *scale, const uint16_t *w0p, const float *b0,
                                   const float *res, const float *gamma0, const float *beta0, const uint16_t *w1p, const float *b1,
                                   const bevmsda_chain_desc *d, float *x_out, float *proj_out, void *stream) {
  return ln_proj_chain_launch(rows, idx, scale, w0p, b0, res, gamma0, beta0, w1p, b1, d, x_out, proj_out, stream, false, nullptr);
}

int bevmsda_proj_ln_proj_chain_train_f32(const float *rows, const uint16_t *w0p, const float *b0, const float *res,
                                         const float *gamma0, const float *beta0, const uint16_t *w1p, const float *b1,
                                         const bevmsda_chain_desc *d, float *x_out, float *proj_out, float *save_z0,
                                         const float *drop0, void *stream) {
  return ln_proj_chain_launch(rows, nullptr, nullptr, w0p, b0, res, gamma0, beta0, w1p, b1, d, x_out, proj_out, stream, true,
                              save_z0, drop0);
}

}  // extern "C"
