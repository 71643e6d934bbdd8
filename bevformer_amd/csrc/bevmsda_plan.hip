// C ABI of libbevmsda.so, per-frame geometry (declared in include/bevmsda.h): the device-side
// frame plan (frame_plan.h).  No torch, no allocation, no global state, no host synchronisation.
#include "../../include/bevmsda.h"
#include "frame_plan.h"

namespace {
inline bool misaligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) != 0; }
inline bool misaligned4(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 3u) != 0; }
}  // namespace

extern "C" {

int64_t bevmsda_frame_plan_scratch(int Nc, int Q) {
  if (Nc <= 0 || Q < 0) return 0;
  return 2 * static_cast<int64_t>(Nc) * ((Q + 255) / 256);
}

int64_t bevmsda_frame_plan_counters(int B, int Nc) {
  if (B <= 0 || Nc <= 0) return 0;
  return 4 + static_cast<int64_t>(B) * Nc + 1;
}

int bevmsda_frame_plan_f32(const float *lidar2img, const float *ref_3d, const int32_t *order,
                           const bevmsda_plan_desc *d, float *ref_cam, uint8_t *bev_mask, float *inv_count,
                           uint8_t *slot, int32_t *block_scratch, int32_t *row_query, int32_t *row_batch, float *row_ref,
                           int32_t *q_rows, int32_t *q_rows2, int32_t *counters, void *stream) {
  if (!d) return BEVMSDA_ERR_NULL_POINTER;
  if (d->B <= 0 || d->Nc <= 0 || d->Q < 0 || d->D <= 0 || d->row_capacity < 0 || d->q_lo < 0 || d->q_hi > d->Q ||
      d->q_lo > d->q_hi)
    return BEVMSDA_ERR_BAD_SHAPE;
  if (d->Nc > bevmsda::kPlanMaxCams || d->D > bevmsda::kPlanMaxAnchors) return BEVMSDA_ERR_UNSUPPORTED;
  if (1LL * d->B * d->Nc * d->Q * d->D >= (1LL << 30)) return BEVMSDA_ERR_TOO_LARGE;
  if (!lidar2img || !ref_3d || !order || !ref_cam || !bev_mask || !inv_count || !slot || !block_scratch || !row_query ||
      !row_batch || !row_ref || !q_rows || !q_rows2 || !counters)
    return BEVMSDA_ERR_NULL_POINTER;
  if (misaligned4(lidar2img) || misaligned4(ref_3d) || misaligned4(order) || misaligned4(ref_cam) ||
      misaligned4(inv_count) || misaligned4(row_query) || misaligned4(row_batch) || misaligned16(row_ref) ||
      misaligned4(q_rows) || misaligned4(q_rows2) || misaligned4(counters))
    return BEVMSDA_ERR_MISALIGNED;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const size_t ncnt = static_cast<size_t>(bevmsda_frame_plan_counters(d->B, d->Nc));
  if (hipMemsetAsync(counters, 0, ncnt * sizeof(int32_t), st) != hipSuccess) return BEVMSDA_ERR_LAUNCH;
  if (d->Q == 0) return BEVMSDA_OK;
  bevmsda::PlanArgs a;
  a.l2i = lidar2img; a.ref3d = ref_3d; a.order = order; a.ref_cam = ref_cam; a.bev_mask = bev_mask;
  a.inv_count = inv_count; a.slot = slot; a.row_query = row_query; a.row_batch = row_batch; a.row_ref = row_ref;
  a.q_rows = q_rows; a.q_rows2 = q_rows2; a.counters = counters;
  const int nblk = (d->Q + 255) / 256;
  a.block_counts = block_scratch;
  a.block_base = block_scratch + static_cast<long>(d->Nc) * nblk;
  // x = p * (hi - lo) + lo with the scale formed in double and rounded once, as the torch statement does
  // with its Python-float operands (encoder.py:102-107)
  a.sx = static_cast<float>(static_cast<double>(d->pc_range[3]) - static_cast<double>(d->pc_range[0]));
  a.sy = static_cast<float>(static_cast<double>(d->pc_range[4]) - static_cast<double>(d->pc_range[1]));
  a.sz = static_cast<float>(static_cast<double>(d->pc_range[5]) - static_cast<double>(d->pc_range[2]));
  a.ox = static_cast<float>(d->pc_range[0]); a.oy = static_cast<float>(d->pc_range[1]);
  a.oz = static_cast<float>(d->pc_range[2]);
  a.img_w = d->img_w; a.img_h = d->img_h;
  a.B = d->B; a.Nc = d->Nc; a.Q = d->Q; a.D = d->D; a.q_lo = d->q_lo; a.q_hi = d->q_hi; a.cap = d->row_capacity;
  hipLaunchKernelGGL(bevmsda::plan_project_kernel, dim3(static_cast<unsigned>(nblk)), dim3(256), 0, st, a);
  hipLaunchKernelGGL(bevmsda::plan_scan_kernel, dim3(1), dim3(256), 0, st, a, nblk);
  hipLaunchKernelGGL(bevmsda::plan_rows_kernel, dim3(static_cast<unsigned>(nblk * d->Nc)), dim3(256), 0, st, a, nblk);
  return hipGetLastError() == hipSuccess ? BEVMSDA_OK : BEVMSDA_ERR_LAUNCH;
}

int bevmsda_fold_extra_rows_f32(float *rows, int64_t ld_rows, const int32_t *q_rows, int64_t slots, int J, int C,
                                const int32_t *n_extra, void *stream) {
  if (slots < 0 || J < 0 || C <= 0 || ld_rows < C) return BEVMSDA_ERR_BAD_SHAPE;
  if (C % 4 != 0 || ld_rows % 4 != 0) return BEVMSDA_ERR_UNSUPPORTED;
  if (slots == 0 || J <= 2) return BEVMSDA_OK;
  if (!rows || !q_rows || !n_extra) return BEVMSDA_ERR_NULL_POINTER;
  if (misaligned16(rows)) return BEVMSDA_ERR_MISALIGNED;
  const long long total = slots * static_cast<long long>(C / 4);
  const long long nb = (total + 255) / 256;
  if (nb >= (1LL << 31)) return BEVMSDA_ERR_TOO_LARGE;
  hipLaunchKernelGGL(bevmsda::fold_extra_rows_kernel, dim3(static_cast<unsigned>(nb)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), rows, static_cast<long>(ld_rows), q_rows,
                     static_cast<long>(slots), J, C, n_extra);
  return hipGetLastError() == hipSuccess ? BEVMSDA_OK : BEVMSDA_ERR_LAUNCH;
}

}  // extern "C"
