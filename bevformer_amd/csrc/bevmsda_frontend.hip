// C ABI of libbevmsda.so, backward of the fused front end (declared in include/bevmsda.h): the two row passes
// of msda_frontend.h.  No torch, no allocation, no global state.
#include "../../include/bevmsda.h"
#include "msda_frontend.h"

namespace {
inline bool mis4(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 3u) != 0; }
inline bool mis8(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 7u) != 0; }

int front_common(const bevmsda_fused_desc *d, bevmsda::FrontArgs &f) {
  if (!d) return BEVMSDA_ERR_NULL_POINTER;
  if (d->R < 0 || d->M <= 0 || d->L <= 0 || d->P <= 0 || d->K <= 0 || d->A <= 0 || d->proj_row <= 0) return BEVMSDA_ERR_BAD_SHAPE;
  if ((d->P != 4 && d->P != 8) || d->L > 4 || d->K > 2 || (d->ref_mode != 0 && d->ref_mode != 1) ||
      d->proj_row % 2 != 0 || d->off_head % 2 != 0 || d->off_k % 2 != 0)
    return BEVMSDA_ERR_UNSUPPORTED;
  if (d->ref_mode == 1 && d->A < d->L) return BEVMSDA_ERR_BAD_SHAPE;
  if (d->R * d->K * d->M * d->P >= (1LL << 37)) return BEVMSDA_ERR_TOO_LARGE;
  f.R = d->R; f.proj_row = d->proj_row; f.M = d->M; f.L = d->L; f.P = d->P; f.Q = d->Q > 0 ? d->Q : 1; f.K = d->K;
  f.A = d->A; f.ref_mode = d->ref_mode; f.off_head = d->off_head; f.off_k = d->off_k; f.lg_head = d->lg_head;
  f.lg_k = d->lg_k; f.vmul = d->vmul; f.vadd = d->vadd;
  return BEVMSDA_OK;
}

template <bool CHAIN>
int front_launch(const bevmsda_fused_desc *d, const bevmsda::FrontArgs &f, void *stream) {
  const long long threads = static_cast<long long>(d->R) * d->M * d->K * d->P;
  const long long nb = (threads + 255) / 256;
  if (nb >= (1LL << 31)) return BEVMSDA_ERR_TOO_LARGE;
  const dim3 grid(static_cast<unsigned>(nb)), block(256);
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (d->P == 8) hipLaunchKernelGGL((bevmsda::frontend_kernel<8, CHAIN>), grid, block, 0, st, f);
  else hipLaunchKernelGGL((bevmsda::frontend_kernel<4, CHAIN>), grid, block, 0, st, f);
  return hipGetLastError() == hipSuccess ? BEVMSDA_OK : BEVMSDA_ERR_LAUNCH;
}
}  // namespace

extern "C" {

static int expand_launch(const float *offs, const float *logits, const float *ref, const int32_t *row_batch,
                         const int32_t *row_src, const int64_t *spatial_shapes, const bevmsda_fused_desc *desc,
                         float *loc, float *attn, int32_t *row_batch_k, void *stream, const int32_t *nrows);

int bevmsda_frontend_expand_f32(const float *offs, const float *logits, const float *ref, const int32_t *row_batch,
                                const int32_t *row_src, const int64_t *spatial_shapes, const bevmsda_fused_desc *desc,
                                float *loc, float *attn, int32_t *row_batch_k, void *stream) {
  return expand_launch(offs, logits, ref, row_batch, row_src, spatial_shapes, desc, loc, attn, row_batch_k, stream, nullptr);
}

int bevmsda_frontend_expand_rows_f32(const float *offs, const float *logits, const float *ref, const int32_t *row_batch,
                                     const int32_t *row_src, const int32_t *nrows, const int64_t *spatial_shapes,
                                     const bevmsda_fused_desc *desc, float *loc, float *attn, int32_t *row_batch_k,
                                     void *stream) {
  if (!nrows) return BEVMSDA_ERR_NULL_POINTER;
  if (desc && desc->K != 1) return BEVMSDA_ERR_UNSUPPORTED;     // (queue-major rows of K > 1 would need the count per entry)
  return expand_launch(offs, logits, ref, row_batch, row_src, spatial_shapes, desc, loc, attn, row_batch_k, stream, nrows);
}

static int expand_launch(const float *offs, const float *logits, const float *ref, const int32_t *row_batch,
                         const int32_t *row_src, const int64_t *spatial_shapes, const bevmsda_fused_desc *desc,
                         float *loc, float *attn, int32_t *row_batch_k, void *stream, const int32_t *nrows) {
  bevmsda::FrontArgs f{};
  const int rc = front_common(desc, f);
  if (rc != BEVMSDA_OK) return rc;
  f.nrows_dev = nrows;
  if (desc->R == 0) return BEVMSDA_OK;
  if (!offs || !logits || !ref || !spatial_shapes || !loc || !attn || !row_batch_k) return BEVMSDA_ERR_NULL_POINTER;
  if (mis8(offs) || mis4(logits) || mis8(ref) || mis8(loc) || mis4(attn) || mis4(row_batch_k)) return BEVMSDA_ERR_MISALIGNED;
  f.offs = offs; f.logits = logits; f.ref = ref; f.row_batch = row_batch; f.row_src = row_src; f.shapes = spatial_shapes;
  f.loc = loc; f.attn = attn; f.row_batch_k = row_batch_k;
  return front_launch<false>(desc, f, stream);
}

int bevmsda_frontend_chain_f32(const float *grad_loc, const float *grad_attn, const float *attn, const int32_t *row_src,
                               const int64_t *spatial_shapes, const bevmsda_fused_desc *desc, float *grad_offs,
                               float *grad_logits, void *stream) {
  bevmsda::FrontArgs f{};
  const int rc = front_common(desc, f);
  if (rc != BEVMSDA_OK) return rc;
  if (desc->R == 0) return BEVMSDA_OK;
  if (!grad_loc || !grad_attn || !attn || !spatial_shapes || !grad_offs || !grad_logits) return BEVMSDA_ERR_NULL_POINTER;
  if (mis8(grad_loc) || mis4(grad_attn) || mis4(attn) || mis8(grad_offs) || mis4(grad_logits)) return BEVMSDA_ERR_MISALIGNED;
  // offs / logits / ref are not read by the chain pass; the row pointers it forms from them must still be valid
  f.offs = grad_offs; f.logits = grad_logits; f.ref = grad_loc; f.row_src = row_src; f.shapes = spatial_shapes;
  f.grad_loc = grad_loc; f.grad_attn = grad_attn; f.attn_in = attn; f.grad_offs = grad_offs; f.grad_logits = grad_logits;
  if (desc->L == 1 || desc->L == 2 || desc->L == 4) {
    const long long threads = static_cast<long long>(desc->R) * desc->M * desc->K * desc->L * desc->P;
    const long long nb = (threads + 255) / 256;
    if (nb >= (1LL << 31)) return BEVMSDA_ERR_TOO_LARGE;
    const dim3 grid(static_cast<unsigned>(nb)), block(256);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (desc->P == 8) hipLaunchKernelGGL((bevmsda::frontend_chain_flat_kernel<8>), grid, block, 0, st, f);
    else hipLaunchKernelGGL((bevmsda::frontend_chain_flat_kernel<4>), grid, block, 0, st, f);
    return hipGetLastError() == hipSuccess ? BEVMSDA_OK : BEVMSDA_ERR_LAUNCH;
  }
  return front_launch<true>(desc, f, stream);
}

int bevmsda_frontend_chain_gather_f32(const float *grad_loc, const float *grad_attn, const float *attn,
                                      const int32_t *q_rows, int64_t slots, int J, const int32_t *n_extra,
                                      const int64_t *spatial_shapes, const bevmsda_fused_desc *desc, float *grad_offs,
                                      float *grad_logits, void *stream) {
  bevmsda::FrontArgs f{};
  const int rc = front_common(desc, f);
  if (rc != BEVMSDA_OK) return rc;
  if (slots < 0 || J < 0) return BEVMSDA_ERR_BAD_SHAPE;
  if (desc->K != 1 || !(desc->L == 1 || desc->L == 2 || desc->L == 4)) return BEVMSDA_ERR_UNSUPPORTED;
  if (slots == 0) return BEVMSDA_OK;
  if (!grad_loc || !grad_attn || !attn || !q_rows || !spatial_shapes || !grad_offs || !grad_logits) return BEVMSDA_ERR_NULL_POINTER;
  if (mis8(grad_loc) || mis4(grad_attn) || mis4(attn) || mis4(q_rows) || mis8(grad_offs) || mis4(grad_logits)) return BEVMSDA_ERR_MISALIGNED;
  f.shapes = spatial_shapes; f.grad_loc = grad_loc; f.grad_attn = grad_attn; f.attn_in = attn;
  f.grad_offs = grad_offs; f.grad_logits = grad_logits;
  const long long threads = static_cast<long long>(slots) * desc->M * desc->L * desc->P;
  const long long nb = (threads + 255) / 256;
  if (nb >= (1LL << 31)) return BEVMSDA_ERR_TOO_LARGE;
  const dim3 grid(static_cast<unsigned>(nb)), block(256);
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (desc->P == 8) hipLaunchKernelGGL((bevmsda::frontend_chain_gather_kernel<8>), grid, block, 0, st, f, q_rows, static_cast<long>(slots), J, n_extra);
  else hipLaunchKernelGGL((bevmsda::frontend_chain_gather_kernel<4>), grid, block, 0, st, f, q_rows, static_cast<long>(slots), J, n_extra);
  return hipGetLastError() == hipSuccess ? BEVMSDA_OK : BEVMSDA_ERR_LAUNCH;
}

}  // extern "C"
