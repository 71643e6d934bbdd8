/* bevmsda — C ABI of the MI355X-native multi-scale deformable attention path.
 *
 * This header is the drop-in boundary for BEVFormer's BEV-encoder hot path.
 * The entry points replace the two functions the reference binds from
 * mmcv-full's `_ext` module
 *     ext_loader.load_ext('_ext', ['ms_deform_attn_backward', 'ms_deform_attn_forward'])
 *     (projects/mmdet3d_plugin/bevformer/modules/multi_scale_deformable_attn_function.py:10-12)
 * and are what `bevformer_amd/ext.py` (the `_ext`-shaped Python module) calls
 * through ctypes.  Plain pointers and sizes only: no torch types, no ownership
 * transfer, no allocation — the caller owns every buffer, exactly as in the
 * reference where Python allocates the gradient buffers
 * (multi_scale_deformable_attn_function.py:146-148).
 *
 * Conventions (identical to the reference op, ibid. :97-112):
 *   value          (N, S, M, D)        contiguous, S = sum_l H_l*W_l
 *   spatial_shapes (L, 2) int64        (H_l, W_l), DEVICE memory
 *   level_start    (L,)   int64        first row of level l in S, DEVICE memory
 *   loc            (N, Q, M, L, P, 2)  fp32 (x, y), normalised to [0,1]
 *   attn           (N, Q, M, L, P)     fp32
 *   out / grad_out (N, Q, M*D)
 * All device pointers must be 16-byte aligned (torch allocations are).
 * `stream` is a hipStream_t (NULL = the null stream).  Every function is
 * asynchronous with respect to the host, re-entrant, keeps no global state and
 * never throws; it returns BEVMSDA_OK or a negative error code.
 */
#ifndef BEVMSDA_H_
#define BEVMSDA_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BEVMSDA_ABI_VERSION 3

enum {
  BEVMSDA_OK = 0,
  BEVMSDA_ERR_NULL_POINTER = -1, /* a required pointer is NULL               */
  BEVMSDA_ERR_BAD_SHAPE = -2,    /* negative dim, or a dim of 0 where illegal */
  BEVMSDA_ERR_TOO_LARGE = -3,    /* S*M*D or Q*M*L*P*2 does not fit int32     */
  BEVMSDA_ERR_MISALIGNED = -4,   /* pointer not 16-byte aligned               */
  BEVMSDA_ERR_LAUNCH = -5,       /* hipLaunchKernel failed (see hipGetLastError) */
  BEVMSDA_ERR_BAD_OPTION = -6,   /* unknown tuning value                      */
  BEVMSDA_ERR_UNSUPPORTED = -7   /* shape not covered by this entry point: use the unfused one */
};

/* Optional launch tuning (benchmark sweeps).  Zero-initialise for defaults. */
typedef struct bevmsda_tuning {
  int32_t variant;   /* 0 = library default; 1 = generic lane-group kernels; 2 = one lane per
                        channel; 3/4/5 = D=32 forward sized for 4/8/2 waves per SIMD */
  int32_t qtile;     /* 0 = default; queries of one head handled by adjacent lane groups */
  int32_t xcd_remap; /* 0 = default, 1 = off, 2 = on: contiguous row ranges per XCD */
  int32_t reserved[5]; /* backward only: [0] = rows per workgroup of the grad_value sort kernel (0 = default; 64 / 128 /
                          256); [1], [2] = low / high word of a DEVICE address of 8 uint64 that receive the phase clocks
                          of that kernel (0 = off; tools/gvprof.py); [3] = 1: bf16 storage takes the 8-byte-lane gather
                          kernel instead of the 16-byte-lane one */
} bevmsda_tuning;

int bevmsda_abi_version(void);
const char *bevmsda_error_string(int code);

/* ms_deform_attn_forward (call site: multi_scale_deformable_attn_function.py:118-124).
 * Writes every element of `out`. */
int bevmsda_forward_f32(const float *value, const int64_t *spatial_shapes,
                        const int64_t *level_start, const float *loc, const float *attn,
                        int N, int S, int M, int D, int L, int Q, int P, float *out,
                        void *stream);

/* ms_deform_attn_backward (call site: ibid. :150-160).
 * grad_value is ACCUMULATED into (the caller zeroes it, ibid. :146);
 * grad_loc and grad_attn are fully overwritten. */
int bevmsda_backward_f32(const float *value, const int64_t *spatial_shapes,
                         const int64_t *level_start, const float *loc, const float *attn,
                         const float *grad_out, int N, int S, int M, int D, int L, int Q, int P,
                         float *grad_value, float *grad_loc, float *grad_attn, void *stream);

/* bf16 storage variants: value / out / grad_out are bfloat16 (uint16_t bit
 * patterns), sampling locations, attention weights and all arithmetic stay
 * fp32; grad_value is accumulated in fp32.  No reference semantics exist for
 * this (the reference always up-casts to fp32, ibid. :93) — tolerance is
 * stated against the fp32 oracle in tests/. */
int bevmsda_forward_bf16(const uint16_t *value, const int64_t *spatial_shapes,
                         const int64_t *level_start, const float *loc, const float *attn,
                         int N, int S, int M, int D, int L, int Q, int P, uint16_t *out,
                         void *stream);
int bevmsda_backward_bf16(const uint16_t *value, const int64_t *spatial_shapes,
                          const int64_t *level_start, const float *loc, const float *attn,
                          const uint16_t *grad_out, int N, int S, int M, int D, int L, int Q,
                          int P, float *grad_value, float *grad_loc, float *grad_attn,
                          void *stream);

/* Ragged batches: R query rows in total, row r samples value[row_batch[r]]
 * (row_batch: int32, DEVICE memory, values in [0,N)); loc is (R, M, L, P, 2),
 * attn (R, M, L, P), out / grad_out (R, M*D).  This is how the encoder runs
 * SpatialCrossAttention without the reference's zero-padded per-camera
 * rebatch (spatial_cross_attention.py:143-153: rows beyond a camera's hit
 * count are computed and thrown away there) — same results, fewer rows. */
int bevmsda_forward_ragged_f32(const float *value, const int64_t *spatial_shapes,
                               const int64_t *level_start, const float *loc, const float *attn,
                               const int32_t *row_batch, int N, int S, int M, int D, int L,
                               int R, int P, float *out, void *stream);
int bevmsda_backward_ragged_f32(const float *value, const int64_t *spatial_shapes,
                                const int64_t *level_start, const float *loc, const float *attn,
                                const int32_t *row_batch, const float *grad_out, int N, int S,
                                int M, int D, int L, int R, int P, float *grad_value,
                                float *grad_loc, float *grad_attn, void *stream);
int bevmsda_forward_ragged_bf16(const uint16_t *value, const int64_t *spatial_shapes,
                                const int64_t *level_start, const float *loc, const float *attn,
                                const int32_t *row_batch, int N, int S, int M, int D, int L,
                                int R, int P, uint16_t *out, void *stream);
int bevmsda_backward_ragged_bf16(const uint16_t *value, const int64_t *spatial_shapes,
                                 const int64_t *level_start, const float *loc, const float *attn,
                                 const int32_t *row_batch, const uint16_t *grad_out, int N, int S,
                                 int M, int D, int L, int R, int P, float *grad_value,
                                 float *grad_loc, float *grad_attn, void *stream);

/* bevmsda_backward_ragged_* with the row count in DEVICE memory (`nrows`, read when the kernels run; csrc/frame_plan.h
 * writes it): R is the CAPACITY of the row arrays (loc, attn, row_batch, grad_out, grad_loc, grad_attn), rows
 * [0, min(*nrows, R)) are processed, the others neither read nor written — no host synchronisation between the frame
 * plan and the backward of multi_scale_deformable_attn_function.py:130-163, so a training step can be captured in a
 * HIP graph.  `grad_value_stride`: floats between two pixels of grad_value (0 = M * D, the dense (N, S, M, D) array;
 * larger, a multiple of 4: the pixel rows of a wider array — the gradients of the encoder layers' value projections of a
 * frame side by side, the operand of ONE input-gradient GEMM).
 * Second-generation kernels only: D = 32, P in {4, 8}, L <= 4, value < 2 GiB; else BEVMSDA_ERR_UNSUPPORTED. */
int bevmsda_backward_rows_f32(const float *value, const int64_t *spatial_shapes, const int64_t *level_start,
                              const float *loc, const float *attn, const int32_t *row_batch, const float *grad_out,
                              const int32_t *nrows, int N, int S, int M, int D, int L, int R, int P,
                              float *grad_value, int64_t grad_value_stride, float *grad_loc, float *grad_attn, void *stream);
int bevmsda_backward_rows_bf16(const uint16_t *value, const int64_t *spatial_shapes, const int64_t *level_start,
                               const float *loc, const float *attn, const int32_t *row_batch, const uint16_t *grad_out,
                               const int32_t *nrows, int N, int S, int M, int D, int L, int R, int P,
                               float *grad_value, int64_t grad_value_stride, float *grad_loc, float *grad_attn, void *stream);

/* bevmsda_backward_rows_* WITHOUT the sampling locations as an operand (round 6): when the forward was
 * bevmsda_fused_forward_rows_save_* with save_loc = NULL, the backward kernels recompute every location from what that
 * forward read, with its own two operations (spatial_cross_attention.py:357-372: reference + offset / (W_l, H_l)):
 *   loc(r, m, l, p) = ref[(r * A + p % A) * 2 + c] + offs[row_src[r] * proj_row + m * off_head + (l * P + p) * 2 + c] / (W_l, H_l)[c]
 * (row_src NULL: row r itself; one queue entry, pillar-anchor references: the K = 1, ref_mode = 0 form of
 * bevmsda_fused_desc).  `attn` stays an operand (the forward's save_attn).  What multi_scale_deformable_attn_function.py
 * :94-128 keeps for backward shrinks from 12 to 4 bytes per sampling point (564 MB of a bevformer_base training step); the
 * step takes the same time — what the forward kernel gains without its location stores (32 us of 335 per layer) the backward
 * kernels spend on the row_src -> offset indirection (profiles/r6/r6t_fused_save_ab.txt, r6s_save_loc_cost.txt).
 * offs / ref 8-byte aligned, proj_row and off_head even; else as bevmsda_backward_rows_*. */
typedef struct bevmsda_loc_source {
  const float *offs;        /* sampling-offset projections (the fused forward's `offs`) */
  const float *ref;         /* reference points (R, A, 2), normalised (x, y) */
  const int32_t *row_src;   /* optional: projection row of operand row r */
  int64_t proj_row;         /* floats between two projection rows */
  int32_t off_head;         /* floats between two heads' offsets inside a row */
  int32_t A;                /* reference points per row: point p uses anchor p % A */
} bevmsda_loc_source;
int bevmsda_backward_rows_offs_f32(const float *value, const int64_t *spatial_shapes, const int64_t *level_start,
                                   const bevmsda_loc_source *loc_source, const float *attn, const int32_t *row_batch,
                                   const float *grad_out, const int32_t *nrows, int N, int S, int M, int D, int L, int R, int P,
                                   float *grad_value, int64_t grad_value_stride, float *grad_loc, float *grad_attn, void *stream);
int bevmsda_backward_rows_offs_bf16(const uint16_t *value, const int64_t *spatial_shapes, const int64_t *level_start,
                                    const bevmsda_loc_source *loc_source, const float *attn, const int32_t *row_batch,
                                    const uint16_t *grad_out, const int32_t *nrows, int N, int S, int M, int D, int L, int R, int P,
                                    float *grad_value, int64_t grad_value_stride, float *grad_loc, float *grad_attn, void *stream);

/* bevmsda_backward_* whose N * Q operand rows SHARE `grad_rows` rows of grad_out: row r reads grad_scale *
 * grad_out[r % grad_rows] — TemporalSelfAttention averages its queue entries (temporal_self_attention.py:257-262), so the
 * N = 2 entries of a query receive the same output gradient times 1 / 2; no repeated, pre-scaled copy of it is formed.
 * `grad_value_stride`: as in bevmsda_backward_rows_*.  Second-generation kernels only (D = 32, P in {4, 8}, L <= 4). */
int bevmsda_backward_shared_f32(const float *value, const int64_t *spatial_shapes, const int64_t *level_start, const float *loc,
                                const float *attn, const float *grad_out, int64_t grad_rows, float grad_scale, int N, int S, int M,
                                int D, int L, int Q, int P, float *grad_value, int64_t grad_value_stride, float *grad_loc,
                                float *grad_attn, void *stream);
int bevmsda_backward_shared_bf16(const uint16_t *value, const int64_t *spatial_shapes, const int64_t *level_start, const float *loc,
                                 const float *attn, const uint16_t *grad_out, int64_t grad_rows, float grad_scale, int N, int S,
                                 int M, int D, int L, int Q, int P, float *grad_value, int64_t grad_value_stride, float *grad_loc,
                                 float *grad_attn, void *stream);

/* Same as above with explicit tuning. */
int bevmsda_forward_f32_ex(const float *value, const int64_t *spatial_shapes,
                           const int64_t *level_start, const float *loc, const float *attn,
                           int N, int S, int M, int D, int L, int Q, int P, float *out,
                           void *stream, const bevmsda_tuning *tuning);
int bevmsda_backward_f32_ex(const float *value, const int64_t *spatial_shapes,
                            const int64_t *level_start, const float *loc, const float *attn,
                            const float *grad_out, int N, int S, int M, int D, int L, int Q,
                            int P, float *grad_value, float *grad_loc, float *grad_attn,
                            void *stream, const bevmsda_tuning *tuning);
int bevmsda_forward_bf16_ex(const uint16_t *value, const int64_t *spatial_shapes,
                            const int64_t *level_start, const float *loc, const float *attn,
                            int N, int S, int M, int D, int L, int Q, int P, uint16_t *out,
                            void *stream, const bevmsda_tuning *tuning);
int bevmsda_backward_bf16_ex(const uint16_t *value, const int64_t *spatial_shapes,
                             const int64_t *level_start, const float *loc, const float *attn,
                             const uint16_t *grad_out, int N, int S, int M, int D, int L, int Q,
                             int P, float *grad_value, float *grad_loc, float *grad_attn,
                             void *stream, const bevmsda_tuning *tuning);

/* Fused front end of the two attention modules (inference path).
 *
 * The reference computes, with separate elementwise launches per layer,
 *     attention_weights = softmax(logits over L*P)          spatial_cross_attention.py:340-348
 *                                                            temporal_self_attention.py:209-211
 *     sampling_locations = reference + offsets / (W_l, H_l)  spatial_cross_attention.py:357-372
 *                                                            temporal_self_attention.py:224-229
 * and, in TemporalSelfAttention, the mean over the two BEV-queue entries after the
 * operator (temporal_self_attention.py:257-262).  This entry point takes the raw
 * projection outputs and does all of it inside the sampling kernel.
 *
 *   offs    sampling-offset projections, element (r, m, q, l, p, c) at
 *           offs[r*proj_row + m*off_head + q*off_k + (l*P + p)*2 + c]
 *   logits  attention logits, element (r, m, q, l, p) at
 *           logits[r*proj_row + m*lg_head + q*lg_k + l*P + p]      (softmax over l, p)
 *   ref     (R, K, A, 2) fp32 normalised reference points; ref_mode 0: point p uses
 *           anchor p % A (the pillar anchors of MSDeformableAttention3D), ref_mode 1:
 *           level l uses ref l (A = L)
 *   row_batch  optional (R,) int32: value batch entry base of row r (else r / Q)
 *   row_src    optional (R,) int32: row of offs / logits used by output row r (else r) —
 *              SpatialCrossAttention projects every BEV query once and each camera that
 *              sees the query reads the same projection row
 *   value batch entry of (row r, queue entry q) = base * vmul + q * vadd
 *   out     (R, M*D) = (1/K) * sum_q sample(value[entry(r, q)], loc(r, q), softmax(r, q))
 * Supported: D = 32, P in {4, 8}, 1 <= L <= 4, K in {1, 2} with P*K <= 8, value < 2 GiB; anything else
 * returns BEVMSDA_ERR_UNSUPPORTED and the caller uses bevmsda_forward_*.  Forward only. */
typedef struct bevmsda_fused_desc {
  int64_t R;          /* output rows */
  int64_t proj_row;   /* row stride of offs / logits, in floats (even) */
  int32_t N, S, M, D, L, P, Q;
  int32_t K, A, ref_mode;
  int32_t off_head, off_k, lg_head, lg_k;
  int32_t vmul, vadd;
  int32_t reserved[6];   /* [0]: 0 = default, 4 / 8 = kernel sized for 4 / 8 waves per SIMD; bf16 entry point
                            only: [1] = 1 selects the 8-byte-lane kernel instead of the 16-byte-lane one,
                            [2] = 1 (16-byte-lane kernel) makes `out` an fp32 (R, M*D) matrix;
                            [5] (fp32 entry points, benchmark knob): bodies with compile-time head / level counts —
                            0 = default (the 8-head, one-level, two-entry shape of TemporalSelfAttention at 128
                            registers), 1 = generic kernels only, 2 = that body at 64 registers, 3 = the 8-head,
                            4-level shape of SpatialCrossAttention specialised too (no gain: profiles/r5), 4 = TemporalSelfAttention's
                            shape on a resident, software-pipelined grid (3 % faster, twice the L2 misses: profiles/r6x), 5 =
                            that shape with the tile's tap lines staged in LDS (csrc/msda_d32.h, msda_fused_d32_tsa_lds_kernel): the
                            rows must be the cells of the sampled grid in raster order (one batch entry, no row_batch / row_src)
                            and [3] (static-row entry points) carries the HOST's copy of that grid's shape, (height << 16) | width,
                            which sizes the launch; anything else runs the default kernel */
} bevmsda_fused_desc;

int bevmsda_fused_forward_f32(const float *value, const int64_t *spatial_shapes,
                              const int64_t *level_start, const float *offs, const float *logits,
                              const float *ref, const int32_t *row_batch, const int32_t *row_src,
                              const bevmsda_fused_desc *desc, float *out, void *stream);
int bevmsda_fused_forward_bf16(const uint16_t *value, const int64_t *spatial_shapes,
                               const int64_t *level_start, const float *offs,
                               const float *logits, const float *ref, const int32_t *row_batch,
                               const int32_t *row_src, const bevmsda_fused_desc *desc,
                               uint16_t *out, void *stream);

/* The same entry points with the ROW COUNT in device memory: desc->R is the capacity of the row
 * arrays (row_batch, row_src, ref, out), `nrows` points to the actual count (int32, DEVICE memory,
 * e.g. counters[0] of bevmsda_frame_plan_f32), read by the kernels when they run.
 * desc->reserved[3] = the caller's HINT of the count (0 = none): rows below the hint are covered by
 * a launch with one workgroup per block of rows like the fixed-count entry point, rows beyond it by
 * a small strided launch — any count <= desc->R is computed correctly, the hint only sizes the
 * grids.  The launch geometry depends on (capacity, hint) only, so one captured HIP graph serves
 * every frame. */
int bevmsda_fused_forward_rows_f32(const float *value, const int64_t *spatial_shapes,
                                   const int64_t *level_start, const float *offs, const float *logits,
                                   const float *ref, const int32_t *row_batch, const int32_t *row_src,
                                   const int32_t *nrows, const bevmsda_fused_desc *desc, float *out,
                                   void *stream);
int bevmsda_fused_forward_rows_bf16(const uint16_t *value, const int64_t *spatial_shapes,
                                    const int64_t *level_start, const float *offs, const float *logits,
                                    const float *ref, const int32_t *row_batch, const int32_t *row_src,
                                    const int32_t *nrows, const bevmsda_fused_desc *desc, uint16_t *out,
                                    void *stream);

/* bevmsda_fused_forward_rows_* that also WRITE what the operator's backward reads (training forward of
 * SpatialCrossAttention's sampling: K = 1, P = 8, L >= 2; else BEVMSDA_ERR_UNSUPPORTED): save_loc (R, M, L, P, 2) = the
 * sampling locations reference + offset / (W_l, H_l) (spatial_cross_attention.py:357-372), save_attn (R, M, L, P) = the
 * softmax over the L * P logits of a head (:340-348) — the operands `sampling_locations` / `attention_weights` that
 * multi_scale_deformable_attn_function.py:94-128 saves for backward.  Rows [0, *nrows) are written.  The backward then
 * needs no bevmsda_frontend_expand_rows_f32 pass.  save_loc may be NULL: the locations are then not written and the
 * backward is bevmsda_backward_rows_offs_* (above), which recomputes them. */
int bevmsda_fused_forward_rows_save_f32(const float *value, const int64_t *spatial_shapes, const int64_t *level_start,
                                        const float *offs, const float *logits, const float *ref, const int32_t *row_batch,
                                        const int32_t *row_src, const int32_t *nrows, const bevmsda_fused_desc *desc, float *out,
                                        float *save_loc, float *save_attn, void *stream);
int bevmsda_fused_forward_rows_save_bf16(const uint16_t *value, const int64_t *spatial_shapes, const int64_t *level_start,
                                         const float *offs, const float *logits, const float *ref, const int32_t *row_batch,
                                         const int32_t *row_src, const int32_t *nrows, const bevmsda_fused_desc *desc,
                                         uint16_t *out, float *save_loc, float *save_attn, void *stream);

/* Backward of bevmsda_fused_forward_f32 in three steps (the autograd path of the modules; reference statements:
 * bevformer/modules/spatial_cross_attention.py:340-372, temporal_self_attention.py:209-229, 257-262 — what
 * autograd records there as softmax / divide / add / view nodes):
 *   1. bevmsda_frontend_expand_f32: the raw projection rows -> sampling locations `loc` (K*R, M, L, P, 2),
 *      attention weights `attn` (K*R, M, L, P) and `row_batch_k` (K*R) = the value batch entry of every (queue entry,
 *      row), QUEUE-MAJOR (row q*R + r): the operands of bevmsda_backward_ragged_f32 (or, with one batch element and
 *      one value batch entry per queue entry, of bevmsda_backward_f32 with N = K, Q = R), recomputed instead of saved
 *      by the forward;
 *   2. the operator's backward over those K*R rows with grad_out[q*R + r] = grad_out[r] / K;
 *   3. bevmsda_frontend_chain_f32: its grad_loc / grad_attn -> the gradient of the raw projection rows (softmax
 *      backward over the L*P logits of a (row, head, queue entry); 1 / (W_l, H_l) on the offsets).  With `row_src`
 *      (several rows share a projection row) the results are ADDED to grad_offs / grad_logits with fp32 atomics — the
 *      caller zeroes them; without it they are stored.  grad_offs / grad_logits address the gradient matrix exactly as
 *      offs / logits address the projection matrix (same desc->proj_row, off_head, off_k, lg_head, lg_k).
 * Supported: P in {4, 8}, L <= 4, K in {1, 2}; else BEVMSDA_ERR_UNSUPPORTED. */
int bevmsda_frontend_expand_f32(const float *offs, const float *logits, const float *ref, const int32_t *row_batch,
                                const int32_t *row_src, const int64_t *spatial_shapes, const bevmsda_fused_desc *desc,
                                float *loc, float *attn, int32_t *row_batch_k, void *stream);
/* Step 1 with the row count in device memory (K = 1): desc->R is the capacity, rows [0, min(*nrows, R)) are written. */
int bevmsda_frontend_expand_rows_f32(const float *offs, const float *logits, const float *ref, const int32_t *row_batch,
                                     const int32_t *row_src, const int32_t *nrows, const int64_t *spatial_shapes,
                                     const bevmsda_fused_desc *desc, float *loc, float *attn, int32_t *row_batch_k,
                                     void *stream);
int bevmsda_frontend_chain_f32(const float *grad_loc, const float *grad_attn, const float *attn, const int32_t *row_src,
                               const int64_t *spatial_shapes, const bevmsda_fused_desc *desc, float *grad_offs,
                               float *grad_logits, void *stream);
/* Step 3 as a gather (K = 1, L in {1, 2, 4}): `q_rows` (slots, J) int32 lists the rows that read projection row
 * `slot` (-1 = none; the frame plan's table, spatial_cross_attention.py:136-153 inverted); every element of the
 * gradient matrix rows [0, slots) is STORED (no atomics, no zeroing by the caller, fixed summation order).
 * `n_extra` (device, may be NULL): the frame plan's count of slots with more than two rows; when it reads 0 only the
 * first two columns of the table are walked (rows fill the columns in order). */
int bevmsda_frontend_chain_gather_f32(const float *grad_loc, const float *grad_attn, const float *attn,
                                      const int32_t *q_rows, int64_t slots, int J, const int32_t *n_extra,
                                      const int64_t *spatial_shapes, const bevmsda_fused_desc *desc, float *grad_offs,
                                      float *grad_logits, void *stream);

/* Row-wise helpers of the encoder layer (csrc/rowops.h), fp32, forward only.
 *
 * out = LayerNorm(x + res) * gamma + beta over the last dimension C (res may be NULL):
 * the "+ identity" of every attention / FFN step followed by the layer's norm
 * (encoder.py:360-404; torch.nn.LayerNorm semantics: biased variance, eps inside the
 * square root).  C must be 256, 512 or 1024. */
int bevmsda_add_layernorm_f32(const float *x, const float *res, const float *gamma,
                              const float *beta, float eps, int64_t rows, int C, float *out,
                              void *stream);

/* Backward of bevmsda_add_layernorm_f32 (what autograd records for `dropout(x) + identity` followed by
 * torch.nn.LayerNorm, encoder.py:360-404): grad_x (rows, C) = the gradient of BOTH x and res (z = x + res is
 * recomputed, the forward saves nothing); grad_gamma_beta (2, C) = [sum_rows g * xhat ; sum_rows g], written (not
 * accumulated).  `scratch`: bevmsda_add_layernorm_backward_partials(rows) * 2 * C floats (per-workgroup partial
 * column sums, reduced by a second small launch).  C in {256, 512}. */
int64_t bevmsda_add_layernorm_backward_partials(int64_t rows);
int bevmsda_add_layernorm_backward_f32(const float *x, const float *res, const float *gamma, const float *grad_out,
                                       float eps, int64_t rows, int C, float *grad_x, float *scratch,
                                       float *grad_gamma_beta, void *stream);
/* The same with the incoming gradient given as TWO addends, grad_out + grad_out2 (grad_out2 may be NULL): a residual
 * branch's gradient and a projection's input gradient meet in front of a LayerNorm without a separate add pass. */
int bevmsda_add_layernorm_backward2_f32(const float *x, const float *res, const float *gamma, const float *grad_out,
                                        const float *grad_out2, float eps, int64_t rows, int C, float *grad_x,
                                        float *scratch, float *grad_gamma_beta, void *stream);
/* out[q, :] = scale[q] * sum_{j<J, idx[q,j]>=0} rows[idx[q,j], :] — the per-camera
 * scatter-add and division by the camera count of SpatialCrossAttention
 * (spatial_cross_attention.py:165-172) as a gather.  idx: (Q, J) int32, -1 = empty. */
int bevmsda_gather_mean_f32(const float *rows, const int32_t *idx, const float *scale, int64_t Q,
                            int J, int C, float *out, void *stream);

/* Backward of bevmsda_gather_mean_f32 w.r.t. `rows`: rows[r, :] = scale[s] * slots[s, :] with s = row_slot[r] (the
 * inverse of idx: the frame plan's row -> BEV query table), for r in [0, min(*nrows, R)) (`nrows`: device-side row
 * count, may be NULL = R); other rows are not touched.  C a multiple of 4. */
int bevmsda_rows_from_slots_f32(const float *slots, int64_t ld_slots, const float *scale, const int32_t *row_slot,
                                const int32_t *nrows, int64_t R, int C, float *rows, void *stream);

/* out[r, :] = bf16(scale * in[r, :]) (round to nearest even) for r in [0, min(*nrows, R)) (`nrows` device-side, may be
 * NULL = R): the incoming gradient rows of the sampling backward converted to the bf16 storage type of its kernels
 * without touching the unused capacity of a device-side plan.  Dense (R, C) matrices, C a multiple of 8. */
int bevmsda_cast_rows_bf16(const float *in, const int32_t *nrows, int64_t R, int C, float scale, uint16_t *out, void *stream);

/* Dense projection on the matrix cores (csrc/linear_mfma.h), fp32 in / fp32 out, forward only:
 *
 *     y[m, n] = act( sum_k A[m, k] * w[n, k] + bias[n] ),   A = [ x0 (+ a0) | x1 (+ a1) ]
 *
 * = torch.nn.functional.linear, the value / sampling_offsets / attention_weights /
 * output_proj Linear layers and the FFN of the encoder layer
 * (temporal_self_attention.py:198,206-211,267; spatial_cross_attention.py:173,334-348;
 * custom_base_transformer_layer.py:157-158).  The optional second source x1 concatenates
 * along K in place (TemporalSelfAttention's cat([value[:bs], query + query_pos], -1),
 * temporal_self_attention.py:186-197); a0 / a1 are optional element-wise addends (query_pos).
 * precision 0: every fp32 operand is split into two bf16 terms and each product is
 * accumulated in fp32 from three bf16 MFMAs (>= 16 mantissa bits per product); precision 1:
 * operands rounded to bf16 (one MFMA), fp32 accumulation.
 * Requirements: K0, K1 multiples of 32 (K1 may be 0 with x1 NULL), input row strides
 * multiples of 4 floats, input pointers 16-byte aligned (y: any float pointer / stride);
 * otherwise BEVMSDA_ERR_UNSUPPORTED / _MISALIGNED and
 * the caller uses the library GEMM. */
typedef struct bevmsda_linear_desc {
  int64_t M;                                  /* rows of A and y */
  int64_t ldx0, lda0, ldx1, lda1, ldw, ldy;   /* row strides in floats */
  int32_t N, K0, K1;
  int32_t relu;                               /* 1: y = max(y, 0) after the bias */
  int32_t precision;                          /* 0 = split-fp32 (3 products), 1 = bf16 inputs */
  int32_t variant;                            /* 0 = library default; 1 = first kernel over the fp32 weight
                                                 (bevmsda_linear_f32), 13 = first kernel over the packed weight
                                                 image, 131 = software-pipelined kernel (packed weight, no
                                                 addends; BEVMSDA_ERR_UNSUPPORTED when it does not cover the call) */
  int32_t group_cols;                         /* 0, or a multiple of 128 dividing N: output column n
                                                 is written to matrix n / group_cols of
                                                 N / group_cols consecutive (M, ldy) matrices at y,
                                                 column n % group_cols — several Linear layers that
                                                 share their input (the value projections of all
                                                 encoder layers) in one pass over that input */
  int32_t out_bf16;                           /* 1: y is a bf16 matrix (the fp32 result rounded to
                                                 nearest even; ldy / group layout in elements) — the
                                                 projected value of the bf16-storage sampling path;
                                                 float4-epilogue variants only, N and ldy % 4 == 0 */
  int32_t reserved[4];                        /* [0] = 1: y += result instead of y = result (fp32 y, N and ldy multiples of
                                                 4, 16-byte aligned y / bias; bevmsda_linear_f32 / _packed_f32 only): the
                                                 input gradients of several projections of one tensor summed in place;
                                                 [1] = 1: the default never picks the software-pipelined kernel;
                                                 [2]: panel shape of bevmsda_linear_panel_f32 */
} bevmsda_linear_desc;

int bevmsda_linear_f32(const float *x0, const float *a0, const float *x1, const float *a1,
                       const float *w, const float *bias, const bevmsda_linear_desc *desc,
                       float *y, void *stream);

/* The same projection with the weight pre-split once per weight version (inference: the
 * weights are constants): bevmsda_linear_pack_weight_f32 writes, for every 128-row tile and
 * 32-deep K chunk of w (N, K), the [hi | lo] bf16 planes in the padded row format the kernel
 * keeps in LDS, bevmsda_linear_packed_bytes(N, K) bytes in all (0 when K % 32 != 0); the kernel
 * then copies 20 KB chunks instead of loading and splitting fp32 weights in every block.
 * Caller-owned blob, 16-byte aligned.  desc->ldw is ignored. */
int64_t bevmsda_linear_packed_bytes(int N, int K);
int bevmsda_linear_pack_weight_f32(const float *w, int64_t ldw, int N, int K, uint16_t *blob,
                                   void *stream);
/* The same image of the (N, K) weight whose TRANSPOSE is what lies in memory: wt (K, ldwt), element (n, k) = wt[k * ldwt + n].
 * The input gradient of a Linear layer, g W, is the projection of g by W^T: its weight image is packed straight from the
 * layer's own weight — no contiguous transpose per training step.  wt 4-byte aligned, ldwt >= N. */
int bevmsda_linear_pack_weight_t_f32(const float *wt, int64_t ldwt, int N, int K, uint16_t *blob, void *stream);
int bevmsda_linear_packed_f32(const float *x0, const float *a0, const float *x1, const float *a1,
                              const uint16_t *wpack, const float *bias,
                              const bevmsda_linear_desc *desc, float *y, void *stream);
/* Input gradient of a Linear that sits behind a ReLU, with the ReLU's backward in the epilogue:
 *     y[m, n] = act[m, n] > 0 ? sum_k g[m, k] * w[n, k] : 0
 * g (M, desc->ldx0) = the gradient w.r.t. the Linear's output (desc->K0 columns), wpack = the packed image of the
 * TRANSPOSED weight (N = the Linear's in_features rows, K0 = out_features columns), act (M, ld_act) = the ReLU's output
 * saved by the forward (mmcv FFN: Linear -> ReLU -> Linear, custom_base_transformer_layer.py:157-158).  fp32 y with N,
 * ldy, ld_act multiples of 4 and 16-byte aligned pointers; no bias, no second source. */
int bevmsda_linear_relu_backward_packed_f32(const float *g, const uint16_t *wpack, const float *act, int64_t ld_act,
                                            float scale, const bevmsda_linear_desc *desc, float *y, void *stream);
/* scale: multiplies the passed elements (1 / (1 - p) when a Dropout sat between the ReLU and the Linear and `act` is
 * the activation AFTER that dropout: zero where dropped or not positive); 1 otherwise. */

/* The packed projection with bevmsda_gather_mean_f32 folded into its A-load:
 *     A[m, :] = scale[m] * sum_{j < 2, idx[m, j] >= 0} rows[idx[m, j], :]      (idx: (M, 2) int32)
 * — SpatialCrossAttention's per-camera scatter-add, camera-count division and output_proj
 * (spatial_cross_attention.py:165-173) in one kernel; same arithmetic as the two-step path
 * (bit-identical result).  desc->K1 must be 0, desc->ldx0 is replaced by ld_rows. */
int bevmsda_linear_gather_packed_f32(const float *rows, int64_t ld_rows, const int32_t *idx,
                                     const float *scale, const uint16_t *wpack, const float *bias,
                                     const bevmsda_linear_desc *desc, float *y, void *stream);

/* Per-frame geometry of the encoder on the device (csrc/frame_plan.h): camera projection of the
 * pillar anchors + visibility (BEVFormerEncoder.point_sampling, encoder.py:88-149), the visible
 * (camera, query) pairs of SpatialCrossAttention as a ragged row list (spatial_cross_attention.py:
 * 136-153, without its per-camera nonzero() host syncs) and 1 / max(#cameras seeing a query, 1)
 * (ibid. :169-172).  Three small launches, no host synchronisation; the row count stays on the device.
 *
 *   lidar2img (B, Nc, 4, 4) fp32;  ref_3d (B, D, Q, 3) normalised anchors (encoder.py:62-71);
 *   order (Q,) int32: position -> BEV query, the row order inside a camera
 * outputs (caller-owned, all DEVICE memory):
 *   ref_cam (Nc, B, Q, D, 2), bev_mask (Nc, B, Q, D) bytes 0/1, inv_count (B, Q),
 *   slot (Nc, Q) scratch bytes, block_scratch (bevmsda_frame_plan_scratch(Nc, Q) int32), row_query / row_batch (row_capacity,) int32 (tile-local slot
 *   j*Qt + q - q_lo with Qt = q_hi - q_lo; value batch entry j*Nc + cam), row_ref (row_capacity, D, 2),
 *   q_rows (B*Qt, Nc) / q_rows2 (B*Qt, 2) int32 rows of every slot (-1 = none),
 *   counters (bevmsda_frame_plan_counters(B, Nc) int32): [0] rows, [1] rows dropped because
 *   row_capacity was too small, [2] slots seen by more than two cameras, [3] rows of one batch
 *   element, [4 ...] first row of every (j, cam) run (+ the end).
 * As in the reference the visible set of a camera is taken from batch element 0.  Only queries in
 * [q_lo, q_hi) produce rows (BEV tiling over GPUs); ref_cam / bev_mask / inv_count cover all. */
typedef struct bevmsda_plan_desc {
  int32_t B, Nc, Q, D;
  double pc_range[6];
  float img_w, img_h;
  int32_t q_lo, q_hi;
  int32_t row_capacity;
  int32_t reserved[5];
} bevmsda_plan_desc;

int64_t bevmsda_frame_plan_counters(int B, int Nc);
int64_t bevmsda_frame_plan_scratch(int Nc, int Q);
int bevmsda_frame_plan_f32(const float *lidar2img, const float *ref_3d, const int32_t *order,
                           const bevmsda_plan_desc *desc, float *ref_cam, uint8_t *bev_mask,
                           float *inv_count, uint8_t *slot, int32_t *block_scratch,
                           int32_t *row_query, int32_t *row_batch,
                           float *row_ref, int32_t *q_rows, int32_t *q_rows2, int32_t *counters,
                           void *stream);
/* rows[q_rows[s, 0]] += sum_{j >= 2} rows[q_rows[s, j]] for slots seen by more than two cameras, so
 * that the two-row gather of bevmsda_linear_gather_packed_f32 sums over all of them; n_extra
 * (DEVICE, counters[2] above) = 0 makes the launch a no-op. */
int bevmsda_fold_extra_rows_f32(float *rows, int64_t ld_rows, const int32_t *q_rows, int64_t slots, int J,
                                int C, const int32_t *n_extra, void *stream);

/* Residual add + LayerNorm that follow a projection (`output_proj` / the FFN's second Linear, "+ identity" and
 * `norms[i]` of an encoder layer: temporal_self_attention.py:267-272, spatial_cross_attention.py:173-175,
 * encoder.py:376-404), as the epilogue descriptor of bevmsda_linear_panel_f32:
 *     y = LayerNorm(A w^T + bias + res) * gamma + beta         over the N = 256 columns of every row
 * torch.nn.LayerNorm semantics (biased variance, eps inside the square root). */
typedef struct bevmsda_layernorm_desc {
  const float *res;      /* (M, ldres) or NULL */
  int64_t ldres;
  const float *gamma;    /* (N) */
  const float *beta;     /* (N) */
  float eps;
  int32_t reserved[3];
} bevmsda_layernorm_desc;

/* Row-panel form of the projections (csrc/linear_panel.h): the same contract and arithmetic as
 * bevmsda_linear_packed_f32 / _gather_packed_f32 / _layernorm_packed_f32 behind one entry point, for the
 * layer shapes K0 + K1 in {256, 512} (K0 in {256, 512}, K1 in {0, 256}).  A workgroup fetches a panel of 64 or
 * 128 complete rows once (LDS-DMA), splits it once, and its wavefronts sweep the N columns without further
 * synchronisation, weight fragments coming straight from L2 in MFMA operand order.  The weight image is its own
 * format: bevmsda_linear_panel_pack_weight_f32 writes bevmsda_linear_panel_packed_bytes(N, K) bytes (0 when
 * K % 256 != 0).  idx / scale: optional two-row gather (then a0 / x1 must be NULL); ln: optional residual +
 * LayerNorm epilogue (N = 256).  desc->reserved[2]: 0 = panel shape by problem shape, 1 = 64-row panels (4
 * wavefronts, 64 x 64 tiles, two workgroups per CU), 2 = 128-row panels (8 wavefronts, 128 x 32 tiles).
 * K = 512 and ln need N <= 256; N % 4 == 0, group_cols % 64 == 0, all pointers 16-byte aligned, row strides
 * multiples of 4; anything else returns BEVMSDA_ERR_UNSUPPORTED / _MISALIGNED and the caller uses the entry
 * points above.  The k order inside an MFMA differs from the first kernel's: results agree to fp32 summation
 * order, not bit for bit.  The epilogue addresses one ROW PANEL (<= 128 rows x ldy) of an output group through a 32-bit
 * raw buffer with a 64-bit base: 128 * ldy * element size must stay below 2 GiB (BEVMSDA_ERR_TOO_LARGE otherwise).  desc->reserved[3] is a BENCHMARK knob (0 in
 * production; tools/gemm_epilogue_ab.py, profiles/r5): 2 / 6 weight-fragment prefetch depth of the 64-row shape;
 * 32 + {1: finished tile stored one piece per k16 step, 2: prefetch depth 4, 3: both, 4: the round-4 epilogue};
 * 64 + n: phase skew of the column sweep (n x 1024 clocks); 97 / 98: one wavefront per SIMD with dripping stores. */
int64_t bevmsda_linear_panel_packed_bytes(int N, int K);
int bevmsda_linear_panel_pack_weight_f32(const float *w, int64_t ldw, int N, int K, uint16_t *blob, void *stream);
/* ... of the (N, K) weight whose transpose lies in memory: wt (K, ldwt), element (n, k) = wt[k * ldwt + n] (backward GEMMs). */
int bevmsda_linear_panel_pack_weight_t_f32(const float *wt, int64_t ldwt, int N, int K, uint16_t *blob, void *stream);
/* Many weight images in ONE launch (round 6; the reference re-reads its nn.Linear weights on every call, e.g.
 * temporal_self_attention.py:197-211 — the images are this library's derived copies and a training step rebuilds all of
 * them from the weights' current values: 52 launches at bevformer_base, one with this entry).  `jobs`: DEVICE array, kept
 * alive and unchanged by the caller while a launch that reads it can run (graph replays included); kind bit 0 = the
 * source is the memory of the transpose (w[k * ldw + n]), bit 1 = row-panel image (bevmsda_linear_panel_pack_weight*),
 * else the first kernel's (bevmsda_linear_pack_weight*); first_block = sum of the block counts
 * (bevmsda_linear_pack_job_blocks) of the jobs before; `blocks` = the total.  Per-job shape / alignment rules are those
 * of the single-image entry points and are the caller's to check. */
typedef struct bevmsda_pack_job {
  const float *w;
  int64_t ldw;
  uint16_t *blob;
  int32_t N, K;
  int32_t kind;
  int32_t first_block;
} bevmsda_pack_job;
int64_t bevmsda_linear_pack_job_blocks(int N, int K, int kind);
int bevmsda_linear_pack_weights_multi_f32(const bevmsda_pack_job *jobs, int njobs, int64_t blocks, void *stream);

int bevmsda_linear_panel_f32(const float *x0, const float *a0, const float *x1, const float *a1, const int32_t *idx,
                             const float *scale, const uint16_t *wpanel, const float *bias,
                             const bevmsda_linear_desc *desc, const bevmsda_layernorm_desc *ln, float *y,
                             void *stream);

/* bevmsda_linear_panel_f32 whose A has TWO ROW BLOCKS in two tensors: rows [0, m_split) = x_lo, rows [m_split, desc->M) =
 * x_hi (row m - m_split; both with row stride desc->ldx0; K1 = 0, no addends / gather / LayerNorm).  The value tensor of
 * TemporalSelfAttention is torch.stack([prev_bev, bev_query]) (encoder.py:216-222 of the reference): its projection reads the
 * history BEV and the current queries where they lie instead of from a stacked copy (82 MB written and read per frame). */
int bevmsda_linear_panel_rows2_f32(const float *x_lo, const float *x_hi, int64_t m_split, const uint16_t *wpanel, const float *bias,
                                   const bevmsda_linear_desc *desc, float *y, void *stream);

/* bevmsda_linear_panel_f32 over ROW SEGMENTS of which only some are needed (BEV tiling over GPUs, SURVEY.md §8e: the
 * camera-feature value projection is a replicated input, but a rank's queries see only some of the cameras): the rows
 * form ceil(M / seg_len) segments of seg_len rows (one per (batch entry, camera)); seg_start (segments + 1, int32, DEVICE
 * memory — the camera starts of bevmsda_frame_plan_f32's counters, read when the kernel runs) tells how many ragged
 * rows sample each segment.  The sampling kernels also ISSUE the taps whose bilinear coefficient is 0 (up to one image
 * row + 1 pixel before / after a level: rows of the neighbouring segment), and 0 x NaN is NaN, so the rows within
 * halo = max_l W_l + 1 of a used segment are computed too (level_shapes: (num_levels, 2) int64 [H, W], DEVICE memory,
 * the operator's spatial_shapes; NULL / 0 levels = no halo).  A workgroup all of whose rows (+- halo) lie in unused
 * segments returns at once and its output rows stay unwritten: nothing reads them.  No host synchronisation,
 * graph-capturable.  Single source, no addend. */
int bevmsda_linear_panel_segments_f32(const float *x0, const uint16_t *wpanel, const float *bias,
                                      const bevmsda_linear_desc *desc, const int32_t *seg_start, int64_t seg_len,
                                      const int64_t *level_shapes, int num_levels, float *y, void *stream);

/* The row-local tail of an encoder layer in one kernel (csrc/linear_chain.h):
 *     x = LayerNorm0(A w0^T + b0 + res)                          attention output projection, "+ identity", norm
 *     y = LayerNorm1(x + relu(x w1^T + b1) w2^T + b2)            FFN (C -> F -> C), "+ identity", norm
 * = SpatialCrossAttention's `output_proj` + residual (spatial_cross_attention.py:165-175; with idx / scale the
 * per-camera scatter-add and camera-count division as a two-row gather: A[m] = scale[m] * (rows[idx[m, 0]] +
 * rows[idx[m, 1]]), -1 = absent), `norms[1]`, the FFN and `norms[2]` of BEVFormerLayer's operation order
 * (encoder.py:376-404).  A workgroup owns 64 complete rows; x and the hidden layer never leave the chip.
 * w0p / w1p / w2p: bevmsda_linear_panel_pack_weight_f32 images of (C, C), (F, C), (C, F).  Supported: C = 256,
 * F = 512; row strides multiples of 4, every pointer 16-byte aligned; else BEVMSDA_ERR_UNSUPPORTED / _MISALIGNED and
 * the caller runs the three launches (bevmsda_linear_panel_f32 x 2 with LayerNorm descriptors + the first Linear).
 * Arithmetic as bevmsda_linear_panel_f32 (precision 0: three bf16 MFMAs per fp32 product; x and the hidden
 * activations are re-split from their fp32 values exactly as a separate launch would split them: same results as the
 * three-launch sequence to fp32 summation order). */
typedef struct bevmsda_chain_desc {
  int64_t M;                     /* rows of y */
  int64_t ld_rows, ld_res, ld_y; /* row strides in floats */
  int32_t C, F;                  /* embedding and hidden width */
  int32_t precision;             /* as bevmsda_linear_desc */
  float eps0, eps1;
  int32_t reserved[5];           /* [0]: bevmsda_proj_ln_proj_chain_f32: row stride of proj_out in floats; [1]: workgroup
                                    shape, 0 = default (by row count: 2, or 1 between 8,192 and 16,384 rows), 1 = 64-row panels (one workgroup per
                                    CU), 2 = 32-row panels (two), 3 = mixed: whole rounds of 256 x 64 rows on shape 1, the
                                    remaining rows on shape 2 (a second launch); [2]: bevmsda_proj_ffn_chain_f32 / _tail_f32: columns J of idx (0 = 2;
                                    2 < J <= 64: idx is (M, J), present rows first, and rows idx[m, 2..] >= 0 are added to
                                    row idx[m, 0]'s values, in column order, before the two-row sum — what
                                    bevmsda_fold_extra_rows_f32 does in a launch of its own) */
} bevmsda_chain_desc;

int bevmsda_proj_ffn_chain_f32(const float *rows, const int32_t *idx, const float *scale, const uint16_t *w0p, const float *b0,
                               const float *res, const float *gamma0, const float *beta0, const uint16_t *w1p, const float *b1,
                               const uint16_t *w2p, const float *b2, const float *gamma1, const float *beta1,
                               const bevmsda_chain_desc *desc, float *y, void *stream);

/* The same launch with the seam to the NEXT layer behind it (csrc/linear_chain.h TP): besides y the workgroup forms
 *     proj_out = [first | y + pos] w3^T + b3          (n3 columns, K = 512)
 * = the next BEVFormerLayer's TemporalSelfAttention `sampling_offsets` / `attention_weights` projection of
 * `cat([value[:bs], query + query_pos], -1)` (temporal_self_attention.py:197-211) with query = y, the rows this launch has
 * just produced: `first` (M, ld_first) = the history BEV's rows, `pos` (M, ld_pos) = the positional encoding or NULL,
 * w3p = the bevmsda_linear_panel_pack_weight_f32 image of the merged (n3, 512) weight, b3 (n3) or NULL.  Replaces one
 * bevmsda_linear_f32 two-source launch per layer (y is not re-read).  Supported: n3 a multiple of 64, <= 256; else
 * BEVMSDA_ERR_UNSUPPORTED and the caller runs the two launches.  Same arithmetic as that launch (split operands, fp32
 * accumulation); results agree to fp32 summation order. */
int bevmsda_proj_ffn_chain_tail_f32(const float *rows, const int32_t *idx, const float *scale, const uint16_t *w0p, const float *b0,
                                    const float *res, const float *gamma0, const float *beta0, const uint16_t *w1p, const float *b1,
                                    const uint16_t *w2p, const float *b2, const float *gamma1, const float *beta1,
                                    const bevmsda_chain_desc *desc, float *y, const float *first, int64_t ld_first,
                                    const float *pos, int64_t ld_pos, const uint16_t *w3p, const float *b3, int n3,
                                    float *proj_out, int64_t ld_proj, void *stream);

/* The same launch as the FORWARD of the autograd path: besides y it stores what the backward of the chain needs and the
 * inference launch keeps on chip — save_z0 (M, 256) = A w0^T + b0 + res (the input of LayerNorm0), save_x (M, 256) =
 * LayerNorm0(...), save_h (M, 512) = relu(x w1^T + b1), save_z1 (M, 256) = x + h w2^T + b2 (the input of LayerNorm1);
 * dense matrices, 16-byte aligned.  Same arithmetic, same y. */
int bevmsda_proj_ffn_chain_train_f32(const float *rows, const int32_t *idx, const float *scale, const uint16_t *w0p, const float *b0,
                                     const float *res, const float *gamma0, const float *beta0, const uint16_t *w1p,
                                     const float *b1, const uint16_t *w2p, const float *b2, const float *gamma1,
                                     const float *beta1, const bevmsda_chain_desc *desc, float *y, float *save_z0,
                                     float *save_x, float *save_h, float *save_z1, const float *drop0, const float *droph,
                                     const float *drop1, void *stream);

/* The BACKWARD of that launch in one kernel (linear_chain.h MODE 2; the rows are complete in a workgroup, so both LayerNorm
 * backwards are row-local like the forward's LayerNorms): from grad_y (M, ld_grad_y) and the saved save_z0 / save_h / save_z1
 *     grad_z1 = LayerNorm1'(save_z1; grad_y)                     grad_gamma_beta1 (2, 256) += [sum g xhat | sum g]
 *     grad_h  = (grad_z1 w2) where save_h > 0                     (M, 512)
 *     grad_z0 = LayerNorm0'(save_z0; grad_h w1 + grad_z1)         grad_gamma_beta0 (2, 256) += ...      (= the residual's gradient)
 *     grad_in = grad_z0 w0                                        (M, 256): the gradient of the seam's (gathered) input rows
 * i.e. encoder.py:376-404 / the mmcv FFN / spatial_cross_attention.py:173-175 differentiated.  w0t_p / w1t_p / w2t_p: the
 * row-panel weight images (bevmsda_linear_panel_pack_weight_f32) of w0^T (256 x 256), w1^T (256 x 512), w2^T (512 x 256).
 * grad_z1 / grad_h / grad_z0 are outputs because the weight gradients read them (bevmsda_linear_wgrad_multi_f32).  The
 * caller zeroes grad_gamma_beta*.  train() mode (the forward ran with drop0 / droph / drop1): drop1 (M, 256) scales grad_z1 on
 * its way into the FFN (grad_z1 then holds the scaled gradient; x still receives the unscaled one), hidden_scale = 1 / (1 - p)
 * of the hidden dropout (save_h is zero where it dropped), drop0 (M, 256) scales grad_z0 on its way into the projection:
 * grad_zp (M, 256) = grad_z0 * drop0 is stored and grad_in = grad_zp w0.  NULL / 1.0 / NULL: no dropout. */
int bevmsda_proj_ffn_chain_backward_f32(const float *grad_y, int64_t ld_grad_y, const float *save_z0, const float *save_h,
                                        const float *save_z1, const float *gamma0, const float *gamma1, const uint16_t *w0t_p,
                                        const uint16_t *w1t_p, const uint16_t *w2t_p, const bevmsda_chain_desc *desc,
                                        float *grad_z1, float *grad_h, float *grad_z0, float *grad_in, float *grad_gamma_beta1,
                                        float *grad_gamma_beta0, const float *drop0, const float *drop1, float hidden_scale,
                                        float *grad_zp, void *stream);

/* The backward of bevmsda_proj_ln_proj_chain_train_f32 in one kernel (linear_chain.h MODE 3):
 *     grad_z0 = LayerNorm0'(save_z0; grad_proj w1 + grad_x)     grad_gamma_beta0 (2, 256) += [sum g xhat | sum g]
 *     grad_in = grad_z0 w0
 * grad_proj (M, ld_grad_proj) with desc->reserved[0] = N2 columns (a multiple of 256, <= 768: N2 / 256 panel passes), grad_x
 * (M, 256) or NULL: the gradient x received directly (it is the next attention's residual).  w0t_p / w1t_p: row-panel images
 * of w0^T (256 x 256) and w1^T (256 x N2).  temporal_self_attention.py:267-272 + the projections of
 * spatial_cross_attention.py:338-348 differentiated.  drop0 / grad_zp: as above (the attention's dropout of train() mode). */
int bevmsda_proj_ln_proj_chain_backward_f32(const float *grad_proj, int64_t ld_grad_proj, const float *grad_x, const float *save_z0,
                                            const float *gamma0, const uint16_t *w0t_p, const uint16_t *w1t_p,
                                            const bevmsda_chain_desc *desc, float *grad_z0, float *grad_in,
                                            float *grad_gamma_beta0, const float *drop0, float *grad_zp, void *stream);
/* drop0 (M, 256), droph (M, 512), drop1 (M, 256): dropout scale tensors (0 or 1 / (1 - p); NULL = inactive) of the three
 * nn.Dropout sites of the chain in train() mode — on the attention's projected output before "+ identity"
 * (spatial_cross_attention.py:175), on the FFN's hidden activations and on its output (mmcv FFN); save_h then holds the
 * hidden rows AFTER their dropout.  Gather form (idx) only when any is given. */

/* The attention-to-attention seam of a layer with the same machinery:
 *     x = LayerNorm0(A w0^T + b0 + res)        TemporalSelfAttention's output projection, "+ identity", norms[0]
 *     p = x w1^T + b1                           the next attention's projection of the same rows — SpatialCrossAttention's
 *                                               merged [sampling_offsets ; attention_weights] Linear (desc->F columns)
 * (temporal_self_attention.py:267-272, encoder.py:376-378, spatial_cross_attention.py:338-348).  x_out (M, ld_y) is stored
 * (it is the next attention's residual), p goes to proj_out (M, desc->reserved[0]); x is projected from its on-chip copy.
 * desc->F: a multiple of 32, at most 768; w1p: the panel image of (F, C).  Other requirements as above. */
int bevmsda_proj_ln_proj_chain_f32(const float *rows, const int32_t *idx, const float *scale, const uint16_t *w0p, const float *b0,
                                   const float *res, const float *gamma0, const float *beta0, const uint16_t *w1p, const float *b1,
                                   const bevmsda_chain_desc *desc, float *x_out, float *proj_out, void *stream);

/* ... and as the forward of the autograd path (plain rows, no gather): save_z0 (M, 256) = the input of LayerNorm0. */
int bevmsda_proj_ln_proj_chain_train_f32(const float *rows, const uint16_t *w0p, const float *b0, const float *res,
                                         const float *gamma0, const float *beta0, const uint16_t *w1p, const float *b1,
                                         const bevmsda_chain_desc *desc, float *x_out, float *proj_out, float *save_z0,
                                         const float *drop0, void *stream);      /* drop0: as above (temporal_self_attention.py:272) */

/* Weight / bias gradient of a Linear layer (csrc/wgrad_mfma.h), the TN form of the projection:
 *     grad_w[n, k] += sum_m g[m, n] * x[m, k]          grad_b[n] += sum_m g[m, n]      (grad_b may be NULL)
 * g (M, ldg) = gradient w.r.t. the layer's output (N columns), x (M, ldx) = the layer's input (K columns).
 * ACCUMULATES (fp32 atomics over row slices): the caller zeroes grad_w / grad_b.  precision as for
 * bevmsda_linear_f32 (0: three bf16 MFMAs per product over split operands, 1: operands rounded to bf16).
 * N, K, ldg, ldx multiples of 4, g / x 16-byte aligned; otherwise BEVMSDA_ERR_UNSUPPORTED / _MISALIGNED. */
int bevmsda_linear_wgrad_f32(const float *g, int64_t ldg, const float *x, int64_t ldx, int64_t M, int N, int K,
                             float *grad_w, int64_t ldgw, float *grad_b, int precision, void *stream);

/* Several weight gradients over the SAME M rows in one launch (the backward of a layer seam needs two or three at
 * once): tiles of all problems share one round of workgroups — longer row slices, the epilogue atomics paid once
 * instead of once per Linear.  Up to 8 problems; each as bevmsda_linear_wgrad_f32 (accumulating; grad_b may be NULL). */
typedef struct bevmsda_wgrad_problem {
  const float *g;       /* (M, ldg): gradient w.r.t. the Linear's output, N columns */
  int64_t ldg;
  const float *x;       /* (M, ldx): the Linear's input, K columns */
  int64_t ldx;
  int32_t N, K;
  float *grad_w;        /* (N, ldgw), accumulated into */
  int64_t ldgw;
  float *grad_b;        /* (N) or NULL */
} bevmsda_wgrad_problem;
int bevmsda_linear_wgrad_multi_f32(const bevmsda_wgrad_problem *probs, int nprob, int64_t M, int precision, int workgroups,
                                   int variant, void *stream);
/* workgroups: 0 = the library's choice (benchmark sweeps: a target count); variant: 0 = operands split once to bf16 planes
 * in LDS, fragments by the transposing LDS read (csrc/wgrad_tr.h), 1 = the first kernel's gathered fragments. */

/* The encoder's caller, PerceptionTransformer.get_bev_features (modules/transformer.py:104-200).
 *
 * bevmsda_rotate_bev_f32: dst = rotate(src) of an (H, W) grid of C-float rows (row p at
 * ptr + p * ld) with nearest sampling and zero fill: torchvision.transforms.functional.rotate
 * as called on prev_bev at transformer.py:146-156.  theta = the 2 x 3 inverse affine matrix in
 * torchvision's normalised form (row 0 divided by W/2, row 1 by H/2: `rescaled_theta` of
 * _gen_affine_grid), computed by the caller from (angle, center).  C = 256 or 512; src != dst.
 *
 * bevmsda_flatten_feats_f32: one feature level feat (bs, Nc, C, hw) -> rows [s0, s0 + hw) of
 * out (Nc, S, bs, C), adding cams_embeds[cam] (Nc, C; may be NULL) and then level_embed (C; may
 * be NULL) — transformer.py:165-184.  C a multiple of 64. */
int bevmsda_rotate_bev_f32(const float *src, int64_t ld_src, float *dst, int64_t ld_dst, int H, int W,
                           int C, const float *theta, void *stream);

/* The same with the six matrix values in DEVICE memory (`theta_dev`, fp32, read when the kernel runs): a captured
 * launch follows the ego pose of every replayed frame (detectors/bevformer.py:253-261 feeding transformer.py:146-156). */
int bevmsda_rotate_bev_dev_f32(const float *src, int64_t ld_src, float *dst, int64_t ld_dst, int H, int W, int C,
                               const float *theta_dev, void *stream);
int bevmsda_flatten_feats_f32(const float *feat, const float *cams_embeds, const float *level_embed,
                              float *out, int bs, int Nc, int C, int hw, int S, int s0, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* BEVMSDA_H_ */
