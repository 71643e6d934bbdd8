/* TEST INFRASTRUCTURE — plain-C CPU restatement of multi-scale deformable
 * attention, forward and backward.  Checker only: the product path never links
 * or calls this file (see oracle/README.md).
 *
 * Restates the operator the reference reaches through
 *   projects/mmdet3d_plugin/bevformer/modules/multi_scale_deformable_attn_function.py:118-124
 *   (ms_deform_attn_forward) and :150-160 (ms_deform_attn_backward),
 * whose implementation lives in mmcv-full==1.4.0 (docs/install.md:27) and is
 * not on disk.  Written from the published algorithm (Deformable DETR,
 * arXiv:2010.04159, eq. 3) as spelled out in SURVEY.md Appendix A:
 *
 *   out[n,q,m,c] = sum_{l,p} A[n,q,m,l,p] * bilinear(V_l[n,:,:,m,c], x, y)
 *   x = loc_x*W_l - 0.5, y = loc_y*H_l - 0.5, zero padding outside the map,
 *   a point contributes only if -1 < y < H_l and -1 < x < W_l.
 *
 * Parity pinning: checked in tests/test_oracle.py against the grid_sample
 * statement in oracle/bevformer_cpu.py (which is itself pinned against the
 * reference's own files) and against torch autograd for the gradients.
 *
 * Layouts (all row-major, as the reference passes them):
 *   value  (N,S,M,D)   shapes (L,2)=(H,W) int64   level_start (L,) int64
 *   loc    (N,Q,M,L,P,2)=(x,y)   attn (N,Q,M,L,P)   out / grad_out (N,Q,M*D)
 * Accumulation is double so the checker is tighter than what it checks.
 */
#include <math.h>
#include <stdint.h>
#include <stddef.h>

#define IDX_V(n, s, m, c) ((((size_t)(n) * S + (size_t)(s)) * M + (m)) * D + (c))

int msda_ref_forward_f32(const float *value, const int64_t *shapes, const int64_t *level_start,
                         const float *loc, const float *attn, int N, int S, int M, int D, int L,
                         int Q, int P, float *out) {
  if (N < 0 || S < 0 || M < 0 || D < 0 || L < 0 || Q < 0 || P < 0) return -1;
#pragma omp parallel for collapse(2) schedule(static)
  for (int n = 0; n < N; ++n)
    for (int q = 0; q < Q; ++q)
      for (int m = 0; m < M; ++m) {
        const size_t row = ((size_t)n * Q + q) * M + m;
        const float *lp = loc + row * L * P * 2;
        const float *ap = attn + row * L * P;
        float *op = out + row * D;
        for (int c = 0; c < D; ++c) {
          double acc = 0.0;
          for (int l = 0; l < L; ++l) {
            const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];
            const int64_t base = level_start[l];
            for (int p = 0; p < P; ++p) {
              const double x = (double)lp[(l * P + p) * 2] * W - 0.5;
              const double y = (double)lp[(l * P + p) * 2 + 1] * H - 0.5;
              if (!(y > -1 && x > -1 && y < H && x < W)) continue;
              const int x0 = (int)floor(x), y0 = (int)floor(y);
              const double fx = x - x0, fy = y - y0;
              double v = 0.0;
              if (y0 >= 0 && x0 >= 0) v += (1 - fy) * (1 - fx) * value[IDX_V(n, base + (int64_t)y0 * W + x0, m, c)];
              if (y0 >= 0 && x0 + 1 < W) v += (1 - fy) * fx * value[IDX_V(n, base + (int64_t)y0 * W + x0 + 1, m, c)];
              if (y0 + 1 < H && x0 >= 0) v += fy * (1 - fx) * value[IDX_V(n, base + (int64_t)(y0 + 1) * W + x0, m, c)];
              if (y0 + 1 < H && x0 + 1 < W) v += fy * fx * value[IDX_V(n, base + (int64_t)(y0 + 1) * W + x0 + 1, m, c)];
              acc += (double)ap[l * P + p] * v;
            }
          }
          op[c] = (float)acc;
        }
      }
  return 0;
}

/* Accumulates into the three caller-zeroed gradient buffers, like the
 * reference op (multi_scale_deformable_attn_function.py:146-160).
 * grad_value is accumulated in a double scratch owned by the caller
 * (size N*S*M*D) so that summation order does not matter to the checker;
 * pass NULL to accumulate directly in float.  One thread per batch entry
 * (scatter inside an entry stays sequential), which is fine for a checker. */
int msda_ref_backward_f32(const float *value, const int64_t *shapes, const int64_t *level_start,
                          const float *loc, const float *attn, const float *grad_out, int N, int S,
                          int M, int D, int L, int Q, int P, float *grad_value, float *grad_loc,
                          float *grad_attn, double *gv_scratch) {
  if (N < 0 || S < 0 || M < 0 || D < 0 || L < 0 || Q < 0 || P < 0) return -1;
  const size_t nv = (size_t)N * S * M * D;
  if (gv_scratch)
    for (size_t i = 0; i < nv; ++i) gv_scratch[i] = 0.0;
  /* batch entries own disjoint slices of every gradient buffer: one thread each */
#pragma omp parallel for schedule(dynamic, 1)
  for (int n = 0; n < N; ++n)
    for (int q = 0; q < Q; ++q)
      for (int m = 0; m < M; ++m) {
        const size_t row = ((size_t)n * Q + q) * M + m;
        const float *go = grad_out + row * D;
        for (int l = 0; l < L; ++l) {
          const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];
          const int64_t base = level_start[l];
          for (int p = 0; p < P; ++p) {
            const size_t pi = row * L * P + (size_t)l * P + p;
            const double x = (double)loc[pi * 2] * W - 0.5;
            const double y = (double)loc[pi * 2 + 1] * H - 0.5;
            if (!(y > -1 && x > -1 && y < H && x < W)) continue;
            const int x0 = (int)floor(x), y0 = (int)floor(y);
            const double fx = x - x0, fy = y - y0;
            const double a = attn[pi];
            double g_a = 0.0, g_x = 0.0, g_y = 0.0;
            for (int c = 0; c < D; ++c) {
              const double g = go[c];
              double v00 = 0, v01 = 0, v10 = 0, v11 = 0;
              if (y0 >= 0 && x0 >= 0) {
                const size_t i = IDX_V(n, base + (int64_t)y0 * W + x0, m, c);
                v00 = value[i];
                if (gv_scratch) gv_scratch[i] += a * g * (1 - fy) * (1 - fx);
                else grad_value[i] += (float)(a * g * (1 - fy) * (1 - fx));
              }
              if (y0 >= 0 && x0 + 1 < W) {
                const size_t i = IDX_V(n, base + (int64_t)y0 * W + x0 + 1, m, c);
                v01 = value[i];
                if (gv_scratch) gv_scratch[i] += a * g * (1 - fy) * fx;
                else grad_value[i] += (float)(a * g * (1 - fy) * fx);
              }
              if (y0 + 1 < H && x0 >= 0) {
                const size_t i = IDX_V(n, base + (int64_t)(y0 + 1) * W + x0, m, c);
                v10 = value[i];
                if (gv_scratch) gv_scratch[i] += a * g * fy * (1 - fx);
                else grad_value[i] += (float)(a * g * fy * (1 - fx));
              }
              if (y0 + 1 < H && x0 + 1 < W) {
                const size_t i = IDX_V(n, base + (int64_t)(y0 + 1) * W + x0 + 1, m, c);
                v11 = value[i];
                if (gv_scratch) gv_scratch[i] += a * g * fy * fx;
                else grad_value[i] += (float)(a * g * fy * fx);
              }
              g_a += g * ((1 - fy) * (1 - fx) * v00 + (1 - fy) * fx * v01 + fy * (1 - fx) * v10 + fy * fx * v11);
              g_x += g * a * ((1 - fy) * (v01 - v00) + fy * (v11 - v10));
              g_y += g * a * ((1 - fx) * (v10 - v00) + fx * (v11 - v01));
            }
            grad_attn[pi] += (float)g_a;
            grad_loc[pi * 2] += (float)(g_x * W);
            grad_loc[pi * 2 + 1] += (float)(g_y * H);
          }
        }
      }
  if (gv_scratch)
    for (size_t i = 0; i < nv; ++i) grad_value[i] += (float)gv_scratch[i];
  return 0;
}
