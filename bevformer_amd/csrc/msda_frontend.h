// The modules' front end of the fused sampling kernels (msda_d32.h: softmax over the L*P logits of a
// (row, head, queue entry), `location = reference + offset / (W_l, H_l)`), as two stand-alone row passes
// for the BACKWARD of the fused forward:
//
//   frontend_expand_kernel : raw projection rows -> sampling locations (K*R, M, L, P, 2), attention weights
//                            (K*R, M, L, P) (row q*R + r = queue entry q of row r) and the value batch entry of each — the
//                            operands of the operator's backward kernels (msda_bwd_gather.h, msda_bwd_lds.h),
//                            recomputed instead of saved by the forward (141 MB per base SCA layer);
//   frontend_chain_kernel  : their grad_loc / grad_attn -> the gradient of the raw projection rows (softmax
//                            backward, 1 / (W_l, H_l)), accumulated over the rows that share a projection row
//                            (row_src: SpatialCrossAttention projects every BEV query once and each camera
//                            that sees it reads the same row).
//
// One thread per (row, head, queue entry, point); the P (4 or 8) threads of a softmax group are consecutive
// lanes and reduce with shuffles.  HBM-bound row passes (base SCA: 235 / 330 MB).  Reference statements:
// spatial_cross_attention.py:340-372, temporal_self_attention.py:209-229, 257-262.
#pragma once
#include "msda_d32.h"

namespace bevmsda {

struct FrontArgs {
  const float *offs, *logits, *ref;
  const int32_t *row_batch, *row_src;
  const int64_t *shapes;
  long R, proj_row;
  int M, L, P, Q, K, A, ref_mode, off_head, off_k, lg_head, lg_k, vmul, vadd;
  // expand
  float *loc, *attn;
  int32_t *row_batch_k;
  // chain
  const float *grad_loc, *grad_attn, *attn_in;
  float *grad_offs, *grad_logits;
  const int32_t *nrows_dev;         // expand: R is the capacity of the row arrays, the count is read here (K = 1)
};

template <int PT>
__device__ __forceinline__ float front_sum(float v) {
  v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64);
  if constexpr (PT == 8) v += __shfl_xor(v, 4, 64);
  return v;
}
template <int PT>
__device__ __forceinline__ float front_max(float v) {
  v = fmaxf(v, __shfl_xor(v, 1, 64)); v = fmaxf(v, __shfl_xor(v, 2, 64));
  if constexpr (PT == 8) v = fmaxf(v, __shfl_xor(v, 4, 64));
  return v;
}

template <int PT, bool CHAIN>
__global__ void __launch_bounds__(256) frontend_kernel(const FrontArgs f) {
  const long t = static_cast<long>(blockIdx.x) * 256 + threadIdx.x;
  const int pj = static_cast<int>(t % PT);
  long g = t / PT;                                  // (row, head, queue entry)
  const int q = static_cast<int>(g % f.K); g /= f.K;
  const int m = static_cast<int>(g % f.M);
  long r = g / f.M;
  long Ract = f.R;
  if (f.nrows_dev) {
    const long n = static_cast<long>(*f.nrows_dev);
    Ract = n < Ract ? (n < 0 ? 0 : n) : Ract;
  }
  const bool active = r < Ract;
  if (__builtin_amdgcn_ballot_w64(active) == 0) return;     // (device-side row count: the grid covers the capacity)
  if (!active) r = Ract - 1;                        // whole groups stay alive for the shuffles
  const long rs = f.row_src ? static_cast<long>(f.row_src[r]) : r;
  const long rk = static_cast<long>(q) * f.R + r;   // queue-major: the rows of one queue entry (one value batch entry per
                                                    // batch element) stay contiguous for the backward kernels
  const int L = f.L;
  const float *lgp = f.logits + rs * f.proj_row + m * f.lg_head + q * f.lg_k + pj;
  const float2 *ofp = reinterpret_cast<const float2 *>(f.offs + rs * f.proj_row + m * f.off_head + q * f.off_k) + pj;
  const float2 *rfp = reinterpret_cast<const float2 *>(f.ref) + (r * f.K + q) * f.A;
  const long o = ((rk * f.M + m) * L) * PT + pj;    // (rk, m, l, pj): + l * PT

  if constexpr (!CHAIN) {
    float e[4];
    float mx = -INFINITY;
#pragma unroll
    for (int l = 0; l < 4; ++l) {
      e[l] = l < L ? lgp[l * PT] : -INFINITY;
      mx = fmaxf(mx, e[l]);
    }
    mx = front_max<PT>(mx);
    float sum = 0.f;
#pragma unroll
    for (int l = 0; l < 4; ++l) {
      e[l] = expf(e[l] - mx);
      sum += e[l];
    }
    sum = front_sum<PT>(sum);
    if (!active) return;
#pragma unroll
    for (int l = 0; l < 4; ++l) {
      if (l >= L) break;
      const float W = static_cast<float>(f.shapes[2 * l + 1]), H = static_cast<float>(f.shapes[2 * l]);
      const float2 of = ofp[l * PT];
      const float2 rf = rfp[f.ref_mode == 0 ? pj % f.A : l];
      reinterpret_cast<float2 *>(f.loc)[o + static_cast<long>(l) * PT] = make_float2(rf.x + of.x / W, rf.y + of.y / H);
      f.attn[o + static_cast<long>(l) * PT] = e[l] / sum;
    }
    if (m == 0 && pj == 0) {
      const long base = f.row_batch ? static_cast<long>(f.row_batch[r]) : r / f.Q;
      f.row_batch_k[rk] = static_cast<int32_t>(base * f.vmul + static_cast<long>(q) * f.vadd);
    }
  } else {
    float aw[4], ga[4];
    float dot = 0.f;
#pragma unroll
    for (int l = 0; l < 4; ++l) {
      aw[l] = (l < L && active) ? f.attn_in[o + static_cast<long>(l) * PT] : 0.f;
      ga[l] = (l < L && active) ? f.grad_attn[o + static_cast<long>(l) * PT] : 0.f;
      dot = fmaf(aw[l], ga[l], dot);
    }
    dot = front_sum<PT>(dot);
    if (!active) return;
    float *glg = f.grad_logits + rs * f.proj_row + m * f.lg_head + q * f.lg_k + pj;
    float *gof = f.grad_offs + rs * f.proj_row + m * f.off_head + q * f.off_k + pj * 2;
#pragma unroll
    for (int l = 0; l < 4; ++l) {
      if (l >= L) break;
      const float W = static_cast<float>(f.shapes[2 * l + 1]), H = static_cast<float>(f.shapes[2 * l]);
      const float2 gl = reinterpret_cast<const float2 *>(f.grad_loc)[o + static_cast<long>(l) * PT];
      const float glogit = aw[l] * (ga[l] - dot);
      const float gx = gl.x / W, gy = gl.y / H;
      if (f.row_src) {                               // several rows (cameras) share a projection row
        unsafeAtomicAdd(glg + l * PT, glogit);
        unsafeAtomicAdd(gof + l * PT * 2, gx);
        unsafeAtomicAdd(gof + l * PT * 2 + 1, gy);
      } else {
        glg[l * PT] = glogit;
        *reinterpret_cast<float2 *>(gof + l * PT * 2) = make_float2(gx, gy);
      }
    }
  }
}

// Chain pass with one thread per (row, head, queue entry, LEVEL, point): the L * PT lanes of a softmax group sit
// side by side, so a wavefront's atomics / stores cover whole contiguous runs of the gradient row (the per-point
// form above spreads every instruction over 8 short runs: 218 us for the base SCA call, 3.5x this form).
// L * PT must be a power of two <= 32 (L = 1, 2, 4).
template <int PT>
__global__ void __launch_bounds__(256) frontend_chain_flat_kernel(const FrontArgs f) {
  const int LP = f.L * PT;
  const long t = static_cast<long>(blockIdx.x) * 256 + threadIdx.x;
  const int lp = static_cast<int>(t % LP);          // l * PT + pj
  long g = t / LP;
  const int q = static_cast<int>(g % f.K); g /= f.K;
  const int m = static_cast<int>(g % f.M);
  long r = g / f.M;
  const bool active = r < f.R;
  if (!active) r = f.R - 1;
  const long rs = f.row_src ? static_cast<long>(f.row_src[r]) : r;
  const long rk = static_cast<long>(q) * f.R + r;
  const long o = (rk * f.M + m) * LP + lp;
  const int l = lp / PT, pj = lp - l * PT;
  const float aw = active ? f.attn_in[o] : 0.f;
  const float ga = active ? f.grad_attn[o] : 0.f;
  float dot = aw * ga;
  for (int s = 1; s < LP; s <<= 1) dot += __shfl_xor(dot, s, 64);
  if (!active) return;
  const float W = static_cast<float>(f.shapes[2 * l + 1]), H = static_cast<float>(f.shapes[2 * l]);
  const float2 gl = reinterpret_cast<const float2 *>(f.grad_loc)[o];
  float *glg = f.grad_logits + rs * f.proj_row + m * f.lg_head + q * f.lg_k + lp;
  float *gof = f.grad_offs + rs * f.proj_row + m * f.off_head + q * f.off_k + lp * 2;
  const float glogit = aw * (ga - dot), gx = gl.x / W, gy = gl.y / H;
  (void)pj;
  if (f.row_src) {
    unsafeAtomicAdd(glg, glogit);
    unsafeAtomicAdd(gof, gx);
    unsafeAtomicAdd(gof + 1, gy);
  } else {
    *glg = glogit;
    *reinterpret_cast<float2 *>(gof) = make_float2(gx, gy);
  }
}

// Chain pass as a GATHER over the rows of a projection row (q_rows (slots, J) int32, -1 = none: the frame plan's
// table of the (camera, query) rows of every BEV query): one thread per (slot, head, level, point) walks the <= J
// rows of its slot and STORES the sums — no atomics, no zero fill of the gradient matrix, a fixed summation order.
// K = 1 (SpatialCrossAttention); L * PT a power of two <= 32.
template <int PT>
__global__ void __launch_bounds__(256) frontend_chain_gather_kernel(const FrontArgs f, const int32_t *__restrict__ q_rows,
                                                                   long slots, int J, const int32_t *__restrict__ n_extra) {
  const int LP = f.L * PT;
  // the plan's device-side count of slots with more than two rows: 0 (the usual frame) -> only the first two columns
  // of the table can hold rows (they fill in order), the walk over the other J - 2 is skipped
  const int Jw = (n_extra && J > 2 && *n_extra == 0) ? 2 : J;
  const long t = static_cast<long>(blockIdx.x) * 256 + threadIdx.x;
  const int lp = static_cast<int>(t % LP);
  long g = t / LP;
  const int m = static_cast<int>(g % f.M);
  long sl = g / f.M;
  const bool active = sl < slots;
  if (!active) sl = slots - 1;
  const int l = lp / PT;
  const float W = static_cast<float>(f.shapes[2 * l + 1]), H = static_cast<float>(f.shapes[2 * l]);
  float glogit = 0.f, gx = 0.f, gy = 0.f;
  for (int j = 0; j < Jw; ++j) {
    const int r = q_rows[sl * J + j];                // uniform over the LP lanes of a group
    if (r < 0) continue;
    const long o = (static_cast<long>(r) * f.M + m) * LP + lp;
    const float aw = f.attn_in[o], ga = f.grad_attn[o];
    float dot = aw * ga;
    for (int s = 1; s < LP; s <<= 1) dot += __shfl_xor(dot, s, 64);
    const float2 gl = reinterpret_cast<const float2 *>(f.grad_loc)[o];
    glogit += aw * (ga - dot);
    gx += gl.x / W;
    gy += gl.y / H;
  }
  if (!active) return;
  f.grad_logits[sl * f.proj_row + m * f.lg_head + lp] = glogit;
  *reinterpret_cast<float2 *>(f.grad_offs + sl * f.proj_row + m * f.off_head + lp * 2) = make_float2(gx, gy);
}

}  // namespace bevmsda
