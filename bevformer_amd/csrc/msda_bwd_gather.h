// grad_sampling_loc / grad_attn_weight of multi-scale deformable attention, D = 32 (the gather half of
// the second-generation backward; the scatter half is msda_bwd_lds.h).
//
// Math (the reference reaches it at multi_scale_deformable_attn_function.py:150-160):
//   grad_attn[r,m,l,p] = sum_c g[c] * bil(V, x, y)[c]
//   grad_loc_x         = W_l * A * sum_c g[c] * ( (1-fy) (v01 - v00) + fy (v11 - v10) )[c]
//   grad_loc_y         = H_l * A * sum_c g[c] * ( (1-fx) (v10 - v00) + fx (v11 - v01) )[c]
// with taps outside the map contributing 0 and points outside (-1, W) x (-1, H) contributing nothing.
//
// Same decomposition as the forward kernel (msda_d32.h): 8 lanes x 4 channels per (row, head), lane j
// owns point j of a level (tap byte offsets — an out-of-map tap gets the out-of-range offset, so its
// load returns 0 —, fractions), parameters broadcast with ds_swizzle, all taps of a batch of points in
// flight before the first FMA.  New here: each lane forms the four dot products g . v_tap over its 4
// channels, combines them into the three partial sums of a point, and the 8 x 3 partials of a level
// are reduced over the 8 lanes with a butterfly REDUCE-SCATTER (12 + 6 + 3 swizzles instead of
// 8 x 3 x 3): lane j ends up with the channel sums of point j and writes one coalesced record.
#pragma once
#include "msda_d32.h"
#include "scalar_ops.h"

namespace bevmsda {

struct GradPointParams {
  uint32_t o00, o01, o10, o11;   // byte offsets of the four taps (kOobOffset: contributes 0)
  float fx, fy;
};

__device__ __forceinline__ GradPointParams grad_point_params(float lx, float ly, int H, int W,
                                                             uint32_t level_base, uint32_t pix_bytes) {
  GradPointParams p;
  const float Wf = static_cast<float>(W), Hf = static_cast<float>(H);
  const float x = lx * Wf - 0.5f, y = ly * Hf - 0.5f;
  const bool inside = (x > -1.f) && (y > -1.f) && (x < Wf) && (y < Hf);
  const float xf = floorf(x), yf = floorf(y);
  const int x0 = static_cast<int>(xf), y0 = static_cast<int>(yf);
  p.fx = x - xf;
  p.fy = y - yf;
  const bool x0ok = x0 >= 0, x1ok = x0 + 1 < W, y0ok = y0 >= 0, y1ok = y0 + 1 < H;
  const uint32_t o = level_base + static_cast<uint32_t>(y0 * W + x0) * pix_bytes;
  const uint32_t dyb = static_cast<uint32_t>(W) * pix_bytes;
  p.o00 = (inside && y0ok && x0ok) ? o : kOobOffset;
  p.o01 = (inside && y0ok && x1ok) ? o + pix_bytes : kOobOffset;
  p.o10 = (inside && y1ok && x0ok) ? o + dyb : kOobOffset;
  p.o11 = (inside && y1ok && x1ok) ? o + dyb + pix_bytes : kOobOffset;
  return p;
}

// (the differences of a point go through sub_scalar: scalar_ops.h says why)
__device__ __forceinline__ float dot4(const f32x4 &a, const f32x4 &b) {
  return fmaf(a[3], b[3], fmaf(a[2], b[2], fmaf(a[1], b[1], a[0] * b[0])));
}

// points J0 .. J0 + CNT - 1 of the level: broadcast, load, partial sums into pa / px / py
template <int J0, int j, int CNT, typename T>
struct GradPoints {
  static __device__ __forceinline__ void issue(const GradPointParams &p, __amdgpu_buffer_rsrc_t r, uint32_t lane_term,
                                               f32x4 (&v)[CNT][4], float (&fx)[CNT], float (&fy)[CNT]) {
    constexpr int J = J0 + j;
    fx[j] = bcast8<J>(p.fx);
    fy[j] = bcast8<J>(p.fy);
    v[j][0] = TapLoad<T>::load(r, bcast8<J>(p.o00) + lane_term);
    v[j][1] = TapLoad<T>::load(r, bcast8<J>(p.o01) + lane_term);
    v[j][2] = TapLoad<T>::load(r, bcast8<J>(p.o10) + lane_term);
    v[j][3] = TapLoad<T>::load(r, bcast8<J>(p.o11) + lane_term);
    if constexpr (j + 1 < CNT) GradPoints<J0, j + 1, CNT, T>::issue(p, r, lane_term, v, fx, fy);
  }
};

template <int J0, int CNT, typename T, int PT>
__device__ __forceinline__ void grad_points(const GradPointParams &p, __amdgpu_buffer_rsrc_t r, uint32_t lane_term,
                                            const f32x4 &g, float (&pa)[PT], float (&px)[PT], float (&py)[PT]) {
  f32x4 v[CNT][4];
  float fx[CNT], fy[CNT];
  GradPoints<J0, 0, CNT, T>::issue(p, r, lane_term, v, fx, fy);
#pragma unroll
  for (int j = 0; j < CNT; ++j) {
    const float s00 = dot4(g, v[j][0]), s01 = dot4(g, v[j][1]), s10 = dot4(g, v[j][2]), s11 = dot4(g, v[j][3]);
    const float gx = sub_scalar(1.f, fx[j]), gy = sub_scalar(1.f, fy[j]);
    pa[J0 + j] = gy * (gx * s00 + fx[j] * s01) + fy[j] * (gx * s10 + fx[j] * s11);
    px[J0 + j] = gy * sub_scalar(s01, s00) + fy[j] * sub_scalar(s11, s10);
    py[J0 + j] = gx * sub_scalar(s10, s00) + fx[j] * sub_scalar(s11, s01);
  }
}

// butterfly reduce-scatter over the 8 lanes of a group: on return lane `lig` holds in v[0] the sum
// over the group of the input element v[lig % PT]
template <int PT>
__device__ __forceinline__ float reduce_scatter8(float (&v)[PT], int lig) {
  float w[4];
  if constexpr (PT == 8) {
    const bool up = lig & 4;
#pragma unroll
    for (int p = 0; p < 4; ++p) w[p] = (up ? v[p + 4] : v[p]) + xor8<4>(up ? v[p] : v[p + 4]);
  } else {
#pragma unroll
    for (int p = 0; p < 4; ++p) w[p] = v[p] + xor8<4>(v[p]);
  }
  float u[2];
  {
    const bool up = lig & 2;
#pragma unroll
    for (int p = 0; p < 2; ++p) u[p] = (up ? w[p + 2] : w[p]) + xor8<2>(up ? w[p] : w[p + 2]);
  }
  const bool up = lig & 1;
  return (up ? u[1] : u[0]) + xor8<1>(up ? u[0] : u[1]);
}

template <typename T, int PT, int WPE>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WPE, WPE)))
msda_gradloc_d32_kernel(const KArgs a) {
  constexpr int D = 32, LPG = 8, GPB = 256 / LPG;
  static_assert(PT == 4 || PT == 8, "PT");
  const int lig = threadIdx.x & 7;
  const long NQ = effective_rows(a);
  const int lb = logical_block_rows(a, NQ);
  if (lb < 0) return;                                        // (device-side row count: a block beyond the actual rows)
  const long G = static_cast<long>(lb) * GPB + (threadIdx.x >> 3);
  long nq; int m;
  map_group(G, a, nq, m);
  const bool active = nq < NQ;
  if (__builtin_amdgcn_ballot_w64(active) == 0) return;     // (device-side row count: the grid covers the capacity)
  if (!active) nq = NQ - 1;         // whole groups stay alive for the swizzles; nothing is stored
  const int L = a.L;
  const long n = a.row_batch ? static_cast<long>(a.row_batch[nq]) : nq / a.Q;
  const long row = nq * a.M + m;
  const uint32_t pix_bytes = static_cast<uint32_t>(a.M) * D * sizeof(T);
  const uint32_t head_base = static_cast<uint32_t>((static_cast<unsigned long long>(n) * a.S * a.M + m) * D * sizeof(T));
  const uint32_t lane_term = lig * 4 * static_cast<uint32_t>(sizeof(T));
  const uint32_t total_bytes = static_cast<uint32_t>(static_cast<unsigned long long>(a.N) * a.S * pix_bytes);
  __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(a.value), 0,
                                                                  static_cast<int>(total_bytes), 0x00020000);
  const int pj = lig % PT;          // the point this lane owns (PT = 4: both halves of the group own 0..3)
  const float2 *__restrict__ lp = loc_records(a, nq, m) + pj;       // (locations, or raw offsets + the reference point below)
  const float2 lrf = loc_reference(a, nq, pj);
  const float *__restrict__ ap = a.attn + row * L * PT + pj;
  float2 *__restrict__ glp = reinterpret_cast<float2 *>(a.grad_loc) + row * L * PT + pj;
  float *__restrict__ gap = a.grad_attn + row * L * PT + pj;

  f32x4 g;
  {
    float t[4];
    const long grow = a.gout_rows > 0 ? (nq % a.gout_rows) * a.M + m : row;
    Io<T, 4>::load(static_cast<const T *>(a.grad_out) + grow * D + lig * 4, t);
    const float gs = a.gout_rows > 0 ? a.gout_scale : 1.f;
    g[0] = t[0] * gs; g[1] = t[1] * gs; g[2] = t[2] * gs; g[3] = t[3] * gs;
  }
  float2 xy = lp[0];
  float aw = ap[0];
  for (int l = 0; l < L; ++l) {
    const int H = static_cast<int>(a.shapes[2 * l]), W = static_cast<int>(a.shapes[2 * l + 1]);
    const uint32_t lbytes = static_cast<uint32_t>(a.lstart[l]) * pix_bytes;
    const float2 at = loc_of_record(a, xy, lrf, H, W);
    const GradPointParams p = grad_point_params(at.x, at.y, H, W, head_base + lbytes, pix_bytes);
    const float aw_l = aw;
    if (l + 1 < L) {                // next level's record travels under this level's taps
      xy = lp[(l + 1) * PT];
      aw = ap[(l + 1) * PT];
    }
    float pa[PT], px[PT], py[PT];
    if constexpr (PT == 8) {
      grad_points<0, 4, T, PT>(p, rsrc, lane_term, g, pa, px, py);
      grad_points<4, 4, T, PT>(p, rsrc, lane_term, g, pa, px, py);
    } else {
      grad_points<0, 4, T, PT>(p, rsrc, lane_term, g, pa, px, py);
    }
    const float ga = reduce_scatter8<PT>(pa, lig);
    const float gx = reduce_scatter8<PT>(px, lig);
    const float gy = reduce_scatter8<PT>(py, lig);
    if (active && lig < PT) {
      gap[l * PT] = ga;
      glp[l * PT] = make_float2(gx * aw_l * static_cast<float>(W), gy * aw_l * static_cast<float>(H));
    }
  }
}

// ------------------------------------------------------------------ bf16 value storage, 16-byte lanes
// The gather kernel above with TapLoad<bf16_t> fetches 8 bytes per lane and tap: half the bytes of fp32 but the same
// number of lane requests, and the kernel is bound by requests (msda_d32.h, K1b).  Here, as in
// msda_fused_d32_bf16x8_kernel, a lane fetches 16 bytes = 8 channels, four lanes cover a tap, and the two halves of
// an 8-lane group fetch the two x-adjacent taps of a bilinear footprint in ONE instruction (a second one for the
// row below): 2 requests per point instead of 4.  A lane then holds, per point, the dot products of its 8 channels
// of grad_out with its top and bottom tap (t, b); its share of the three sums of the point is linear in (t, b)
//   lower half (x0):      attn  gy gx t + fy gx b     d/dx  -(gy t + fy b)     d/dy  gx (b - t)
//   upper half (x0 + 1):  attn  gy fx t + fy fx b     d/dx  +(gy t + fy b)     d/dy  fx (b - t)
// so the same butterfly reduce-scatter over the 8 lanes finishes the job.
__device__ __forceinline__ float dot8_bf16(const float (&g)[8], const uint4 &w) {
  float s = g[0] * bf16_lo(w.x);
  s = fmaf(g[1], bf16_hi(w.x), s);
  s = fmaf(g[2], bf16_lo(w.y), s); s = fmaf(g[3], bf16_hi(w.y), s);
  s = fmaf(g[4], bf16_lo(w.z), s); s = fmaf(g[5], bf16_hi(w.z), s);
  s = fmaf(g[6], bf16_lo(w.w), s); s = fmaf(g[7], bf16_hi(w.w), s);
  return s;
}

template <int J0, int j, int CNT>
struct GradPointsB8 {
  static __device__ __forceinline__ void issue(const GradPointParams &p, __amdgpu_buffer_rsrc_t r, uint32_t lane_term,
                                               bool upper, uint4 (&v)[CNT][2], float (&fx)[CNT], float (&fy)[CNT]) {
    constexpr int J = J0 + j;
    fx[j] = bcast8<J>(p.fx);
    fy[j] = bcast8<J>(p.fy);
    const uint32_t o00 = bcast8<J>(p.o00), o01 = bcast8<J>(p.o01), o10 = bcast8<J>(p.o10), o11 = bcast8<J>(p.o11);
    // (an out-of-map tap carries kOobOffset: the load returns 0 and touches no line)
    v[j][0] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(r, static_cast<int>((upper ? o01 : o00) + lane_term), 0, 0));
    v[j][1] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(r, static_cast<int>((upper ? o11 : o10) + lane_term), 0, 0));
    if constexpr (j + 1 < CNT) GradPointsB8<J0, j + 1, CNT>::issue(p, r, lane_term, upper, v, fx, fy);
  }
};

template <int J0, int CNT, int PT>
__device__ __forceinline__ void grad_points_b8(const GradPointParams &p, __amdgpu_buffer_rsrc_t r, uint32_t lane_term,
                                               bool upper, const float (&g)[8], float (&pa)[PT], float (&px)[PT],
                                               float (&py)[PT]) {
  uint4 v[CNT][2];
  float fx[CNT], fy[CNT];
  GradPointsB8<J0, 0, CNT>::issue(p, r, lane_term, upper, v, fx, fy);
#pragma unroll
  for (int j = 0; j < CNT; ++j) {
    const float t = dot8_bf16(g, v[j][0]), b = dot8_bf16(g, v[j][1]);
    const float gy = sub_scalar(1.f, fy[j]);
    const float wx = upper ? fx[j] : sub_scalar(1.f, fx[j]);   // my x weight
    const float col = gy * t + fy[j] * b;                   // my column of the footprint, y-interpolated
    pa[J0 + j] = wx * col;
    px[J0 + j] = upper ? col : -col;
    py[J0 + j] = wx * sub_scalar(b, t);
  }
}

template <int PT, int WPE>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WPE, WPE)))
msda_gradloc_d32_bf16x8_kernel(const KArgs a) {
  constexpr int D = 32, LPG = 8, GPB = 256 / LPG;
  static_assert(PT == 4 || PT == 8, "PT");
  const int lig = threadIdx.x & 7;
  const bool upper = lig >= 4;
  const long NQ = effective_rows(a);
  const int lb = logical_block_rows(a, NQ);
  if (lb < 0) return;                                        // (device-side row count: a block beyond the actual rows)
  const long G = static_cast<long>(lb) * GPB + (threadIdx.x >> 3);
  long nq; int m;
  map_group(G, a, nq, m);
  const bool active = nq < NQ;
  if (__builtin_amdgcn_ballot_w64(active) == 0) return;
  if (!active) nq = NQ - 1;
  const int L = a.L;
  const long n = a.row_batch ? static_cast<long>(a.row_batch[nq]) : nq / a.Q;
  const long row = nq * a.M + m;
  const uint32_t pix_bytes = static_cast<uint32_t>(a.M) * D * 2;
  const uint32_t head_base = static_cast<uint32_t>((static_cast<unsigned long long>(n) * a.S * a.M + m) * D * 2);
  const uint32_t lane_term = (lig & 3) * 16;
  const uint32_t total_bytes = static_cast<uint32_t>(static_cast<unsigned long long>(a.N) * a.S * pix_bytes);
  __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(a.value), 0,
                                                                  static_cast<int>(total_bytes), 0x00020000);
  const int pj = lig % PT;
  const float2 *__restrict__ lp = loc_records(a, nq, m) + pj;       // (locations, or raw offsets + the reference point below)
  const float2 lrf = loc_reference(a, nq, pj);
  const float *__restrict__ ap = a.attn + row * L * PT + pj;
  float2 *__restrict__ glp = reinterpret_cast<float2 *>(a.grad_loc) + row * L * PT + pj;
  float *__restrict__ gap = a.grad_attn + row * L * PT + pj;

  float g[8];
  {
    const long grow = a.gout_rows > 0 ? (nq % a.gout_rows) * a.M + m : row;
    const uint4 t = *reinterpret_cast<const uint4 *>(static_cast<const bf16_t *>(a.grad_out) + grow * D + (lig & 3) * 8);
    const float gs = a.gout_rows > 0 ? a.gout_scale : 1.f;
    g[0] = bf16_lo(t.x) * gs; g[1] = bf16_hi(t.x) * gs; g[2] = bf16_lo(t.y) * gs; g[3] = bf16_hi(t.y) * gs;
    g[4] = bf16_lo(t.z) * gs; g[5] = bf16_hi(t.z) * gs; g[6] = bf16_lo(t.w) * gs; g[7] = bf16_hi(t.w) * gs;
  }
  float2 xy = lp[0];
  float aw = ap[0];
  for (int l = 0; l < L; ++l) {
    const int H = static_cast<int>(a.shapes[2 * l]), W = static_cast<int>(a.shapes[2 * l + 1]);
    const uint32_t lbytes = static_cast<uint32_t>(a.lstart[l]) * pix_bytes;
    const float2 at = loc_of_record(a, xy, lrf, H, W);
    const GradPointParams p = grad_point_params(at.x, at.y, H, W, head_base + lbytes, pix_bytes);
    const float aw_l = aw;
    if (l + 1 < L) {
      xy = lp[(l + 1) * PT];
      aw = ap[(l + 1) * PT];
    }
    float pa[PT], px[PT], py[PT];
    grad_points_b8<0, PT, PT>(p, rsrc, lane_term, upper, g, pa, px, py);
    const float ga = reduce_scatter8<PT>(pa, lig);
    const float gx = reduce_scatter8<PT>(px, lig);
    const float gy = reduce_scatter8<PT>(py, lig);
    if (active && lig < PT) {
      gap[l * PT] = ga;
      glp[l * PT] = make_float2(gx * aw_l * static_cast<float>(W), gy * aw_l * static_cast<float>(H));
    }
  }
}

}  // namespace bevmsda
