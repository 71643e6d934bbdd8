// Multi-scale deformable attention, head size D = 32 (every BEVFormer config):
// the forward sampling kernel of the BEV-encoder hot path for CDNA4 (gfx950).
//
// Same math as msda_kernels.h (operator reached by the
// reference at bevformer/modules/multi_scale_deformable_attn_function.py:118-124).
// What is specific to this kernel:
//
//   * lane group = 8 lanes x 4 channels (one 16-byte fp32 / 8-byte bf16 request per
//     lane and bilinear tap: one tap of one head is one contiguous 128 B / 64 B line);
//     a wavefront carries 8 (query row, head) pairs, a 256-thread block 32;
//   * the sampling parameters of a level's points are computed ONCE, lane j of a
//     group owning point j (floor, bilinear coefficients x attention weight,
//     validity folded into the coefficients, byte offset of the top-left tap), and
//     are then broadcast inside the group with ds_swizzle BROADCAST(8, j) — the
//     LDS crossbar without an address VGPR — instead of being recomputed by all 8
//     lanes (the generic kernel spends ~60 VALU instructions per point on that);
//   * value is read through ONE raw buffer descriptor over the whole tensor with
//     hardware bounds checking: a point outside the map gets the byte offset
//     0x80000000 (beyond num_records -> the load returns 0 and touches no cache
//     line), a tap outside the map keeps its in-range neighbour address and a
//     coefficient of exactly 0.  The inner loop therefore has no branches: all
//     16 requests of 4 points are issued back to back (16 x 1 KiB per wave in
//     flight) before the first FMA waits;
//   * blocks are dealt to XCDs in contiguous row ranges (msda_kernels.h).
//
// Exactness note: an out-of-map tap contributes fma(0, v, acc) with v a finite
// neighbouring feature value instead of being skipped; identical results for
// finite feature maps (the reference's CPU fallback, grid_sample with zero
// padding, multiplies by the out-of-range mask the same way).
#pragma once
#include "msda_kernels.h"

namespace bevmsda {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr uint32_t kOobOffset = 0x80000000u;  // > num_records of any eligible tensor
// cache-policy bits of the tap loads (A/B builds: -DBEVMSDA_TAP_AUX=1 sc0, 2 nt, 16 sc1; measured in round 6, see DESIGN K1f)
// diagnostic build only (-DBEVMSDA_DIAG_NO_SAVE_LOC=1): the training forward keeps its attention weights but not its sampling
// locations (wrong gradients by construction) — what the location stores cost the kernel
#ifndef BEVMSDA_DIAG_NO_SAVE_LOC
#define BEVMSDA_DIAG_NO_SAVE_LOC 0
#endif
#ifndef BEVMSDA_TAP_AUX
#define BEVMSDA_TAP_AUX 0
#endif

template <typename T> struct TapLoad;
template <> struct TapLoad<float> {
  static constexpr int kBytes = 16;
  static __device__ __forceinline__ f32x4 load(__amdgpu_buffer_rsrc_t r, uint32_t off) {
    auto v = __builtin_amdgcn_raw_buffer_load_b128(r, static_cast<int>(off), 0, BEVMSDA_TAP_AUX);
    f32x4 o;
    o[0] = __uint_as_float(v[0]); o[1] = __uint_as_float(v[1]);
    o[2] = __uint_as_float(v[2]); o[3] = __uint_as_float(v[3]);
    return o;
  }
};
template <> struct TapLoad<bf16_t> {
  static constexpr int kBytes = 8;
  static __device__ __forceinline__ f32x4 load(__amdgpu_buffer_rsrc_t r, uint32_t off) {
    auto v = __builtin_amdgcn_raw_buffer_load_b64(r, static_cast<int>(off), 0, BEVMSDA_TAP_AUX);
    f32x4 o;
    o[0] = bf16_lo(v[0]); o[1] = bf16_hi(v[0]); o[2] = bf16_lo(v[1]); o[3] = bf16_hi(v[1]);
    return o;
  }
};

template <int J>
__device__ __forceinline__ float bcast8(float v) {
  return __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(v), (J << 5) | 0x18));
}
template <int J>
__device__ __forceinline__ uint32_t bcast8(uint32_t v) {
  return static_cast<uint32_t>(__builtin_amdgcn_ds_swizzle(static_cast<int>(v), (J << 5) | 0x18));
}

// Sampling parameters of one point, held by the lane that owns the point.
struct PointParams {
  float k00, k01, k10, k11;  // bilinear coefficient x attention weight, 0 when the tap is outside
  uint32_t off;              // byte offset of value[n, level, y0, x0, m, 0], or kOobOffset
};

// (x0, y0): the footprint's top-left pixel, for callers that address the taps themselves (the LDS-tile kernel)
__device__ __forceinline__ PointParams point_params_xy(float lx, float ly, float aw, int H, int W,
                                                       uint32_t level_base, uint32_t pix_bytes, int &x0, int &y0) {
  PointParams p;
  const float Wf = static_cast<float>(W), Hf = static_cast<float>(H);
  const float x = lx * Wf - 0.5f, y = ly * Hf - 0.5f;
  const bool inside = (x > -1.f) && (y > -1.f) && (x < Wf) && (y < Hf);
  const float xf = floorf(x), yf = floorf(y);
  x0 = static_cast<int>(xf); y0 = static_cast<int>(yf);
  const float fx = x - xf, fy = y - yf;
  const float gx = 1.f - fx, gy = 1.f - fy;
  const bool x0ok = x0 >= 0, x1ok = x0 + 1 < W, y0ok = y0 >= 0, y1ok = y0 + 1 < H;
  // `inside` is false for NaN coordinates too: every coefficient is then an exact 0
  p.k00 = (inside && y0ok && x0ok) ? gy * gx * aw : 0.f;
  p.k01 = (inside && y0ok && x1ok) ? gy * fx * aw : 0.f;
  p.k10 = (inside && y1ok && x0ok) ? fy * gx * aw : 0.f;
  p.k11 = (inside && y1ok && x1ok) ? fy * fx * aw : 0.f;
  // y0*W + x0 >= -(W+1): two's-complement wrap keeps the sum right whenever the tap
  // itself is in range; otherwise the coefficient is 0 or the load is out of range
  const uint32_t o = level_base + static_cast<uint32_t>(y0 * W + x0) * pix_bytes;
  p.off = inside ? o : kOobOffset;
  return p;
}

__device__ __forceinline__ PointParams point_params(float lx, float ly, float aw, int H, int W,
                                                    uint32_t level_base, uint32_t pix_bytes) {
  int x0, y0;
  return point_params_xy(lx, ly, aw, H, W, level_base, pix_bytes, x0, y0);
}

// Broadcast the parameters of points J0 .. J0+CNT-1 (held by lanes J0.. of every
// group), issue their 4*CNT tap requests back to back, then accumulate.
template <int J0, int j, int CNT, typename T>
struct IssuePoints {
  static __device__ __forceinline__ void run(const PointParams &p, __amdgpu_buffer_rsrc_t r,
                                             uint32_t lane_term, uint32_t dxb, uint32_t dyb,
                                             f32x4 (&v)[CNT][4], float (&k)[CNT][4]) {
    constexpr int J = J0 + j;
    // p.off carries no lane term: every lane adds its own channel offset
    // (kOobOffset + lane_term stays beyond num_records < 2^31)
    const uint32_t o = bcast8<J>(p.off) + lane_term;
    k[j][0] = bcast8<J>(p.k00); k[j][1] = bcast8<J>(p.k01);
    k[j][2] = bcast8<J>(p.k10); k[j][3] = bcast8<J>(p.k11);
    v[j][0] = TapLoad<T>::load(r, o);
    v[j][1] = TapLoad<T>::load(r, o + dxb);
    v[j][2] = TapLoad<T>::load(r, o + dyb);
    v[j][3] = TapLoad<T>::load(r, o + dyb + dxb);
    if constexpr (j + 1 < CNT) IssuePoints<J0, j + 1, CNT, T>::run(p, r, lane_term, dxb, dyb, v, k);
  }
};

template <int J0, int CNT, typename T>
__device__ __forceinline__ void sample_points(const PointParams &p, __amdgpu_buffer_rsrc_t r,
                                              uint32_t lane_term, uint32_t dxb, uint32_t dyb,
                                              f32x4 &acc) {
  f32x4 v[CNT][4];
  float k[CNT][4];
  IssuePoints<J0, 0, CNT, T>::run(p, r, lane_term, dxb, dyb, v, k);
#pragma unroll
  for (int j = 0; j < CNT; ++j) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      acc[0] = fmaf(k[j][t], v[j][t][0], acc[0]);
      acc[1] = fmaf(k[j][t], v[j][t][1], acc[1]);
      acc[2] = fmaf(k[j][t], v[j][t][2], acc[2]);
      acc[3] = fmaf(k[j][t], v[j][t][3], acc[3]);
    }
  }
}

// PT = points per level (4 or 8).  WPE = waves per SIMD the register allocation is
// sized for: it is the knob that decides how many of a level's 4*PT tap requests hipcc
// keeps in flight per wave (8 -> 64 VGPRs, ~8 requests; 4 -> 128 VGPRs, ~24; 2 -> all 32).
template <typename T, int PT, int WPE>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WPE, WPE)))
msda_fwd_d32_kernel(const KArgs a) {
  constexpr int D = 32, LPG = 8, GPB = 256 / LPG;
  static_assert(PT == 4 || PT == 8, "PT");
  const int lig = threadIdx.x & 7;
  const long G = static_cast<long>(logical_block(a)) * GPB + (threadIdx.x >> 3);
  long nq; int m;
  map_group(G, a, nq, m);
  const bool active = nq < a.NQ;
  if (!active) nq = a.NQ - 1;       // whole groups stay alive for the swizzles; nothing is stored
  const int L = a.L;
  const long n = a.row_batch ? static_cast<long>(a.row_batch[nq]) : nq / a.Q;
  const long row = nq * a.M + m;
  const uint32_t pix_bytes = static_cast<uint32_t>(a.M) * D * sizeof(T);
  // byte offset of value[n, 0, m, 0]; the tensor is < 2 GiB (checked by the host side)
  const uint32_t head_base = static_cast<uint32_t>((static_cast<unsigned long long>(n) * a.S * a.M + m) * D * sizeof(T));
  const uint32_t lane_term = lig * 4 * static_cast<uint32_t>(sizeof(T));
  const uint32_t total_bytes = static_cast<uint32_t>(static_cast<unsigned long long>(a.N) * a.S * pix_bytes);
  __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(a.value), 0,
                                                                  static_cast<int>(total_bytes), 0x00020000);
  // lane j of the group owns point j of every level (lanes >= PT own nothing)
  const bool owner = lig < PT;
  const float2 *__restrict__ lp = reinterpret_cast<const float2 *>(a.loc) + row * L * PT + (owner ? lig : 0);
  const float *__restrict__ ap = a.attn + row * L * PT + (owner ? lig : 0);
  const bool live = active && owner;

  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  float2 xy = lp[0];
  float aw = ap[0];
  for (int l = 0; l < L; ++l) {
    const int H = static_cast<int>(a.shapes[2 * l]), W = static_cast<int>(a.shapes[2 * l + 1]);
    const uint32_t lbytes = static_cast<uint32_t>(a.lstart[l]) * pix_bytes;
    const uint32_t dyb = static_cast<uint32_t>(W) * pix_bytes;
    const PointParams p = point_params(xy.x, xy.y, live ? aw : 0.f, H, W, head_base + lbytes, pix_bytes);
    if (l + 1 < L) {                // next level's locations travel under this level's taps
      xy = lp[(l + 1) * PT];
      aw = ap[(l + 1) * PT];
    }
    sample_points<0, PT, T>(p, rsrc, lane_term, pix_bytes, dyb, acc);
  }
  if (active) {
    T *op = static_cast<T *>(a.out) + row * D + lig * 4;
    if constexpr (sizeof(T) == 4) {
      *reinterpret_cast<float4 *>(op) = make_float4(acc[0], acc[1], acc[2], acc[3]);
    } else {
      uint2 t;
      t.x = f32_to_bf16(acc[0]) | (f32_to_bf16(acc[1]) << 16);
      t.y = f32_to_bf16(acc[2]) | (f32_to_bf16(acc[3]) << 16);
      *reinterpret_cast<uint2 *>(op) = t;
    }
  }
}

// ------------------------------------------------------------------ fused front end
// The modules feed the operator with softmax(attention logits) and with
// reference point + offset / (W_l, H_l) (spatial_cross_attention.py:340-372,
// temporal_self_attention.py:209-229).  In the reference those are separate
// elementwise launches over (rows, M, L, P[, 2]) tensors; measured here they cost as
// much as the sampling itself (profiles/r1).  This kernel takes the RAW projection
// output instead (one row of the merged [offsets | logits] GEMM per query row) and does
// the softmax (lane j owns point j: 1 exp per level, 8-lane butterflies for max / sum)
// and the location arithmetic in its prologue; with K = 2 it also averages the two
// BEV-queue entries of TemporalSelfAttention (temporal_self_attention.py:257-262).
struct FusedArgs {
  KArgs k;              // value, shapes, lstart, out, row_batch, NQ = output rows R, N, S, M, L, Q, P, launch fields
  const float *offs;    // sampling-offset projections: offs[r*proj_row + m*off_head + q*off_k + (l*P + p)*2 + c]
  const float *logits;  // attention logits:            logits[r*proj_row + m*lg_head + q*lg_k + l*P + p]
  const float *ref;     // reference points (R, K, A, 2), normalised (x, y)
  const int32_t *row_src;  // optional: projection row of output row r (else r) — SCA projects each
                           // BEV query once and every camera that sees it reads the same row
  long proj_row;        // row stride (floats) of offs / logits
  int off_head, off_k, lg_head, lg_k;
  int K;                // queue entries averaged into one output row (1 or 2)
  int A;                // reference points per (row, queue entry)
  int ref_mode;         // 0: point p uses anchor p % A (pillar anchors); 1: level l uses ref l
  int vmul, vadd;       // value batch entry of (row, q) = base * vmul + q * vadd, base = row_batch[r] or r / Q
  float out_scale;      // 1 / K
  int out_f32;          // bf16-storage kernels only: write the output rows as fp32 (the consumer is the
                        // fp32 output projection) instead of bf16
  const int32_t *nrows; // optional DEVICE row count (frame_plan.h: counters[0]): when set, k.NQ is only the
                        // capacity of the row arrays and the launch geometry comes from `launch_rows`
                        // (see DynRows), so one captured launch serves every frame
  int launch_rows;      // host hint of the row count (<= capacity)
  float *save_loc;      // SAVE kernels (training forward, K = 1): the sampling locations (R, M, L, P, 2) and attention weights
  float *save_attn;     // (R, M, L, P) the backward kernels read — written here instead of recomputed by the expand pass
};

// Launches whose row count lives on the device.  Two kernels cover the rows:
//   * head: rows [0, min(count, launch_rows)) — one workgroup per logical block like the fixed-count
//     kernel, the block map (XCD-contiguous ranges) computed from the clamped count; `launch_rows` is
//     the host's HINT of the count (frame plan of an earlier frame + margin), it sizes the grid;
//   * tail: rows [launch_rows, count) — a small fixed grid striding over whatever the hint missed
//     (normally nothing: the launch exits at once).  Correct for any count <= capacity; only speed
//     depends on the hint.
struct DynRows {
  long NQ;        // rows in all (clamped to the capacity)
  long row0;      // first row of this launch's share
  int nblocks, per;
};
__device__ __forceinline__ DynRows dyn_rows(const FusedArgs &f, bool tail) {
  DynRows d;
  const int n = *f.nrows;
  const int cap = static_cast<int>(f.k.NQ), hint = f.launch_rows;
  const int total = n < cap ? n : cap;
  const int head = total < hint ? total : hint;
  d.NQ = tail ? total : head;
  d.row0 = tail ? head : 0;
  const int mine = tail ? total - head : head;
  const int tiles = (mine + f.k.qtile - 1) / f.k.qtile;
  d.nblocks = (tiles * f.k.qtile * f.k.M + 31) / 32;
  d.per = (d.nblocks + 7) >> 3;
  return d;
}

template <int X>
__device__ __forceinline__ float xor8(float v) {   // value of lane ^ X (X < 8)
  return __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(v), (X << 10) | 0x1f));
}
// reductions over the W (4 or 8) consecutive lanes that hold the points of one softmax
template <int W>
__device__ __forceinline__ float lanes_max(float v) {
  v = fmaxf(v, xor8<1>(v)); v = fmaxf(v, xor8<2>(v));
  if constexpr (W == 8) v = fmaxf(v, xor8<4>(v));
  return v;
}
template <int W>
__device__ __forceinline__ float lanes_sum(float v) {
  v += xor8<1>(v); v += xor8<2>(v);
  if constexpr (W == 8) v += xor8<4>(v);
  return v;
}

// PT points per level, KT queue entries; the PT*KT (<= 8) "virtual points" of a level are
// owned by the lanes of the group: lane j -> queue entry j / PT, point j % PT.  With
// KT = 2 (TemporalSelfAttention: PT = 4) both queue entries are sampled in the same
// round and summed into the same accumulator (their mean is the output).
// LC / MC > 0: the level count / the head count as COMPILE-TIME constants (the launcher checks the descriptor: 8 heads, query
// tiles of 8 rows, LC levels): the group -> (row, head) map is shifts, a pixel is 1 KiB (the x-adjacent tap is an immediate
// offset of its load), the level loop unrolls — round 5: the generic body spends 552 instructions per level of 32 tap
// loads (1,169 per row for TemporalSelfAttention's single level), much of it 64-bit address and division arithmetic on
// runtime strides.  0 = the generic body.
template <typename T, int PT, int KT, bool SAVE = false, int LC = 0, int MC = 0>
__device__ __forceinline__ void msda_fused_d32_body(const FusedArgs &f, int lblock, long NQ, int tid, long row0 = 0) {
  static_assert(!SAVE || KT == 1, "SAVE: one queue entry");
  constexpr int D = 32, LPG = 8, GPB = 256 / LPG, NP = PT * KT;
  static_assert((PT == 4 || PT == 8) && (KT == 1 || KT == 2) && NP <= 8, "PT/KT");
  static_assert(MC == 0 || MC == 8, "MC");
  const KArgs &a = f.k;
  const int lig = tid & 7;
  const long G = static_cast<long>(lblock) * GPB + (tid >> 3);
  long r; int m;
  if constexpr (MC == 8) {                              // M = 8 heads, qtile = 8 rows: 64 groups per tile
    const long tile = G >> 6;
    const int rr = static_cast<int>(G & 63);
    m = rr >> 3;
    r = (tile << 3) + (rr & 7);
  } else {
    map_group(G, a, r, m);
  }
  r += row0;
  const bool active = r < NQ;
  if (!active) r = NQ - 1;
  const int L = LC > 0 ? LC : a.L;                      // 1..4 (host-checked)
  const int Mh = MC > 0 ? MC : a.M;
  const long base = a.row_batch ? static_cast<long>(a.row_batch[r]) : r / a.Q;
  const uint32_t pix_bytes = static_cast<uint32_t>(Mh) * D * sizeof(T);
  const uint32_t lane_term = lig * 4 * static_cast<uint32_t>(sizeof(T));
  const uint32_t total_bytes = static_cast<uint32_t>(static_cast<unsigned long long>(a.N) * a.S * pix_bytes);
  __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(a.value), 0,
                                                                  static_cast<int>(total_bytes), 0x00020000);
  const bool owner = lig < NP;
  const bool live = active && owner;
  const int q = owner ? lig / PT : 0;                    // my queue entry
  const int pj = owner ? lig % PT : 0;                   // my point
  const long n = base * f.vmul + static_cast<long>(q) * f.vadd;
  const uint32_t head_base = static_cast<uint32_t>((static_cast<unsigned long long>(n) * a.S * Mh + m) * D * sizeof(T));
  const long rs = f.row_src ? static_cast<long>(f.row_src[r]) : r;
  const float *__restrict__ lgp = f.logits + rs * f.proj_row + m * f.lg_head + q * f.lg_k + pj;
  const float2 *__restrict__ ofp =
      reinterpret_cast<const float2 *>(f.offs + rs * f.proj_row + m * f.off_head + q * f.off_k) + pj;
  const float2 *__restrict__ rfp = reinterpret_cast<const float2 *>(f.ref) + (r * f.K + q) * f.A;

  // softmax over the L*PT logits of my (row, head, queue entry): ONE batch of loads (my
  // point's logit of every level), one exp per level, butterflies over the PT lanes
#ifndef BEVMSDA_SCA_DIAG
#define BEVMSDA_SCA_DIAG 0          // diagnostic builds (wrong results): 1 = no softmax arithmetic, 2 = no front-end loads either
#endif
#if BEVMSDA_SCA_DIAG >= 2
  float e0 = 0.1f * pj, e1 = 0.2f, e2 = 0.3f, e3 = 0.4f;
  float2 of = make_float2(1.f + pj, 2.f - pj);
  float2 rf = make_float2(0.3f + 1e-5f * (r & 1023), 0.4f + 1e-5f * m);
#else
  float e0 = lgp[0];
  float e1 = L > 1 ? lgp[PT] : -INFINITY;
  float e2 = L > 2 ? lgp[2 * PT] : -INFINITY;
  float e3 = L > 3 ? lgp[3 * PT] : -INFINITY;
  float2 of = ofp[0];
  float2 rf = rfp[f.ref_mode == 0 ? pj % f.A : 0];
#endif
#if BEVMSDA_SCA_DIAG >= 1
  const float sum = 32.f;
#else
  const float mx = lanes_max<PT>(fmaxf(fmaxf(e0, e1), fmaxf(e2, e3)));
  e0 = expf(e0 - mx); e1 = expf(e1 - mx); e2 = expf(e2 - mx); e3 = expf(e3 - mx);   // exp(-inf) = 0
  const float sum = lanes_sum<PT>((e0 + e1) + (e2 + e3));
#endif

  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  // (a known level count is NOT unrolled: four levels' taps in one body spill at 128 registers — 1.6 KB of scratch, measured)
#pragma nounroll
  for (int l = 0; l < (LC > 0 ? LC : L); ++l) {
    const int H = static_cast<int>(a.shapes[2 * l]), W = static_cast<int>(a.shapes[2 * l + 1]);
    const uint32_t lbytes = static_cast<uint32_t>(a.lstart[l]) * pix_bytes;
    const float lx = rf.x + of.x / static_cast<float>(W);
    const float ly = rf.y + of.y / static_cast<float>(H);
    const float e = l == 0 ? e0 : (l == 1 ? e1 : (l == 2 ? e2 : e3));
    const float aw = live ? e / sum : 0.f;
    const PointParams p = point_params(lx, ly, aw, H, W, head_base + lbytes, pix_bytes);
    if constexpr (SAVE) {
      if (live) {                   // (r, m, l, pj): 64 + 32 contiguous bytes per (row, head, level)
        const long o = ((r * Mh + m) * L + l) * PT + pj;
#if !BEVMSDA_DIAG_NO_SAVE_LOC
        if (f.save_loc) reinterpret_cast<float2 *>(f.save_loc)[o] = make_float2(lx, ly);     // (nullptr: the backward recomputes them)
#endif
        f.save_attn[o] = aw;
      }
    }
#if BEVMSDA_SCA_DIAG < 2
    if (l + 1 < L) {                // next level's record travels under this level's taps
      of = ofp[(l + 1) * PT];
      if (f.ref_mode == 1) rf = rfp[l + 1];
    }
#endif
    sample_points<0, NP, T>(p, rsrc, lane_term, pix_bytes, static_cast<uint32_t>(W) * pix_bytes, acc);
  }
  if (active) {
    T *op = static_cast<T *>(a.out) + (r * Mh + m) * D + lig * 4;
    const float sc = f.out_scale;
    if constexpr (sizeof(T) == 4) {
      *reinterpret_cast<float4 *>(op) = make_float4(acc[0] * sc, acc[1] * sc, acc[2] * sc, acc[3] * sc);
    } else {
      uint2 t;
      t.x = f32_to_bf16(acc[0] * sc) | (f32_to_bf16(acc[1] * sc) << 16);
      t.y = f32_to_bf16(acc[2] * sc) | (f32_to_bf16(acc[3] * sc) << 16);
      *reinterpret_cast<uint2 *>(op) = t;
    }
  }
}

template <typename T, int PT, int KT, int WPE, int LC = 0, int MC = 0>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WPE, WPE)))
msda_fused_d32_kernel(const FusedArgs f) {
  msda_fused_d32_body<T, PT, KT, false, LC, MC>(f, logical_block(f.k), f.k.NQ, threadIdx.x);
}

// TemporalSelfAttention's shape (fp32, 8 heads, ONE level, 2 queue entries x 4 points, qtile 8) in a resident grid with the
// front end of the NEXT block under the taps of this one (round 6).  With one level a lane group's life is: parameter loads
// (one trip to memory) -> softmax / locations -> 32 taps -> store; nothing of it overlaps inside a wavefront, and the
// wavefronts of a CU spend ~40 % of their time with no tap in flight (the kernel ran at 52 % of the gather ceiling where
// SpatialCrossAttention's four-level loop, which prefetches the next level's record, reaches 68 %).  Here 1,024 workgroups
// (4 per CU) stay resident; workgroup w of XCD x takes the logical blocks x * per + w, + 128, + 256 .. of that XCD's range
// (the window of blocks an XCD works on at any time is the same as with one workgroup per block), and the logits / offsets /
// reference of block i + 1 are requested BEFORE the taps of block i.  Same arithmetic, same order of sums: bit-equal to
// msda_fused_d32_kernel<float, 4, 2, 4, 1, 8>.
// The launcher takes this kernel for ONE batch entry without row indirection (R == Q, no row_batch, no row_src: the
// encoder's call): consecutive blocks of a workgroup are then 4 x (workgroups per XCD) rows apart with the same head and
// lane roles, so the front end of the next block is three loads at fixed strides from this one's — a handful of live
// registers across the taps (a first version recomputed every address per block: at 128 registers hipcc spilled
// lane-constant 64-bit temporaries, and each reload from scratch drew an `s_waitcnt vmcnt(0)` in front of the prefetch).
template <typename T, int WPE>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WPE, WPE)))
msda_fused_d32_tsa_pipe_kernel(const FusedArgs f) {
  constexpr int D = 32, PT = 4, NP = 8, Mh = 8;
  const KArgs &a = f.k;
  const int tid = threadIdx.x, lig = tid & 7;
  const int per = (a.nblocks + 7) >> 3;                      // logical blocks per XCD range (as logical_block())
  const int wpx = static_cast<int>(gridDim.x >> 3);          // workgroups per XCD
  const int x = blockIdx.x & 7, w = blockIdx.x >> 3;
  const int hi = (x + 1) * per < a.nblocks ? (x + 1) * per : a.nblocks;
  int lb = x * per + w;
  if (lb >= hi) return;
  const int H = static_cast<int>(a.shapes[0]), W = static_cast<int>(a.shapes[1]);
  const uint32_t pix_bytes = static_cast<uint32_t>(Mh) * D * sizeof(T);
  const uint32_t lane_term = lig * 4 * static_cast<uint32_t>(sizeof(T));
  const uint32_t lbytes = static_cast<uint32_t>(a.lstart[0]) * pix_bytes;
  const uint32_t total_bytes = static_cast<uint32_t>(static_cast<unsigned long long>(a.N) * a.S * pix_bytes);
  __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(a.value), 0,
                                                                  static_cast<int>(total_bytes), 0x00020000);
  const float sc = f.out_scale;
  const float inv_w = static_cast<float>(W), inv_h = static_cast<float>(H);
  // my (row, head) of the first block; the next ones are `rstep` rows further
  const uint32_t G = static_cast<uint32_t>(lb) * 32u + static_cast<uint32_t>(tid >> 3);
  const uint32_t rr = G & 63u, m = rr >> 3;
  uint32_t r = ((G >> 6) << 3) + (rr & 7u);
  const uint32_t NQ = static_cast<uint32_t>(a.NQ);
  const uint32_t rstep = 4u * static_cast<uint32_t>(wpx);
  const uint32_t q = lig / PT, pj = lig % PT;
  const uint32_t n = q * static_cast<uint32_t>(f.vadd);                                        // (batch entry 0)
  const uint32_t level_base = static_cast<uint32_t>((static_cast<unsigned long long>(n) * a.S * Mh + m) * D * sizeof(T)) + lbytes;
  const uint32_t rc = r < NQ ? r : NQ - 1;
  // (front-end operands through raw buffer descriptors: one 32-bit byte offset per stream instead of a 64-bit address pair)
  const auto whole = [](const void *p) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, 0x7fffffff, 0x00020000); };
  const __amdgpu_buffer_rsrc_t lg_rs = whole(f.logits), of_rs = whole(f.offs), rf_rs = whole(f.ref);
  uint32_t lg_b = (rc * static_cast<uint32_t>(f.proj_row) + m * f.lg_head + q * f.lg_k + pj) * 4u;
  uint32_t of_b = (rc * static_cast<uint32_t>(f.proj_row) + m * f.off_head + q * f.off_k + 2u * pj) * 4u;
  uint32_t rf_b = ((rc * f.K + q) * f.A + (f.ref_mode == 0 ? pj % f.A : 0)) * 8u;
  uint32_t out_i = (r * Mh + m) * D + lig * 4;
  const uint32_t lg_s = rstep * static_cast<uint32_t>(f.proj_row) * 4u, rf_s = rstep * f.K * f.A * 8u, out_s = rstep * Mh * D;
  const auto ld1 = [](__amdgpu_buffer_rsrc_t rs, uint32_t b) { return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, static_cast<int>(b), 0, 0)); };
  const auto ld2 = [](__amdgpu_buffer_rsrc_t rs, uint32_t b) {
    const auto v = __builtin_amdgcn_raw_buffer_load_b64(rs, static_cast<int>(b), 0, 0);
    return make_float2(__uint_as_float(v[0]), __uint_as_float(v[1]));
  };
  float lg = ld1(lg_rs, lg_b);
  float2 of = ld2(of_rs, of_b), rf = ld2(rf_rs, rf_b);
#pragma nounroll
  for (;;) {
    const bool more = lb + wpx < hi;                          // (uniform)
    const bool active = r < NQ;
    const float lg0 = lg;
    const float2 of0 = of, rf0 = rf;
    const uint32_t out0 = out_i;
    if (more) {                                               // the next block's front end, under this block's taps
      lb += wpx;
      r += rstep;
      out_i += out_s;
      if (r < NQ) { lg_b += lg_s; of_b += lg_s; rf_b += rf_s; }     // (rows beyond the last re-read the previous block's: valid memory)
      lg = ld1(lg_rs, lg_b);
      of = ld2(of_rs, of_b);
      rf = ld2(rf_rs, rf_b);
    }
    // softmax over the PT logits of my (row, head, queue entry)
    const float mx = lanes_max<PT>(lg0);
    const float e = expf(lg0 - mx);
    const float sum = lanes_sum<PT>(e);
    const float lx = rf0.x + of0.x / inv_w;
    const float ly = rf0.y + of0.y / inv_h;
    const float aw = active ? e / sum : 0.f;
    const PointParams p = point_params(lx, ly, aw, H, W, level_base, pix_bytes);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    sample_points<0, NP, T>(p, rsrc, lane_term, pix_bytes, static_cast<uint32_t>(W) * pix_bytes, acc);
    if (active) {
      T *op = static_cast<T *>(a.out) + out0;
      *reinterpret_cast<float4 *>(op) = make_float4(acc[0] * sc, acc[1] * sc, acc[2] * sc, acc[3] * sc);
    }
    if (!more) break;
  }
}

// TemporalSelfAttention's shape with the tile's tap lines staged in LDS (round 6; the north star's "feature maps staged through
// LDS tiles", for the one call of the path whose taps re-use lines densely).  TSA samples the BEV grid itself: the taps of a
// TX x TY tile of queries fall within a few pixels of the tile (offset bias grid: point j at most j + 1 pixels out) — for the
// history entry shifted by the ego motion.  A 128-byte tap line reaches the lanes at <= 60 B / clk / CU from the vector L1
// (tools/probes/gather_path_probe.hip) and at 117-133 B / clk / CU from LDS, 97-112 when every staged line serves 4-6 taps
// (tools/probes/lds_gather_probe.hip).  So: one workgroup (512 threads, two per CU) owns a 16 x 8 tile of queries of ONE head;
// per queue entry it stages the (TX + 2 HALO + 2) x (TY + 2 HALO + 2) lines of that head around the tile (LDS-DMA, 70 KiB;
// lines beyond the map repeat the border pixel: finite values under the zero coefficients of taps outside, as the global form's
// in-range neighbour) and every 8-lane group serves its two rows' 16 taps of the entry with ds_read_b128.  A point whose
// footprint leaves the staged region (offsets beyond the halo, points outside the map) sends its WAVEFRONT through the
// global-memory taps of msda_fused_d32_kernel for that (row, queue entry) — any offsets are exact, only slower.
// Same parameters, same coefficients, same order of sums: bit-equal to msda_fused_d32_kernel<float, 4, 2, 4, 1, 8>.
#ifndef BEVMSDA_TSA_LDS_TY
#define BEVMSDA_TSA_LDS_TY 8          // A/B builds: 16 (98 KiB regions, one workgroup per CU, four rows per lane group)
#endif
#ifndef BEVMSDA_TSA_LDS_DIAG
#define BEVMSDA_TSA_LDS_DIAG 0        // diagnostic builds (wrong results): bit 0 no staging, bit 1 no taps, bit 2 no front-end loads
#endif
constexpr int kTsaLdsTX = 16, kTsaLdsTY = BEVMSDA_TSA_LDS_TY, kTsaLdsHalo = 5;
constexpr int kTsaLdsRW = kTsaLdsTX + 2 * kTsaLdsHalo + 2, kTsaLdsRH = kTsaLdsTY + 2 * kTsaLdsHalo + 2;   // 28 x 20 lines
constexpr uint32_t kTsaLdsNoLine = 0xffffffffu;

template <int J0, int j, int CNT>
struct IssuePointsLds {
  static __device__ __forceinline__ void run(const PointParams &p, uint32_t my_line, const unsigned char *region, uint32_t lane_b,
                                             f32x4 (&v)[CNT][4], float (&k)[CNT][4]) {
    constexpr int J = J0 + j;
    constexpr uint32_t dy = kTsaLdsRW * 128;
    const uint32_t o = bcast8<J>(my_line) + lane_b;
    k[j][0] = bcast8<J>(p.k00); k[j][1] = bcast8<J>(p.k01);
    k[j][2] = bcast8<J>(p.k10); k[j][3] = bcast8<J>(p.k11);
    v[j][0] = *reinterpret_cast<const f32x4 *>(region + o);
    v[j][1] = *reinterpret_cast<const f32x4 *>(region + o + 128);
    v[j][2] = *reinterpret_cast<const f32x4 *>(region + o + dy);
    v[j][3] = *reinterpret_cast<const f32x4 *>(region + o + dy + 128);
    if constexpr (j + 1 < CNT) IssuePointsLds<J0, j + 1, CNT>::run(p, my_line, region, lane_b, v, k);
  }
};

template <int J0, int CNT>
__device__ __forceinline__ void sample_points_lds(const PointParams &p, uint32_t my_line, const unsigned char *region, uint32_t lane_b,
                                                  f32x4 &acc) {
  f32x4 v[CNT][4];
  float k[CNT][4];
  IssuePointsLds<J0, 0, CNT>::run(p, my_line, region, lane_b, v, k);
#pragma unroll
  for (int j = 0; j < CNT; ++j) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      acc[0] = fmaf(k[j][t], v[j][t][0], acc[0]);
      acc[1] = fmaf(k[j][t], v[j][t][1], acc[1]);
      acc[2] = fmaf(k[j][t], v[j][t][2], acc[2]);
      acc[3] = fmaf(k[j][t], v[j][t][3], acc[3]);
    }
  }
}

template <int WPE>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(WPE, WPE)))
msda_fused_d32_tsa_lds_kernel(const FusedArgs f) {
  using T = float;
  constexpr int D = 32, PT = 4, Mh = 8, TX = kTsaLdsTX, TY = kTsaLdsTY, RW = kTsaLdsRW, RH = kTsaLdsRH, HALO = kTsaLdsHalo;
  constexpr int NLINES = RW * RH;                              // 560 lines = 70 KiB
  static_assert(NLINES % 8 == 0, "whole wavefront DMA instructions (8 lines each)");
  __shared__ __attribute__((aligned(16))) unsigned char region[NLINES * 128];
  const KArgs &a = f.k;
  const int tid = threadIdx.x, lig = tid & 7, grp = tid >> 3, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int H = static_cast<int>(a.shapes[0]), W = static_cast<int>(a.shapes[1]);
  const int tiles_x = (W + TX - 1) / TX, tiles_y = (H + TY - 1) / TY;
  const int nb = tiles_x * tiles_y * Mh;
  const int per = (nb + 7) >> 3;                                // XCD x walks the logical blocks [x per, (x + 1) per): one band of the grid
  if (static_cast<int>(blockIdx.x >> 3) >= per) return;
  const int lb = static_cast<int>(blockIdx.x & 7) * per + static_cast<int>(blockIdx.x >> 3);
  if (lb >= nb) return;
  const int m = lb & 7, tile = lb >> 3;                         // (the 8 heads of a tile run back to back on one XCD: same pixels)
  const int x0 = (tile % tiles_x) * TX, y0 = (tile / tiles_x) * TY;
  const uint32_t pix_bytes = static_cast<uint32_t>(Mh) * D * sizeof(T);
  const uint32_t lane_term = lig * 4 * static_cast<uint32_t>(sizeof(T));
  const uint32_t lbytes = static_cast<uint32_t>(a.lstart[0]) * pix_bytes;
  const uint32_t total_bytes = static_cast<uint32_t>(static_cast<unsigned long long>(a.N) * a.S * pix_bytes);
  __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(a.value), 0,
                                                                  static_cast<int>(total_bytes), 0x00020000);
  const float sc = f.out_scale;
  const uint32_t q = lig / PT, pj = lig % PT;                   // my queue entry / point
  const uint32_t nq = q * static_cast<uint32_t>(f.vadd);        // (batch entry 0)
  const uint32_t level_base = static_cast<uint32_t>((static_cast<unsigned long long>(nq) * a.S * Mh + m) * D * sizeof(T)) + lbytes;
  const float2 *__restrict__ rf2 = reinterpret_cast<const float2 *>(f.ref);
  // the staged regions' origins: HALO pixels up and left of the tile's first query as the queue entry sees it
  int ox[2], oy[2];
  {
    const long rc = static_cast<long>(y0) * W + x0;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const float2 c = rf2[(rc * f.K + e) * f.A];
      ox[e] = static_cast<int>(floorf(c.x * static_cast<float>(W) - 0.5f)) - HALO;
      oy[e] = static_cast<int>(floorf(c.y * static_cast<float>(H) - 0.5f)) - HALO;
    }
  }
  // my NR rows: (x0 + gx, y0 + gy) and every 4th grid row below
  constexpr int NR = TX * TY / 64;
  const int gx = grp & 15, gy = grp >> 4;
  PointParams pp[NR];
  uint32_t line[NR];
  uint32_t orow[NR];
  bool act[NR];
#pragma unroll
  for (int i = 0; i < NR; ++i) {
    const int x = x0 + gx, y = y0 + gy + 4 * i;
    act[i] = x < W && y < H;
    const long r = act[i] ? static_cast<long>(y) * W + x : a.NQ - 1;
    orow[i] = static_cast<uint32_t>((r * Mh + m) * D + lig * 4);
#if BEVMSDA_TSA_LDS_DIAG & 4
    const float lg = 0.1f * pj;
    const float2 of = make_float2(1.f + pj, 0.5f * q);
    const float2 rf = make_float2((x + 0.5f) / W, (y + 0.5f) / H);
#else
    const float lg = f.logits[r * f.proj_row + m * f.lg_head + q * f.lg_k + pj];
    const float2 of = reinterpret_cast<const float2 *>(f.offs + r * f.proj_row + m * f.off_head + q * f.off_k)[pj];
    const float2 rf = rf2[(r * f.K + q) * f.A];
#endif
    const float mx = lanes_max<PT>(lg);
    const float e = expf(lg - mx);
    const float sum = lanes_sum<PT>(e);
    const float lx = rf.x + of.x / static_cast<float>(W);
    const float ly = rf.y + of.y / static_cast<float>(H);
    const float aw = act[i] ? e / sum : 0.f;
    int fx0, fy0;
    pp[i] = point_params_xy(lx, ly, aw, H, W, level_base, pix_bytes, fx0, fy0);
    // the footprint's top-left pixel inside my queue entry's region
    const int px = fx0 - (q ? ox[1] : ox[0]);
    const int py = fy0 - (q ? oy[1] : oy[0]);
    const bool in = pp[i].off != kOobOffset && px >= 0 && px + 1 < RW && py >= 0 && py + 1 < RH;
    line[i] = in ? static_cast<uint32_t>(py * RW + px) * 128u : kTsaLdsNoLine;
  }
  f32x4 acc[NR];
#pragma unroll
  for (int i = 0; i < NR; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  const char *vbase = static_cast<const char *>(a.value);
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    if (e) __syncthreads();                                     // every wavefront is done with the first entry's lines
    {
      const unsigned long long ebase = (static_cast<unsigned long long>(e) * f.vadd * a.S + static_cast<unsigned long long>(a.lstart[0])) * pix_bytes
                                       + static_cast<unsigned long long>(m) * D * sizeof(T) + lane_term;
#pragma nounroll
      for (int l0 = wave * 8; l0 < ((BEVMSDA_TSA_LDS_DIAG & 1) ? 0 : NLINES); l0 += 64) {          // 8 lines per wavefront instruction
        const int li = l0 + (lane >> 3);
        const int ry = li / RW, rx = li - ry * RW;
        int sx = ox[e] + rx, sy = oy[e] + ry;
        sx = sx < 0 ? 0 : (sx >= W ? W - 1 : sx);
        sy = sy < 0 ? 0 : (sy >= H ? H - 1 : sy);
        const char *src = vbase + ebase + static_cast<unsigned long long>(sy * W + sx) * pix_bytes;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src),
                                         (__attribute__((address_space(3))) void *)(region + l0 * 128), 16, 0, 0);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < ((BEVMSDA_TSA_LDS_DIAG & 2) ? 0 : NR); ++i) {
      // (lanes 4 e .. 4 e + 3 of a group own this entry's points; `line` of the other lanes is not read)
      const bool mine = (lig >> 2) == e;
      const bool slow = __builtin_amdgcn_ballot_w64(mine && line[i] == kTsaLdsNoLine) != 0;
      if (slow) {
        if (e == 0) sample_points<0, PT, T>(pp[i], rsrc, lane_term, pix_bytes, static_cast<uint32_t>(W) * pix_bytes, acc[i]);
        else sample_points<PT, PT, T>(pp[i], rsrc, lane_term, pix_bytes, static_cast<uint32_t>(W) * pix_bytes, acc[i]);
      } else {
        if (e == 0) sample_points_lds<0, PT>(pp[i], line[i], region, lane_term, acc[i]);
        else sample_points_lds<PT, PT>(pp[i], line[i], region, lane_term, acc[i]);
      }
    }
  }
#pragma unroll
  for (int i = 0; i < NR; ++i)
    if (act[i])
      *reinterpret_cast<float4 *>(static_cast<T *>(a.out) + orow[i]) = make_float4(acc[i][0] * sc, acc[i][1] * sc, acc[i][2] * sc, acc[i][3] * sc);
}

// The same kernel over a device-side row count (DynRows): head = one workgroup per logical block of
// the hinted count, tail = a small strided grid for rows beyond the hint.
template <typename T, int PT, int KT, int WPE, bool SAVE = false, int LC = 0, int MC = 0>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WPE, WPE)))
msda_fused_d32_head_kernel(const FusedArgs f) {
  const DynRows d = dyn_rows(f, false);
  const int b = blockIdx.x;
  if ((b >> 3) >= d.per) return;
  msda_fused_d32_body<T, PT, KT, SAVE, LC, MC>(f, (b & 7) * d.per + (b >> 3), d.NQ, threadIdx.x);
}

template <typename T, int PT, int KT, int WPE, bool SAVE = false>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WPE, WPE)))
msda_fused_d32_dyn_kernel(const FusedArgs f) {
  const DynRows d = dyn_rows(f, true);
  if (d.nblocks <= 0) return;
#pragma nounroll
  for (int pb = blockIdx.x; pb < d.per * 8; pb += gridDim.x) {
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    msda_fused_d32_body<T, PT, KT, SAVE>(f, (pb & 7) * d.per + (pb >> 3), d.NQ, tid, d.row0);
  }
}

// ------------------------------------------------------------------ bf16 storage, 16-byte lanes
// With bf16 storage a tap of one head is 64 bytes.  The 8-byte-per-lane form above (TapLoad<bf16_t>)
// halves the bytes but not the number of lane requests, and the kernel is bound by the request
// rate, not by bytes (no gain).  Here a lane fetches 16 bytes = 8 channels, four
// lanes cover a tap, and the two halves of an 8-lane group fetch the two x-adjacent taps of a
// bilinear footprint in ONE instruction (lanes 0-3: x0, lanes 4-7: x0 + 1; a second instruction
// for the row below): 2 requests per point instead of 4, the same 16 bytes per lane as the fp32
// kernel.  Each half accumulates its taps over 8 channels; one xor-4 exchange at the end adds
// the halves.  Point ownership, parameter broadcast, softmax / location prologue and the queue
// mean are those of msda_fused_d32_kernel.
template <int J0, int j, int CNT>
struct IssuePointsB8 {
  static __device__ __forceinline__ void run(const PointParams &p, __amdgpu_buffer_rsrc_t r, uint32_t lane_term,
                                             bool upper, uint32_t dyb, uint4 (&v)[CNT][2], float (&k)[CNT][2]) {
    constexpr int J = J0 + j;
    const uint32_t o = bcast8<J>(p.off) + lane_term;           // lane_term already holds "+ dxb" for the upper half
    const float k00 = bcast8<J>(p.k00), k01 = bcast8<J>(p.k01);
    const float k10 = bcast8<J>(p.k10), k11 = bcast8<J>(p.k11);
    k[j][0] = upper ? k01 : k00;
    k[j][1] = upper ? k11 : k10;
    v[j][0] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(r, static_cast<int>(o), 0, 0));
    v[j][1] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(r, static_cast<int>(o + dyb), 0, 0));
    if constexpr (j + 1 < CNT) IssuePointsB8<J0, j + 1, CNT>::run(p, r, lane_term, upper, dyb, v, k);
  }
};

__device__ __forceinline__ void fma8_bf16(float kk, const uint4 &w, float (&acc)[8]) {
  acc[0] = fmaf(kk, bf16_lo(w.x), acc[0]); acc[1] = fmaf(kk, bf16_hi(w.x), acc[1]);
  acc[2] = fmaf(kk, bf16_lo(w.y), acc[2]); acc[3] = fmaf(kk, bf16_hi(w.y), acc[3]);
  acc[4] = fmaf(kk, bf16_lo(w.z), acc[4]); acc[5] = fmaf(kk, bf16_hi(w.z), acc[5]);
  acc[6] = fmaf(kk, bf16_lo(w.w), acc[6]); acc[7] = fmaf(kk, bf16_hi(w.w), acc[7]);
}

template <int CNT>
__device__ __forceinline__ void sample_points_b8(const PointParams &p, __amdgpu_buffer_rsrc_t r, uint32_t lane_term,
                                                 bool upper, uint32_t dyb, float (&acc)[8]) {
  uint4 v[CNT][2];
  float k[CNT][2];
  IssuePointsB8<0, 0, CNT>::run(p, r, lane_term, upper, dyb, v, k);
#pragma unroll
  for (int j = 0; j < CNT; ++j) {
    fma8_bf16(k[j][0], v[j][0], acc);
    fma8_bf16(k[j][1], v[j][1], acc);
  }
}

template <int PT, int KT, bool SAVE = false>
__device__ __forceinline__ void msda_fused_d32_bf16x8_body(const FusedArgs &f, int lblock, long NQ, int tid, long row0 = 0) {
  static_assert(!SAVE || KT == 1, "SAVE: one queue entry");
  constexpr int D = 32, LPG = 8, GPB = 256 / LPG, NP = PT * KT;
  static_assert((PT == 4 || PT == 8) && (KT == 1 || KT == 2) && NP <= 8, "PT/KT");
  const KArgs &a = f.k;
  const int lig = tid & 7;
  const bool upper = lig >= 4;                           // this lane's taps: x0 + 1
  const long G = static_cast<long>(lblock) * GPB + (tid >> 3);
  long r; int m;
  map_group(G, a, r, m);
  r += row0;
  const bool active = r < NQ;
  if (!active) r = NQ - 1;
  const int L = a.L;
  const long base = a.row_batch ? static_cast<long>(a.row_batch[r]) : r / a.Q;
  const uint32_t pix_bytes = static_cast<uint32_t>(a.M) * D * 2;
  const uint32_t lane_term = (lig & 3) * 16 + (upper ? pix_bytes : 0u);
  const uint32_t total_bytes = static_cast<uint32_t>(static_cast<unsigned long long>(a.N) * a.S * pix_bytes);
  __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(a.value), 0,
                                                                  static_cast<int>(total_bytes), 0x00020000);
  const bool owner = lig < NP;
  const bool live = active && owner;
  const int q = owner ? lig / PT : 0;
  const int pj = owner ? lig % PT : 0;
  const long n = base * f.vmul + static_cast<long>(q) * f.vadd;
  const uint32_t head_base = static_cast<uint32_t>((static_cast<unsigned long long>(n) * a.S * a.M + m) * D * 2);
  const long rs = f.row_src ? static_cast<long>(f.row_src[r]) : r;
  const float *__restrict__ lgp = f.logits + rs * f.proj_row + m * f.lg_head + q * f.lg_k + pj;
  const float2 *__restrict__ ofp =
      reinterpret_cast<const float2 *>(f.offs + rs * f.proj_row + m * f.off_head + q * f.off_k) + pj;
  const float2 *__restrict__ rfp = reinterpret_cast<const float2 *>(f.ref) + (r * f.K + q) * f.A;

  float e0 = lgp[0];
  float e1 = L > 1 ? lgp[PT] : -INFINITY;
  float e2 = L > 2 ? lgp[2 * PT] : -INFINITY;
  float e3 = L > 3 ? lgp[3 * PT] : -INFINITY;
  float2 of = ofp[0];
  float2 rf = rfp[f.ref_mode == 0 ? pj % f.A : 0];
  const float mx = lanes_max<PT>(fmaxf(fmaxf(e0, e1), fmaxf(e2, e3)));
  e0 = expf(e0 - mx); e1 = expf(e1 - mx); e2 = expf(e2 - mx); e3 = expf(e3 - mx);
  const float sum = lanes_sum<PT>((e0 + e1) + (e2 + e3));

  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int l = 0; l < L; ++l) {
    const int H = static_cast<int>(a.shapes[2 * l]), W = static_cast<int>(a.shapes[2 * l + 1]);
    const uint32_t lbytes = static_cast<uint32_t>(a.lstart[l]) * pix_bytes;
    const float lx = rf.x + of.x / static_cast<float>(W);
    const float ly = rf.y + of.y / static_cast<float>(H);
    const float e = l == 0 ? e0 : (l == 1 ? e1 : (l == 2 ? e2 : e3));
    const float aw = live ? e / sum : 0.f;
    const PointParams p = point_params(lx, ly, aw, H, W, head_base + lbytes, pix_bytes);
    if constexpr (SAVE) {
      if (live) {
        const long o = ((r * a.M + m) * L + l) * PT + pj;
#if !BEVMSDA_DIAG_NO_SAVE_LOC
        if (f.save_loc) reinterpret_cast<float2 *>(f.save_loc)[o] = make_float2(lx, ly);     // (nullptr: the backward recomputes them)
#endif
        f.save_attn[o] = aw;
      }
    }
    if (l + 1 < L) {
      of = ofp[(l + 1) * PT];
      if (f.ref_mode == 1) rf = rfp[l + 1];
    }
    sample_points_b8<NP>(p, rsrc, lane_term, upper, static_cast<uint32_t>(W) * pix_bytes, acc);
  }
  // add the two halves (lane ^ 4 holds the other x tap's partial sums over the same 8 channels)
#pragma unroll
  for (int c = 0; c < 8; ++c) acc[c] += xor8<4>(acc[c]);
  if (active && !upper && f.out_f32) {
    const float sc = f.out_scale;
    float *op = static_cast<float *>(a.out) + (r * a.M + m) * D + (lig & 3) * 8;
    *reinterpret_cast<float4 *>(op) = make_float4(acc[0] * sc, acc[1] * sc, acc[2] * sc, acc[3] * sc);
    *reinterpret_cast<float4 *>(op + 4) = make_float4(acc[4] * sc, acc[5] * sc, acc[6] * sc, acc[7] * sc);
  } else if (active && !upper) {
    const float sc = f.out_scale;
    bf16_t *op = static_cast<bf16_t *>(a.out) + (r * a.M + m) * D + (lig & 3) * 8;
    uint4 t;
    t.x = f32_to_bf16(acc[0] * sc) | (f32_to_bf16(acc[1] * sc) << 16);
    t.y = f32_to_bf16(acc[2] * sc) | (f32_to_bf16(acc[3] * sc) << 16);
    t.z = f32_to_bf16(acc[4] * sc) | (f32_to_bf16(acc[5] * sc) << 16);
    t.w = f32_to_bf16(acc[6] * sc) | (f32_to_bf16(acc[7] * sc) << 16);
    *reinterpret_cast<uint4 *>(op) = t;
  }
}

template <int PT, int KT, int WPE>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WPE, WPE)))
msda_fused_d32_bf16x8_kernel(const FusedArgs f) {
  msda_fused_d32_bf16x8_body<PT, KT>(f, logical_block(f.k), f.k.NQ, threadIdx.x);
}

template <int PT, int KT, int WPE, bool SAVE = false>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WPE, WPE)))
msda_fused_d32_bf16x8_head_kernel(const FusedArgs f) {
  const DynRows d = dyn_rows(f, false);
  const int b = blockIdx.x;
  if ((b >> 3) >= d.per) return;
  msda_fused_d32_bf16x8_body<PT, KT, SAVE>(f, (b & 7) * d.per + (b >> 3), d.NQ, threadIdx.x);
}

template <int PT, int KT, int WPE, bool SAVE = false>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WPE, WPE)))
msda_fused_d32_bf16x8_dyn_kernel(const FusedArgs f) {
  const DynRows d = dyn_rows(f, true);
  if (d.nblocks <= 0) return;
#pragma nounroll
  for (int pb = blockIdx.x; pb < d.per * 8; pb += gridDim.x) {
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    msda_fused_d32_bf16x8_body<PT, KT, SAVE>(f, (pb & 7) * d.per + (pb >> 3), d.NQ, tid, d.row0);
  }
}

}  // namespace bevmsda
