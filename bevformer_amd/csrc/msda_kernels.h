// Multi-scale deformable attention for CDNA4 (gfx950): device code.
//
// Math (operator reached by the reference at
// projects/mmdet3d_plugin/bevformer/modules/multi_scale_deformable_attn_function.py:118-124,150-160):
//   out[n,q,m,:] = sum_{l,p} A[n,q,m,l,p] * bilinear(V_l[n,:,:,m,:], x*W_l-0.5, y*H_l-0.5)
// with zero padding; gradients w.r.t. value (scatter), location and weight.
//
// Work decomposition (wave64-first, not a 32-thread-warp design):
//   * one "lane group" of LPG = D/CPL lanes owns one (n,q,m) output row; each
//     lane keeps CPL consecutive channels in registers and moves them with one
//     16-byte (fp32, CPL=4 / bf16, CPL=8) global load per bilinear tap, so one
//     tap of one head (128 B fp32 / 64 B bf16 at D=32) is one contiguous request;
//   * a wavefront therefore carries 64/LPG rows (8 at D=32 fp32); which rows is
//     decided by `qtile`: 1 -> the M heads of one query (output store is one
//     contiguous 1 KiB), T>1 -> T consecutive queries of the same head in
//     adjacent lane groups (taps of neighbouring BEV cells are neighbouring
//     pixels of one head slice -> better cache-line reuse);
//   * sampling locations / weights of a row are read once, coalesced (lane j
//     of the group reads point j), then broadcast inside the group with
//     ds_bpermute (`__shfl` width LPG) instead of 64 redundant loads;
//   * backward reduces grad_loc / grad_attn over the channels of a head with an
//     in-register xor butterfly over the LPG lanes and scatters grad_value with
//     hardware fp32 atomics (global_atomic_add_f32, no return);
//   * blocks are remapped so each XCD (block id mod 8) walks one contiguous
//     range of rows: its private 4 MiB L2 then sees one camera region.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "scalar_ops.h"

namespace bevmsda {

struct KArgs {
  const void *value;
  const int64_t *shapes;
  const int64_t *lstart;
  const float *loc;
  const float *attn;
  void *out;             // fwd: output; bwd: unused
  const void *grad_out;  // bwd only
  float *grad_value;
  float *grad_loc;
  float *grad_attn;
  const int32_t *row_batch;  // ragged mode: value batch entry of every query row (else nq / Q)
  long NQ;  // N*Q (ragged: total rows)
  int N, S, M, D, L, Q, P;
  int qtile;      // >=1
  int xcd_remap;  // 0/1
  int nblocks;    // logical blocks
  int variant;    // 0 default, 1 generic lane-group kernels only, 2 scalar fallback
  int mshift;     // log2(M) when M is a power of two, else -1
  int qshift;     // log2(qtile) when qtile is a power of two, else -1
  int gv_rows;    // backward: rows per workgroup of the grad_value sort kernel (0 = default; 64 / 128 / 256)
  int bf16_lanes8;  // backward, bf16 storage: 1 = the 8-byte-lane gather kernel instead of the 16-byte-lane one
  unsigned long long *gv_prof;   // backward: phase clocks of the sort kernel (tools/gvprof.py), else nullptr
  const int32_t *nrows_dev;      // backward (second-generation D = 32 kernels): NQ is the CAPACITY of the row arrays and the
                                 // actual row count is read here, on the device (frame_plan.h); nullptr: NQ rows
  long gv_stride;                // backward (second-generation D = 32 kernels): elements between two pixels of grad_value (0 = M * D:
                                 // dense); > M * D: the L value-projection gradients of a frame side by side in one array
  long gout_rows;                // backward (second-generation D = 32 kernels): > 0: grad_out has this many rows and row r of
  float gout_scale;              // the operands reads scale * grad_out[r % gout_rows] — the queue entries of
                                 // TemporalSelfAttention share one output row (their mean: scale = 1 / entries); 0: one
                                 // grad_out row per operand row, scale ignored
  // backward (second-generation D = 32 kernels, rows form): loc == nullptr — the sampling locations are not an operand but
  // recomputed from what the fused forward kernel read (msda_d32.h, K = 1, ref_mode 0), with the forward's own expression:
  //   loc(r, m, l, p) = loc_ref[r * loc_A + p % loc_A] + loc_offs[rs * loc_proj_row + m * loc_off_head + (l * P + p) * 2 + {0, 1}] / (W_l, H_l)
  // (rs = loc_row_src ? loc_row_src[r] : r) — the training forward then keeps 4 bytes per point (its attention weight) instead of 12
  const float *loc_offs, *loc_ref;
  const int32_t *loc_row_src;
  long loc_proj_row;
  int loc_off_head, loc_A;
};

// the (L, P) float2 records of (row nq, head m) — locations, or (loc == nullptr) raw offsets — and the reference point of point p
__device__ __forceinline__ const float2 *loc_records(const KArgs &a, long nq, int m) {
  if (a.loc) return reinterpret_cast<const float2 *>(a.loc) + (nq * a.M + m) * a.L * a.P;
  const long rs = a.loc_row_src ? static_cast<long>(a.loc_row_src[nq]) : nq;
  return reinterpret_cast<const float2 *>(a.loc_offs + rs * a.loc_proj_row + static_cast<long>(m) * a.loc_off_head);
}
__device__ __forceinline__ float2 loc_reference(const KArgs &a, long nq, int p) {
  if (a.loc) return make_float2(0.f, 0.f);
  return reinterpret_cast<const float2 *>(a.loc_ref)[nq * a.loc_A + p % a.loc_A];
}
// a record -> the location the forward sampled at (msda_fused_d32_body: lx = rf.x + of.x / W, the same two operations)
__device__ __forceinline__ float2 loc_of_record(const KArgs &a, float2 rec, float2 rf, int H, int W) {
  if (a.loc) return rec;
  return make_float2(rf.x + rec.x / static_cast<float>(W), rf.y + rec.y / static_cast<float>(H));
}

// rows the launch has to process: the host's NQ, or the device-side count below it
__device__ __forceinline__ long effective_rows(const KArgs &a) {
  long nq = a.NQ;
  if (a.nrows_dev) {
    const long n = static_cast<long>(*a.nrows_dev);
    nq = n < nq ? (n < 0 ? 0 : n) : nq;
  }
  return nq;
}

typedef uint16_t bf16_t;

template <typename T, int CPL>
struct Io;

template <>
struct Io<float, 4> {
  static __device__ __forceinline__ void load(const float *p, float (&v)[4]) {
    const float4 t = *reinterpret_cast<const float4 *>(p);
    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
  }
  static __device__ __forceinline__ void store(float *p, const float (&v)[4]) {
    *reinterpret_cast<float4 *>(p) = make_float4(v[0], v[1], v[2], v[3]);
  }
};

template <>
struct Io<float, 1> {
  static __device__ __forceinline__ void load(const float *p, float (&v)[1]) { v[0] = *p; }
  static __device__ __forceinline__ void store(float *p, const float (&v)[1]) { *p = v[0]; }
};

__device__ __forceinline__ float bf16_lo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf16_hi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }
// round-to-nearest-even fp32 -> bf16 (NaN kept quiet)
__device__ __forceinline__ uint32_t f32_to_bf16(float f) {
  uint32_t u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (u >> 16) | 0x40u;
  return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}

template <>
struct Io<bf16_t, 8> {
  static __device__ __forceinline__ void load(const bf16_t *p, float (&v)[8]) {
    const uint4 t = *reinterpret_cast<const uint4 *>(p);
    v[0] = bf16_lo(t.x); v[1] = bf16_hi(t.x); v[2] = bf16_lo(t.y); v[3] = bf16_hi(t.y);
    v[4] = bf16_lo(t.z); v[5] = bf16_hi(t.z); v[6] = bf16_lo(t.w); v[7] = bf16_hi(t.w);
  }
  static __device__ __forceinline__ void store(bf16_t *p, const float (&v)[8]) {
    uint4 t;
    t.x = f32_to_bf16(v[0]) | (f32_to_bf16(v[1]) << 16);
    t.y = f32_to_bf16(v[2]) | (f32_to_bf16(v[3]) << 16);
    t.z = f32_to_bf16(v[4]) | (f32_to_bf16(v[5]) << 16);
    t.w = f32_to_bf16(v[6]) | (f32_to_bf16(v[7]) << 16);
    *reinterpret_cast<uint4 *>(p) = t;
  }
};

template <>
struct Io<bf16_t, 1> {
  static __device__ __forceinline__ void load(const bf16_t *p, float (&v)[1]) {
    v[0] = __uint_as_float(static_cast<uint32_t>(*p) << 16);
  }
  static __device__ __forceinline__ void store(bf16_t *p, const float (&v)[1]) {
    *p = static_cast<bf16_t>(f32_to_bf16(v[0]));
  }
};

// logical block id: XCD k (= physical block id mod 8, observed placement,
// speed only) gets the k-th contiguous eighth of the row range.
__device__ __forceinline__ int logical_block(const KArgs &a) {
  int b = blockIdx.x;
  if (a.xcd_remap) {
    const int per = (a.nblocks + 7) >> 3;
    b = (b & 7) * per + (b >> 3);
  }
  return b;
}

// The same with a device-side row count (launches sized by the CAPACITY of the row arrays): the XCD ranges are cut over
// the blocks that hold actual rows — cut over the capacity, all work would land on the first XCDs — and the surplus
// blocks report -1 (the caller returns).  32 lane groups (row, head) per block.
__device__ __forceinline__ int logical_block_rows(const KArgs &a, long rows) {
  if (!a.nrows_dev) return logical_block(a);
  const long tiles = (rows + a.qtile - 1) / a.qtile;
  const int nb = static_cast<int>((tiles * a.qtile * a.M + 31) >> 5);
  int b = blockIdx.x;
  if (a.xcd_remap) {
    const int per = (nb + 7) >> 3;
    if ((b >> 3) >= per) return -1;
    b = (b & 7) * per + (b >> 3);
  }
  return b < nb ? b : -1;
}

// lane-group index -> (n*Q+q, m)
__device__ __forceinline__ void map_group(long G, const KArgs &a, long &nq, int &m) {
  if (a.mshift >= 0 && a.qshift >= 0) {  // shifts instead of 64-bit divisions
    const long tile = G >> (a.mshift + a.qshift);
    const int r = static_cast<int>(G & ((1L << (a.mshift + a.qshift)) - 1));
    m = r >> a.qshift;
    nq = (tile << a.qshift) + (r & ((1 << a.qshift) - 1));
    return;
  }
  if (a.qtile <= 1) {
    nq = G / a.M;
    m = static_cast<int>(G - nq * a.M);
  } else {
    const long span = static_cast<long>(a.qtile) * a.M;
    const long tile = G / span;
    const int r = static_cast<int>(G - tile * span);
    m = r / a.qtile;
    nq = tile * a.qtile + (r - m * a.qtile);
  }
}

struct Tap {
  float w00, w01, w10, w11;  // bilinear weights (no attention weight)
  int o00;                   // element offset of tap (y0,x0) inside the level
  int dx, dy;                // element strides to x0+1 / y0+1
  bool ok00, ok01, ok10, ok11;
  float fx, fy;
};

// Returns false when the point contributes nothing (outside (-1,W)x(-1,H)).
__device__ __forceinline__ bool make_tap(float lx, float ly, int H, int W, int pix_stride, Tap &t) {
  const float Wf = static_cast<float>(W), Hf = static_cast<float>(H);
  const float x = lx * Wf - 0.5f, y = fma_scalar(ly, Hf, -0.5f);   // (scalar_ops.h: no (x, y) pair with swapped halves)
  if (!(x > -1.f && y > -1.f && x < Wf && y < Hf)) return false;
  const float xf = floorf(x), yf = floorf(y);
  const int x0 = static_cast<int>(xf), y0 = static_cast<int>(yf);
  t.fx = x - xf;
  t.fy = y - yf;
  const float gx = 1.f - t.fx, gy = 1.f - t.fy;
  t.w00 = gy * gx; t.w01 = gy * t.fx; t.w10 = t.fy * gx; t.w11 = t.fy * t.fx;
  t.o00 = (y0 * W + x0) * pix_stride;
  t.dx = pix_stride;
  t.dy = W * pix_stride;
  const bool x0ok = x0 >= 0, x1ok = x0 + 1 < W, y0ok = y0 >= 0, y1ok = y0 + 1 < H;
  t.ok00 = y0ok && x0ok; t.ok01 = y0ok && x1ok; t.ok10 = y1ok && x0ok; t.ok11 = y1ok && x1ok;
  return true;
}

// ----------------------------------------------------------------- forward
template <typename T, int CPL, int LPG, int PT>
__global__ void __launch_bounds__(256) msda_fwd_kernel(const KArgs a) {
  constexpr int GPB = 256 / LPG;
  const int D = a.D;
  const int lig = threadIdx.x % LPG;
  const long G = static_cast<long>(logical_block(a)) * GPB + threadIdx.x / LPG;
  long nq; int m;
  map_group(G, a, nq, m);
  if (nq >= a.NQ) return;
  const int P = PT ? PT : a.P;
  const int L = a.L;
  const long n = a.row_batch ? static_cast<long>(a.row_batch[nq]) : nq / a.Q;
  const long row = nq * a.M + m;
  const int pix_stride = a.M * D;
  const T *__restrict__ vb = static_cast<const T *>(a.value) + static_cast<size_t>(n) * a.S * pix_stride + m * D + lig * CPL;
  const float2 *__restrict__ lp = reinterpret_cast<const float2 *>(a.loc) + row * L * P;
  const float *__restrict__ ap = a.attn + row * L * P;

  float acc[CPL];
#pragma unroll
  for (int c = 0; c < CPL; ++c) acc[c] = 0.f;

  for (int l = 0; l < L; ++l) {
    const int H = static_cast<int>(a.shapes[2 * l]), W = static_cast<int>(a.shapes[2 * l + 1]);
    const T *__restrict__ vl = vb + static_cast<size_t>(a.lstart[l]) * pix_stride;
#pragma unroll
    for (int p0 = 0; p0 < P; p0 += LPG) {
      float2 mxy = make_float2(-4.f, -4.f);
      float ma = 0.f;
      if (p0 + lig < P) {
        mxy = lp[l * P + p0 + lig];
        ma = ap[l * P + p0 + lig];
      }
      const int cnt = (P - p0) < LPG ? (P - p0) : LPG;
#pragma unroll
      for (int j = 0; j < cnt; ++j) {
        const float lx = __shfl(mxy.x, j, LPG), ly = __shfl(mxy.y, j, LPG), aw = __shfl(ma, j, LPG);
        Tap t;
        if (make_tap(lx, ly, H, W, pix_stride, t)) {
          float v[CPL];
          if (t.ok00) {
            Io<T, CPL>::load(vl + t.o00, v);
            const float w = t.w00 * aw;
#pragma unroll
            for (int c = 0; c < CPL; ++c) acc[c] = fmaf(w, v[c], acc[c]);
          }
          if (t.ok01) {
            Io<T, CPL>::load(vl + t.o00 + t.dx, v);
            const float w = t.w01 * aw;
#pragma unroll
            for (int c = 0; c < CPL; ++c) acc[c] = fmaf(w, v[c], acc[c]);
          }
          if (t.ok10) {
            Io<T, CPL>::load(vl + t.o00 + t.dy, v);
            const float w = t.w10 * aw;
#pragma unroll
            for (int c = 0; c < CPL; ++c) acc[c] = fmaf(w, v[c], acc[c]);
          }
          if (t.ok11) {
            Io<T, CPL>::load(vl + t.o00 + t.dy + t.dx, v);
            const float w = t.w11 * aw;
#pragma unroll
            for (int c = 0; c < CPL; ++c) acc[c] = fmaf(w, v[c], acc[c]);
          }
        }
      }
    }
  }
  Io<T, CPL>::store(static_cast<T *>(a.out) + row * D + lig * CPL, acc);
}

// ---------------------------------------------------------------- backward
template <int LPG>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
  for (int s = 1; s < LPG; s <<= 1) v += __shfl_xor(v, s, LPG);
  return v;
}

template <typename T, int CPL, int LPG, int PT>
__global__ void __launch_bounds__(256) msda_bwd_kernel(const KArgs a) {
  constexpr int GPB = 256 / LPG;
  const int D = a.D;
  const int lig = threadIdx.x % LPG;
  const long G = static_cast<long>(logical_block(a)) * GPB + threadIdx.x / LPG;
  long nq; int m;
  map_group(G, a, nq, m);
  if (nq >= a.NQ) return;
  const int P = PT ? PT : a.P;
  const int L = a.L;
  const long n = a.row_batch ? static_cast<long>(a.row_batch[nq]) : nq / a.Q;
  const long row = nq * a.M + m;
  const int pix_stride = a.M * D;
  const size_t boff = static_cast<size_t>(n) * a.S * pix_stride + m * D + lig * CPL;
  const T *__restrict__ vb = static_cast<const T *>(a.value) + boff;
  float *__restrict__ gvb = a.grad_value + boff;
  const float2 *__restrict__ lp = reinterpret_cast<const float2 *>(a.loc) + row * L * P;
  const float *__restrict__ ap = a.attn + row * L * P;
  float2 *__restrict__ glp = reinterpret_cast<float2 *>(a.grad_loc) + row * L * P;
  float *__restrict__ gap = a.grad_attn + row * L * P;

  float g[CPL];
  Io<T, CPL>::load(static_cast<const T *>(a.grad_out) + row * D + lig * CPL, g);

  for (int l = 0; l < L; ++l) {
    const int H = static_cast<int>(a.shapes[2 * l]), W = static_cast<int>(a.shapes[2 * l + 1]);
    const size_t lo = static_cast<size_t>(a.lstart[l]) * pix_stride;
    const T *__restrict__ vl = vb + lo;
    float *__restrict__ gvl = gvb + lo;
#pragma unroll
    for (int p0 = 0; p0 < P; p0 += LPG) {
      float2 mxy = make_float2(-4.f, -4.f);
      float ma = 0.f;
      if (p0 + lig < P) {
        mxy = lp[l * P + p0 + lig];
        ma = ap[l * P + p0 + lig];
      }
      float out_ga = 0.f, out_gx = 0.f, out_gy = 0.f;
      const int cnt = (P - p0) < LPG ? (P - p0) : LPG;
#pragma unroll
      for (int j = 0; j < cnt; ++j) {
        const float lx = __shfl(mxy.x, j, LPG), ly = __shfl(mxy.y, j, LPG), aw = __shfl(ma, j, LPG);
        float ga = 0.f, gx = 0.f, gy = 0.f;
        Tap t;
        if (make_tap(lx, ly, H, W, pix_stride, t)) {
          float v00[CPL], v01[CPL], v10[CPL], v11[CPL];
#pragma unroll
          for (int c = 0; c < CPL; ++c) v00[c] = v01[c] = v10[c] = v11[c] = 0.f;
          if (t.ok00) Io<T, CPL>::load(vl + t.o00, v00);
          if (t.ok01) Io<T, CPL>::load(vl + t.o00 + t.dx, v01);
          if (t.ok10) Io<T, CPL>::load(vl + t.o00 + t.dy, v10);
          if (t.ok11) Io<T, CPL>::load(vl + t.o00 + t.dy + t.dx, v11);
          const float hx = 1.f - t.fx, hy = 1.f - t.fy;
#pragma unroll
          for (int c = 0; c < CPL; ++c) {
            const float gc = g[c];
            ga = fmaf(gc, t.w00 * v00[c] + t.w01 * v01[c] + t.w10 * v10[c] + t.w11 * v11[c], ga);
            gx = fmaf(gc, hy * sub_scalar(v01[c], v00[c]) + t.fy * sub_scalar(v11[c], v10[c]), gx);
            gy = fmaf(gc, hx * sub_scalar(v10[c], v00[c]) + t.fx * sub_scalar(v11[c], v01[c]), gy);
          }
          gx *= aw * static_cast<float>(W);
          gy = mul_scalar(gy, aw * static_cast<float>(H));     // (scalar_ops.h)
          if (t.ok00) {
            const float w = t.w00 * aw;
#pragma unroll
            for (int c = 0; c < CPL; ++c) unsafeAtomicAdd(gvl + t.o00 + c, w * g[c]);
          }
          if (t.ok01) {
            const float w = t.w01 * aw;
#pragma unroll
            for (int c = 0; c < CPL; ++c) unsafeAtomicAdd(gvl + t.o00 + t.dx + c, w * g[c]);
          }
          if (t.ok10) {
            const float w = t.w10 * aw;
#pragma unroll
            for (int c = 0; c < CPL; ++c) unsafeAtomicAdd(gvl + t.o00 + t.dy + c, w * g[c]);
          }
          if (t.ok11) {
            const float w = t.w11 * aw;
#pragma unroll
            for (int c = 0; c < CPL; ++c) unsafeAtomicAdd(gvl + t.o00 + t.dy + t.dx + c, w * g[c]);
          }
        }
        ga = group_sum<LPG>(ga);
        gx = group_sum<LPG>(gx);
        gy = group_sum<LPG>(gy);
        if (lig == j) { out_ga = ga; out_gx = gx; out_gy = gy; }
      }
      if (p0 + lig < P) {
        gap[l * P + p0 + lig] = out_ga;
        glp[l * P + p0 + lig] = make_float2(out_gx, out_gy);
      }
    }
  }
}


// -------------------------------------------- backward, D = 32 (the encoder's head size)
// Measured on MI355X (tools/probes/atomic_probe.hip): fp32 global atomics retire
// ~10 G (instruction x 128-byte line) operations per second no matter how many
// dwords of the line an instruction carries — 80 G atomics/s when 8 lanes touch a
// line (the generic kernel above), 326 G/s when 32 consecutive lanes cover the
// whole line.  This kernel therefore keeps the float4 lane-group layout for the
// value loads and the channel reductions, and re-shapes only the scatter: the
// bilinear coefficient and element offset of every (row, tap) are broadcast to a
// 32-lane half-wave whose lane i owns channel i of that row's grad_out, so each
// atomic instruction updates two complete (pixel, head) lines.
template <>
struct Io<bf16_t, 4> {
  static __device__ __forceinline__ void load(const bf16_t *p, float (&v)[4]) {
    const uint2 t = *reinterpret_cast<const uint2 *>(p);
    v[0] = bf16_lo(t.x); v[1] = bf16_hi(t.x); v[2] = bf16_lo(t.y); v[3] = bf16_hi(t.y);
  }
};

// SCATTER = false: grad_loc / grad_attn only (grad_value comes from msda_bwd_lds.h).
template <typename T, int PT, bool SCATTER = true>
__global__ void __launch_bounds__(256) msda_bwd_d32_kernel(const KArgs a) {
  constexpr int LPG = 8, CPL = 4, D = 32, GPB = 256 / LPG;
  const int lane = threadIdx.x & 63;
  const int lig = lane & 7;
  const int half_lane = lane & 31;            // channel owned in the scatter phase
  const int half_base = lane & 32;            // first lane of my half-wave
  const long G = static_cast<long>(logical_block(a)) * GPB + threadIdx.x / LPG;
  long nq; int m;
  map_group(G, a, nq, m);
  const bool active = nq < a.NQ;
  if (!active) nq = a.NQ - 1;                 // keep the lanes alive for the cross-group shuffles
  const int P = PT ? PT : a.P;
  const int L = a.L;
  const long n = a.row_batch ? static_cast<long>(a.row_batch[nq]) : nq / a.Q;
  const long row = nq * a.M + m;
  const int pix_stride = a.M * D;
  // element offset of (n, :, m, 0) — fits in 63 bits, kept as two shuffled halves
  const long long boff = static_cast<long long>(n) * a.S * pix_stride + m * D;
  const T *__restrict__ vb = static_cast<const T *>(a.value) + boff + lig * CPL;
  const float2 *__restrict__ lp = reinterpret_cast<const float2 *>(a.loc) + row * L * P;
  const float *__restrict__ ap = a.attn + row * L * P;
  float2 *__restrict__ glp = reinterpret_cast<float2 *>(a.grad_loc) + row * L * P;
  float *__restrict__ gap = a.grad_attn + row * L * P;

  float g[CPL];
  Io<T, CPL>::load(static_cast<const T *>(a.grad_out) + row * D + lig * CPL, g);

  // scatter-side state: for each of the 4 rows handled by my half-wave, my
  // channel of its grad_out and the base of its (n, m) gradient slice
  float gd[4];
  long long gbase[4];
#pragma unroll
  for (int k = 0; SCATTER && k < 4; ++k) {
    const int src = half_base + k * 8;        // first lane of that row's group
    const float e0 = __shfl(g[0], src + (half_lane >> 2), 64), e1 = __shfl(g[1], src + (half_lane >> 2), 64);
    const float e2 = __shfl(g[2], src + (half_lane >> 2), 64), e3 = __shfl(g[3], src + (half_lane >> 2), 64);
    const int c = half_lane & 3;
    gd[k] = c == 0 ? e0 : (c == 1 ? e1 : (c == 2 ? e2 : e3));
    const int lo = __shfl(static_cast<int>(boff & 0xffffffffLL), src, 64);
    const int hi = __shfl(static_cast<int>(boff >> 32), src, 64);
    const int ok = __shfl(active ? 1 : 0, src, 64);
    gbase[k] = (static_cast<long long>(hi) << 32) | static_cast<unsigned int>(lo);
    if (!ok) gd[k] = 0.f;                     // rows past the end scatter nothing
  }

  for (int l = 0; l < L; ++l) {
    const int H = static_cast<int>(a.shapes[2 * l]), W = static_cast<int>(a.shapes[2 * l + 1]);
    const long long lo_off = static_cast<long long>(a.lstart[l]) * pix_stride;
    const T *__restrict__ vl = vb + lo_off;
#pragma unroll
    for (int p0 = 0; p0 < P; p0 += LPG) {
      float2 mxy = make_float2(-4.f, -4.f);
      float ma = 0.f;
      if (p0 + lig < P) {
        mxy = lp[l * P + p0 + lig];
        ma = ap[l * P + p0 + lig];
      }
      float out_ga = 0.f, out_gx = 0.f, out_gy = 0.f;
      const int cnt = (P - p0) < LPG ? (P - p0) : LPG;
#pragma unroll
      for (int j = 0; j < cnt; ++j) {
        const float lx = __shfl(mxy.x, j, LPG), ly = __shfl(mxy.y, j, LPG), aw = __shfl(ma, j, LPG);
        float ga = 0.f, gx = 0.f, gy = 0.f;
        // per-tap scatter coefficients (0 = nothing to add) and offsets of my row
        float c00 = 0.f, c01 = 0.f, c10 = 0.f, c11 = 0.f;
        int o00 = 0, dx = 0, dy = 0;
        Tap t;
        if (make_tap(lx, ly, H, W, pix_stride, t)) {
          float v00[CPL], v01[CPL], v10[CPL], v11[CPL];
#pragma unroll
          for (int c = 0; c < CPL; ++c) v00[c] = v01[c] = v10[c] = v11[c] = 0.f;
          if (t.ok00) Io<T, CPL>::load(vl + t.o00, v00);
          if (t.ok01) Io<T, CPL>::load(vl + t.o00 + t.dx, v01);
          if (t.ok10) Io<T, CPL>::load(vl + t.o00 + t.dy, v10);
          if (t.ok11) Io<T, CPL>::load(vl + t.o00 + t.dy + t.dx, v11);
          const float hx = 1.f - t.fx, hy = 1.f - t.fy;
#pragma unroll
          for (int c = 0; c < CPL; ++c) {
            const float gc = g[c];
            ga = fmaf(gc, t.w00 * v00[c] + t.w01 * v01[c] + t.w10 * v10[c] + t.w11 * v11[c], ga);
            gx = fmaf(gc, hy * sub_scalar(v01[c], v00[c]) + t.fy * sub_scalar(v11[c], v10[c]), gx);
            gy = fmaf(gc, hx * sub_scalar(v10[c], v00[c]) + t.fx * sub_scalar(v11[c], v01[c]), gy);
          }
          gx *= aw * static_cast<float>(W);
          gy = mul_scalar(gy, aw * static_cast<float>(H));     // (scalar_ops.h)
          if (t.ok00) c00 = t.w00 * aw;
          if (t.ok01) c01 = t.w01 * aw;
          if (t.ok10) c10 = t.w10 * aw;
          if (t.ok11) c11 = t.w11 * aw;
          o00 = t.o00; dx = t.dx; dy = t.dy;
        }
        ga = group_sum<LPG>(ga);
        gx = group_sum<LPG>(gx);
        gy = group_sum<LPG>(gy);
        if (lig == j) { out_ga = ga; out_gx = gx; out_gy = gy; }

        // scatter: 4 rows x 4 taps per half-wave, one full line per atomic
#pragma unroll
        for (int k = 0; SCATTER && k < 4; ++k) {
          const int src = half_base + k * 8;
          const float k00 = __shfl(c00, src, 64), k01 = __shfl(c01, src, 64);
          const float k10 = __shfl(c10, src, 64), k11 = __shfl(c11, src, 64);
          const int q00 = __shfl(o00, src, 64), qdx = __shfl(dx, src, 64), qdy = __shfl(dy, src, 64);
          float *__restrict__ gp = a.grad_value + gbase[k] + lo_off + q00 + half_lane;
          const float gk = gd[k];
          if (k00 != 0.f) unsafeAtomicAdd(gp, k00 * gk);
          if (k01 != 0.f) unsafeAtomicAdd(gp + qdx, k01 * gk);
          if (k10 != 0.f) unsafeAtomicAdd(gp + qdy, k10 * gk);
          if (k11 != 0.f) unsafeAtomicAdd(gp + qdy + qdx, k11 * gk);
        }
      }
      if (active && p0 + lig < P) {
        gap[l * P + p0 + lig] = out_ga;
        glp[l * P + p0 + lig] = make_float2(out_gx, out_gy);
      }
    }
  }
}

// --------------------------------------------- any-D fallback (1 lane / channel)
template <typename T>
__global__ void __launch_bounds__(256) msda_fwd_scalar_kernel(const KArgs a) {
  const long tid = static_cast<long>(blockIdx.x) * 256 + threadIdx.x;
  const long total = a.NQ * a.M * a.D;
  if (tid >= total) return;
  const int D = a.D, P = a.P, L = a.L;
  const int c = static_cast<int>(tid % D);
  const long row = tid / D;
  const int m = static_cast<int>(row % a.M);
  const long n = a.row_batch ? static_cast<long>(a.row_batch[row / a.M]) : (row / a.M) / a.Q;
  const int pix_stride = a.M * D;
  const T *vb = static_cast<const T *>(a.value) + static_cast<size_t>(n) * a.S * pix_stride + m * D + c;
  float acc[1] = {0.f};
  for (int l = 0; l < L; ++l) {
    const int H = static_cast<int>(a.shapes[2 * l]), W = static_cast<int>(a.shapes[2 * l + 1]);
    const T *vl = vb + static_cast<size_t>(a.lstart[l]) * pix_stride;
    for (int p = 0; p < P; ++p) {
      const long pi = (row * L + l) * P + p;
      const float aw = a.attn[pi];
      Tap t;
      if (!make_tap(a.loc[2 * pi], a.loc[2 * pi + 1], H, W, pix_stride, t)) continue;
      float v[1];
      if (t.ok00) { Io<T, 1>::load(vl + t.o00, v); acc[0] = fmaf(t.w00 * aw, v[0], acc[0]); }
      if (t.ok01) { Io<T, 1>::load(vl + t.o00 + t.dx, v); acc[0] = fmaf(t.w01 * aw, v[0], acc[0]); }
      if (t.ok10) { Io<T, 1>::load(vl + t.o00 + t.dy, v); acc[0] = fmaf(t.w10 * aw, v[0], acc[0]); }
      if (t.ok11) { Io<T, 1>::load(vl + t.o00 + t.dy + t.dx, v); acc[0] = fmaf(t.w11 * aw, v[0], acc[0]); }
    }
  }
  Io<T, 1>::store(static_cast<T *>(a.out) + tid, acc);
}

// grad_loc / grad_attn must be zero on entry for this kernel (capi zeroes them).
template <typename T>
__global__ void __launch_bounds__(256) msda_bwd_scalar_kernel(const KArgs a) {
  const long tid = static_cast<long>(blockIdx.x) * 256 + threadIdx.x;
  const long total = a.NQ * a.M * a.D;
  if (tid >= total) return;
  const int D = a.D, P = a.P, L = a.L;
  const int c = static_cast<int>(tid % D);
  const long row = tid / D;
  const int m = static_cast<int>(row % a.M);
  const long n = a.row_batch ? static_cast<long>(a.row_batch[row / a.M]) : (row / a.M) / a.Q;
  const int pix_stride = a.M * D;
  const size_t boff = static_cast<size_t>(n) * a.S * pix_stride + m * D + c;
  const T *vb = static_cast<const T *>(a.value) + boff;
  float *gvb = a.grad_value + boff;
  float g[1];
  Io<T, 1>::load(static_cast<const T *>(a.grad_out) + tid, g);
  for (int l = 0; l < L; ++l) {
    const int H = static_cast<int>(a.shapes[2 * l]), W = static_cast<int>(a.shapes[2 * l + 1]);
    const size_t lo = static_cast<size_t>(a.lstart[l]) * pix_stride;
    for (int p = 0; p < P; ++p) {
      const long pi = (row * L + l) * P + p;
      const float aw = a.attn[pi];
      Tap t;
      if (!make_tap(a.loc[2 * pi], a.loc[2 * pi + 1], H, W, pix_stride, t)) continue;
      float v00[1] = {0.f}, v01[1] = {0.f}, v10[1] = {0.f}, v11[1] = {0.f};
      if (t.ok00) { Io<T, 1>::load(vb + lo + t.o00, v00); unsafeAtomicAdd(gvb + lo + t.o00, t.w00 * aw * g[0]); }
      if (t.ok01) { Io<T, 1>::load(vb + lo + t.o00 + t.dx, v01); unsafeAtomicAdd(gvb + lo + t.o00 + t.dx, t.w01 * aw * g[0]); }
      if (t.ok10) { Io<T, 1>::load(vb + lo + t.o00 + t.dy, v10); unsafeAtomicAdd(gvb + lo + t.o00 + t.dy, t.w10 * aw * g[0]); }
      if (t.ok11) { Io<T, 1>::load(vb + lo + t.o00 + t.dy + t.dx, v11); unsafeAtomicAdd(gvb + lo + t.o00 + t.dy + t.dx, t.w11 * aw * g[0]); }
      const float hx = 1.f - t.fx, hy = 1.f - t.fy;
      unsafeAtomicAdd(a.grad_attn + pi, g[0] * (t.w00 * v00[0] + t.w01 * v01[0] + t.w10 * v10[0] + t.w11 * v11[0]));
      unsafeAtomicAdd(a.grad_loc + 2 * pi, g[0] * aw * W * (hy * sub_scalar(v01[0], v00[0]) + t.fy * sub_scalar(v11[0], v10[0])));
      unsafeAtomicAdd(a.grad_loc + 2 * pi + 1, g[0] * aw * H * (hx * sub_scalar(v10[0], v00[0]) + t.fx * sub_scalar(v11[0], v01[0])));
    }
  }
}

}  // namespace bevmsda
