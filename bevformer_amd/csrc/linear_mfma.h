// Dense projections of the encoder layer on the CDNA4 matrix cores (gfx950).
//
//   y[m, n] = act( sum_k A[m, k] * W[n, k] + bias[n] ),   A = [ x0 (+ a0) | x1 (+ a1) ]
//
// i.e. torch.nn.functional.linear over a row-major activation matrix whose K axis may be the
// concatenation of two sources, each with an optional element-wise addend.  These are the
// value / sampling-offset / attention-weight / output projections and the two FFN layers of
// BEVFormerLayer (temporal_self_attention.py:186-211,267; spatial_cross_attention.py:173,
// 334-348; mmcv FFN), all fp32 in the reference.  The two-source form is TemporalSelfAttention's
// `cat([value[:bs], query + query_pos], -1)` (temporal_self_attention.py:186-197) read in
// place: no concatenated copy, no separate add.
//
// Why not the f32 MFMA: gfx950 has no xf32/TF32 path and v_mfma_f32_32x32x2_f32 runs at the
// f32 vector rate (157 TFLOP/s), 1/16 of the bf16 matrix rate.  With K = 256..512 these GEMMs
// are then compute-bound at ~100 TFLOP/s although their HBM time is 3-4x shorter.  Here each
// f32 operand is split in registers into two bf16 terms
//       x = hi + lo,   hi = bf16(x),   lo = bf16(x - hi)        (|x - hi - lo| <= 2^-17 |x|)
// and the product is accumulated in f32 from three bf16 MFMAs
//       hi*hi + hi*lo + lo*hi                                    (dropped lo*lo <= 2^-16 |x||w|)
// on v_mfma_f32_32x32x16_bf16: 16/3 = 5.3x the f32 MFMA rate at >= 16 mantissa bits per
// product (TF32, which the reference's A100 GEMMs use by default on its pinned torch 1.9,
// keeps 10).  NPROD = 1 is the plain bf16-input variant (one product, f32 accumulate).
//
// Tiling: 256 threads = 4 wavefronts in a 2 x 2 grid, block tile 128 x 128, wavefront tile
// 64 x 64 = 2 x 2 MFMA tiles of 32 x 32 (64 accumulator VGPRs), K chunks of 32.  Both
// operands are K-contiguous rows ("NT" GEMM), so A and B fragments are built the same way:
// lane l of a wave holds row (l & 31), k = 8*(l >> 5) .. +7 of the 32 x 16 sub-tile — one
// 16-byte ds_read_b128 from a [row][32 + 8 pad] bf16 image (80-byte row stride: the 16 lanes
// of every ds_read_b128 service group hit 16 distinct 4-bank slots).  Staging: thread t loads
// 8 consecutive k of row t/4 (two float4; 4 lanes cover a 128-byte line), splits, and writes
// 16 bytes per plane; the loads of chunk c+1 are issued before the MFMAs of chunk c.
// Block -> tile map is XCD-aware: XCD x (= blockIdx % 8) owns the row panels x, x+8, ... and
// walks all column tiles of a panel back to back, so a panel is fetched from HBM once and
// re-read from that XCD's L2.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace bevmsda {

typedef __bf16 lin_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 lin_bf16x2 __attribute__((ext_vector_type(2)));
typedef float lin_f32x2 __attribute__((ext_vector_type(2)));
typedef float lin_f32x16 __attribute__((ext_vector_type(16)));

struct LinArgs {
  const float *x0, *a0, *x1, *a1;  // A sources (a* optional addends); x1 == nullptr when K1 == 0
  long ldx0, lda0, ldx1, lda1;     // row strides in floats
  const float *w;                  // (N, K0 + K1) row-major
  long ldw;
  const float *bias;               // (N) or nullptr
  float *y;
  long ldy;
  long M;
  int N, K0, K1;
  int relu;
  int nblk_m, nblk_n;
};

constexpr int kLinBM = 128, kLinBN = 128, kLinBK = 32;
constexpr int kLinRow = 40;                 // bf16 elements per LDS row: 32 + 8 pad (80 bytes)
constexpr int kLinPlane = 128 * kLinRow;    // one operand plane (128 rows)

__device__ __forceinline__ uint32_t lin_pack2(float a, float b) {
  lin_f32x2 v = {a, b};
  lin_bf16x2 r = __builtin_convertvector(v, lin_bf16x2);   // v_cvt_pk_bf16_f32, round-nearest-even
  return __builtin_bit_cast(uint32_t, r);
}

// 8 floats -> 8 bf16 "hi" (+ 8 bf16 "lo" residuals), element j in bits [16j%32 ..] of word j/2
template <bool LO>
__device__ __forceinline__ void lin_split8(const float4 &p, const float4 &q, uint4 &hi, uint4 &lo) {
  hi.x = lin_pack2(p.x, p.y);
  hi.y = lin_pack2(p.z, p.w);
  hi.z = lin_pack2(q.x, q.y);
  hi.w = lin_pack2(q.z, q.w);
  if (LO) {
    lo.x = lin_pack2(p.x - __uint_as_float(hi.x << 16), p.y - __uint_as_float(hi.x & 0xffff0000u));
    lo.y = lin_pack2(p.z - __uint_as_float(hi.y << 16), p.w - __uint_as_float(hi.y & 0xffff0000u));
    lo.z = lin_pack2(q.x - __uint_as_float(hi.z << 16), q.y - __uint_as_float(hi.z & 0xffff0000u));
    lo.w = lin_pack2(q.z - __uint_as_float(hi.w << 16), q.w - __uint_as_float(hi.w & 0xffff0000u));
  }
}

__device__ __forceinline__ float4 lin_add4(const float4 &a, const float4 &b) {
  return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
}

template <int NPROD, bool ADD>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3)))
linear_splitbf16_kernel(const LinArgs a) {
  static_assert(NPROD == 1 || NPROD == 3, "NPROD: 1 = bf16 inputs, 3 = split-f32");
  constexpr bool LO = NPROD == 3;
  // planes: [0] A hi, [1] W hi, [2] A lo, [3] W lo
  __shared__ __attribute__((aligned(16))) uint16_t lds[(LO ? 4 : 2) * kLinPlane];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;

  // XCD-aware tile map (see header)
  const int xcd = blockIdx.x & 7;
  const int seq = blockIdx.x >> 3;
  const int mt = (seq / a.nblk_n) * 8 + xcd;
  const int nt = seq % a.nblk_n;
  if (mt >= a.nblk_m) return;
  const long m0 = static_cast<long>(mt) * kLinBM;
  const int n0 = nt * kLinBN;

  // staging assignment: row srow (+64 on the second pass), k offset skq inside the chunk
  const int srow = tid >> 2;
  const int skq = (tid & 3) * 8;
  long gm[2];
  const float *wp[2];
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    const long m = m0 + p * 64 + srow;
    gm[p] = m < a.M ? m : a.M - 1;          // clamped rows are computed and never stored
    const int n = n0 + p * 64 + srow;
    wp[p] = a.w + static_cast<long>(n < a.N ? n : a.N - 1) * a.ldw + skq;
  }
  const int K = a.K0 + a.K1;

  // running source pointers of the current A segment (re-based once, where the K axis
  // switches from x0 to x1)
  const float *xp[2], *ap[2];
  bool has_add = false;
  auto set_segment = [&](bool second) {
    const float *xs = second ? a.x1 : a.x0;
    const float *as = second ? a.a1 : a.a0;
    const long ldx = second ? a.ldx1 : a.ldx0;
    const long lda = second ? a.lda1 : a.lda0;
    has_add = ADD && as != nullptr;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      xp[p] = xs + gm[p] * ldx + skq;
      ap[p] = has_add ? as + gm[p] * lda + skq : xp[p];
    }
  };

  float4 xr[2][2], wr[2][2], ar[2][2];
  bool staged_add = false;                   // the chunk held in xr has an addend in ar
  auto load_chunk = [&]() {
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      xr[p][0] = reinterpret_cast<const float4 *>(xp[p])[0];
      xr[p][1] = reinterpret_cast<const float4 *>(xp[p])[1];
      xp[p] += kLinBK;
      if (ADD && has_add) {
        ar[p][0] = reinterpret_cast<const float4 *>(ap[p])[0];
        ar[p][1] = reinterpret_cast<const float4 *>(ap[p])[1];
        ap[p] += kLinBK;
      }
      wr[p][0] = reinterpret_cast<const float4 *>(wp[p])[0];
      wr[p][1] = reinterpret_cast<const float4 *>(wp[p])[1];
      wp[p] += kLinBK;
    }
    staged_add = has_add;
  };

  lin_f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // fragment addresses (bf16 element offsets inside a plane)
  const int frow = lane & 31;
  const int fk = (lane >> 5) * 8;
  const int a_off = (wm * 64 + frow) * kLinRow + fk;
  const int b_off = (wn * 64 + frow) * kLinRow + fk;

  set_segment(false);
  load_chunk();
  for (int kc = 0; kc < K; kc += kLinBK) {
    // registers -> (+ addend) -> split -> LDS
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      uint4 hi, lo;
      const int off = (p * 64 + srow) * kLinRow + skq;
      if (ADD && staged_add) {
        xr[p][0] = lin_add4(xr[p][0], ar[p][0]);
        xr[p][1] = lin_add4(xr[p][1], ar[p][1]);
      }
      lin_split8<LO>(xr[p][0], xr[p][1], hi, lo);
      *reinterpret_cast<uint4 *>(&lds[0 * kLinPlane + off]) = hi;
      if (LO) *reinterpret_cast<uint4 *>(&lds[2 * kLinPlane + off]) = lo;
      lin_split8<LO>(wr[p][0], wr[p][1], hi, lo);
      *reinterpret_cast<uint4 *>(&lds[1 * kLinPlane + off]) = hi;
      if (LO) *reinterpret_cast<uint4 *>(&lds[3 * kLinPlane + off]) = lo;
    }
    __syncthreads();
    if (kc + kLinBK < K) {                          // in flight under the MFMAs below
      if (kc + kLinBK == a.K0) set_segment(true);
      load_chunk();
    }

#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      lin_bf16x8 ah[2], bh[2], al[2], bl[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int ao = a_off + t * 32 * kLinRow + ks * 16;
        const int bo = b_off + t * 32 * kLinRow + ks * 16;
        ah[t] = *reinterpret_cast<const lin_bf16x8 *>(&lds[0 * kLinPlane + ao]);
        bh[t] = *reinterpret_cast<const lin_bf16x8 *>(&lds[1 * kLinPlane + bo]);
        if (LO) {
          al[t] = *reinterpret_cast<const lin_bf16x8 *>(&lds[2 * kLinPlane + ao]);
          bl[t] = *reinterpret_cast<const lin_bf16x8 *>(&lds[3 * kLinPlane + bo]);
        }
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          if (LO) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
          }
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
        }
    }
    __syncthreads();
  }

  // epilogue: D tile element (row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5), col = lane & 31);
  // for a fixed r the 32 lanes of a half-wave store one contiguous 128-byte row segment
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int n = n0 + wn * 64 + j * 32 + (lane & 31);
    const bool nok = n < a.N;
    const float bv = (a.bias && nok) ? a.bias[n] : 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const long mb = m0 + wm * 64 + i * 32 + 4 * (lane >> 5);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const long m = mb + (r & 3) + 8 * (r >> 2);
        float v = acc[i][j][r] + bv;
        if (a.relu) v = v < 0.f ? 0.f : v;      // NaN stays NaN, as torch.relu
        if (nok && m < a.M) a.y[m * a.ldy + n] = v;
      }
    }
  }
}

}  // namespace bevmsda
