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
  const float *w;                  // (N, K0 + K1) row-major (WMODE 0)
  long ldw;
  const uint16_t *wpack;           // pre-split weight image (WMODE > 0), see lin_pack_weight_kernel
  const int32_t *gidx;             // gather mode (ADD kernels): A[m, :] = gscale[m] * sum_{j < 2, gidx[m, j] >= 0}
  const float *gscale;             //   x0[gidx[m, j], :]  — SpatialCrossAttention's camera mean folded into
                                   //   the A-load of its output projection (spatial_cross_attention.py:165-173)
  const float *bias;               // (N) or nullptr
  float *y;
  long ldy;
  long M;
  int N, K0, K1;
  int relu;
  int group_cols;                  // > 0: output column n goes to matrix n / group_cols (each (M, ldy))
  int out_bf16;                    // 1: y holds bf16 (round-to-nearest-even of the fp32 result), ldy in elements
  int accum;                       // 1: y += result (fp32 y, float4 epilogue): gradients of a tensor with several consumers
  const float *mask;               // (M, ldmask) or nullptr: y = mask > 0 ? result : 0 (float4 epilogue) — the backward of a
  long ldmask;                     //   ReLU folded into the input-gradient GEMM of the Linear behind it (mask = the ReLU's output)
  float mask_scale;                // ... times this (1 / (1 - p) when a dropout sat between the ReLU and the Linear)
  int nblk_m, nblk_n;
#ifdef BEVMSDA_LIN_DIAG
  int diag;                        // tools/gemm_diag only: bit 0 no MFMA, 1 no stores, 2 A loads of chunk 0 only,
                                   //   3 W copy of chunk 0 only, 4 no A split / LDS write after chunk 0
#endif
};
#ifdef BEVMSDA_LIN_DIAG
#define LIN_DIAG(a, bit) (((a).diag >> (bit)) & 1)
#else
#define LIN_DIAG(a, bit) 0
#endif

constexpr int kLinBM = 128, kLinBN = 128;   // kLinBN: column-tile granularity of the packed weight image
constexpr int kLinKGran = 32;               // K0 and K1 must be multiples of this

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

// NPROD: 3 = split-f32, 1 = bf16 inputs.  ADD: an addend pointer may be present.
// BK: K elements staged per barrier pair (32: 40 KB LDS, 3 blocks / CU; 64: 72 KB, 2 blocks / CU
// with twice the bytes in flight per thread and half the barriers).
// WMODE: where the W operand comes from.  0 = the fp32 weight matrix, split in registers like
// the activations; 1..3 = the pre-split bf16 image written once per weight version by
// lin_pack_weight_kernel (per 128-row x 32-k chunk: [hi plane | lo plane] already in the padded
// LDS row format, 20 KB contiguous), copied 1 = through registers, 2 = by LDS-DMA
// (global_load_lds_dwordx4) into a double-buffered W area, 3 = by LDS-DMA into a single W area
// (issued after the chunk's last fragment read).  WMODE > 0 requires BK = 32.
// SWAP: the MFMA computes the transposed tile (W rows as the A operand), which leaves 4
// consecutive output columns in 4 consecutive accumulator registers of a lane -> 16-byte
// stores (4x fewer store instructions); SWAP = false stores dwords in 128-byte row segments.
// BN: block tile width.  256 (packed weights by LDS-DMA only) doubles the wavefront tile to
// 64 x 128: the activation tile is loaded, split and written to LDS once per 256 output columns
// instead of twice, 12 fragment reads feed 24 MFMAs instead of 8 feeding 12; 128 accumulator
// VGPRs and 60 KB of LDS leave 2 blocks per CU.
constexpr int lin_waves_per_eu(int bk, int wmode, bool add, int bn) {
  return bk == 64 || wmode == 2 || bn == 256 ? 2 : (wmode == 3 && !add ? 4 : 3);   // by LDS bytes and VGPR need
}

// FRAGS: all MFMA fragments of a chunk (both k-steps, both operands: 64 VGPRs) are read into
// registers first, ONE barrier follows, and the LDS-DMA of the next weight chunk is issued
// before the 24 MFMAs instead of after them: the MFMA phase has no LDS dependence and both
// operand streams of the next chunk are in flight under it (3 blocks / CU by VGPRs).
// BM: rows per block.  64 (packed weights by LDS-DMA only): the four wavefronts sit side by side
// along N (wavefront tile 64 x 32, 32 accumulator VGPRs), the activation tile is 64 rows (10 KB):
// 30 KB of LDS and < 100 VGPRs let 5 blocks = 20 waves share a CU, and twice as many, smaller
// blocks even out the last wave of blocks on the short (M = 40 k) projections.
// (The residual + LayerNorm epilogue of round 2 lived here on 128 / 64 x 256 tiles; it lost to two launches and is
// retired — tools/experimental/linear_mfma_layernorm_epilogue.inc; the row-panel kernel carries that fusion now.)
template <int NPROD, bool ADD, int BK, bool SWAP, int WMODE, int BN = 128, bool FRAGS = false, int BM = 128>
__global__ void __launch_bounds__(256)
__attribute__((amdgpu_waves_per_eu(BM == 64 ? (BN == 256 ? 3 : (ADD ? 4 : 5)) : (FRAGS ? 3 : lin_waves_per_eu(BK, WMODE, ADD, BN)),
                                   BM == 64 ? (BN == 256 ? 3 : (ADD ? 4 : 5)) : (FRAGS ? 3 : lin_waves_per_eu(BK, WMODE, ADD, BN)))))
linear_splitbf16_kernel(const LinArgs a) {
  static_assert(BM == 128 || (BM == 64 && WMODE == 3 && BK == 32 && !FRAGS),
                "BM = 64: packed weights by LDS-DMA, 64 x (128 | 256) x 32");
  static_assert(!FRAGS || (WMODE == 3 && BN == 128 && BK == 32), "FRAGS: packed weights by LDS-DMA, 128 x 128 x 32");
  static_assert(BN == 128 || (BN == 256 && WMODE == 3), "BN = 256 needs the packed weight image by LDS-DMA");
  constexpr int NTW = BN / 128;             // packed 128-row weight tiles per block
  constexpr int WNW = BM == 64 ? 4 : 2;     // wavefronts along N (x 4 / WNW along M)
  constexpr int NJ = BN / WNW / 32;         // 32-column MFMA tiles per wavefront
  static_assert(NPROD == 1 || NPROD == 3, "NPROD: 1 = bf16 inputs, 3 = split-f32");
  static_assert(BK == 32 || BK == 64, "BK");
  static_assert(WMODE >= 0 && WMODE <= 3 && (WMODE == 0 || BK == 32), "WMODE");
  constexpr bool LO = NPROD == 3;
  constexpr int ROW = BK + 8;               // bf16 elements per LDS row (16-byte pad: the 16 lanes
                                            // of a ds_read_b128 group hit 16 distinct 4-bank slots)
  constexpr int PLANE = 128 * ROW;          // one 128-row plane (weight tiles are always 128 rows)
  constexpr int APLANE = BM * ROW;          // one activation plane
  constexpr int TPR = BK / 8;               // staging threads per row
  constexpr int RPP = 256 / TPR;            // rows per staging pass
  constexpr int NP = BM / RPP;              // staging passes
  constexpr int NPL = LO ? 2 : 1;           // planes per operand: hi (, lo)
  constexpr int WBUFS = WMODE == 2 ? 2 : 1;
  // [A hi | A lo] then WBUFS x [W hi | W lo]
  __shared__ __attribute__((aligned(16))) uint16_t lds[NPL * APLANE + WBUFS * NTW * NPL * PLANE];
  uint16_t *const lds_a = lds;
  uint16_t *const lds_w = lds + NPL * APLANE;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = BM == 64 ? 0 : wave >> 1, wn = BM == 64 ? wave : wave & 1;

  // XCD-aware tile map (see header)
  const int xcd = blockIdx.x & 7;
  const int seq = blockIdx.x >> 3;
  const int mt = (seq / a.nblk_n) * 8 + xcd;
  const int nt = seq % a.nblk_n;
  if (mt >= a.nblk_m) return;
  const long m0 = static_cast<long>(mt) * BM;
  const int n0 = nt * BN;

  // staging assignment: row srow (+RPP per pass), k offset skq inside the chunk
  const int srow = tid / TPR;
  const int skq = (tid % TPR) * 8;
  long gm[NP];
  const float *wp[NP];
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    const long m = m0 + p * RPP + srow;
    gm[p] = m < a.M ? m : a.M - 1;          // clamped rows are computed and never stored
    const int n = n0 + p * RPP + srow;
    wp[p] = WMODE == 0 ? a.w + static_cast<long>(n < a.N ? n : a.N - 1) * a.ldw + skq : nullptr;
  }
  // packed weight image: chunk (nt, kc / 32) is NPLANES_PACKED x PLANE bf16 = 20 KB contiguous
  constexpr int WCH16 = NPL * PLANE / 8;     // 16-byte pieces of a chunk that this NPROD uses
  constexpr int WPIECES = (WCH16 + 255) / 256;
  const uint4 *wchunk = WMODE > 0
      ? reinterpret_cast<const uint4 *>(a.wpack) + static_cast<long>(nt) * NTW * ((a.K0 + a.K1) / 32) * (2 * PLANE / 8)
      : nullptr;
  const long wtile_stride = static_cast<long>((a.K0 + a.K1) / 32) * (2 * PLANE / 8);   // uint4 per 128-row tile
  const int K = a.K0 + a.K1;

  // gather mode: the (up to two) source rows of every staged row and its scale, once per block
  const bool gather = ADD && a.gidx != nullptr;
  int gi0[NP], gi1[NP];
  float gs[NP];
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    gi0[p] = gather ? a.gidx[gm[p] * 2] : 0;
    gi1[p] = gather ? a.gidx[gm[p] * 2 + 1] : 0;
    gs[p] = gather ? a.gscale[gm[p]] : 1.f;
  }

  // running source pointers of the current A segment (re-based once, where the K axis
  // switches from x0 to x1); a chunk never straddles the switch (K0 % BK == 0 is checked
  // by the launcher for the BK in use)
  const float *xp[NP], *ap[NP];
  bool has_add = false;
  auto set_segment = [&](bool second) {
    if (ADD && gather) {      // single source: the two gathered rows play the parts of x and addend
      has_add = true;
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        xp[p] = a.x0 + static_cast<long>(gi0[p] < 0 ? 0 : gi0[p]) * a.ldx0 + skq;
        ap[p] = a.x0 + static_cast<long>(gi1[p] < 0 ? 0 : gi1[p]) * a.ldx0 + skq;
      }
      return;
    }
    const float *xs = second ? a.x1 : a.x0;
    const float *as = second ? a.a1 : a.a0;
    const long ldx = second ? a.ldx1 : a.ldx0;
    const long lda = second ? a.lda1 : a.lda0;
    has_add = ADD && as != nullptr;
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      xp[p] = xs + gm[p] * ldx + skq;
      ap[p] = has_add ? as + gm[p] * lda + skq : xp[p];
    }
  };

  float4 xr[NP][2], wr[NP][2], ar[NP][2];
  uint4 wq[WPIECES];                         // WMODE 1: packed W pieces in flight
  bool staged_add = false;                   // the chunk held in xr has an addend in ar
  auto load_chunk = [&]() {
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      xr[p][0] = reinterpret_cast<const float4 *>(xp[p])[0];
      xr[p][1] = reinterpret_cast<const float4 *>(xp[p])[1];
      xp[p] += BK;
      if (ADD && has_add) {
        ar[p][0] = reinterpret_cast<const float4 *>(ap[p])[0];
        ar[p][1] = reinterpret_cast<const float4 *>(ap[p])[1];
        ap[p] += BK;
      }
      if (WMODE == 0) {
        wr[p][0] = reinterpret_cast<const float4 *>(wp[p])[0];
        wr[p][1] = reinterpret_cast<const float4 *>(wp[p])[1];
        wp[p] += BK;
      }
    }
    staged_add = has_add;
  };
  // packed W chunk `c` -> registers (WMODE 1) or straight into W buffer `buf` (WMODE 2, 3:
  // LDS-DMA; destination = wave-uniform base + lane * 16, which is exactly the chunk image)
  auto load_w = [&](int c, int buf) {
#pragma unroll
    for (int t = 0; t < NTW; ++t) {
      const uint4 *src = wchunk + t * wtile_stride + static_cast<long>(c) * (2 * PLANE / 8);
#pragma unroll
      for (int i = 0; i < WPIECES; ++i) {
        const int idx = i * 256 + tid;
        if (WCH16 % 256 == 0 || (i * 256 + (tid & ~63)) < WCH16) {     // wave-uniform tail guard
          if (WMODE == 1) {
            wq[i] = src[idx];
          } else {
            uint16_t *dst = lds_w + (buf * NTW + t) * NPL * PLANE + (i * 256 + (tid & ~63)) * 8;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + idx),
                                             (__attribute__((address_space(3))) void *)(dst), 16, 0, 0);
          }
        }
      }
    }
  };

  lin_f32x16 acc[2][NJ];                     // [m tile][n tile] of the 64 x (BN / 2) wavefront tile
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // fragment addresses (bf16 element offsets inside a plane)
  const int frow = lane & 31;
  const int fk = (lane >> 5) * 8;
  const int a_off = (wm * 64 + frow) * ROW + fk;
  // BN = 128: wave column wn reads rows wn * 64 .. of the one W tile; BN = 256: W tile wn entirely
  // BM = 64, BN = 256: four wave columns of 64, two per packed W tile
  const int b_off = (BN == 256 ? (BM == 64 ? (wn >> 1) * NPL * PLANE + (wn & 1) * 64 * ROW : wn * NPL * PLANE)
                               : wn * (BN / WNW) * ROW) + frow * ROW + fk;

  set_segment(false);
  load_chunk();
  if (WMODE > 0) load_w(0, 0);
  for (int kc = 0, c = 0; kc < K; kc += BK, ++c) {
    // registers -> (+ addend) -> split -> LDS
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      if (LIN_DIAG(a, 4) && kc > 0) break;
      uint4 hi, lo;
      const int off = (p * RPP + srow) * ROW + skq;
      if (ADD && staged_add) {
        if (gather) {           // (row0 + row1) * scale with absent rows as zeros: gather_mean's arithmetic
          const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const float4 r0 = gi0[p] >= 0 ? xr[p][h] : z, r1 = gi1[p] >= 0 ? ar[p][h] : z;
            xr[p][h] = make_float4((r0.x + r1.x) * gs[p], (r0.y + r1.y) * gs[p], (r0.z + r1.z) * gs[p],
                                   (r0.w + r1.w) * gs[p]);
          }
        } else {
          xr[p][0] = lin_add4(xr[p][0], ar[p][0]);
          xr[p][1] = lin_add4(xr[p][1], ar[p][1]);
        }
      }
      lin_split8<LO>(xr[p][0], xr[p][1], hi, lo);
      *reinterpret_cast<uint4 *>(&lds_a[off]) = hi;
      if (LO) *reinterpret_cast<uint4 *>(&lds_a[APLANE + off]) = lo;
      if (WMODE == 0) {
        lin_split8<LO>(wr[p][0], wr[p][1], hi, lo);
        *reinterpret_cast<uint4 *>(&lds_w[off]) = hi;
        if (LO) *reinterpret_cast<uint4 *>(&lds_w[PLANE + off]) = lo;
      }
    }
    if (WMODE == 1) {
#pragma unroll
      for (int i = 0; i < WPIECES; ++i)
        if (WCH16 % 256 == 0 || i * 256 + tid < WCH16)
          *reinterpret_cast<uint4 *>(&lds_w[(i * 256 + tid) * 8]) = wq[i];
    }
    if (FRAGS && kc + BK < K) {                     // xr is free again: next activations first
      if (kc + BK == a.K0) set_segment(true);
      load_chunk();
    }
    __syncthreads();          // staged chunk visible (an LDS-DMA in flight is drained here too)
    const uint16_t *wcur = lds_w + (WMODE == 2 ? (c & 1) * NTW * NPL * PLANE : 0);
    if (!FRAGS && kc + BK < K) {                    // in flight under the MFMAs below
      if (kc + BK == a.K0) set_segment(true);
      if (!LIN_DIAG(a, 2)) load_chunk();
      if (WMODE == 1) load_w(c + 1, 0);
      if (WMODE == 2) load_w(c + 1, (c + 1) & 1);
    }

    if (FRAGS) {
      lin_bf16x8 ah[2][2], bh[2][2], al[2][2], bl[2][2];     // [k-step][tile]
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const int ao = a_off + t * 32 * ROW + ks * 16;
          const int bo = b_off + t * 32 * ROW + ks * 16;
          ah[ks][t] = *reinterpret_cast<const lin_bf16x8 *>(&lds_a[ao]);
          bh[ks][t] = *reinterpret_cast<const lin_bf16x8 *>(&wcur[bo]);
          if (LO) {
            al[ks][t] = *reinterpret_cast<const lin_bf16x8 *>(&lds_a[APLANE + ao]);
            bl[ks][t] = *reinterpret_cast<const lin_bf16x8 *>(&wcur[PLANE + bo]);
          }
        }
      __syncthreads();        // every fragment of this chunk is in registers: both LDS areas are free
      if (kc + BK < K) load_w(c + 1, 0);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            if (SWAP) {
              if (LO) {
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh[ks][j], al[ks][i], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bl[ks][j], ah[ks][i], acc[i][j], 0, 0, 0);
              }
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh[ks][j], ah[ks][i], acc[i][j], 0, 0, 0);
            } else {
              if (LO) {
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[ks][i], bh[ks][j], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[ks][i], bl[ks][j], acc[i][j], 0, 0, 0);
              }
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[ks][i], bh[ks][j], acc[i][j], 0, 0, 0);
            }
          }
      continue;
    }

#pragma unroll
    for (int ks = 0; ks < BK / 16; ++ks) {
      lin_bf16x8 ah[2], bh[NJ], al[2], bl[NJ];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int ao = a_off + t * 32 * ROW + ks * 16;
        ah[t] = *reinterpret_cast<const lin_bf16x8 *>(&lds_a[ao]);
        if (LO) al[t] = *reinterpret_cast<const lin_bf16x8 *>(&lds_a[APLANE + ao]);
      }
#pragma unroll
      for (int t = 0; t < NJ; ++t) {
        const int bo = b_off + t * 32 * ROW + ks * 16;
        bh[t] = *reinterpret_cast<const lin_bf16x8 *>(&wcur[bo]);
        if (LO) bl[t] = *reinterpret_cast<const lin_bf16x8 *>(&wcur[PLANE + bo]);
      }
      if (LIN_DIAG(a, 0) && kc > 0) continue;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          if (SWAP) {     // D[n][m]: W fragment as the A operand
            if (LO) {
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh[j], al[i], acc[i][j], 0, 0, 0);
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bl[j], ah[i], acc[i][j], 0, 0, 0);
            }
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh[j], ah[i], acc[i][j], 0, 0, 0);
          } else {        // D[m][n]
            if (LO) {
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
            }
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
          }
        }
    }
    __syncthreads();          // every fragment of this chunk has been read
    if (WMODE == 3 && kc + BK < K && !LIN_DIAG(a, 3)) load_w(c + 1, 0);
  }

  // grouped output: the N columns are `N / group_cols` consecutive (M, ldy) matrices (one
  // projection per encoder layer from a single pass over the shared input); a 128-column
  // tile never straddles two groups (group_cols % 128 == 0, checked by the launcher)
  const int grp = a.group_cols > 0 ? n0 / a.group_cols : 0;
  float *const yg = a.y + static_cast<long>(grp) * a.M * a.ldy;
  const int ncol0 = grp * a.group_cols;     // column of y that output column 0 of this group maps to

  // Epilogue.  MFMA D tile: lane holds (row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5), col = lane & 31).
  if (SWAP) {
    // D rows are output columns: registers 4g .. 4g+3 of a lane are n = nb + 8g .. +3 of output
    // row m = lane & 31 -> one float4 store per g (16-byte path needs N, ldy multiples of 4 and
    // a 16-byte aligned y / bias: checked once, uniform)
    const bool vec = (a.N & 3) == 0 && (a.ldy & 3) == 0 && (a.group_cols & 3) == 0 && (reinterpret_cast<uintptr_t>(a.y) & 15u) == 0 &&
                     (!a.bias || (reinterpret_cast<uintptr_t>(a.bias) & 15u) == 0);
    if (vec && a.bias && !a.mask && !a.accum && !a.out_bf16) {
      // The plain projection with a bias (optional ReLU, fp32 out; without a bias the loop below has no load either): NO load and no branch between the stores.  A load placed
      // between two stores is followed by s_waitcnt vmcnt(0) — bias may alias y as far as the compiler knows, so it keeps
      // program order — and that wait also covers the store in front of it: through round 4 every 16-byte piece sat
      // through the previous piece's store round trip, 16 per workgroup (found in the ISA, round 5).  The tile's bias
      // fragments are loaded up front; bias / ReLU are selects, not branches (a uniform branch between stores makes the
      // wait-count pass merge paths and fall back to vmcnt(0) as well).
      // ... and the arithmetic of ALL pieces comes first, in place, in straight-line code: the one wait for the bias
      // fragments then sits in front of the first store instead of inside every piece's bounds-check block.
      const bool rl = a.relu != 0;
      constexpr bool hb = true;
      const float *bsrc = a.bias;
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        if (j) __builtin_amdgcn_sched_barrier(0);   // (one MFMA tile column's 4 fragments at a time: the 128-register variants)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int n = n0 + wn * (BN / WNW) + j * 32 + 4 * (lane >> 5) + 8 * g;
          const float4 b4 = *reinterpret_cast<const float4 *>(bsrc + ((hb && n < a.N) ? n : 0));
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            float v0 = acc[i][j][4 * g], v1 = acc[i][j][4 * g + 1], v2 = acc[i][j][4 * g + 2], v3 = acc[i][j][4 * g + 3];
            v0 = hb ? v0 + b4.x : v0; v1 = hb ? v1 + b4.y : v1; v2 = hb ? v2 + b4.z : v2; v3 = hb ? v3 + b4.w : v3;
            acc[i][j][4 * g] = (rl && v0 < 0.f) ? 0.f : v0;          // NaN stays NaN, as torch.relu
            acc[i][j][4 * g + 1] = (rl && v1 < 0.f) ? 0.f : v1;
            acc[i][j][4 * g + 2] = (rl && v2 < 0.f) ? 0.f : v2;
            acc[i][j][4 * g + 3] = (rl && v3 < 0.f) ? 0.f : v3;
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const long m = m0 + wm * 64 + i * 32 + (lane & 31);
        float *yrow = yg + (m < a.M ? m : 0) * a.ldy - ncol0;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          const int nb = n0 + wn * (BN / WNW) + j * 32 + 4 * (lane >> 5);
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int n = nb + 8 * g;
            const float4 v = make_float4(acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]);
            if (LIN_DIAG(a, 1) && v.x != 1.2345e30f) continue;
            if (m < a.M && n < a.N) *reinterpret_cast<float4 *>(yrow + n) = v;      // N % 4 == 0: n < N covers n .. n+3
          }
        }
      }
    } else if (vec) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const long m = m0 + wm * 64 + i * 32 + (lane & 31);
        float *yrow = yg + (m < a.M ? m : 0) * a.ldy - ncol0;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          const int nb = n0 + wn * (BN / WNW) + j * 32 + 4 * (lane >> 5);
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int n = nb + 8 * g;
            if (m < a.M && n < a.N) {         // N % 4 == 0: n < N covers n .. n+3
              float4 v = make_float4(acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2],
                                     acc[i][j][4 * g + 3]);
              if (LIN_DIAG(a, 1) && v.x != 1.2345e30f) continue;
              if (a.bias) v = lin_add4(v, *reinterpret_cast<const float4 *>(a.bias + n));
              if (a.relu) {                   // NaN stays NaN, as torch.relu
                v.x = v.x < 0.f ? 0.f : v.x;
                v.y = v.y < 0.f ? 0.f : v.y;
                v.z = v.z < 0.f ? 0.f : v.z;
                v.w = v.w < 0.f ? 0.f : v.w;
              }
              if (a.mask) {                   // ReLU backward: pass where the forward's activation was positive
                const float4 mk = *reinterpret_cast<const float4 *>(a.mask + m * a.ldmask + n);
                v.x = mk.x > 0.f ? v.x * a.mask_scale : 0.f;
                v.y = mk.y > 0.f ? v.y * a.mask_scale : 0.f;
                v.z = mk.z > 0.f ? v.z * a.mask_scale : 0.f;
                v.w = mk.w > 0.f ? v.w * a.mask_scale : 0.f;
              }
              if (a.out_bf16) {             // same element offsets, 2-byte elements
                uint2 pk;
                pk.x = lin_pack2(v.x, v.y);
                pk.y = lin_pack2(v.z, v.w);
                uint16_t *yb = reinterpret_cast<uint16_t *>(a.y) + (yrow - a.y) + n;
                *reinterpret_cast<uint2 *>(yb) = pk;
              } else {
                if (a.accum) v = lin_add4(v, *reinterpret_cast<const float4 *>(yrow + n));
                *reinterpret_cast<float4 *>(yrow + n) = v;
              }
            }
          }
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const long m = m0 + wm * 64 + i * 32 + (lane & 31);
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          const int nb = n0 + wn * (BN / WNW) + j * 32 + 4 * (lane >> 5);
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int n = nb + 8 * (r >> 2) + (r & 3);
            if (m < a.M && n < a.N) {
              float t = acc[i][j][r] + (a.bias ? a.bias[n] : 0.f);
              if (a.relu) t = t < 0.f ? 0.f : t;
              yg[m * a.ldy + n - ncol0] = t;
            }
          }
        }
      }
    }
  } else {
    // for a fixed r the 32 lanes of a half-wave store one contiguous 128-byte row segment
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int n = n0 + wn * (BN / WNW) + j * 32 + (lane & 31);
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
          if (nok && m < a.M) yg[m * a.ldy + n - ncol0] = v;
        }
      }
    }
  }
}

// Weight image for WMODE > 0: for every (128-row tile nt, 32-deep chunk kc) a contiguous
// [plane hi | plane lo][128 rows][32 + 8 pad] bf16 block (rows >= N and the pad are zero), i.e.
// exactly the LDS image the kernel reads its B fragments from.  One thread per 8 k of a row.
// TRANSPOSED: element (n, k) of the weight is w[k * ldw + n] — the image of W^T packed straight from W (the operand of an
// input-gradient GEMM g W: no contiguous transpose in between; the matrices are a few hundred squared, the strided reads
// cost nothing next to a second launch).
template <bool TRANSPOSED>
__device__ __forceinline__ void lin_pack_weight_thread(long t, const float *__restrict__ w, long ldw, int N, int K,
                                                       uint16_t *__restrict__ blob) {
  constexpr int ROW = 40, PLANE = 128 * ROW;
  const int kch = K / 32;
  const long rows = static_cast<long>((N + 127) / 128) * 128;
  if (t >= rows * (K / 8)) return;
  const int n = static_cast<int>(t / (K / 8));
  const int k = static_cast<int>(t % (K / 8)) * 8;
  uint4 hi = make_uint4(0, 0, 0, 0), lo = hi;
  if (n < N) {
    if constexpr (TRANSPOSED) {
      const float *src = w + static_cast<long>(k) * ldw + n;
      lin_split8<true>(make_float4(src[0], src[ldw], src[2 * ldw], src[3 * ldw]),
                       make_float4(src[4 * ldw], src[5 * ldw], src[6 * ldw], src[7 * ldw]), hi, lo);
    } else {
      const float4 *src = reinterpret_cast<const float4 *>(w + static_cast<long>(n) * ldw + k);
      lin_split8<true>(src[0], src[1], hi, lo);
    }
  }
  uint16_t *chunk = blob + (static_cast<long>(n / 128) * kch + k / 32) * (2 * PLANE);
  const int off = (n % 128) * ROW + (k % 32);
  *reinterpret_cast<uint4 *>(chunk + off) = hi;
  *reinterpret_cast<uint4 *>(chunk + PLANE + off) = lo;
  if (k % 32 == 24) {       // the 16-byte row pad
    *reinterpret_cast<uint4 *>(chunk + off + 8) = make_uint4(0, 0, 0, 0);
    *reinterpret_cast<uint4 *>(chunk + PLANE + off + 8) = make_uint4(0, 0, 0, 0);
  }
}

template <bool TRANSPOSED = false>
__global__ void __launch_bounds__(256) lin_pack_weight_kernel(const float *__restrict__ w, long ldw, int N,
                                                             int K, uint16_t *__restrict__ blob) {
  lin_pack_weight_thread<TRANSPOSED>(static_cast<long>(blockIdx.x) * 256 + threadIdx.x, w, ldw, N, K, blob);
}

}  // namespace bevmsda
