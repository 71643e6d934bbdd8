// Dense projection, second kernel: both operands reach LDS by LDS-DMA, one barrier per K chunk.
//
// Same contract as linear_splitbf16_kernel (linear_mfma.h): y = act([x0 | x1] W^T + b), fp32 in / out,
// every product from three bf16 MFMAs over split operands (NPROD = 3) or one over rounded operands
// (NPROD = 1), packed weight image.  What changed is the way the ACTIVATIONS travel.  The first kernel
// loads them into registers, splits them, writes the bf16 planes to LDS and synchronises twice per
// 32-deep chunk; measured, its time barely moves when the MFMA work is cut to a third (--gemm bf16:
// 2.3 vs 2.6 ms per frame), i.e. the loop is bound by that load -> convert -> write -> barrier chain,
// not by the matrix cores.  Here the fp32 activations go global -> LDS directly
// (global_load_lds_dwordx4: no VGPRs, no VALU, no ds_write), into a 2-stage ring next to the weight
// chunks; the DMA of chunk c + 1 is issued right after the one barrier of chunk c and lands under its
// MFMAs; each wavefront converts its own A fragments fp32 -> (hi, lo) bf16 in registers after the
// ds_read (VALU work that co-issues with the MFMAs).
//
// LDS image of an activation chunk: [128 rows][8 x 16 B], lane-linear as the DMA requires, with the
// 16-byte pieces of row r stored at position p = c ^ ((r >> 1) & 7) (the swizzle is applied to the
// SOURCE address of the DMA and again on the fragment read): the 16 lanes of every ds_read_b128 service
// group then hit 16 distinct 4-bank slots.
//
// Not covered (the caller keeps linear_splitbf16_kernel): element-wise addends, the gather A-load,
// fp32 (unpacked) weights.
#pragma once
#include "linear_mfma.h"

namespace bevmsda {

constexpr int kDmaBM = 128, kDmaBN = 128, kDmaBK = 32;
constexpr int kDmaAStage = kDmaBM * kDmaBK * 4;                  // 16 KB: fp32 activations of a chunk
constexpr int kDmaWRow = kDmaBK + 8;                             // bf16 elements per packed weight row
constexpr int kDmaWPlane = 128 * kDmaWRow;                       // one plane of a packed weight chunk

template <int NPROD>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2)))
linear_dma_kernel(const LinArgs a) {
  static_assert(NPROD == 1 || NPROD == 3, "NPROD");
  constexpr bool LO = NPROD == 3;
  constexpr int NPL = LO ? 2 : 1;
  constexpr int WSTAGE = NPL * kDmaWPlane * 2;                   // bytes of a weight chunk in LDS
  constexpr int WPIECES = WSTAGE / 16;                           // 1280 (640) 16-byte pieces
  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * kDmaAStage + 2 * WSTAGE];
  unsigned char *const lds_a = lds;
  unsigned char *const lds_w = lds + 2 * kDmaAStage;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  // XCD-aware tile map (linear_mfma.h)
  const int xcd = blockIdx.x & 7, seq = blockIdx.x >> 3;
  const int mt = (seq / a.nblk_n) * 8 + xcd;
  const int nt = seq % a.nblk_n;
  if (mt >= a.nblk_m) return;
  const long m0 = static_cast<long>(mt) * kDmaBM;
  const int n0 = nt * kDmaBN;
  const int K = a.K0 + a.K1;
  const int nchunks = K / kDmaBK;

  // DMA assignment, activations: 1024 pieces per chunk = 4 per thread; piece id -> (row, position)
  const float *arow0[4], *arow1[4];
  int asrc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int id = i * 256 + tid;
    const int row = id >> 3, pos = id & 7;
    long m = m0 + row;
    if (m >= a.M) m = a.M - 1;                                   // clamped rows are computed, never stored
    arow0[i] = a.x0 + m * a.ldx0;
    arow1[i] = a.K1 > 0 ? a.x1 + m * a.ldx1 : nullptr;
    asrc[i] = (pos ^ ((row >> 1) & 7)) * 4;                      // source float offset inside the chunk
  }
  const uint4 *wchunk0 = reinterpret_cast<const uint4 *>(a.wpack) +
                         static_cast<long>(nt) * nchunks * (2 * kDmaWPlane / 8);

  auto issue = [&](int c, int stage) {
    const int kc = c * kDmaBK;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float *src = kc < a.K0 ? arow0[i] + kc + asrc[i] : arow1[i] + (kc - a.K0) + asrc[i];
      unsigned char *dst = lds_a + stage * kDmaAStage + (i * 256 + (tid & ~63)) * 16;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                       (__attribute__((address_space(3))) void *)dst, 16, 0, 0);
    }
    const uint4 *wsrc = wchunk0 + static_cast<long>(c) * (2 * kDmaWPlane / 8);
#pragma unroll
    for (int i = 0; i < (WPIECES + 255) / 256; ++i) {
      if (WPIECES % 256 == 0 || (i * 256 + (tid & ~63)) < WPIECES) {   // wave-uniform tail guard
        unsigned char *dst = lds_w + stage * WSTAGE + (i * 256 + (tid & ~63)) * 16;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(wsrc + i * 256 + tid),
                                         (__attribute__((address_space(3))) void *)dst, 16, 0, 0);
      }
    }
  };

  lin_f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // fragment addresses
  const int frow = lane & 31, fk8 = lane >> 5;
  const int b_off = (wn * 64 + frow) * kDmaWRow + fk8 * 8;       // bf16 elements inside a plane

  issue(0, 0);
  for (int c = 0; c < nchunks; ++c) {
    __syncthreads();            // chunk c has landed (the compiler drains the DMA here); stage (c + 1) & 1 is free
    if (c + 1 < nchunks) issue(c + 1, (c + 1) & 1);
    const unsigned char *as = lds_a + (c & 1) * kDmaAStage;
    const uint16_t *ws = reinterpret_cast<const uint16_t *>(lds_w + (c & 1) * WSTAGE);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      lin_bf16x8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int r = wm * 64 + t * 32 + frow;
        const int c0 = ks * 4 + fk8 * 2;
        const int sw = (r >> 1) & 7;
        const float4 p = *reinterpret_cast<const float4 *>(as + r * 128 + ((c0 ^ sw) << 4));
        const float4 q = *reinterpret_cast<const float4 *>(as + r * 128 + (((c0 + 1) ^ sw) << 4));
        uint4 hi, lo;
        lin_split8<LO>(p, q, hi, lo);
        ah[t] = __builtin_bit_cast(lin_bf16x8, hi);
        if (LO) al[t] = __builtin_bit_cast(lin_bf16x8, lo);
        const int bo = b_off + t * 32 * kDmaWRow + ks * 16;
        bh[t] = *reinterpret_cast<const lin_bf16x8 *>(&ws[bo]);
        if (LO) bl[t] = *reinterpret_cast<const lin_bf16x8 *>(&ws[kDmaWPlane + bo]);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {     // D[n][m]: W fragment as the A operand (float4 epilogue)
          if (LO) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh[j], al[i], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bl[j], ah[i], acc[i][j], 0, 0, 0);
          }
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh[j], ah[i], acc[i][j], 0, 0, 0);
        }
    }
  }

  // epilogue: as the transposed-tile float4 epilogue of linear_splitbf16_kernel (N, ldy, group_cols multiples
  // of 4 and a 16-byte aligned y / bias are launch conditions of this kernel)
  const int grp = a.group_cols > 0 ? n0 / a.group_cols : 0;
  float *const yg = a.y + static_cast<long>(grp) * a.M * a.ldy;
  const int ncol0 = grp * a.group_cols;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const long m = m0 + wm * 64 + i * 32 + (lane & 31);
    float *yrow = yg + (m < a.M ? m : 0) * a.ldy - ncol0;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int nb = n0 + wn * 64 + j * 32 + 4 * (lane >> 5);
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int n = nb + 8 * g;
        if (m < a.M && n < a.N) {
          float4 v = make_float4(acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]);
          if (a.bias) v = lin_add4(v, *reinterpret_cast<const float4 *>(a.bias + n));
          if (a.relu) {
            v.x = v.x < 0.f ? 0.f : v.x;
            v.y = v.y < 0.f ? 0.f : v.y;
            v.z = v.z < 0.f ? 0.f : v.z;
            v.w = v.w < 0.f ? 0.f : v.w;
          }
          if (a.out_bf16) {
            uint2 pk;
            pk.x = lin_pack2(v.x, v.y);
            pk.y = lin_pack2(v.z, v.w);
            uint16_t *yb = reinterpret_cast<uint16_t *>(a.y) + (yrow - a.y) + n;
            *reinterpret_cast<uint2 *>(yb) = pk;
          } else {
            *reinterpret_cast<float4 *>(yrow + n) = v;
          }
        }
      }
    }
  }
}

}  // namespace bevmsda
