// Dense projection, third kernel: weight-stationary, activations straight into MFMA fragments.
//
// Same contract as linear_splitbf16_kernel (linear_mfma.h): y = act(x W^T + b), fp32 in / out, every
// product from three bf16 MFMAs over split operands (NPROD = 3) or one over rounded operands (NPROD = 1),
// packed weight image.  What the measurements of the first kernel say (tools/gemm_diag, profiles/r2):
// with its 128 x 128 x 32 tiles staged through LDS and two barriers per K chunk, the LDS pipe (activation
// planes written and read back, weight chunk copied and read) and the MFMA pipe each need about the same
// number of cycles per chunk and do NOT overlap — a projection of the base frame spends 25 us in data
// movement with the MFMAs switched off, 10-14 us more with them on, and all blocks of a launch march in
// phase, so the epilogue's stores (at HBM write speed on their own) add to the time instead of hiding
// under other blocks' MFMAs.  The encoder's projections have a short reduction (K = 256 or 512) and a
// long row axis (40 k - 185 k rows); this kernel is shaped for that:
//
//   * W stationary: a workgroup (8 wavefronts, one per CU) copies the packed weight image of its column
//     tile — ALL of K, hi and lo planes, 132 KB for 128 columns x 256 — into LDS once and then streams
//     rows; no weight traffic and NO barrier after that copy;
//   * activations never touch LDS: the A-operand layout of v_mfma_f32_32x32x16_bf16 holds, per lane,
//     8 consecutive k of one row, so a lane loads its own 64 contiguous bytes of a row per 32-deep chunk
//     (lanes l and l + 32 cover the row's 128-byte line), splits them into (hi, lo) bf16 in registers
//     (VALU under the MFMAs) and feeds the matrix cores directly.  The k order inside a chunk is
//     permuted (lane half h owns k = 16 h .. 16 h + 15); the weight fragment is read with the same
//     permutation, the sum over k does not care;
//   * a wavefront owns a contiguous run of rows and walks it in 64-row tiles (x 128 columns: 128
//     accumulator registers), independent of the other wavefronts: compute and store phases of the 8
//     wavefronts of a CU drift apart, stores hide under MFMAs;
//   * LDS traffic per MFMA drops to a third (16 fragment reads per 48 MFMAs, nothing else).
//
// Covered: one activation source without addend (K0 = K, K1 = 0), K a multiple of 32 with
// 128 x (K + 8) x planes x 2 bytes <= 150 KB, float4 epilogue conditions (as linear_dma.h).
#pragma once
#include "linear_mfma.h"

namespace bevmsda {

constexpr int kWsBN = 128;
constexpr int kWsThreads = 512;
constexpr int kWsLdsLimit = 150 * 1024;

inline long ws_lds_bytes(int K, int nprod) { return static_cast<long>(nprod == 3 ? 2 : 1) * kWsBN * (K + 8) * 2; }

struct WsArgs {
  LinArgs l;
  int rows_per_wave;   // multiple of 32
  int slabs;           // row slabs (workgroups per column tile)
};

template <int NPROD>
__global__ void __launch_bounds__(kWsThreads) __attribute__((amdgpu_waves_per_eu(2, 2)))
linear_ws_kernel(const WsArgs g) {
  static_assert(NPROD == 1 || NPROD == 3, "NPROD");
  constexpr bool LO = NPROD == 3;
  constexpr int NPL = LO ? 2 : 1;
  constexpr int PLANE = 128 * 40;                      // bf16 elements of one plane of a packed chunk
  extern __shared__ __attribute__((aligned(16))) uint16_t ws_lds[];   // [NPL][128][K + 8]
  const LinArgs &a = g.l;
  const int K = a.K0;
  const int WROW = K + 8;
  const int nchunks = K / 32;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

  // blocks of one row slab (all column tiles) sit on one XCD: its L2 serves the re-reads of the rows
  const int xcd = blockIdx.x & 7, seq = blockIdx.x >> 3;
  const int nt = seq % a.nblk_n;
  const int slab = (seq / a.nblk_n) * 8 + xcd;
  if (slab >= g.slabs) return;
  const int n0 = nt * kWsBN;

  {  // weight image of my column tile -> LDS, once
    const uint4 *wt = reinterpret_cast<const uint4 *>(a.wpack) + static_cast<long>(nt) * nchunks * (2 * PLANE / 8);
    const int per_row = K / 8;                         // 16-byte pieces per row
    const int total = NPL * 128 * per_row;
    for (int d = tid; d < total; d += kWsThreads) {
      const int q = d % per_row, rp = d / per_row;
      const int row = rp & 127, p = rp >> 7;
      const uint4 v = wt[((q >> 2) * 2 + p) * (PLANE / 8) + row * 5 + (q & 3)];
      *reinterpret_cast<uint4 *>(&ws_lds[(p * 128 + row) * WROW + q * 8]) = v;
    }
  }
  __syncthreads();

  const int frow = lane & 31, h = lane >> 5;
  const long worker = static_cast<long>(slab) * (kWsThreads / 64) + wave;
  const long r_begin = worker * g.rows_per_wave;
  long r_end = r_begin + g.rows_per_wave;
  if (r_end > a.M) r_end = a.M;
  const uint16_t *wfrag = ws_lds + frow * WROW + h * 16;      // + j * 32 * WROW + c * 32 + ks * 8 (+ plane)

  const int grp = a.group_cols > 0 ? n0 / a.group_cols : 0;
  float *const yg = a.y + static_cast<long>(grp) * a.M * a.ldy;
  const int ncol0 = grp * a.group_cols;

  for (long r0 = r_begin; r0 < r_end; r0 += 64) {
    const bool two = r0 + 32 < r_end;                  // wave-uniform: the run ends with a 32-row half tile
    const float *xp[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      long m = r0 + i * 32 + frow;
      if (m >= a.M) m = a.M - 1;                       // clamped rows are computed, never stored
      xp[i] = a.x0 + m * a.ldx0 + h * 16;
    }
    float4 raw[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int t = 0; t < 4; ++t) raw[i][t] = reinterpret_cast<const float4 *>(xp[i])[t];

    lin_f32x16 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    for (int c = 0; c < nchunks; ++c) {
      lin_bf16x8 ah[2][2], al[2][2];                   // [row tile][k-step]
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          uint4 hi, lo;
          lin_split8<LO>(raw[i][2 * ks], raw[i][2 * ks + 1], hi, lo);
          ah[i][ks] = __builtin_bit_cast(lin_bf16x8, hi);
          if (LO) al[i][ks] = __builtin_bit_cast(lin_bf16x8, lo);
        }
      if (c + 1 < nchunks) {                           // next chunk's 64 bytes per row, in flight under the MFMAs
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int t = 0; t < 4; ++t) raw[i][t] = reinterpret_cast<const float4 *>(xp[i] + (c + 1) * 32)[t];
      }
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const uint16_t *wf = wfrag + j * 32 * WROW + c * 32 + ks * 8;
          const lin_bf16x8 bh = *reinterpret_cast<const lin_bf16x8 *>(wf);
          lin_bf16x8 bl;
          if (LO) bl = *reinterpret_cast<const lin_bf16x8 *>(wf + 128 * WROW);
          // D[n][m]: W fragment as the A operand (float4 epilogue)
          if (LO) {
            acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh, al[0][ks], acc[0][j], 0, 0, 0);
            acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bl, ah[0][ks], acc[0][j], 0, 0, 0);
          }
          acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh, ah[0][ks], acc[0][j], 0, 0, 0);
          if (two) {
            if (LO) {
              acc[1][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh, al[1][ks], acc[1][j], 0, 0, 0);
              acc[1][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bl, ah[1][ks], acc[1][j], 0, 0, 0);
            }
            acc[1][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh, ah[1][ks], acc[1][j], 0, 0, 0);
          }
        }
    }

    // epilogue: the transposed-tile float4 epilogue of linear_splitbf16_kernel
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      if (i == 1 && !two) break;
      const long m = r0 + i * 32 + frow;
      const bool mok = m < r_end;
      float *yrow = yg + (mok ? m : 0) * a.ldy - ncol0;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int nb = n0 + j * 32 + 4 * h;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int n = nb + 8 * q;
          if (mok && n < a.N) {
            float4 v = make_float4(acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]);
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
}

}  // namespace bevmsda
