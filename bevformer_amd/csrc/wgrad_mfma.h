// Weight and bias gradients of the encoder's Linear layers on the matrix cores:
//     grad_W[n, k] += sum_m G[m, n] * X[m, k]        grad_b[n] += sum_m G[m, n]
// (the TN form of the projection GEMM: the reduction runs over the ROWS of both operands — 40 k BEV
// queries, 185 k camera pixels — and the output is a few hundred squared).  The library's TN GEMMs take
// 130-560 us per layer and Linear at these shapes (tools/wgrad_ref.py), 16 ms of a 48 ms forward +
// backward base frame; their HBM floor is 10-40 us.
//
// Structure: a workgroup (256 threads, 2 x 2 wavefronts, 64 x 64 each) owns a 128 (n) x 128 (k) tile of
// grad_W and a SLICE of the rows; per 32-row chunk both operand tiles (32 x 128 fp32, natural row-major
// layout) go global -> LDS by LDS-DMA into a 2-stage ring, one barrier per chunk.  The MFMA wants, per
// lane, 8 consecutive elements ALONG THE REDUCTION (rows m) of one column — a transposed access; with
// the fp32 tile in LDS that is 8 ds_read_b32 whose 32 lanes read 32 consecutive columns (32 distinct
// banks: conflict-free with no padding).  The gathered fp32 values are split hi / lo in registers and
// every product is three bf16 MFMAs accumulated in fp32, as in the forward kernel (linear_mfma.h);
// precision 1 rounds to bf16 (one MFMA).  The 128 x 128 partial is added to grad_W with 128-byte-line
// fp32 atomics (32 lanes of a half-wave = 32 consecutive k); the column sums of G (bias gradient) ride
// along in the wavefronts of the first k tile.  Caller zeroes grad_W / grad_b.
#pragma once
#include "linear_mfma.h"

namespace bevmsda {

struct WgradArgs {
  const float *g;      // (M, ldg): gradient w.r.t. the Linear's output, N columns
  const float *x;      // (M, ldx): the Linear's input, K columns
  long ldg, ldx;
  float *gw;           // (N, ldgw)
  long ldgw;
  float *gb;           // (N) or nullptr
  long M;
  int N, K;
  int rows_per_block;  // multiple of 32
  int tiles_n, tiles_k;
};

constexpr int kWgStage = 32 * 128 * 4;       // bytes of one operand tile of a chunk

// One 128 x 128 output tile (tile index `tile` of the problem) over row slice `slice`.
template <int NPROD>
__device__ __forceinline__ void wgrad_tile(const WgradArgs &a, int tile, int slice, unsigned char *lds) {
  constexpr bool LO = NPROD == 3;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wn = wave >> 1, wk = wave & 1;
  const int tn = tile / a.tiles_k, tk = tile - tn * a.tiles_k;
  const int n0 = tn * 128, k0 = tk * 128;
  const long m_begin = static_cast<long>(slice) * a.rows_per_block;
  if (m_begin >= a.M) return;
  const long m_end = m_begin + a.rows_per_block < a.M ? m_begin + a.rows_per_block : a.M;
  const int nchunks = static_cast<int>((m_end - m_begin + 31) / 32);

  // DMA assignment: a chunk tile = 32 rows x 32 pieces of 16 B; 1024 pieces = 4 per thread and operand
  int prow[4], pcol_g[4], pcol_x[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int id = i * 256 + tid;
    prow[i] = id >> 5;
    const int c4 = (id & 31) * 4;
    // columns past the matrix are clamped (their products land in rows / columns that are never stored)
    pcol_g[i] = n0 + c4 < a.N ? n0 + c4 : (a.N - 4);
    pcol_x[i] = k0 + c4 < a.K ? k0 + c4 : (a.K - 4);
  }
  auto issue = [&](int c, int stage) {
    const long mb = m_begin + static_cast<long>(c) * 32;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      long m = mb + prow[i];
      if (m >= a.M) m = a.M - 1;                    // rows past the end: read a valid row, masked at the gather
      unsigned char *dg = lds + (stage * 2) * kWgStage + (i * 256 + (tid & ~63)) * 16;
      unsigned char *dx = lds + (stage * 2 + 1) * kWgStage + (i * 256 + (tid & ~63)) * 16;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(a.g + m * a.ldg + pcol_g[i]),
                                       (__attribute__((address_space(3))) void *)dg, 16, 0, 0);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(a.x + m * a.ldx + pcol_x[i]),
                                       (__attribute__((address_space(3))) void *)dx, 16, 0, 0);
    }
  };

  lin_f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  float colsum[2] = {0.f, 0.f};                     // bias gradient: my column of each of my two n tiles

  const int fcol = lane & 31, fm8 = (lane >> 5) * 8;
  issue(0, 0);
  for (int c = 0; c < nchunks; ++c) {
    __syncthreads();            // chunk c has landed (the DMA is drained at the barrier); the other stage is free
    if (c + 1 < nchunks) issue(c + 1, (c + 1) & 1);
    const float *gs = reinterpret_cast<const float *>(lds + ((c & 1) * 2) * kWgStage);
    const float *xs = reinterpret_cast<const float *>(lds + ((c & 1) * 2 + 1) * kWgStage);
    const long mrem = m_end - (m_begin + static_cast<long>(c) * 32);      // valid rows of this chunk (>= 1)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      lin_bf16x8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        float ga[8], xa[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int m = ks * 16 + fm8 + e;
          const bool ok = m < mrem;
          const float gv = gs[m * 128 + wn * 64 + t * 32 + fcol];
          const float xv = xs[m * 128 + wk * 64 + t * 32 + fcol];
          ga[e] = ok ? gv : 0.f;
          xa[e] = ok ? xv : 0.f;
        }
        colsum[t] += ((ga[0] + ga[1]) + (ga[2] + ga[3])) + ((ga[4] + ga[5]) + (ga[6] + ga[7]));
        uint4 hi, lo;
        lin_split8<LO>(make_float4(ga[0], ga[1], ga[2], ga[3]), make_float4(ga[4], ga[5], ga[6], ga[7]), hi, lo);
        ah[t] = __builtin_bit_cast(lin_bf16x8, hi);
        if (LO) al[t] = __builtin_bit_cast(lin_bf16x8, lo);
        lin_split8<LO>(make_float4(xa[0], xa[1], xa[2], xa[3]), make_float4(xa[4], xa[5], xa[6], xa[7]), hi, lo);
        bh[t] = __builtin_bit_cast(lin_bf16x8, hi);
        if (LO) bl[t] = __builtin_bit_cast(lin_bf16x8, lo);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {       // D[n][k]: G fragment as the A operand, X fragment as B
          if (LO) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
          }
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
        }
    }
  }

  // epilogue: D tile lane layout: column (k) = lane & 31, row (n) = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int k = k0 + wk * 64 + j * 32 + (lane & 31);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = n0 + wn * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (n < a.N && k < a.K) unsafeAtomicAdd(a.gw + static_cast<long>(n) * a.ldgw + k, acc[i][j][r]);
      }
    }
  if (a.gb && tk == 0 && wk == 0) {
    // lanes l and l + 32 hold the same column over different rows
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const float s = colsum[t] + __shfl_xor(colsum[t], 32, 64);
      const int n = n0 + wn * 64 + t * 32 + fcol;
      if (lane < 32 && n < a.N) unsafeAtomicAdd(a.gb + n, s);
    }
  }
}

template <int NPROD>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2)))
wgrad_splitbf16_kernel(const WgradArgs a) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[4 * kWgStage];     // [stage][G | X]
  // XCD-aware map: workgroup b runs on XCD b % 8 with its private L2.  All output tiles of one row slice read the
  // same rows of g and x, so they go to ONE XCD, back to back (tile fastest inside an XCD's sequence): the slice
  // is fetched from HBM once and re-read from that L2 — with tiles dealt round-robin over the XCDs every tile
  // fetched its operands from HBM again (L2 hit rate 3 %, 346 MB per launch instead of 123 MB: r2n_backward_pmc.json)
  const int ntile = a.tiles_n * a.tiles_k;
  const int xcd = blockIdx.x & 7, seq = blockIdx.x >> 3;
  wgrad_tile<NPROD>(a, seq % ntile, (seq / ntile) * 8 + xcd, lds);
}

// Several weight gradients over the SAME rows in one launch (round 4): the backward of a layer seam needs the
// gradients of two or three Linear layers at once (FFN fc1 / fc2 and the output projection; the output projection and
// the next attention's projection; ...).  One launch each costs every one of them a full round of workgroups' worth
// of epilogue atomics (workgroups x 16,384 element atomics: 26 us of a 65 us launch at 512 workgroups) and its own
// ramp; together their tiles share ONE round: the row slices get longer (fewer partial tiles per output element) and
// the atomics are paid once.  Tiles of all problems are numbered consecutively (tile0[i] = first tile of problem i).
constexpr int kWgMaxProblems = 8;
struct WgradMultiArgs {
  WgradArgs p[kWgMaxProblems];       // (M, rows_per_block identical in all of them)
  int tile0[kWgMaxProblems + 1];
  int nprob;
};

template <int NPROD>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2)))
wgrad_multi_kernel(const WgradMultiArgs a) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[4 * kWgStage];
  const int ntile = a.tile0[a.nprob];
  const int xcd = blockIdx.x & 7, seq = blockIdx.x >> 3;
  const int tile = seq % ntile;
  const int slice = (seq / ntile) * 8 + xcd;
  // (uniform selects over constant indices instead of a dynamically indexed kernel-argument array: no scratch copy,
  // one copy of the tile body)
  WgradArgs w = a.p[0];
  int t0 = 0;
#pragma unroll
  for (int i = 1; i < kWgMaxProblems; ++i)
    if (i < a.nprob && tile >= a.tile0[i]) {
      w = a.p[i];
      t0 = a.tile0[i];
    }
  wgrad_tile<NPROD>(w, tile - t0, slice, lds);
}

}  // namespace bevmsda
