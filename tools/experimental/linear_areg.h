// Dense projection, activation-stationary form for the LONG-N projections (the hoisted value projections of all
// encoder layers: N = 1536, K = 256, 80 k - 185 k rows, 1.1 GB of output).
//
// Same contract as linear_splitbf16_kernel (linear_mfma.h).  What tools/gemm_diag measured on that kernel for this
// shape: every 128 x 128 tile re-loads and re-splits its activation rows (12 times per row for N = 1536), and the
// 64 KB of stores at the end of a tile do not overlap anything.  Here a wavefront keeps ITS 32 ROWS — all of
// K = 256, split into (hi, lo) bf16 MFMA fragments: 128 VGPRs — in registers for the whole kernel: a row is read
// from memory once and split once.  The 8 wavefronts of a workgroup (256 rows) walk the N / 128 column tiles
// together; the packed weight image streams through a 6-stage LDS ring by LDS-DMA, five chunks ahead, across
// column-tile boundaries (one barrier per 32-deep chunk, no LDS write traffic at all); a wavefront's 32 x 128
// accumulator is stored at the end of a column tile and the next tile's MFMAs run while those stores drain.
// k order inside a chunk is permuted as in linear_ws.h (lane half h owns k = 16 h .. 16 h + 15).
//
// Covered: K = 256 exactly, one activation source without addend, float4-epilogue conditions (linear_dma.h).
#pragma once
#include <type_traits>
#include <utility>
#include "linear_mfma.h"

namespace bevmsda {

constexpr int kAregThreads = 512;
constexpr int kAregRows = 256;                        // rows per workgroup (32 per wavefront)
constexpr int kAregRow = 40, kAregPlane = 128 * kAregRow;

template <class F, int... Is>
__device__ __forceinline__ void areg_static_for(F &&f, std::integer_sequence<int, Is...>) {
  (f(std::integral_constant<int, Is>{}), ...);
}

// s_waitcnt vmcnt(n) for a wave-uniform n known only at run time (the immediate has to be a constant)
template <int N>
__device__ __forceinline__ void areg_wait_vm_c() {
  __builtin_amdgcn_s_waitcnt(0x0F70 | (N & 15) | ((N >> 4) << 14));
}
__device__ __forceinline__ void areg_wait_vm(int n) {
  switch (n) {            // (LEAD - 1) * {1, 2, 3} LDS-DMA instructions (+ 16 stores)
    case 4: areg_wait_vm_c<4>(); break;
    case 8: areg_wait_vm_c<8>(); break;
    case 12: areg_wait_vm_c<12>(); break;
    case 20: areg_wait_vm_c<20>(); break;
    case 24: areg_wait_vm_c<24>(); break;
    case 28: areg_wait_vm_c<28>(); break;
    default: areg_wait_vm_c<0>(); break;
  }
}

template <int NPROD>
__global__ void __launch_bounds__(kAregThreads) __attribute__((amdgpu_waves_per_eu(2, 2)))
linear_areg_kernel(const LinArgs a) {
  static_assert(NPROD == 1 || NPROD == 3, "NPROD");
  constexpr bool LO = NPROD == 3;
  constexpr int NPL = LO ? 2 : 1;
  constexpr int NCH = 8;                              // K = 256
  constexpr int STAGE = NPL * kAregPlane;             // bf16 elements of a weight chunk
  constexpr int WPIECES = STAGE / 8;                  // 16-byte pieces (1280 / 640)
  constexpr int WITER = (WPIECES + kAregThreads - 1) / kAregThreads;    // LDS-DMA instructions per thread and chunk
  constexpr int NSTORE = 16;                          // float4 stores per wavefront and column tile
  // weight ring: the LDS-DMA of a chunk is issued LEAD chunks before its MFMAs.  Two ahead (3 stages) left every
  // iteration waiting on the L2 -> LDS latency (2.3 us per chunk measured, against 0.64 us of MFMA work): the
  // stores of the column tiles keep the memory pipeline busy
  constexpr int NSTAGE = 6, LEAD = NSTAGE - 1;
  extern __shared__ __attribute__((aligned(16))) uint16_t ring[];     // [NSTAGE][STAGE]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int frow = lane & 31, h = lane >> 5;
  const long m0 = static_cast<long>(blockIdx.x) * kAregRows + wave * 32;
  const long mrow = m0 + frow;
  const bool mok = mrow < a.M;
  const int ntiles = a.nblk_n;
  const int total = ntiles * NCH;                     // weight chunks in walking order

  // my 32 rows, all of K, as MFMA operand fragments
  lin_bf16x8 ah[2 * NCH], al[2 * NCH];
  {
    const float *xp = a.x0 + (mok ? mrow : a.M - 1) * a.ldx0 + h * 16;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const float4 *p = reinterpret_cast<const float4 *>(xp + c * 32);
      const float4 v0 = p[0], v1 = p[1], v2 = p[2], v3 = p[3];
      uint4 hi, lo;
      lin_split8<LO>(v0, v1, hi, lo);
      ah[2 * c] = __builtin_bit_cast(lin_bf16x8, hi);
      if (LO) al[2 * c] = __builtin_bit_cast(lin_bf16x8, lo);
      lin_split8<LO>(v2, v3, hi, lo);
      ah[2 * c + 1] = __builtin_bit_cast(lin_bf16x8, hi);
      if (LO) al[2 * c + 1] = __builtin_bit_cast(lin_bf16x8, lo);
    }
  }

  const uint4 *wbase = reinterpret_cast<const uint4 *>(a.wpack);     // chunk g at + g * (2 * PLANE / 8) uint4
  auto dma_w = [&](int g, int stage) {
    const uint4 *src = wbase + static_cast<long>(g) * (2 * kAregPlane / 8);
    uint16_t *dst0 = ring + stage * STAGE;
#pragma unroll
    for (int i = 0; i < WITER; ++i) {
      if (WPIECES % kAregThreads == 0 || (i * kAregThreads + (tid & ~63)) < WPIECES) {    // wave-uniform tail guard
        uint16_t *dst = dst0 + (i * kAregThreads + (tid & ~63)) * 8;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + i * kAregThreads + tid),
                                         (__attribute__((address_space(3))) void *)(dst), 16, 0, 0);
      }
    }
  };
  // LDS-DMA instructions THIS wavefront issues per chunk (the tail guard is wave-uniform: 3 or 2 with two planes,
  // 2 or 1 with one): the wait counts below are per wavefront
  const int nd = __builtin_amdgcn_readfirstlane((WPIECES / 64 - wave + 7) / 8);
  const int nst = m0 < a.M ? NSTORE : 0;              // a wavefront without rows issues no stores (N % 128 == 0:
                                                      // every other wavefront issues all 16 per column tile)

  lin_f32x16 acc[4];
  const int b_off = frow * kAregRow + h * 16;         // + j * 32 * ROW + ks * 8 (+ plane)

  // (Tried and dropped: delaying workgroup b by (b mod 8) / 8 of a column-tile period with s_sleep so that the
  // store bursts of the workgroups spread over the period instead of coinciding — no change, 702 vs 693 us.)

  for (int g = 0; g < LEAD && g < total; ++g) dma_w(g, g);
  int stage = 0;                                      // stage of chunk g
  for (int nt = 0; nt < ntiles; ++nt) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    areg_static_for([&](auto ctag) {
      constexpr int c = decltype(ctag)::value;
      const int g = nt * NCH + c;
      // chunk g has landed: behind its LDS-DMA this wavefront issued those of chunks g + 1 .. g + LEAD - 1 and,
      // when a column tile ended in between, that tile's stores
      if (g + LEAD - 1 < total) areg_wait_vm(nd * (LEAD - 1) + ((c <= LEAD - 1 && nt > 0) ? nst : 0));
      else areg_wait_vm(0);                           // the last chunks: everything
      asm volatile("s_barrier" ::: "memory");
      if (g + LEAD < total) dma_w(g + LEAD, stage >= 1 ? stage - 1 : NSTAGE - 1);   // = (stage + LEAD) % NSTAGE
      __builtin_amdgcn_sched_barrier(0);
      const uint16_t *ws = ring + stage * STAGE;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const uint16_t *wf = ws + b_off + j * 32 * kAregRow + ks * 8;
          const lin_bf16x8 bh = *reinterpret_cast<const lin_bf16x8 *>(wf);
          lin_bf16x8 bl;
          if (LO) bl = *reinterpret_cast<const lin_bf16x8 *>(wf + kAregPlane);
          // D[n][m]: W fragment as the A operand (float4 epilogue)
          if (LO) {
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh, al[2 * c + ks], acc[j], 0, 0, 0);
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bl, ah[2 * c + ks], acc[j], 0, 0, 0);
          }
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh, ah[2 * c + ks], acc[j], 0, 0, 0);
        }
      stage = stage == NSTAGE - 1 ? 0 : stage + 1;
    }, std::make_integer_sequence<int, NCH>{});

    // this column tile of my rows: 16 float4 stores (always issued — masked lanes aside — so that the wait
    // counts above hold)
    const int n0 = nt * 128;
    const int grp = a.group_cols > 0 ? n0 / a.group_cols : 0;
    float *yrow = a.y + (static_cast<long>(grp) * a.M + (mok ? mrow : 0)) * a.ldy - grp * a.group_cols;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int nb = n0 + j * 32 + 4 * h;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int n = nb + 8 * q;
        float4 v = make_float4(acc[j][4 * q], acc[j][4 * q + 1], acc[j][4 * q + 2], acc[j][4 * q + 3]);
        const bool ok = mok && n < a.N;
        if (a.bias && n < a.N) v = lin_add4(v, *reinterpret_cast<const float4 *>(a.bias + n));
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
          if (ok) *reinterpret_cast<uint2 *>(yb) = pk;
        } else {
          if (ok) *reinterpret_cast<float4 *>(yrow + n) = v;
        }
      }
    }
  }
}

}  // namespace bevmsda
