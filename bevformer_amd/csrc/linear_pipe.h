// Dense projection, software-pipelined form of linear_splitbf16_kernel (linear_mfma.h).
//
// Same contract: y = act([x0 | x1] W^T + b), fp32 in / out, every product from three bf16 MFMAs over
// split operands (NPROD = 3) or one over rounded operands (NPROD = 1), packed weight image, 128 x 128 x 32
// block-chunks, 2 x 2 wavefronts of 64 x 64, transposed-tile float4 epilogue.  What tools/gemm_diag measured
// on the first kernel is that its phases ADD UP: a chunk is staged (global -> registers -> split -> LDS,
// weight chunk by LDS-DMA), a barrier, 16 fragment reads, 24 MFMAs, a barrier — the LDS pipe idles under
// the MFMAs, the matrix cores idle under the fragment reads, and a load has one chunk (~0.3 us) to come
// back.  Here every wavefront keeps all three streams in flight at once:
//
//   iteration c:   barrier (the only one)
//                  activations of chunk c + 2 (loaded two iterations ago): split -> LDS stage c & 1
//                  LDS-DMA of weight chunk c + 2        -> LDS stage  c & 1
//                  fragment reads of chunk c + 1        <- LDS stage (c + 1) & 1   -> fragment set (c + 1) & 1
//                  global loads of the activations of chunk c + 4 -> the registers just freed
//                  24 MFMAs of chunk c from fragment set c & 1 (read during iteration c - 1)
//
// Two LDS stages of [A hi | A lo] and of [W hi | W lo] (80 KB), two fragment register sets (128 VGPRs) and two
// activation register sets (32 VGPRs): 2 workgroups = 8 wavefronts per CU.  The LDS-DMA has no register
// the compiler could wait on, so its completion is awaited by count: it is issued FIRST in an iteration
// (a scheduling barrier keeps the order), the only vector-memory operations after it are the activation
// loads, and `s_waitcnt vmcnt(<their number>)` before the barrier therefore covers it.
//
// Covered: what linear_dma.h covers (packed weights, no addends / gather, float4-epilogue conditions).
#pragma once
#include <type_traits>
#include <utility>
#include "linear_mfma.h"

namespace bevmsda {

template <class F, int... Is>
__device__ __forceinline__ void pipe_static_for(F &&f, std::integer_sequence<int, Is...>) {
  (f(std::integral_constant<int, Is>{}), ...);
}
// s_waitcnt immediates (gfx9 encoding: vmcnt[3:0] | expcnt[6:4] | lgkmcnt[11:8] | vmcnt[5:4] at [15:14]); issued
// through the builtin so that the compiler's own wait-count bookkeeping sees them
constexpr int kWaitLgkm0 = 0xC07F;     // lgkmcnt(0)
constexpr int kWaitVm0 = 0x0F70;       // vmcnt(0)
constexpr int kWaitVm4 = 0x0F74;       // vmcnt(4)
// s_barrier that the compiler may not move memory operations across (the builtin alone is no fence: the
// fragment reads of the NEXT chunk were sunk below it, next to their MFMAs — and into a race with the stage's
// refill), without the vmcnt(0) a workgroup-scope fence would cost
__device__ __forceinline__ void pipe_barrier() { asm volatile("s_barrier" ::: "memory"); }
// LDS-DMA issued from inline assembly: through the builtin, the compiler puts s_waitcnt vmcnt(0) in front of fragment
// reads of the OTHER stage (it cannot tell them from reads that alias the DMA in flight) — half of the iterations
// waited for the weight chunk they had just requested.  The kernel's own counted waits order the stages.
__device__ __forceinline__ void pipe_dma16(const void *src, uint16_t *lds_dst) {
  const unsigned d = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(reinterpret_cast<size_t>((__attribute__((address_space(3))) uint16_t *)lds_dst)));
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(src), "s"(d) : "memory", "m0");
}

constexpr int kPipeRow = 40;                         // bf16 elements per LDS row (32 + 8 pad)
constexpr int kPipePlane = 128 * kPipeRow;           // one plane

// NCH = K / 32 at compile time (8: K = 256, 16: K = 512): the K loop is straight-line code, so the compiler's
// wait counts between the three streams are exact (with a real loop it falls back to vmcnt(0) / lgkmcnt(0)
// in front of every use, which serialises the streams again).
template <int NPROD, int NCH>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2)))
linear_pipe_kernel(const LinArgs a) {
  static_assert(NPROD == 1 || NPROD == 3, "NPROD");
  constexpr bool LO = NPROD == 3;
  constexpr int NPL = LO ? 2 : 1;
  constexpr int STAGE = NPL * kPipePlane;            // bf16 elements of one operand's stage
  constexpr int WPIECES = NPL * kPipePlane / 8;      // 16-byte pieces of a weight chunk
  constexpr int WITER = (WPIECES + 255) / 256;
  // two distinct LDS objects: the activation stages (written with ds_write) and the weight stages (written by
  // LDS-DMA).  In ONE array the compiler has to assume that a ds_write may alias an LDS-DMA in flight and puts
  // vmcnt(0) in front of every staging write.
  __shared__ __attribute__((aligned(16))) uint16_t pipe_a[2 * STAGE];
  extern __shared__ __attribute__((aligned(16))) uint16_t pipe_w[];      // [2][STAGE]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int xcd = blockIdx.x & 7, seq = blockIdx.x >> 3;
  const int mt = (seq / a.nblk_n) * 8 + xcd;
  const int nt = seq % a.nblk_n;
  if (mt >= a.nblk_m) return;
  const long m0 = static_cast<long>(mt) * 128;
  const int n0 = nt * 128;
  constexpr int nch = NCH;

  // staging assignment: rows srow and srow + 64, 8 consecutive k at skq
  const int srow = tid >> 2, skq = (tid & 3) * 8;
  const float *r0[2], *r1[2];
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    long m = m0 + p * 64 + srow;
    if (m >= a.M) m = a.M - 1;                       // clamped rows are computed, never stored
    r0[p] = a.x0 + m * a.ldx0 + skq;
    r1[p] = a.K1 > 0 ? a.x1 + m * a.ldx1 + skq : nullptr;
  }
  const uint4 *wchunk0 = reinterpret_cast<const uint4 *>(a.wpack) + static_cast<long>(nt) * nch * (2 * kPipePlane / 8);

  float4 X[2][2][2];                                 // [set][pass][half]
  auto load_x = [&](int c, int set) {
    const int kc = c * 32;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const float *src = kc < a.K0 ? r0[p] + kc : r1[p] + (kc - a.K0);
      X[set][p][0] = reinterpret_cast<const float4 *>(src)[0];
      X[set][p][1] = reinterpret_cast<const float4 *>(src)[1];
    }
  };
  auto stage_x = [&](int set, int stage) {
    uint16_t *as = pipe_a + stage * STAGE;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      // the loaded values become visible to the optimiser HERE: without this the split (pure ALU) is hoisted to
      // right behind the loads, two iterations early, and the wavefront waits for them there
#pragma unroll
      for (int h = 0; h < 2; ++h)
        asm volatile("" : "+v"(X[set][p][h].x), "+v"(X[set][p][h].y), "+v"(X[set][p][h].z), "+v"(X[set][p][h].w));
      uint4 hi, lo;
      lin_split8<LO>(X[set][p][0], X[set][p][1], hi, lo);
      const int off = (p * 64 + srow) * kPipeRow + skq;
      *reinterpret_cast<uint4 *>(&as[off]) = hi;
      if (LO) *reinterpret_cast<uint4 *>(&as[kPipePlane + off]) = lo;
    }
  };
  auto dma_w = [&](int c, int stage) {
    const uint4 *src = wchunk0 + static_cast<long>(c) * (2 * kPipePlane / 8);
    uint16_t *ws = pipe_w + stage * STAGE;
#pragma unroll
    for (int i = 0; i < WITER; ++i) {
      if (WPIECES % 256 == 0 || (i * 256 + (tid & ~63)) < WPIECES) {       // wave-uniform tail guard
        uint16_t *dst = ws + (i * 256 + (tid & ~63)) * 8;
        pipe_dma16(src + i * 256 + tid, dst);
      }
    }
  };

  // fragment sets: [set][k-step][tile] for A hi / lo, W hi / lo
  lin_bf16x8 ah[2][2][2], al[2][2][2], bh[2][2][2], bl[2][2][2];
  const int frow = lane & 31, fk = (lane >> 5) * 8;
  const int a_off = (wm * 64 + frow) * kPipeRow + fk;
  const int b_off = (wn * 64 + frow) * kPipeRow + fk;
  auto read_frags = [&](int set, int stage) {
    const uint16_t *sa = pipe_a + stage * STAGE;
    const uint16_t *sw = pipe_w + stage * STAGE;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int ao = a_off + t * 32 * kPipeRow + ks * 16;
        const int bo = b_off + t * 32 * kPipeRow + ks * 16;
        ah[set][ks][t] = *reinterpret_cast<const lin_bf16x8 *>(&sa[ao]);
        bh[set][ks][t] = *reinterpret_cast<const lin_bf16x8 *>(&sw[bo]);
        if (LO) {
          al[set][ks][t] = *reinterpret_cast<const lin_bf16x8 *>(&sa[kPipePlane + ao]);
          bl[set][ks][t] = *reinterpret_cast<const lin_bf16x8 *>(&sw[kPipePlane + bo]);
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
  auto mfma_chunk = [&](int set) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {       // D[n][m]: W fragment as the A operand (float4 epilogue)
          if (LO) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh[set][ks][j], al[set][ks][i], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bl[set][ks][j], ah[set][ks][i], acc[i][j], 0, 0, 0);
          }
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh[set][ks][j], ah[set][ks][i], acc[i][j], 0, 0, 0);
        }
  };

  // prologue: chunks 0 and 1 staged, chunks 2 and 3 of the activations in flight, fragments of chunk 0 read
  static_assert(NCH >= 4, "NCH");
  dma_w(0, 0);
  dma_w(1, 1);
  __builtin_amdgcn_sched_barrier(0);
  load_x(0, 0);
  load_x(1, 1);
  stage_x(0, 0);
  load_x(2, 0);
  stage_x(1, 1);
  load_x(3, 1);
  __builtin_amdgcn_sched_barrier(0);
  __builtin_amdgcn_s_waitcnt(kWaitVm4 & ~0xF | 8);   // vmcnt(8): both weight DMAs (issued first) have landed
  __builtin_amdgcn_s_waitcnt(kWaitLgkm0);
  pipe_barrier();
  read_frags(0, 0);

  // one iteration (c at compile time)
  pipe_static_for([&](auto ctag) {
    constexpr int c = decltype(ctag)::value;
    constexpr int SET = c & 1;
    // everyone has the fragments of chunk c in registers and chunk c + 1 (stage SET ^ 1) is complete
    __builtin_amdgcn_s_waitcnt(kWaitLgkm0);
    pipe_barrier();
    // the staging ds_writes come BEFORE this iteration's LDS-DMA: the compiler has to assume that a ds_write
    // may alias an LDS-DMA in flight and would wait for it (vmcnt(0): the full DMA latency, every iteration)
    if constexpr (c + 2 < nch) stage_x(SET, SET);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (c + 2 < nch) dma_w(c + 2, SET);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (c + 1 < nch) read_frags(SET ^ 1, SET ^ 1);
    if constexpr (c + 4 < nch) load_x(c + 4, SET);
    __builtin_amdgcn_sched_barrier(0);
    mfma_chunk(SET);
    // pin the MFMAs of this chunk in this iteration (they touch registers only, so nothing else keeps the
    // scheduler from sinking them below the barrier — behind the memory operations they are meant to cover)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) asm volatile("" : "+v"(acc[i][j]));
    // weight chunk c + 2 landed: at most the 4 activation loads issued after its DMA may stay in flight
    if constexpr (c + 2 < nch) {
      if constexpr (c + 4 < nch) __builtin_amdgcn_s_waitcnt(kWaitVm4);
      else __builtin_amdgcn_s_waitcnt(kWaitVm0);
    }
  }, std::make_integer_sequence<int, NCH>{});

  // epilogue: as linear_dma.h
  const int grp = a.group_cols > 0 ? n0 / a.group_cols : 0;
  float *const yg = a.y + static_cast<long>(grp) * a.M * a.ldy;
  const int ncol0 = grp * a.group_cols;
  // bias first, for the whole tile, in place and in straight-line code: a bias load between two stores draws s_waitcnt
  // vmcnt(0) (the pointer may alias y), which also waits for the store in front of it (linear_mfma.h, round 5)
  if (a.bias) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int n = n0 + wn * 64 + j * 32 + 4 * (lane >> 5) + 8 * q;
        const float4 b4 = *reinterpret_cast<const float4 *>(a.bias + (n < a.N ? n : 0));
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          acc[i][j][4 * q] += b4.x; acc[i][j][4 * q + 1] += b4.y; acc[i][j][4 * q + 2] += b4.z; acc[i][j][4 * q + 3] += b4.w;
        }
      }
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const long m = m0 + wm * 64 + i * 32 + (lane & 31);
    float *yrow = yg + (m < a.M ? m : 0) * a.ldy - ncol0;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int nb = n0 + wn * 64 + j * 32 + 4 * (lane >> 5);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int n = nb + 8 * q;
        if (m < a.M && n < a.N) {
          float4 v = make_float4(acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]);
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
