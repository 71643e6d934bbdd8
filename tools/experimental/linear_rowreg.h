// The row-local seams of an encoder layer (linear_chain.h: MODE 0 = output projection + residual + LayerNorm + FFN +
// residual + LayerNorm; MODE 1 = output projection + residual + LayerNorm + a second projection) with the ROWS RESIDENT
// IN REGISTERS (K3cr).
//
// linear_chain.h keeps a 64-row panel in LDS and hands every stage's result to the next one through LDS planes: its
// phase clocks (tools/gemm_diag/chain_run.py) put 53 % of a workgroup's cycles outside the GEMMs — the panel's HBM
// round trip, three accumulator -> plane write-backs with their barriers, LayerNorm statistics exchanged through
// LDS.  Here a WAVEFRONT owns 32 complete rows for the whole chain and nothing of a row ever leaves its registers:
//
//   * The MFMA is run transposed (the weight fragment is the A operand, the rows are the B operand), so
//     lane (r = lane & 31, h = lane >> 5) ends up with output features 32 t + 8 g + 4 h + e (g, e = 0..3) of ROW r in
//     accumulator register 4 g + e of tile t.  The B operand of k16-step s wants 8 k values of row r in the same
//     lane; the order of k inside a dot product is free as long as both operands agree, so k slot (s, h, j) is
//     DEFINED as feature 16 s + 8 (j >> 2) + 4 h + (j & 3): registers 8 u .. 8 u + 7 of tile t are, as they lie, the
//     fragment of step 2 t + u.  lin_rowreg_pack_weight_kernel writes the weights in that k order.  An accumulator
//     tile becomes the next GEMM's operand by a bf16 split in registers — no LDS, no barrier, no transposition.
//   * LayerNorm is wave-local: a row's 256 features live in two lanes (r, 0) and (r, 1): one cross-lane add.
//   * Only the WEIGHTS travel through LDS: 32 KiB chunks (16 fragment units of [1 KiB hi | 1 KiB lo]) by LDS-DMA into
//     a 4-slot ring, three chunks ahead, ONE barrier per chunk; the four wavefronts of the workgroup (one per SIMD,
//     512 registers each) read every chunk once: a quarter of linear_chain.h's weight traffic from L2 per row.
//       GEMM 0 (256 -> 256):            chunk = one 32-feature output tile, 16 k16 steps           (8 chunks)
//       FFN, per 32 hidden features:    chunk A = that hidden tile of W1 (16 steps over x), then
//                                       chunk B = W2[all 8 output tiles, the 2 steps of this hidden tile]   (32 chunks)
//     so the hidden layer exists only as one 32 x 32 tile per wavefront (16 registers of fragments), and the FFN
//     accumulators start from x + b2 (the residual is never reloaded).
//       MODE 1, second projection:      chunk = one 32-feature output tile, bias, store
//   * In the main loop a wavefront issues no vector-memory instruction but the LDS-DMAs (MODE 1: + its output stores),
//     so the in-order vmcnt counter is a plain chunk counter (waits count only the DMAs a wavefront certainly issued:
//     anything else in flight only makes a wait stricter).
//
// Registers at the FFN peak: x fragments 128 + FFN accumulators 128 + hidden fragments 16 + hidden accumulator 16 +
// weight fragments: one wavefront per SIMD (amdgpu_waves_per_eu(1, 1)).  128 rows per workgroup; the launcher gives this
// kernel whole rounds of workgroups and linear_chain.h the remainder.
//
// RETIRED (round 3): parity-green, at par with linear_chain.h per row — at one wavefront per SIMD the wavefront's vector /
// scalar instructions are not hidden behind its own MFMAs (profiles/r3/r3l_register_resident_chain_notes.txt).  Not built
// into the library; tools/gemm_diag/chain_run.py 3 <rows> still runs it.
#pragma once
#include <utility>

#include "linear_chain.h"

namespace bevmsda {

constexpr int kRowRegRows = 128;          // rows per workgroup: 4 wavefronts x 32
constexpr int kRowRegStages = 4;          // weight ring slots

// s_waitcnt vmcnt(n), n a multiple of 4 up to 16 (wave-uniform, known at run time)
__device__ __forceinline__ void rowreg_wait_vm(int n) {
  switch (n) {
    case 16: __builtin_amdgcn_s_waitcnt(0x0F70 | (16 & 15) | ((16 >> 4) << 14)); break;
    case 8: __builtin_amdgcn_s_waitcnt(0x0F70 | 8); break;
    case 4: __builtin_amdgcn_s_waitcnt(0x0F70 | 4); break;
    default: __builtin_amdgcn_s_waitcnt(0x0F70); break;
  }
}

// LDS-DMA by hand.  hipcc cannot tell a ds_read of the ring from one that aliases an LDS-DMA in flight (there always is
// one: the prefetch) and puts s_waitcnt vmcnt(0) in front of the first read of every chunk — the whole prefetch
// distance gone.  (Issuing the READS from inline assembly instead does not work: the compiler takes an asm output for
// available at once and moves it to another register before the data has arrived.)  A DMA issued from inline
// assembly is invisible to that pass; the kernel's own chunk counter (rowreg_wait_vm + barrier) orders ring reads
// behind the DMA that feeds them, and the "memory" clobbers keep the compiler from moving LDS reads across either.
// Waits the compiler computes for ITS vector-memory instructions stay correct: unknown newer operations in flight only
// make an in-order vmcnt wait stricter.  src: per-lane global address; lds_dst: wave-uniform LDS byte address (lane l's
// 16 bytes land at lds_dst + 16 l).
__device__ __forceinline__ void rowreg_dma16(const void *src, unsigned lds_dst) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(src), "s"(lds_dst) : "memory", "m0");
}

template <class F, int... Is>
__device__ __forceinline__ void rowreg_static_for_impl(F &&f, std::integer_sequence<int, Is...>) {
  (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void rowreg_static_for(F &&f) {
  rowreg_static_for_impl(static_cast<F &&>(f), std::make_integer_sequence<int, N>{});
}

// Weight image of linear_rowreg_chain_kernel: unit (T, s) = [64 lanes x 8 bf16 hi | the same lo], lane (m, h) holds
// w[32 T + m][16 s + 8 (j >> 2) + 4 h + (j & 3)], j = 0..7.  Unit order: tile-major (T * K/16 + s) or, kmajor, by pairs
// of k16 steps ((s >> 1) * (2 N/32) + 2 T + (s & 1)): 16 consecutive units = one 32 KiB chunk either way (K = 256
// tile-major; N = 256 k-major).  One thread per (unit, lane).
__global__ void __launch_bounds__(256) lin_rowreg_pack_weight_kernel(const float *__restrict__ w, long ldw, int N, int K,
                                                                    int kmajor, uint16_t *__restrict__ blob) {
  const int nstep = K / 16, ntile = N / 32;
  const long t = static_cast<long>(blockIdx.x) * 256 + threadIdx.x;
  if (t >= static_cast<long>(ntile) * nstep * 64) return;
  const int lane = static_cast<int>(t & 63);
  const int s = static_cast<int>((t >> 6) % nstep);
  const int T = static_cast<int>((t >> 6) / nstep);
  const int m = lane & 31, h = lane >> 5;
  const float *src = w + static_cast<long>(32 * T + m) * ldw + 16 * s + 4 * h;
  uint4 hi, lo;
  lin_split8<true>(*reinterpret_cast<const float4 *>(src), *reinterpret_cast<const float4 *>(src + 8), hi, lo);
  const long unit = kmajor ? static_cast<long>(s >> 1) * (2 * ntile) + 2 * T + (s & 1) : static_cast<long>(T) * nstep + s;
  uint4 *dst = reinterpret_cast<uint4 *>(blob) + unit * 2 * 64 + lane;
  dst[0] = hi;
  dst[64] = lo;
}

template <int NPROD, int PRE, int MODE>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
linear_rowreg_chain_kernel(const ChainArgs a) {
  static_assert(NPROD == 1 || NPROD == 3, "NPROD");
  static_assert(PRE == 0 || PRE == 2, "PRE: 0 plain rows, 2 two-row gather");
  constexpr bool LO = NPROD == 3;
  constexpr int NPL = LO ? 2 : 1;
  constexpr int UB = NPL * 1024;               // a fragment unit in LDS
  constexpr int CHB = 16 * UB;                 // a chunk in LDS
  constexpr int NST = kRowRegStages;
  constexpr int ND = 4 * NPL;                  // LDS-DMA instructions per wavefront and chunk
  constexpr int NCST = 4 * kChainC + kChainMaxN2 + 2 * kChainC;       // gamma0, beta0, gamma1, beta1 | b1 | b0, b2
  constexpr int RINGB = NST * CHB > 4 * 32768 ? NST * CHB : 4 * 32768;       // (also the row staging area: 32 KiB per wavefront)
  __shared__ __attribute__((aligned(16))) unsigned char ring[RINGB];
  __shared__ __attribute__((aligned(16))) float cst[NCST];
  float *const c_g0 = cst, *const c_be0 = cst + 256, *const c_g1 = cst + 512, *const c_be1 = cst + 768;
  float *const c_b1 = cst + 1024, *const c_b0 = cst + 1024 + kChainMaxN2, *const c_b2 = c_b0 + 256;
  const int nb1 = MODE == 1 ? a.N2 : kChainF;
  const int nchunk = 8 + (MODE == 1 ? a.N2 / 32 : 2 * (kChainF / 32));

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 31, fh = lane >> 5;
  const long m0 = static_cast<long>(blockIdx.x) * kRowRegRows + wave * 32;
  const long mrow = m0 + fr;
  const bool mok = mrow < a.M;
  const long mclamp = mok ? mrow : a.M - 1;    // clamped rows are computed and never stored
#ifdef BEVMSDA_CHAIN_PROF
  // tools/gemm_diag/chain_run.py: the kernel cut short after phase `stop` (launch times of the prefixes; clock stamps
  // with their atomics would sit in the in-order vmcnt queue of the chunk counter and distort what they measure)
  const long stop = MODE == 0 ? a.ld_y2 : 0;
#define ROWREG_STOP(k) do { if (stop == (k)) return; } while (0)
#else
#define ROWREG_STOP(k) do {} while (0)
#endif

  // chunk c of the walk -> its 32 KiB in the weight images
  auto chunk_src = [&](int c) __attribute__((always_inline)) -> const unsigned char * {
    if (c < 8) return reinterpret_cast<const unsigned char *>(a.w0) + static_cast<long>(c) * 32768;
    const int cc = c - 8;
    if (MODE == 1) return reinterpret_cast<const unsigned char *>(a.w1) + static_cast<long>(cc) * 32768;
    // FFN walk: A0, (A1, B0), (A2, B1), ... (A15, B14), B15   (A = hidden tile of W1, B = its two k16 steps of W2)
    if (cc == 0) return reinterpret_cast<const unsigned char *>(a.w1);
    if (cc == 31) return reinterpret_cast<const unsigned char *>(a.w2) + 15L * 32768;
    return (cc & 1) ? reinterpret_cast<const unsigned char *>(a.w1) + static_cast<long>((cc + 1) >> 1) * 32768
                    : reinterpret_cast<const unsigned char *>(a.w2) + static_cast<long>((cc - 2) >> 1) * 32768;
  };
  const unsigned ring_base = static_cast<unsigned>(reinterpret_cast<size_t>((__attribute__((address_space(3))) unsigned char *)ring));
  auto dma = [&](int c) __attribute__((always_inline)) {
    const unsigned char *src = chunk_src(c) + lane * 16;
    const unsigned dst = __builtin_amdgcn_readfirstlane(ring_base + (c & (NST - 1)) * CHB + 4 * wave * UB);
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int pl = 0; pl < NPL; ++pl)
        rowreg_dma16(src + ((4 * wave + u) * 2 + pl) * 1024, dst + (u * NPL + pl) * 1024);
  };
  // chunk c has landed in every wavefront's share and slot (c - 1) % NST is free ...
  auto chunk_wait = [&](int c) __attribute__((always_inline)) -> const unsigned char * {
    const int left = nchunk - 1 - c;
    rowreg_wait_vm(ND * (left < NST - 2 ? left : NST - 2));
    asm volatile("s_barrier" ::: "memory");
    return ring + (c & (NST - 1)) * CHB + lane * 16;
  };
  // ... refill it with chunk c + NST - 1 (issued behind the chunk's first fragment reads: their LDS latency runs under
  // the address arithmetic of the DMA instead of in front of the first MFMA)
  auto chunk_refill = [&](int c) __attribute__((always_inline)) {
    if (c + NST - 1 < nchunk) dma(c + NST - 1);
  };

  {
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int t4 = tid * 4; t4 < NCST; t4 += 256 * 4) {
      const float *src = nullptr;
      if (t4 < 256) src = a.gamma0 + t4;
      else if (t4 < 512) src = a.beta0 + (t4 - 256);
      else if (t4 < 768) src = MODE == 0 ? a.gamma1 + (t4 - 512) : nullptr;
      else if (t4 < 1024) src = MODE == 0 ? a.beta1 + (t4 - 768) : nullptr;
      else if (t4 < 1024 + kChainMaxN2) src = (a.b1 && t4 - 1024 < nb1) ? a.b1 + (t4 - 1024) : nullptr;
      else if (t4 < 1024 + kChainMaxN2 + 256) src = a.b0 ? a.b0 + (t4 - 1024 - kChainMaxN2) : nullptr;
      else src = (MODE == 0 && a.b2) ? a.b2 + (t4 - 1024 - kChainMaxN2 - 256) : nullptr;
      *reinterpret_cast<float4 *>(cst + t4) = src ? *reinterpret_cast<const float4 *>(src) : z4;
    }
  }
  // ------------------------------------------------------------------ my 32 rows: operand fragments + residual
  // A lane needs 16-byte pieces of ITS row: loaded directly, every instruction touches 32 rows (32 cache lines for
  // 1 KiB: the L1 path serves that at a fraction of its rate — 14 us of prologue per workgroup).  Instead every row
  // travels as ONE coalesced LDS-DMA instruction (64 lanes x 16 bytes = the row) into the still-empty ring (32 KiB per
  // wavefront) and is read back in fragment layout.  Row i is stored rotated by i pieces (lane l fetches piece
  // (l - i) mod 64), so the 32 rows a ds_read_b128 touches sit in different banks.
  lin_bf16x8 xh[16], xl[LO ? 16 : 1];
  lin_f32x16 v[8];                             // residual, then the pre-LayerNorm sum, x, and the FFN accumulators
  {
    int g0 = static_cast<int>(mclamp), g1 = -1;
    float gs = 1.f;
    if (PRE == 2) {
      g0 = a.gidx[mclamp * 2];
      g1 = a.gidx[mclamp * 2 + 1];
      gs = a.gscale[mclamp];
    }
    const unsigned stage_lds = __builtin_amdgcn_readfirstlane(ring_base + wave * 32768);
    const unsigned char *stage = ring + wave * 32768 + fr * 1024;
    auto stage_rows = [&](int rowidx) __attribute__((always_inline)) {      // rowidx: the source row of lane (r, .)'s row
#pragma unroll
      for (int i = 0; i < 32; ++i) {
        const int ri = __builtin_amdgcn_readlane(rowidx, i);
        rowreg_dma16(a.rows + static_cast<long>(ri) * a.ld_rows + ((lane - i) & 63) * 4, stage_lds + i * 1024);
      }
    };
    auto stage_read = [&](float4 (&q)[32]) __attribute__((always_inline)) {
#pragma unroll
      for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g)
          q[4 * t + g] = *reinterpret_cast<const float4 *>(stage + ((8 * t + 2 * g + fh + fr) & 63) * 16);
    };
    stage_rows(PRE == 2 ? (g0 < 0 ? 0 : g0) : g0);
    if (a.res) {                               // (one branch around all the loads, not one per load; needed after GEMM 0)
      const float *pr = a.res + mclamp * a.ld_res + 4 * fh;
#pragma unroll
      for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const float4 r4 = *reinterpret_cast<const float4 *>(pr + 32 * t + 8 * g);
          v[t][4 * g] = r4.x; v[t][4 * g + 1] = r4.y; v[t][4 * g + 2] = r4.z; v[t][4 * g + 3] = r4.w;
        }
    } else {
#pragma unroll
      for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) v[t][r] = 0.f;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float4 q[32];
    stage_read(q);
    if (PRE == 2) {
      if (__builtin_amdgcn_ballot_w64(g1 >= 0) != 0) {        // (wave-uniform: most rows are seen by one camera)
        stage_rows(g1 < 0 ? 0 : g1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        float4 q1[32];
        stage_read(q1);
#pragma unroll
        for (int j = 0; j < 32; ++j) q[j] = panel_gsum(q[j], g0 >= 0, q1[j], g1 >= 0, gs);
      } else {
        const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int j = 0; j < 32; ++j) q[j] = panel_gsum(q[j], g0 >= 0, z4, false, gs);
      }
    }
    __syncthreads();                           // every wavefront is done with its staging area; constants written
#pragma unroll 1
    for (int c = 0; c < NST - 1; ++c) dma(c);  // (the weights' L2 round trip runs under the split below)
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        uint4 hi, lo;
        lin_split8<LO>(q[4 * t + 2 * u], q[4 * t + 2 * u + 1], hi, lo);
        xh[2 * t + u] = __builtin_bit_cast(lin_bf16x8, hi);
        if (LO) xl[LO ? 2 * t + u : 0] = __builtin_bit_cast(lin_bf16x8, lo);
      }
  }

  // The 16 fragment units of a chunk, two at a time: `fn(p, a_hi, a_lo, b_hi, b_lo)` issues the MFMAs of units
  // ua(p) and ub(p) INTERLEAVED on two different accumulators — a 32x32x16 MFMA that depends on its predecessor's
  // accumulator issues at half rate (measured: one accumulator chain per chunk ran at 1.5 us per chunk against 0.64 us
  // of MFMA issue time).  Fragments one pair ahead of the MFMAs.
  auto walk_pairs = [&](const unsigned char *slot, int c, auto fenced, auto &&ua, auto &&ub, auto &&fn) __attribute__((always_inline)) {
    constexpr bool FENCED = decltype(fenced)::value;
    constexpr int PD = 2;                      // fragment pairs in flight ahead of the MFMAs
    lin_bf16x8 wf[PD + 1][2][NPL];
    auto wload = [&](int set, int qa, int qb) __attribute__((always_inline)) {
#pragma unroll
      for (int pl = 0; pl < NPL; ++pl) {
        wf[set][0][pl] = *reinterpret_cast<const lin_bf16x8 *>(slot + qa * UB + pl * 1024);
        wf[set][1][pl] = *reinterpret_cast<const lin_bf16x8 *>(slot + qb * UB + pl * 1024);
      }
    };
#pragma unroll
    for (int p = 0; p < PD; ++p) wload(p, ua(p), ub(p));
    chunk_refill(c);
    rowreg_static_for<8>([&](auto ptag) __attribute__((always_inline)) {
      constexpr int p = decltype(ptag)::value;
      if constexpr (p + PD < 8) wload((p + PD) % (PD + 1), ua(p + PD), ub(p + PD));
      if constexpr (FENCED) __builtin_amdgcn_sched_barrier(0);
      fn(ptag, wf[p % (PD + 1)][0][0], wf[p % (PD + 1)][0][LO ? 1 : 0], wf[p % (PD + 1)][1][0], wf[p % (PD + 1)][1][LO ? 1 : 0]);
      if constexpr (FENCED) __builtin_amdgcn_sched_barrier(0);
    });
  };
  // acc += W[tile of this chunk, 16 k16 steps] x^T   (units s = 0 .. 15 of the chunk; even / odd steps on two accumulators)
  auto gemm_tile = [&](lin_f32x16 &acc, const unsigned char *slot, int c, auto fenced) __attribute__((always_inline)) {
    lin_f32x16 odd;
#pragma unroll
    for (int r = 0; r < 16; ++r) odd[r] = 0.f;
    walk_pairs(slot, c, fenced, [](int p) { return 2 * p; }, [](int p) { return 2 * p + 1; },
               [&](auto ptag, lin_bf16x8 ah, lin_bf16x8 al, lin_bf16x8 bh, lin_bf16x8 bl) __attribute__((always_inline)) {
      constexpr int s = 2 * decltype(ptag)::value;
      if constexpr (LO) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, xl[LO ? s : 0], acc, 0, 0, 0);
        odd = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh, xl[LO ? s + 1 : 0], odd, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, xh[s], acc, 0, 0, 0);
        odd = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bl, xh[s + 1], odd, 0, 0, 0);
      }
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, xh[s], acc, 0, 0, 0);
      odd = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh, xh[s + 1], odd, 0, 0, 0);
    });
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] += odd[r];
  };
  // t[.] += vec[32 T + 8 g + 4 h + e] (per-feature constants from LDS)
  auto add_feat = [&](lin_f32x16 &t, const float *vec, int T) __attribute__((always_inline)) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float4 c4 = *reinterpret_cast<const float4 *>(vec + 32 * T + 8 * g + 4 * fh);
      t[4 * g] += c4.x; t[4 * g + 1] += c4.y; t[4 * g + 2] += c4.z; t[4 * g + 3] += c4.w;
    }
  };
  // LayerNorm over the 256 features of my rows, in place (torch.nn.LayerNorm: biased variance, eps inside the root)
  auto layernorm = [&](const float *gamma, const float *beta, float eps) __attribute__((always_inline)) {
    float sum = 0.f;
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) sum += v[t][r];
    sum += __shfl_xor(sum, 32, 64);
    const float mean = sum * (1.0f / kChainC);
    float ss = 0.f;
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float d = v[t][r] - mean;
        ss = fmaf(d, d, ss);
      }
    ss += __shfl_xor(ss, 32, 64);
    const float rstd = rsqrtf(ss * (1.0f / kChainC) + eps);
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 ga = *reinterpret_cast<const float4 *>(gamma + 32 * t + 8 * g + 4 * fh);
        const float4 be = *reinterpret_cast<const float4 *>(beta + 32 * t + 8 * g + 4 * fh);
        v[t][4 * g] = (v[t][4 * g] - mean) * rstd * ga.x + be.x;
        v[t][4 * g + 1] = (v[t][4 * g + 1] - mean) * rstd * ga.y + be.y;
        v[t][4 * g + 2] = (v[t][4 * g + 2] - mean) * rstd * ga.z + be.z;
        v[t][4 * g + 3] = (v[t][4 * g + 3] - mean) * rstd * ga.w + be.w;
      }
  };
  auto store_rows = [&](float *out, long ld) __attribute__((always_inline)) {
    if (!mok) return;
    float *yrow = out + mrow * ld + 4 * fh;
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        *reinterpret_cast<float4 *>(yrow + 32 * t + 8 * g) = make_float4(v[t][4 * g], v[t][4 * g + 1], v[t][4 * g + 2], v[t][4 * g + 3]);
  };

  ROWREG_STOP(1);                              // constants, row loads, split
  // ------------------------------------------------------------------ x = LN0(A W0^T + b0 + res)
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    const unsigned char *slot = chunk_wait(t);
    lin_f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    gemm_tile(acc, slot, t, std::true_type{});
    add_feat(acc, c_b0, t);
#pragma unroll
    for (int r = 0; r < 16; ++r) v[t][r] += acc[r];
  }
  ROWREG_STOP(2);                              // GEMM 0 (8 chunks)
  layernorm(c_g0, c_be0, a.eps0);
  if constexpr (MODE == 1) store_rows(a.y, a.ld_y);          // x is the next attention's residual
  // x as operand fragments (registers 8 u .. 8 u + 7 of tile t = step 2 t + u)
#pragma unroll
  for (int t = 0; t < 8; ++t)
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      uint4 hi, lo;
      lin_split8<LO>(make_float4(v[t][8 * u], v[t][8 * u + 1], v[t][8 * u + 2], v[t][8 * u + 3]),
                     make_float4(v[t][8 * u + 4], v[t][8 * u + 5], v[t][8 * u + 6], v[t][8 * u + 7]), hi, lo);
      xh[2 * t + u] = __builtin_bit_cast(lin_bf16x8, hi);
      if (LO) xl[LO ? 2 * t + u : 0] = __builtin_bit_cast(lin_bf16x8, lo);
    }

  ROWREG_STOP(3);                              // LayerNorm 0 + split
  if constexpr (MODE == 1) {
    // ---------------------------------------------------------------- p = x W1^T + b1, one 32-feature tile per chunk
    const int ntile = a.N2 / 32;
#pragma unroll 1
    for (int T = 0; T < ntile; ++T) {
      const unsigned char *slot = chunk_wait(8 + T);
      lin_f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      gemm_tile(acc, slot, 8 + T, std::true_type{});
      add_feat(acc, c_b1, T);
      if (mok) {
        float *yrow = a.y2 + mrow * a.ld_y2 + 32 * T + 4 * fh;
#pragma unroll
        for (int g = 0; g < 4; ++g)
          *reinterpret_cast<float4 *>(yrow + 8 * g) = make_float4(acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]);
      }
    }
  } else {
    // ---------------------------------------------------------------- y = LN1(x + relu(x W1^T + b1) W2^T + b2)
#pragma unroll
    for (int t = 0; t < 8; ++t) add_feat(v[t], c_b2, t);     // the FFN accumulators start from x + b2
    // Software pipeline over the hidden tiles: while the matrix cores run tile h + 1's GEMM, the vector ALU turns
    // tile h's accumulator into operand fragments (bias, ReLU, bf16 split: ~250 instructions that would otherwise sit
    // between two MFMA phases — at one wavefront per SIMD nothing else fills those issue slots).
    lin_f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    gemm_tile(acc, chunk_wait(8), 8, std::true_type{});
    auto ffn_step = [&](int hT, auto last) __attribute__((always_inline)) {
      constexpr bool LAST = decltype(last)::value;             // (peeled: the pipelined region must be ONE basic block)
      lin_f32x16 accn;
#pragma unroll
      for (int r = 0; r < 16; ++r) accn[r] = 0.f;
      const unsigned char *slot = nullptr;
      if constexpr (!LAST) slot = chunk_wait(9 + 2 * hT);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (!LAST) gemm_tile(accn, slot, 9 + 2 * hT, std::false_type{});
      add_feat(acc, c_b1, hT);
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = acc[r] < 0.f ? 0.f : acc[r];     // NaN stays NaN, as torch.relu
      lin_bf16x8 hh[2], hl[LO ? 2 : 1];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        uint4 hi, lo;
        lin_split8<LO>(make_float4(acc[8 * u], acc[8 * u + 1], acc[8 * u + 2], acc[8 * u + 3]),
                       make_float4(acc[8 * u + 4], acc[8 * u + 5], acc[8 * u + 6], acc[8 * u + 7]), hi, lo);
        hh[u] = __builtin_bit_cast(lin_bf16x8, hi);
        if (LO) hl[LO ? u : 0] = __builtin_bit_cast(lin_bf16x8, lo);
      }
      // (pins the fragments HERE: without a use in front of the next chunk's barrier the compiler sinks their
      // computation below it, right in front of the MFMAs that consume them)
      asm volatile("" : "+v"(hh[0]), "+v"(hh[1]), "+v"(hl[0]), "+v"(hl[LO ? 1 : 0]));
      // the interleave asked of the scheduler for this region: per fragment pair its LDS reads, then every MFMA
      // followed by a few vector instructions
#pragma unroll
      for (int p = 0; p < 8; ++p) {
        __builtin_amdgcn_sched_group_barrier(0x100, 2 * NPL, 0);
#pragma unroll
        for (int q = 0; q < 2 * NPROD; ++q) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      const int cb = LAST ? 39 : 10 + 2 * hT;
      slot = chunk_wait(cb);
      // v[T] += W2[tile T, the two k16 steps of this hidden tile] h^T
      // (pair p: output tiles 2 (p >> 1) and 2 (p >> 1) + 1, k16 step u = p & 1 of this hidden tile)
      walk_pairs(slot, cb, std::true_type{}, [](int p) { return 4 * (p >> 1) + (p & 1); }, [](int p) { return 4 * (p >> 1) + 2 + (p & 1); },
                 [&](auto ptag, lin_bf16x8 ah, lin_bf16x8 al, lin_bf16x8 bh, lin_bf16x8 bl) __attribute__((always_inline)) {
        constexpr int p = decltype(ptag)::value, T = 2 * (p >> 1), u = p & 1;
        if constexpr (LO) {
          v[T] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, hl[LO ? u : 0], v[T], 0, 0, 0);
          v[T + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh, hl[LO ? u : 0], v[T + 1], 0, 0, 0);
          v[T] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, hh[u], v[T], 0, 0, 0);
          v[T + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bl, hh[u], v[T + 1], 0, 0, 0);
        }
        v[T] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, hh[u], v[T], 0, 0, 0);
        v[T + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh, hh[u], v[T + 1], 0, 0, 0);
      });
      acc = accn;
    };
#pragma unroll 1
    for (int hT = 0; hT + 1 < kChainF / 32; ++hT) ffn_step(hT, std::false_type{});
    ffn_step(kChainF / 32 - 1, std::true_type{});
    ROWREG_STOP(4);                            // FFN (32 chunks)
    layernorm(c_g1, c_be1, a.eps1);
    ROWREG_STOP(5);                            // LayerNorm 1
    store_rows(a.y, a.ld_y);
  }
}

}  // namespace bevmsda
