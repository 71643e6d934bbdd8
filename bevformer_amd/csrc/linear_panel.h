// Dense projections, row-panel form (K3r): y = act([x0 (+ a0) | x1 (+ a1)] W^T + b) [-> + res -> LayerNorm].
//
// Same contract and arithmetic as linear_splitbf16_kernel (linear_mfma.h): fp32 operands split into two bf16
// terms, three v_mfma_f32_32x32x16_bf16 per product (NPROD = 3) or one over rounded operands (NPROD = 1),
// fp32 accumulation — the Linear layers of BEVFormerLayer (temporal_self_attention.py:198,206-211,267;
// spatial_cross_attention.py:173,334-348; mmcv FFN) whose K is 256 or 512.  What differs is the schedule,
// built on what tools/gemm_diag measured on the 128 x 128 x 32 kernels (their per-chunk phases — stage,
// barrier, fragment reads, MFMAs, barrier — add up instead of overlapping, and the epilogue stores of a launch
// march in phase):
//
//   * a workgroup owns a PANEL of BM = 32 MT complete rows and sweeps all N columns.  The panel (BM x 256 fp32
//     = BM KB) is fetched ONCE, whole, by LDS-DMA (every row's eight 128-byte lines in flight at the same time:
//     the memory-level parallelism a K loop of 8 dependent chunk loads never has), split ONCE into [hi | lo]
//     bf16 planes in place by the lanes that fetched it, and followed by ONE barrier;
//   * after that barrier the wavefronts never synchronise again: each walks its own column tiles (32 NT
//     columns), reading activation fragments from the LDS planes (conflict-free ds_read_b128) and WEIGHT
//     fragments straight from L2 into registers — the weight image is stored in MFMA fragment order, so a
//     fragment is one fully coalesced 1 KiB buffer load; no LDS staging, no ds_write, no barrier for W;
//   * wavefronts therefore drift out of phase on their own and one wavefront's epilogue stores overlap the
//     other wavefronts' MFMAs on the same SIMD; the first weight fragments of the next column tile are
//     requested BEFORE the stores (the vector-memory counter retires in order);
//   * K = 512 (two sources, or FFN fc2) runs as two panel passes over the same accumulators;
//   * a workgroup holds complete rows, so the residual add + LayerNorm that follow output_proj / fc2
//     (encoder.py:376-404) are an epilogue (LN) instead of a second launch over the grid.
//
// K order: an MFMA only needs A and B to agree on which k sits in which operand slot.  DMA wants full
// 128-byte lines per 8 lanes and the split wants 8 values per lane, so a lane's two 16-byte DMA slots hold
// k = 32 (2p) + 4c .. +3 and 32 (2p + 1) + 4c .. +3 of its row (line pair p, 16-byte column c); these 8 values
// are "group" j = 8 p + c, and k16-step s of the MFMA consumes groups 2s (lanes 0-31) and 2s + 1 (lanes
// 32-63).  lin_panel_pack_weight_kernel writes W in exactly that order.  Results therefore agree with the
// first kernel to fp32 summation order, not bit for bit.
//
// LDS image (one __shared__ array): pair (q, p) = 2 KiB at ((q * 4 + p) * 2048): [1 KiB hi slots | 1 KiB lo
// slots] (during the DMA: the fp32 granules of line 2p | line 2p + 1); q = row block (row bits 2, 4, 5, 6), the 8
// rows of a block are row bits 0, 1, 3; slot of (row, c) inside a 1 KiB half = rl * 8 + (c ^ (row & 7)) with
// rl = (row & 3) * 2 + bit 3 of row.  With that map the 16 lanes of every ds_read_b128 service group
// ({0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, + 32) hit 16 distinct 16-byte slots of the 256-byte bank row
// (tests/test_linear_layout_model.py replays the arithmetic).
#pragma once
#include "linear_mfma.h"

namespace bevmsda {

struct PanelArgs {
  const float *x0, *a0, *x1, *a1;   // A sources (a* optional addends); x1 for k >= K0
  long ldx0, lda0, ldx1, lda1;
  const int32_t *gidx;              // gather mode: A[m, :] = gscale[m] * (x0[gidx[m, 0]] + x0[gidx[m, 1]]), -1 = absent
  const float *gscale;
  const uint16_t *wp;               // fragment-order weight image (lin_panel_pack_weight_kernel)
  unsigned wp_bytes;
  const float *bias;
  float *y;
  long ldy;
  long M;
  int N, K0, K1;
  int relu, group_cols, out_bf16;
  const float *res;                 // LN: residual (M, ldres) or nullptr
  long ldres;
  const float *gamma, *beta;
  float eps;
  // row segments (e.g. one per camera): a workgroup whose rows all lie in segments with no entry — seg_start[s + 1] ==
  // seg_start[s], read from DEVICE memory when the kernel runs — computes and stores nothing
  // The consumer (msda_d32.h) issues its zero-coefficient taps too: up to one image row + 1 pixel outside a level, so
  // rows within max_l W_l + 1 of a used segment are computed as well (level_shapes: (levels, 2) int64 [H, W], DEVICE).
  const int32_t *seg_start;         // (ceil(M / seg_len) + 1) or nullptr
  long seg_len;
  const int64_t *level_shapes;
  int num_levels;
  // two row blocks (PRE = 0, K1 = 0): rows [m_split, M) of A come from xb (same row stride as x0), row m - m_split —
  // TSA's value [history BEV ; current queries] read from its two tensors instead of from a stacked copy
  const float *xb;
  long m_split;
  // phase skew of the column sweep, in units of 1024 clocks (0 = none): the second wavefront of every SIMD (8-wavefront
  // shape) / every second workgroup of a CU (4-wavefront shape, by the observed block placement) sleeps this long before
  // its first column tile, so that one half of a CU's wavefronts stores while the other half issues MFMAs — left alone all
  // wavefronts of a workgroup (and all workgroups of the launch) finish their tiles together and the chip alternates
  // between an MFMA phase and a store phase
  int skew;
#ifdef BEVMSDA_PANEL_DIAG
  int diag;                         // tools/gemm_diag only: bit 0 no MFMA, 1 no stores, 2 weight fragments of step 0 only,
                                    //   3 activation fragments of step 0 only, 4 no panel fetch / split
  unsigned long long *prof;         // tools/gemm_diag only: per-phase shader clocks summed over the wavefronts (PANEL_CLK)
#endif
};
#ifdef BEVMSDA_PANEL_DIAG
#define PANEL_DIAG(a, bit) (((a).diag >> (bit)) & 1)
// phase clocks: [0] panel DMA issue (+ address set-up), [1] wait for the DMA, [2] split + barrier, [3] k loops of the column
// tiles, [4] tile epilogues (bias, stores issued), [5] everything else, [7] wavefronts counted
// (sampled: wavefront 0 and wavefront NW - 1 of every 64th workgroup — stamps on every wavefront serialise on the counters)
#define PANEL_CLK(k)                                                                        \
  do {                                                                                      \
    if (stamping_) {                                                                        \
      const unsigned long long now_ = __builtin_amdgcn_s_memtime();                         \
      if (lane == 0) atomicAdd(a.prof + (k), now_ - t_prev_);                               \
      t_prev_ = now_;                                                                       \
    }                                                                                       \
  } while (0)
#else
#define PANEL_DIAG(a, bit) 0
#define PANEL_CLK(k) do { } while (0)
#endif

constexpr int kPanelK = 256;        // k per panel pass (8 lines of 128 bytes per row)

__device__ __forceinline__ int panel_row_of(int q, int rl) {
  return ((rl >> 1) & 3) | ((rl & 1) << 3) | ((q & 1) << 2) | ((q >> 1) << 4);
}

__device__ __forceinline__ float4 panel_gsum(const float4 &r0, bool h0, const float4 &r1, bool h1, float sc) {
  return make_float4(((h0 ? r0.x : 0.f) + (h1 ? r1.x : 0.f)) * sc, ((h0 ? r0.y : 0.f) + (h1 ? r1.y : 0.f)) * sc,
                     ((h0 ? r0.z : 0.f) + (h1 ? r1.z : 0.f)) * sc, ((h0 ? r0.w : 0.f) + (h1 ? r1.w : 0.f)) * sc);
}

typedef unsigned panel_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned panel_u32x2 __attribute__((ext_vector_type(2)));
// 16- / 8-byte global store with cache policy bits (the raw-buffer builtin carries them; a null-offset descriptor over the
// whole address space would need 64-bit offsets, so the descriptor is rebuilt on the element's own address: base = p, offset 0)
template <int AUX>
__device__ __forceinline__ void panel_store(float *p, const float4 &v) {
  if constexpr (AUX == 0) {
    *reinterpret_cast<float4 *>(p) = v;
  } else {
    const panel_u32x4 d = {__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)};
    if constexpr (AUX == 2) asm volatile("global_store_dwordx4 %0, %1, off nt\n\ts_nop 1" ::"v"(p), "v"(d) : "memory");
    else if constexpr (AUX == 16) asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(d) : "memory");
    else asm volatile("global_store_dwordx4 %0, %1, off sc1 nt\n\ts_nop 1" ::"v"(p), "v"(d) : "memory");
  }
}
template <int AUX>
__device__ __forceinline__ void panel_store(uint16_t *p, const uint2 &v) {
  if constexpr (AUX == 0) {
    *reinterpret_cast<uint2 *>(p) = v;
  } else {
    const panel_u32x2 d = {v.x, v.y};
    if constexpr (AUX == 2) asm volatile("global_store_dwordx2 %0, %1, off nt\n\ts_nop 1" ::"v"(p), "v"(d) : "memory");
    else if constexpr (AUX == 16) asm volatile("global_store_dwordx2 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(d) : "memory");
    else asm volatile("global_store_dwordx2 %0, %1, off sc1 nt\n\ts_nop 1" ::"v"(p), "v"(d) : "memory");
  }
}

// The two LDS-DMA requests of line pair P of a row block: global source = src + P * 256 (+ 128) bytes as the instruction's
// immediate offset — which the hardware adds to the LDS address too, hence the "- offset" on the destination.
template <int P, int AUX>
__device__ __forceinline__ void panel_dma_pair(const float *src, unsigned char *dst) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src),
                                   (__attribute__((address_space(3))) void *)(dst - P * 256), 16, P * 256, AUX);
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src),
                                   (__attribute__((address_space(3))) void *)(dst + 1024 - (P * 256 + 128)), 16, P * 256 + 128, AUX);
}

// MT x NT MFMA tiles of 32 x 32 per wavefront tile, NW wavefronts, BM = 32 MT rows per workgroup.
// PRE: what the split pass adds to the fetched rows: 0 nothing, 1 an addend matrix (a0 / a1, each optional), 2 gather mode.
// STAUX / LDAUX: cache policy bits of the epilogue stores / the panel fetch (0 = default, 2 = nt, 16 = sc1, 18 = both).  The
// output streams through once; left at the default policy its lines stay in the XCD's L2 and push out the weight image
// every wavefront re-reads (tools/gemm_diag/panel_run.py).
// DRIP: a finished column tile's accumulators move to a second register set and are stored ONE 16-byte piece per k16 step
// of the next tile instead of as a burst of 16 stores at the tile's end.
// WD: weight fragments in flight, in k16 steps ahead of the MFMAs that consume them (ring of WD + 1 stages).
// OLDEPI (A/B record only): the epilogue as it stood through round 4 — every 16-byte piece loaded its bias right before its
// store, and since the bias pointer may alias y the loads could not move above the stores: load, s_waitcnt vmcnt(0), store,
// 16 times per column tile, each wait covering the PREVIOUS piece's store (and the prefetched weight fragments): a
// wavefront sat through 16 store round trips per tile, which is why the MFMA and the store phases added up instead of
// overlapping (round 5: found in the ISA).  Now the tile's bias fragments are loaded at the top of the tile's k loop and the
// epilogue is 16 stores back to back with no wait.
template <int NPROD, int MT, int NT, int NW, bool LN, int PRE, int STAUX = 0, int LDAUX = 0, bool DRIP = false, int WD = 2,
          bool OLDEPI = false>
// (MT x NT = 4 x 2 with 4 wavefronts: ONE wavefront per SIMD with the whole register file — the dripping-store form, whose
// second accumulator set does not fit 256 registers)
__global__ void __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu((MT * NT == 8 && NW == 4) ? 1 : 2, (MT * NT == 8 && NW == 4) ? 1 : 2)))
linear_panel_kernel(const PanelArgs a) {
  static_assert(NPROD == 1 || NPROD == 3, "NPROD");
  static_assert((MT == 2 || MT == 4) && (NT == 1 || NT == 2), "wavefront tile");
  constexpr bool LO = NPROD == 3;
  constexpr int NPL = LO ? 2 : 1;
  constexpr int BM = MT * 32;
  constexpr int NPAIR = (BM / 8) * 4;          // (row block q, line pair p)
  constexpr int TW = NT * 32;                  // columns per wavefront tile
  constexpr int PANEL_BYTES = NPAIR * 2048;
  constexpr int STAT_FLOATS = LN ? NW * BM : 0;
  static_assert(NPAIR % NW == 0, "pairs per wavefront");
  __shared__ __attribute__((aligned(16))) unsigned char lds[PANEL_BYTES + STAT_FLOATS * 4];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // SGPR: LDS-DMA bases and buffer soffsets are scalar operands
#ifdef BEVMSDA_PANEL_DIAG
  const bool stamping_ = a.prof != nullptr && (blockIdx.x & 63) == 0 && (wave == 0 || wave == NW - 1);
  unsigned long long t_prev_ = stamping_ ? __builtin_amdgcn_s_memtime() : 0ULL;
  if (stamping_ && lane == 0) atomicAdd(a.prof + 7, 1ULL);
#endif
  // (one workgroup per row panel.  A persistent grid — what is resident at once, a workgroup walking panels blockIdx.x,
  // + gridDim.x, ... — was measured in round 5 and changed nothing: 488.7 vs 480.6 us and 240.6 vs 240.4 us on the hoisted
  // projections, profiles/r5/r5g_persistent_panels_ab.txt; its loop also cost the prologue its registers.)
  const long m0 = static_cast<long>(blockIdx.x) * BM;
  if (a.seg_start != nullptr) {                // unused row segments: nobody will read their outputs
    long halo = 0;
    for (int l = 0; l < a.num_levels; ++l) halo = a.level_shapes[2 * l + 1] > halo ? a.level_shapes[2 * l + 1] : halo;
    halo += a.num_levels > 0 ? 1 : 0;
    const long first = m0 - halo > 0 ? m0 - halo : 0;
    const long last = (m0 + BM + halo < a.M ? m0 + BM + halo : a.M) - 1;
    const int s_lo = static_cast<int>(first / a.seg_len), s_hi = static_cast<int>(last / a.seg_len);
    int used = 0;
    for (int sg = s_lo; sg <= s_hi; ++sg) used |= a.seg_start[sg + 1] - a.seg_start[sg];
    if (used == 0) return;                     // (uniform over the workgroup)
  }
  const int K = a.K0 + a.K1;
  const int nhalf = K / kPanelK;
  const int nstep = K / 16;                    // k16 steps of the whole K axis (weight image stride)
  const int nct = (a.N + TW - 1) / TW;         // column tiles

  // ---- DMA / split assignment of this lane: slot l of every pair the wavefront owns
  const int d_rl = lane >> 3, d_cc = lane & 7;

  // ---- fragment read addresses (bytes): lane -> row (lane & 31) of MFMA tile i, k half h = lane >> 5
  const int f_r = lane & 31, f_h = lane >> 5;
  const int f_q0 = ((f_r >> 2) & 1) | ((f_r >> 4) << 1);             // row bits 2, 4 (tile i adds bits 5, 6)
  const int f_rl = ((f_r & 3) << 1) | ((f_r >> 3) & 1);
  const int f_x = f_r & 7;
  unsigned f_addr[4];                          // per (s & 3): base of tile 0, pair p = 0, hi plane
  // (f_addr / e_row are FILLED behind the panel pass, from a laundered lane id: computed up front they stay live across the
  // DMA section, whose 64-bit source addresses then spill — and a scratch reload between the two batches of eight DMAs is
  // followed by s_waitcnt vmcnt(0): the second batch went out only after the first had landed, two HBM round trips in a
  // row at the head of every workgroup, 8 % of its clocks (tools/gemm_diag/panel_phases.py, round 5))
  auto fill_lane_tables = [&](int ll, unsigned (&fa)[4]) {
    const int r_ = ll & 31, h_ = ll >> 5;
    const int q0_ = ((r_ >> 2) & 1) | ((r_ >> 4) << 1);
    const int rl_ = ((r_ & 3) << 1) | ((r_ >> 3) & 1);
#pragma unroll
    for (int sc = 0; sc < 4; ++sc)
      fa[sc] = static_cast<unsigned>(q0_ * 4 * 2048 + (rl_ * 8 + (((2 * sc + h_) ^ (r_ & 7)))) * 16);
  };
  (void)f_q0; (void)f_rl; (void)f_x; (void)f_h;

  __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t *>(a.wp), 0,
                                                                   static_cast<int>(a.wp_bytes), 0x00020000);
  const int wlane = lane * 16;

  lin_f32x16 acc[MT][NT];
  constexpr int RS = WD + 1;
  lin_bf16x8 wf[RS][NT][NPL];                  // weight fragments: ring over k16 steps (WD in flight)

  // weight fragments of (column tile ct, global step sg) -> ring stage st
  auto wload = [&](int st, int ct, int sg) {
    if (PANEL_DIAG(a, 2) && (sg & 15) > 1) return;
#pragma unroll
    for (int jn = 0; jn < NT; ++jn)
#pragma unroll
      for (int pl = 0; pl < NPL; ++pl) {
        const int so = (((ct * NT + jn) * nstep + sg) * 2 + pl) * 1024;     // wave-uniform byte offset
        wf[st][jn][pl] = __builtin_bit_cast(lin_bf16x8, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, wlane, so, 0));
      }
  };

  // one 16-byte piece of a finished column tile: piece pc = (MFMA tile i, j, register group g): bias, ReLU, store.
  // MFMA D tile (W fragment as the A operand): lane holds output row m = lane & 31 and, in registers 4g .. 4g + 3,
  // columns nb + 8g .. + 3 with nb = 4 (lane >> 5)
  constexpr int NPIECE = MT * NT * 4;
  // bias fragments of one column tile: the lane's 4 consecutive columns of register group g of MFMA tile j.  Loaded at the
  // top of the tile's k loop (``bias_load``), long before the epilogue reads them: no load, and so no wait, between the
  // epilogue's stores (see OLDEPI above)
  float4 bfr[OLDEPI ? 1 : NT][OLDEPI ? 1 : 4];
  auto bias_load = [&](int tct) {
    if constexpr (!OLDEPI) {
      if (a.bias == nullptr) return;           // (uniform)
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int n = tct * TW + j * 32 + 4 * (lane >> 5) + 8 * g;
          bfr[j][g] = *reinterpret_cast<const float4 *>(a.bias + (n < a.N ? n : 0));     // (columns >= N are never stored)
        }
    }
  };
  // A piece's address: the output group (n0 / group_cols) is ONE raw buffer per column tile whose records end with the
  // workgroup's last row — rows >= M are dropped by the bounds check, so no per-lane predicate and no 64-bit address
  // arithmetic per piece: voffset = this lane's (row, 4 (lane >> 5)) element of row tile i (computed once per kernel),
  // soffset = the tile's first column, the piece's own column offset (32 j + 8 g) an immediate.  Through round 4 the
  // epilogue of a column tile was ~700 executed instructions (64-bit multiplies and a chain of uniform branches per
  // piece) against the tile's 192 MFMAs — and a wavefront issues no MFMA while it walks them (found in the ISA, round 5).
  const unsigned ldy_u = static_cast<unsigned>(a.ldy);
  unsigned e_row[MT];
  const long rows_here = a.M - m0 < BM ? a.M - m0 : BM;
  auto tile_rsrc = [&](int tct, int &soff_elems) {
    const int n0 = tct * TW;
    const int grp = a.group_cols > 0 ? n0 / a.group_cols : 0;
    soff_elems = n0 - grp * a.group_cols;
    const long first = (static_cast<long>(grp) * a.M + m0) * a.ldy;                 // element of (row m0, column 0) of the group
    const int es = a.out_bf16 ? 2 : 4;
    unsigned char *base = reinterpret_cast<unsigned char *>(a.y) + first * es;
    return __builtin_amdgcn_make_buffer_rsrc(base, 0, static_cast<int>(rows_here * a.ldy * es), 0x00020000);
  };
  auto store_piece = [&](const lin_f32x16 (&t)[MT][NT], int tct, int pc, bool add_bias = true) {
    const int i = pc / (NT * 4), j = (pc / 4) % NT, g = pc % 4;
    if constexpr (OLDEPI) {
      const int n0 = tct * TW;
      const int grp = a.group_cols > 0 ? n0 / a.group_cols : 0;
      const long m = m0 + i * 32 + (lane & 31);
      const int n = n0 + j * 32 + 4 * (lane >> 5) + 8 * g;
      if (m >= a.M || n >= a.N) return;          // N % 4 == 0: n < N covers n .. n + 3
      float4 v = make_float4(t[i][j][4 * g], t[i][j][4 * g + 1], t[i][j][4 * g + 2], t[i][j][4 * g + 3]);
      if (PANEL_DIAG(a, 1) && v.x != 1.2345e30f) return;
      if (a.bias) v = lin_add4(v, *reinterpret_cast<const float4 *>(a.bias + n));
      if (a.relu) {                              // NaN stays NaN, as torch.relu
        v.x = v.x < 0.f ? 0.f : v.x;
        v.y = v.y < 0.f ? 0.f : v.y;
        v.z = v.z < 0.f ? 0.f : v.z;
        v.w = v.w < 0.f ? 0.f : v.w;
      }
      const long off = (static_cast<long>(grp) * a.M + m) * a.ldy + (n - grp * a.group_cols);
      if (a.out_bf16) {
        uint2 pk;
        pk.x = lin_pack2(v.x, v.y);
        pk.y = lin_pack2(v.z, v.w);
        panel_store<STAUX>(reinterpret_cast<uint16_t *>(a.y) + off, pk);
      } else {
        panel_store<STAUX>(a.y + off, v);
      }
    }
  };
  // one piece of a tile whose bias / ReLU are already applied (the dripping epilogue), fp32 or bf16 by the uniform flag
  auto store_one = [&](const lin_f32x16 (&t)[MT][NT], __amdgpu_buffer_rsrc_t yr, int soff_e, int pc) {
    const int i = pc / (NT * 4), j = (pc / 4) % NT, g = pc % 4;
    if (a.out_bf16) {
      const panel_u32x2 d = {lin_pack2(t[i][j][4 * g], t[i][j][4 * g + 1]), lin_pack2(t[i][j][4 * g + 2], t[i][j][4 * g + 3])};
      __builtin_amdgcn_raw_buffer_store_b64(d, yr, static_cast<int>(e_row[i] * 2u) + (j * 32 + 8 * g) * 2, soff_e * 2, STAUX);
    } else {
      const panel_u32x4 d = {__float_as_uint(t[i][j][4 * g]), __float_as_uint(t[i][j][4 * g + 1]),
                             __float_as_uint(t[i][j][4 * g + 2]), __float_as_uint(t[i][j][4 * g + 3])};
      __builtin_amdgcn_raw_buffer_store_b128(d, yr, static_cast<int>(e_row[i] * 4u) + (j * 32 + 8 * g) * 4, soff_e * 4, STAUX);
    }
  };
  // the lean epilogue of one column tile (all NPIECE pieces): uniform decisions once per tile, straight-line pieces
  auto store_tile = [&](lin_f32x16 (&t)[MT][NT], int tct, bool add_bias) {
    int soff_e;
    __amdgpu_buffer_rsrc_t yr = tile_rsrc(tct, soff_e);
    const bool jok[2] = {tct * TW < a.N, tct * TW + 32 < a.N};     // (N % 32 == 0 is not required: a half tile may hang over)
    if (a.bias && add_bias) {
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            t[i][j][4 * g] += bfr[j][g].x; t[i][j][4 * g + 1] += bfr[j][g].y;
            t[i][j][4 * g + 2] += bfr[j][g].z; t[i][j][4 * g + 3] += bfr[j][g].w;
          }
    }
    if (a.relu) {                                // NaN stays NaN, as torch.relu
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) t[i][j][r] = t[i][j][r] < 0.f ? 0.f : t[i][j][r];
    }
    if (PANEL_DIAG(a, 1)) return;
    if (tct * TW + TW > a.N) {
      // the last column tile of an N that is not a multiple of the tile width: per-lane column checks (N % 4 == 0, so a
      // piece is in or out as a whole); rare (N = 100 in the tests), kept off the straight-line path below
#pragma unroll
      for (int pc = 0; pc < NPIECE; ++pc) {
        const int j = (pc / 4) % NT, g = pc % 4;
        if (tct * TW + j * 32 + 4 * (lane >> 5) + 8 * g < a.N) store_one(t, yr, soff_e, pc);
      }
      return;
    }
    if (a.out_bf16) {
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          if (!jok[j]) continue;                 // (uniform)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const panel_u32x2 d = {lin_pack2(t[i][j][4 * g], t[i][j][4 * g + 1]), lin_pack2(t[i][j][4 * g + 2], t[i][j][4 * g + 3])};
            __builtin_amdgcn_raw_buffer_store_b64(d, yr, static_cast<int>(e_row[i] * 2u) + (j * 32 + 8 * g) * 2, soff_e * 2, STAUX);
          }
        }
    } else {
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          if (!jok[j]) continue;
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const panel_u32x4 d = {__float_as_uint(t[i][j][4 * g]), __float_as_uint(t[i][j][4 * g + 1]),
                                   __float_as_uint(t[i][j][4 * g + 2]), __float_as_uint(t[i][j][4 * g + 3])};
            __builtin_amdgcn_raw_buffer_store_b128(d, yr, static_cast<int>(e_row[i] * 4u) + (j * 32 + 8 * g) * 4, soff_e * 4, STAUX);
          }
        }
    }
  };
  lin_f32x16 prev[DRIP ? MT : 1][DRIP ? NT : 1];
  int prev_ct = 0;
  bool have_prev = false;
  int prev_soff = 0;
  __amdgpu_buffer_rsrc_t prev_rs = tile_rsrc(0, prev_soff);
  static_assert(!(DRIP && OLDEPI), "the dripping epilogue applies the bias when a tile retires");

  for (int half = 0; half < nhalf; ++half) {
    // ---------------------------------------------------------------- panel pass: fetch, split, one barrier
    const int kb = half * kPanelK;
    const bool second = kb >= a.K0;
    const float *xs = second ? a.x1 + (kb - a.K0) : a.x0 + kb;
    const long ldx = second ? a.ldx1 : a.ldx0;
    const float *as = second ? a.a1 : a.a0;
    const long lda = second ? a.lda1 : a.lda0;
    if (as) as += second ? (kb - a.K0) : kb;
    if (half > 0) __syncthreads();             // every wavefront is done with the previous pass's planes
    constexpr int PPW = NPAIR / NW;            // pairs per wavefront = QPW row blocks x 4 line pairs
    constexpr int QPW = PPW / 4;
    const bool add = PRE == 1 && as != nullptr;
    long srow[QPW], arow[QPW];                 // DMA source row; addend row / second gathered row
    const float *xsu[QPW];                     // DMA source matrix of the row (the second row block has its own)
    int g0[QPW], g1[QPW];
    float gs[QPW];
    int cx[QPW];                               // 16-byte column of this lane's slot: c = d_cc ^ (row & 7)
#pragma unroll
    for (int u = 0; u < QPW; ++u) {
      if constexpr (PRE == 0) break;             // (the plain projection forms its addresses inside the DMA loop below)
      const int row = panel_row_of(wave * QPW + u, d_rl);
      cx[u] = d_cc ^ (row & 7);
      long gm = m0 + row;
      if (gm >= a.M) gm = a.M - 1;             // clamped rows are computed and never stored
      srow[u] = arow[u] = gm;
      xsu[u] = xs;
      if (PRE == 0 && a.xb != nullptr && gm >= a.m_split) {
        xsu[u] = a.xb + kb;
        srow[u] = gm - a.m_split;
      }
      g0[u] = 0; g1[u] = -1; gs[u] = 1.f;
      if (PRE == 2) {
        g0[u] = a.gidx[gm * 2];
        g1[u] = a.gidx[gm * 2 + 1];
        gs[u] = a.gscale[gm];
      }
    }
    if (PRE == 2) {
#pragma unroll
      for (int u = 0; u < QPW; ++u) {
        srow[u] = g0[u] < 0 ? 0 : g0[u];
        arow[u] = g1[u] < 0 ? 0 : g1[u];
      }
    }
    __builtin_amdgcn_sched_barrier(0);         // (nothing of the later set-up is to be scheduled into the DMA section)
#pragma unroll
    for (int u = 0; u < QPW; ++u) {
      if (PANEL_DIAG(a, 4)) break;
      // ONE 64-bit source address per row block; the line pair and the half travel as the instruction's immediate
      // offset (p * 256 + {0, 128} bytes).  The immediate is added to the LDS address as well, so the destination
      // handed over is the real one minus that offset (panel_dma_pair).
      const float *src;
      if constexpr (PRE == 0) {
        // (nothing of this survives the loop: the split pass below works on LDS slots only — kept in arrays, the 64-bit
        // row addresses of both row blocks were spilled around the DMA section)
        const int row = panel_row_of(wave * QPW + u, d_rl);
        long gm = m0 + row;
        if (gm >= a.M) gm = a.M - 1;             // clamped rows are computed and never stored
        const bool second_block = a.xb != nullptr && gm >= a.m_split;
        src = (second_block ? a.xb + kb + (gm - a.m_split) * ldx : xs + gm * ldx) + (d_cc ^ (row & 7)) * 4;
      } else {
        src = xsu[u] + srow[u] * ldx + cx[u] * 4;
      }
      unsigned char *dst = lds + ((wave * QPW + u) * 4) * 2048;
      panel_dma_pair<0, LDAUX>(src, dst);
      panel_dma_pair<1, LDAUX>(src, dst + 2048);
      panel_dma_pair<2, LDAUX>(src, dst + 2 * 2048);
      panel_dma_pair<3, LDAUX>(src, dst + 3 * 2048);
    }
    __builtin_amdgcn_sched_barrier(0);
    float4 ad[PRE ? PPW : 1][2];               // addend / second gathered row: plain loads, under the DMA
    if (PRE == 2 || add) {
#pragma unroll
      for (int u = 0; u < QPW; ++u)
#pragma unroll
        for (int p = 0; p < 4; ++p) {
          const float *s1 = (PRE == 2 ? xs + arow[u] * ldx : as + arow[u] * lda) + (2 * p) * 32 + cx[u] * 4;
          ad[PRE ? u * 4 + p : 0][0] = *reinterpret_cast<const float4 *>(s1);
          ad[PRE ? u * 4 + p : 0][1] = *reinterpret_cast<const float4 *>(s1 + 32);
        }
    }
    // the first weight fragments travel under the panel fetch too
    int ct = wave;
    if (half == 0) {
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    }
    if (ct < nct) {
#pragma unroll
      for (int k = 0; k < WD; ++k) wload(k, ct, half * 16 + k);
    }
    PANEL_CLK(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // my own DMA slots have landed (no other lane reads them yet)
    PANEL_CLK(1);
#pragma unroll
    for (int u = 0; u < QPW; ++u)
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        if (PANEL_DIAG(a, 4)) break;
        unsigned char *slot = lds + ((wave * QPW + u) * 4 + p) * 2048 + lane * 16;
        float4 va = *reinterpret_cast<const float4 *>(slot);
        float4 vb = *reinterpret_cast<const float4 *>(slot + 1024);
        if (PRE == 2) {                        // (row0 + row1) * scale, absent rows as zeros: gather_mean's arithmetic
          va = panel_gsum(va, g0[u] >= 0, ad[PRE ? u * 4 + p : 0][0], g1[u] >= 0, gs[u]);
          vb = panel_gsum(vb, g0[u] >= 0, ad[PRE ? u * 4 + p : 0][1], g1[u] >= 0, gs[u]);
        } else if (add) {
          va = lin_add4(va, ad[PRE ? u * 4 + p : 0][0]);
          vb = lin_add4(vb, ad[PRE ? u * 4 + p : 0][1]);
        }
        uint4 hi, lo;
        lin_split8<LO>(va, vb, hi, lo);
        *reinterpret_cast<uint4 *>(slot) = hi;
        if (LO) *reinterpret_cast<uint4 *>(slot + 1024) = lo;
      }
    __syncthreads();                           // planes complete
    PANEL_CLK(2);
    {
      int ll = lane;
      asm volatile("" : "+v"(ll));              // (keeps the two tables out of the DMA section's live set: see f_addr)
      fill_lane_tables(ll, f_addr);
#pragma unroll
      for (int i = 0; i < MT; ++i) e_row[i] = (static_cast<unsigned>(i * 32 + (ll & 31)) * ldy_u + 4u * (ll >> 5));
    }

    // ---------------------------------------------------------------- column sweep: no synchronisation
    const bool last_half = half == nhalf - 1;
    if (a.skew > 0 && half == 0 && (NW == 8 ? wave >= 4 : ((blockIdx.x >> 8) & 1) != 0)) {
      for (int z = 0; z < a.skew; ++z) __builtin_amdgcn_s_sleep(16);       // 16 x 64 clocks
    }
    for (; ct < nct; ct += NW) {
      const int ct_next = ct + NW;
      lin_bf16x8 af[2][MT][NPL];               // activation fragments: step s in set s & 1
      auto aload = [&](int set, int s) {
        if (PANEL_DIAG(a, 3) && s > 1) return;
        const unsigned base = f_addr[s & 3] + (s >> 2) * 2048;
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int pl = 0; pl < NPL; ++pl)
            af[set][i][pl] = *reinterpret_cast<const lin_bf16x8 *>(lds + base + i * (4 * 4 * 2048) + pl * 1024);
      };
      aload(0, 0);
      if (!LN && last_half) bias_load(ct);
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        // weight fragments WD steps ahead (ring stage (s + WD) % RS); past the tile's end: the next tile's first WD
        if (s + WD < 16) {
          wload((s + WD) % RS, ct, half * 16 + s + WD);
        } else if (ct_next < nct) {
          wload((s + WD) % RS, ct_next, half * 16 + s + WD - 16);
        }
        if (s + 1 < 16) aload((s + 1) & 1, s + 1);
        if constexpr (DRIP && !LN) {
          static_assert(!DRIP || 16 % NPIECE == 0 || NPIECE % 16 == 0, "pieces per tile and the 16 steps must divide one another");
          if constexpr (NPIECE >= 16) {
            if (have_prev) {
#pragma unroll
              for (int u = 0; u < NPIECE / 16; ++u) store_one(prev, prev_rs, prev_soff, s * (NPIECE / 16) + u);
            }
          } else {
            if ((s % (16 / NPIECE)) == 0 && have_prev) store_one(prev, prev_rs, prev_soff, s / (16 / NPIECE));
          }
        }
        __builtin_amdgcn_sched_barrier(0);     // requests first: left alone, hipcc sinks them to the end of the step
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
          for (int i = 0; i < MT; ++i) {
            if (PANEL_DIAG(a, 0) && s > 0) {   // operands stay live, the matrix pipe idles
              asm volatile("" ::"v"(wf[s % RS][j][0]), "v"(af[s & 1][i][0]));
              if (LO) asm volatile("" ::"v"(wf[s % RS][j][1]), "v"(af[s & 1][i][1]));
              continue;
            }
            // D[n][m]: the W fragment is the MFMA's A operand (4 consecutive output columns per lane -> float4 stores)
            if (LO) {
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[s % RS][j][0], af[s & 1][i][1], acc[i][j], 0, 0, 0);
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[s % RS][j][1], af[s & 1][i][0], acc[i][j], 0, 0, 0);
            }
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[s % RS][j][0], af[s & 1][i][0], acc[i][j], 0, 0, 0);
          }
        __builtin_amdgcn_sched_barrier(0);     // ... and hoists every fragment read of the tile to its top
      }
      PANEL_CLK(3);
      // the next tile's steps 0 .. WD - 1 sit in ring stages (16 + k) % RS: rotate them to stages 0 .. WD - 1
      if (ct_next < nct && (16 % RS) != 0) {
        lin_bf16x8 tmp[WD][NT][NPL];
#pragma unroll
        for (int k = 0; k < WD; ++k)
#pragma unroll
          for (int jn = 0; jn < NT; ++jn)
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl) tmp[k][jn][pl] = wf[(16 + k) % RS][jn][pl];
#pragma unroll
        for (int k = 0; k < WD; ++k)
#pragma unroll
          for (int jn = 0; jn < NT; ++jn)
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl) wf[k][jn][pl] = tmp[k][jn][pl];
      }
      if (!last_half) continue;                // (two passes: one column tile per wavefront, checked by the launcher)
      if constexpr (!LN) {
        if constexpr (DRIP) {
#pragma unroll
          for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) {
#pragma unroll
              for (int r = 0; r < 16; ++r) {
                // (the tile retires with its bias: the dripped pieces of the next tile's loop are plain stores)
                const float bb = a.bias ? (r & 3) == 0 ? bfr[j][r >> 2].x : (r & 3) == 1 ? bfr[j][r >> 2].y
                                                   : (r & 3) == 2 ? bfr[j][r >> 2].z : bfr[j][r >> 2].w : 0.f;
                const float pv = a.bias ? acc[i][j][r] + bb : acc[i][j][r];
                prev[i][j][r] = (a.relu && pv < 0.f) ? 0.f : pv;
                acc[i][j][r] = 0.f;
              }
            }
          prev_ct = ct;
          have_prev = true;
          prev_rs = tile_rsrc(ct, prev_soff);
        } else {
          if constexpr (OLDEPI) {
#pragma unroll
            for (int pc = 0; pc < NPIECE; ++pc) store_piece(acc, ct, pc);
          } else {
            store_tile(acc, ct, true);
          }
          PANEL_CLK(4);
#pragma unroll
          for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
              for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        }
      }
    }
  }
  if constexpr (DRIP && !LN) {
    if (have_prev) {
#pragma unroll
      for (int pc = 0; pc < NPIECE; ++pc) store_one(prev, prev_rs, prev_soff, pc);
    }
  }

  if constexpr (LN) {
    // y = LayerNorm(acc + bias + res) * gamma + beta over the 256 columns of a row (N = NW * TW: one column tile per
    // wavefront, checked by the launcher).  A row lives in 2 lanes (l, l ^ 32) of each of the NW wavefronts; two-pass
    // statistics (mean, then centred sum of squares) exchanged through LDS; torch.nn.LayerNorm semantics.
    float *stat = reinterpret_cast<float *>(lds + PANEL_BYTES);      // [wave][BM]
    const int n0 = wave * TW;
    float mean[MT], rstd[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const long m = m0 + i * 32 + (lane & 31);
      const bool mok = m < a.M;
      const float *rrow = a.res ? a.res + (mok ? m : 0) * a.ldres : nullptr;
      float sum = 0.f;
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int nb = n0 + j * 32 + 4 * (lane >> 5);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int n = nb + 8 * g;
          float4 v = make_float4(acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]);
          if (a.bias) v = lin_add4(v, *reinterpret_cast<const float4 *>(a.bias + n));
          if (rrow) v = lin_add4(v, *reinterpret_cast<const float4 *>(rrow + n));
          acc[i][j][4 * g] = v.x; acc[i][j][4 * g + 1] = v.y; acc[i][j][4 * g + 2] = v.z; acc[i][j][4 * g + 3] = v.w;
          sum += (v.x + v.y) + (v.z + v.w);
        }
      }
      sum += __shfl_xor(sum, 32, 64);
      mean[i] = sum;
    }
    if (lane < 32) {
#pragma unroll
      for (int i = 0; i < MT; ++i) stat[wave * BM + i * 32 + lane] = mean[i];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      float t = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) t += stat[w * BM + i * 32 + (lane & 31)];
      mean[i] = t * (1.0f / static_cast<float>(NW * TW));
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      float ss = 0.f;
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float d = acc[i][j][r] - mean[i];
          ss = fmaf(d, d, ss);
        }
      ss += __shfl_xor(ss, 32, 64);
      rstd[i] = ss;
    }
    if (lane < 32) {
#pragma unroll
      for (int i = 0; i < MT; ++i) stat[wave * BM + i * 32 + lane] = rstd[i];
    }
    __syncthreads();
    // gamma / beta fragments of this wavefront's columns: loaded once, in front of the first store (a load between two
    // stores waits for the store in front of it: OLDEPI note above)
    float4 gaf[NT][4], bef[NT][4];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int n = n0 + j * 32 + 4 * (lane >> 5) + 8 * g;
        gaf[j][g] = *reinterpret_cast<const float4 *>(a.gamma + n);
        bef[j][g] = *reinterpret_cast<const float4 *>(a.beta + n);
      }
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      float t = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) t += stat[w * BM + i * 32 + (lane & 31)];
      rstd[i] = rsqrtf(t * (1.0f / static_cast<float>(NW * TW)) + a.eps);
    }
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const long m = m0 + i * 32 + (lane & 31);
      if (m >= a.M) continue;
      float *yrow = a.y + m * a.ldy;
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int nb = n0 + j * 32 + 4 * (lane >> 5);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int n = nb + 8 * g;
          const float4 ga = gaf[j][g];
          const float4 be = bef[j][g];
          float4 v;
          v.x = (acc[i][j][4 * g] - mean[i]) * rstd[i] * ga.x + be.x;
          v.y = (acc[i][j][4 * g + 1] - mean[i]) * rstd[i] * ga.y + be.y;
          v.z = (acc[i][j][4 * g + 2] - mean[i]) * rstd[i] * ga.z + be.z;
          v.w = (acc[i][j][4 * g + 3] - mean[i]) * rstd[i] * ga.w + be.w;
          panel_store<STAUX>(yrow + n, v);
        }
      }
    }
  }
}

// Weight image of the row-panel kernel: for every 32-row tile T of w (N, K), k16 step sg and plane (hi, lo) the
// 64 lanes x 8 bf16 of one MFMA operand, contiguous (1 KiB):
//   blob[(((T * K/16 + sg) * 2 + plane) * 64 + lane) * 8 + e] = plane(w[T*32 + (lane & 31)][k])
//   k = 256 (sg / 16) + 32 (2 p + (e >> 2)) + 4 c + (e & 3),  8 p + c = 2 (sg % 16) + (lane >> 5)
// Rows >= N are zero (N is padded to the launcher's column-tile width).  One thread per (T, sg, lane).
// TRANSPOSED: element (n, k) of the weight is w[k * ldw + n] (the image of W^T from W where it lies: backward GEMMs).
template <bool TRANSPOSED>
__device__ __forceinline__ void lin_panel_pack_weight_thread(long t, const float *__restrict__ w, long ldw, int N, int K,
                                                             int n_tiles32, uint16_t *__restrict__ blob) {
  const int nstep = K / 16;
  if (t >= static_cast<long>(n_tiles32) * nstep * 64) return;
  const int lane = static_cast<int>(t & 63);
  const int sg = static_cast<int>((t >> 6) % nstep);
  const int T = static_cast<int>((t >> 6) / nstep);
  const int n = T * 32 + (lane & 31);
  const int j = 2 * (sg % 16) + (lane >> 5);
  const int p = j >> 3, c = j & 7;
  const int k = (sg / 16) * kPanelK + (2 * p) * 32 + 4 * c;
  uint4 hi = make_uint4(0, 0, 0, 0), lo = hi;
  if (n < N) {
    if constexpr (TRANSPOSED) {
      const float *src = w + static_cast<long>(k) * ldw + n;
      lin_split8<true>(make_float4(src[0], src[ldw], src[2 * ldw], src[3 * ldw]),
                       make_float4(src[32 * ldw], src[33 * ldw], src[34 * ldw], src[35 * ldw]), hi, lo);
    } else {
      const float *src = w + static_cast<long>(n) * ldw + k;
      lin_split8<true>(*reinterpret_cast<const float4 *>(src), *reinterpret_cast<const float4 *>(src + 32), hi, lo);
    }
  }
  uint4 *dst = reinterpret_cast<uint4 *>(blob) + ((static_cast<long>(T) * nstep + sg) * 2) * 64 + lane;
  dst[0] = hi;
  dst[64] = lo;
}

template <bool TRANSPOSED = false>
__global__ void __launch_bounds__(256) lin_panel_pack_weight_kernel(const float *__restrict__ w, long ldw, int N, int K,
                                                                   int n_tiles32, uint16_t *__restrict__ blob) {
  lin_panel_pack_weight_thread<TRANSPOSED>(static_cast<long>(blockIdx.x) * 256 + threadIdx.x, w, ldw, N, K, n_tiles32, blob);
}

// Many weight images in ONE launch (round 6): the training step re-packs the images of every trainable weight from the
// weights' current values — 52 launches of ~5 us each at base, now one.  `jobs` lives in DEVICE memory (a captured graph
// replays the launch with the table it was captured with); job j owns the blocks [first_block[j], first_block[j + 1]).
struct PackJob {
  const float *w;         // the weight matrix, or (kind & 1) the memory its transpose lies in
  long long ldw;
  unsigned short *blob;
  int N, K;
  int kind;               // bit 0: transposed source; bit 1: row-panel image (fragment order), else the first kernel's
  int first_block;
};

__global__ void __launch_bounds__(256) lin_pack_weights_multi_kernel(const PackJob *__restrict__ jobs, int njobs) {
  int lo = 0, hi = njobs - 1;                   // the job whose block range holds blockIdx.x (uniform: scalar loads)
  const int b = static_cast<int>(blockIdx.x);
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (jobs[mid].first_block <= b) lo = mid; else hi = mid - 1;
  }
  const PackJob j = jobs[lo];
  const long t = static_cast<long>(b - j.first_block) * 256 + threadIdx.x;
  uint16_t *blob = reinterpret_cast<uint16_t *>(j.blob);
  switch (j.kind) {
    case 0: lin_pack_weight_thread<false>(t, j.w, j.ldw, j.N, j.K, blob); break;
    case 1: lin_pack_weight_thread<true>(t, j.w, j.ldw, j.N, j.K, blob); break;
    case 2: lin_panel_pack_weight_thread<false>(t, j.w, j.ldw, j.N, j.K, ((j.N + 63) / 64) * 2, blob); break;
    default: lin_panel_pack_weight_thread<true>(t, j.w, j.ldw, j.N, j.K, ((j.N + 63) / 64) * 2, blob); break;
  }
}

}  // namespace bevmsda
