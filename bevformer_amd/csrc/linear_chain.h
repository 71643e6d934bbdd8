// The row-local tail of an encoder layer as ONE kernel (K3c):
//
//     x = LayerNorm0( A W0^T + b0 + res )                 output projection of the attention + "+ identity" + norm
//     y = LayerNorm1( x + relu(x W1^T + b1) W2^T + b2 )   FFN (256 -> 512 -> 256) + "+ identity" + norm
//
// i.e. SpatialCrossAttention's `output_proj` and residual (spatial_cross_attention.py:165-175, with the per-camera
// scatter-add and camera-count division as a two-row gather in the A-load), `norms[1]`, the mmcv FFN and `norms[2]` of
// BEVFormerLayer's operation order (encoder.py:376-404).  Every one of those ops is local to a BEV row, so a workgroup
// that owns a panel of 64 complete rows can run the whole chain without the grid ever leaving the chip: as separate
// launches the chain moves ~10 passes of the (Q, 256) grid through HBM (x written and read twice, the 512-wide hidden
// layer written and read), here it reads the sampled rows and the residual and writes y — 3.15 passes.
//
// Built from the row-panel kernel's parts (linear_panel.h): the panel is fetched whole by LDS-DMA and split once into
// [hi | lo] bf16 planes; weight fragments come straight from L2 in MFMA operand order; 8 wavefronts, each owning ONE
// 32-column tile of every stage's output (N = 256 = 8 x 32 for all of them once the hidden layer is taken in two halves
// of 256), so there is no column loop and the accumulators of a stage ARE the rows the next stage needs:
//
//   stage 0   planes(A) in buffer 0  ->  acc = A W0^T; + b0 + res; LayerNorm0 (row statistics exchanged through LDS)
//             -> x kept in registers (fp32: the FFN's residual) and written as planes into buffer 1
//   half h    acc = x W1[256h .. 256h+255]^T; + b1, relu -> planes into buffer 0      (the hidden layer never exists
//             acc2 += h_half W2[:, 256h ..]^T                                          beyond one 64 x 256 half in LDS)
//   final     acc2 + b2 + x -> LayerNorm1 -> y
//
// MODE 1 — the attention-to-attention seam of the layer, same machinery:
//     x = LayerNorm0( A W0^T + b0 + res )    TemporalSelfAttention's output projection + "+ identity" + norms[0]
//     p = x W1^T + b1                         SpatialCrossAttention's merged [sampling_offsets | attention_weights]
//                                             projection of the same rows (N2 = 768 columns at base)
// x is stored (it is SpatialCrossAttention's residual) AND kept as planes, so the second projection never re-reads it;
// a wavefront walks the 32-column tiles t = wave, wave + 8, ... of p.
//
// TP = true (MODE 0, inference; round 6) — the seam BETWEEN two layers on the same machinery: behind y the workgroup also
// forms the NEXT layer's TemporalSelfAttention offset / weight projection of the rows it has just produced,
//     p = [first | y + pos] W3^T + b3        (temporal_self_attention.py:197-211: cat([value[:bs], query + query_pos], -1))
// `first` (the history BEV's rows: the K columns 0 .. 255) is fetched by LDS-DMA into buffer 1 under LayerNorm1, y + pos
// goes into buffer 0 as planes straight from the accumulators (y is never re-read from HBM), and a wavefront walks the
// 32 NT-column tiles t = wave, wave + NW, ... of p over K = 512 (32 k16 steps across the two buffers).  The stand-alone
// projection it replaces re-reads y and runs at 20 % MFMA utilisation in its own launch (DESIGN §4 K3c "tail projection").
//
// LDS: 2 x 64 KiB plane buffers + 2 KiB of row statistics: one workgroup (512 threads, 2 wavefronts per SIMD) per CU.
// The accumulator -> plane write uses the same slot map the DMA + split pass produces, so the fragment reads of every
// stage are the conflict-free ones of linear_panel.h (tests/test_linear_layout_model.py replays the arithmetic).
#pragma once
#include <type_traits>
#include "linear_panel.h"
#include "scalar_ops.h"

namespace bevmsda {

struct ChainArgs {
  const float *rows;                // A source rows (ld_rows); with gidx: gathered (two rows per output row)
  long ld_rows;
  const int32_t *gidx;              // (M, gstride) or nullptr: A[m] = rows[m]
  int gstride;                      // 0 / 2: two-row gather; J > 2: columns 2.. (while >= 0) are added to the first row's
                                    // values before the two-row sum — the plan's slots seen by more than two cameras
  const float *gscale;
  const uint16_t *w0, *w1, *w2;     // fragment-order weight images: (256, 256), (512, 256), (256, 512)
  const float *b0, *b1, *b2;
  const float *res;                 // (M, ld_res) residual of stage 0, or nullptr
  long ld_res;
  const float *gamma0, *beta0, *gamma1, *beta1;
  float eps0, eps1;
  float *y;
  long ld_y;
  long M;
  float *y2;                        // MODE 1: the second projection's output (M, ld_y2), N2 columns
  long ld_y2;
  int N2;
  // SAVE = true (the forward of the autograd path): what the backward needs of the rows that otherwise never leave
  // the chip, stored as dense (M, 256) / (M, 512) matrices:
  float *sv_z0;                     // A W0^T + b0 + res, the input of LayerNorm0
  float *sv_x;                      // MODE 0: x = LayerNorm0(...) (MODE 1 stores it as `y` anyway)
  float *sv_h;                      // MODE 0: relu(x W1^T + b1), (M, 512)
  float *sv_z1;                     // MODE 0: x + h W2^T + b2, the input of LayerNorm1
  // DROP = true (train() mode with active dropout): per-element scale tensors, 0 or 1 / (1 - p), dense like the saves;
  // nullptr = that dropout is inactive.  The reference's nn.Dropout sites of the chain:
  const float *dk0;                 // (M, 256): on A W0^T + b0, before "+ res" (spatial_cross_attention.py:175 /
                                    //           temporal_self_attention.py:272)
  const float *dkh;                 // MODE 0, (M, 512): on relu(x W1^T + b1) (mmcv FFN: Linear, ReLU, Dropout)
  const float *dk1;                 // MODE 0, (M, 256): on h W2^T + b2, before "+ x"
  // MODE 2 (the backward of MODE 0, round 4): the incoming gradient, the four tensors SAVE stored, the gradients out.
  // w0 / w1 / w2 are then the images of W0^T (256 x 256), W2^T (512 x 256) and W1^T (256 x 512).
  const float *bw_gy;               // (M, bw_ld_gy): gradient of y
  long bw_ld_gy;
  const float *bw_z1, *bw_h, *bw_z0;   // dense (M, 256), (M, 512), (M, 256)
  float *bw_gz1;                    // (M, 256) gradient of z1 = LayerNorm1's input (also the FFN output's and, added in, x's)
  float *bw_gh;                     // (M, 512) gradient of the hidden pre-activation (ReLU mask applied)
  float *bw_gz0;                    // (M, 256) gradient of z0 = LayerNorm0's input (= the residual's gradient)
  float *bw_din;                    // (M, 256) gz0 W0: gradient of the seam's input rows (A)
  float *bw_dgb1, *bw_dgb0;         // (2, 256) each: [grad gamma | grad beta] of LayerNorm1 / 0, ADDED to (caller zeroes)
  // train() mode (the forward ran with DROP): dk1 / dk0 above are the scale tensors of the FFN-output / attention dropouts —
  // bw_gz1 then holds gz1 * dk1 (the FFN output's gradient; x still receives gz1), gh is scaled by bw_hscale = 1 / (1 - p) of
  // the hidden dropout (h is zero where dropped), and the projection's gradient gz0 * dk0 goes to bw_gzp (din = that W0)
  float bw_hscale;
  float *bw_gzp;
  // TP = true: the next layer's [first | y + pos] projection (header comment)
  const float *tp_first;            // (M, ld_tp_first): K columns 0 .. 255 of the projection's input
  long ld_tp_first;
  const float *tp_pos;              // (M, ld_tp_pos) addend of y (the positional encoding), or nullptr
  long ld_tp_pos;
  const uint16_t *w3;               // fragment-order image of the (N3, 512) weight
  const float *b3;                  // (N3) or nullptr
  float *y3;                        // (M, ld_y3): the projection
  long ld_y3;
  int N3;                           // a multiple of 32 NT, <= 256
#ifdef BEVMSDA_CHAIN_PROF
  unsigned long long *prof;         // tools/gemm_diag: 12 phase clocks, summed over workgroups (lane 0 of wavefront 0)
#endif
};
#ifdef BEVMSDA_CHAIN_PROF
#define CHAIN_STAMP(k)                                                           \
  do {                                                                           \
    if (tid == 0) {                                                              \
      const unsigned long long now_ = __builtin_readcyclecounter();             \
      atomicAdd(a.prof + (k), now_ - t_prev_);                                   \
      t_prev_ = now_;                                                            \
    }                                                                            \
  } while (0)
#else
#define CHAIN_STAMP(k) do {} while (0)
#endif

constexpr int kChainC = 256, kChainF = 512;
constexpr int kChainMaxN2 = 768;   // MODE 1: columns of the second projection (its bias is staged in LDS)

// MODE 0: projection + norm + FFN + norm; MODE 1: projection + norm (stored) + a second projection of N2 columns.
// Workgroup shape: MT x NT MFMA tiles of 32 x 32 per wavefront, NW wavefronts with NW * NT = 8 (the 256 columns of
// every stage), BM = 32 MT rows:
//   <2, 1, 8>  64 rows, 512 threads, 133 KiB of LDS: one workgroup per CU, every weight fragment feeds two row tiles;
//   <1, 2, 4>  32 rows, 256 threads,  69 KiB of LDS: TWO workgroups per CU — one's fetch / LayerNorm / plane-write
//              phases (58 % of a workgroup's cycles in the first shape: tools/gemm_diag/chain_run.py) run under the
//              other's MFMAs, and 1,250 half-size workgroups quantise better over 256 CUs than 625 — at twice the
//              weight traffic from L2 per row.
template <int NPROD, int PRE, int MODE = 0, int MT = 2, int NT = 1, int NW = 8, bool SAVE = false, bool DROP = false, bool TP = false>
__global__ void __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu(2, 2)))
linear_chain_kernel(const ChainArgs a) {
  static_assert(NPROD == 1 || NPROD == 3, "NPROD");
  static_assert(!TP || (MODE == 0 && !SAVE && !DROP), "TP: the inference form of MODE 0");
  static_assert(PRE == 0 || PRE == 2, "PRE: 0 plain rows, 2 two-row gather");
  static_assert(MODE < 2 || (MT == 1 && NT == 2 && PRE == 0 && !SAVE && !DROP), "MODE 2 / 3: 32-row workgroups of 4 wavefronts");
  static_assert(NW * NT == 8 && (MT == 1 || MT == 2) && (NT == 1 || NT == 2), "workgroup shape");
  constexpr bool LO = NPROD == 3;
  constexpr int NPL = LO ? 2 : 1;
  constexpr int BM = MT * 32;
  constexpr int NTHREADS = NW * 64;
  constexpr int BUF = (BM / 8) * 4 * 2048;     // one plane buffer: BM KiB
  constexpr int PPW = (BM / 8) * 4 / NW;       // (row block, line pair) DMA pairs per wavefront
  constexpr int NCST = 4 * kChainC + kChainMaxN2 + 2 * kChainC;       // gamma0, beta0, gamma1, beta1 | b1 | b0, b2
  constexpr int NSTAT = MODE >= 2 ? 2 : 1;    // (the LayerNorm backward exchanges two row sums at once)
  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * BUF + NSTAT * NW * BM * 4 + NCST * 4];
  unsigned char *const buf0 = lds, *const buf1 = lds + BUF;
  float *const stat = reinterpret_cast<float *>(lds + 2 * BUF);       // [wave][row]
  // the per-column constants of every epilogue, copied once: read from LDS instead of as an L2 round trip per stage
  float *const cst = stat + NSTAT * NW * BM;
  float *const c_g0 = cst, *const c_be0 = cst + 256, *const c_g1 = cst + 512, *const c_be1 = cst + 768;
  float *const c_b1 = cst + 1024, *const c_b0 = cst + 1024 + kChainMaxN2, *const c_b2 = c_b0 + 256;
  const int nb1 = MODE == 1 ? a.N2 : kChainF;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const long m0 = static_cast<long>(blockIdx.x) * BM;
#ifdef BEVMSDA_CHAIN_PROF
  unsigned long long t_prev_ = __builtin_readcyclecounter();
#endif

  {
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int t4 = tid * 4; t4 < NCST; t4 += NTHREADS * 4) {
      const float *src = nullptr;
      if (t4 < 256) src = a.gamma0 + t4;
      else if (t4 < 512) src = MODE >= 2 ? nullptr : a.beta0 + (t4 - 256);
      else if (t4 < 768) src = (MODE == 0 || MODE == 2) ? a.gamma1 + (t4 - 512) : nullptr;
      else if (t4 < 1024) src = MODE == 0 ? a.beta1 + (t4 - 768) : nullptr;
      else if (TP && t4 >= 1024 + kChainF && t4 < 1024 + kChainMaxN2)         // (b3 behind the 512 values of b1)
        src = (a.b3 && t4 - 1024 - kChainF < a.N3) ? a.b3 + (t4 - 1024 - kChainF) : nullptr;
      else if (t4 < 1024 + kChainMaxN2) src = (a.b1 && t4 - 1024 < nb1) ? a.b1 + (t4 - 1024) : nullptr;
      else if (t4 < 1024 + kChainMaxN2 + 256) src = a.b0 ? a.b0 + (t4 - 1024 - kChainMaxN2) : nullptr;
      else src = (MODE == 0 && a.b2) ? a.b2 + (t4 - 1024 - kChainMaxN2 - 256) : nullptr;
      *reinterpret_cast<float4 *>(cst + t4) = src ? *reinterpret_cast<const float4 *>(src) : z4;
    }
  }                                            // (visible after the barrier that closes the panel fetch)

  // fragment read addresses (linear_panel.h)
  const int f_r = lane & 31, f_h = lane >> 5;
  const int f_q0 = ((f_r >> 2) & 1) | ((f_r >> 4) << 1);
  const int f_rl = ((f_r & 3) << 1) | ((f_r >> 3) & 1);
  const int f_x = f_r & 7;
  unsigned f_addr[4];
#pragma unroll
  for (int sc = 0; sc < 4; ++sc)
    f_addr[sc] = static_cast<unsigned>(f_q0 * 4 * 2048 + (f_rl * 8 + (((2 * sc + f_h) ^ f_x))) * 16);
  // accumulator -> plane write addresses: this lane's value (row i * 32 + f_r, column 32 (NT w + j) + 4 f_h + 8 g + e)
  // is k = that column of the next stage: line NT w + j (pair line >> 1, slot half line & 1), 16-byte column f_h + 2 g
  unsigned p_addr[NT][4];
#pragma unroll
  for (int j = 0; j < NT; ++j)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int line = NT * wave + j;
      p_addr[j][g] = static_cast<unsigned>((f_q0 * 4 + (line >> 1)) * 2048 + (f_rl * 8 + ((f_h + 2 * g) ^ f_x)) * 16 + (line & 1) * 8);
    }

  const int wlane = lane * 16;
  lin_f32x16 acc[MT][NT], xk[MT][NT], acc2[MODE != 1 ? MT : 1][MODE != 1 ? NT : 1];

#ifndef BEVMSDA_CHAIN_WD
#define BEVMSDA_CHAIN_WD 2
#endif
  constexpr int WD = BEVMSDA_CHAIN_WD;         // weight fragments in flight (k16 steps ahead); 4 measured no faster at 40,000 rows
  lin_bf16x8 wf[WD + 1][NT][NPL];              // ring over k16 steps
  // column tile `tile` (32 NT columns) of a weight image with `nstep` k16 steps per 32-row tile
  auto wload = [&](__amdgpu_buffer_rsrc_t wrsrc, int tile, int nstep, int st, int sg) {
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int pl = 0; pl < NPL; ++pl)
        wf[st][j][pl] = __builtin_bit_cast(lin_bf16x8, __builtin_amdgcn_raw_buffer_load_b128(
                                                         wrsrc, wlane, (((tile * NT + j) * nstep + sg) * 2 + pl) * 1024, 0));
  };
  // the first WD weight fragments of a stage, requested before the previous stage's epilogue / the barrier in front
  // of it (an L2 round trip that would otherwise open every stage)
  auto wprefetch = [&](__amdgpu_buffer_rsrc_t wrsrc, int tile, int nstep, int sg0) {
#pragma unroll
    for (int k = 0; k < WD; ++k) wload(wrsrc, tile, nstep, k, sg0 + k);
  };
  // c += planes(buf) x W[column tile, k16 steps sg0 .. sg0 + 15]^T   (K = 256); ring stages 0 .. WD - 1 hold steps
  // sg0 .. sg0 + WD - 1 already (wprefetch)
  auto gemm16 = [&](auto &c, const unsigned char *buf, __amdgpu_buffer_rsrc_t wrsrc, int tile, int nstep, int sg0) {
    lin_bf16x8 af[2][MT][NPL];
    auto aload = [&](int set, int s) {
      const unsigned base = f_addr[s & 3] + (s >> 2) * 2048;
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl)
          af[set][i][pl] = *reinterpret_cast<const lin_bf16x8 *>(buf + base + i * (4 * 4 * 2048) + pl * 1024);
    };
    aload(0, 0);
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      if (s + WD < 16) wload(wrsrc, tile, nstep, (s + WD) % (WD + 1), sg0 + s + WD);
      if (s + 1 < 16) aload((s + 1) & 1, s + 1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int i = 0; i < MT; ++i) {
          if constexpr (LO) {
            c[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[s % (WD + 1)][j][0], af[s & 1][i][1], c[i][j], 0, 0, 0);
            c[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[s % (WD + 1)][j][1], af[s & 1][i][0], c[i][j], 0, 0, 0);
          }
          c[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[s % (WD + 1)][j][0], af[s & 1][i][0], c[i][j], 0, 0, 0);
        }
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  auto zero = [&](auto &c) {
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) c[i][j][r] = 0.f;
  };
  // the tile as [hi | lo] planes of the next stage's activation panel
  auto to_planes = [&](const auto &c, unsigned char *buf) {
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const float v0 = c[i][j][4 * g], v1 = c[i][j][4 * g + 1], v2 = c[i][j][4 * g + 2], v3 = c[i][j][4 * g + 3];
          uint2 hi, lo;
          hi.x = lin_pack2(v0, v1);
          hi.y = lin_pack2(v2, v3);
          unsigned char *dst = buf + p_addr[j][g] + i * (4 * 4 * 2048);
          *reinterpret_cast<uint2 *>(dst) = hi;
          if (LO) {
            lo.x = lin_pack2(v0 - __uint_as_float(hi.x << 16), v1 - __uint_as_float(hi.x & 0xffff0000u));
            lo.y = lin_pack2(v2 - __uint_as_float(hi.y << 16), v3 - __uint_as_float(hi.y & 0xffff0000u));
            *reinterpret_cast<uint2 *>(dst + 1024) = lo;
          }
        }
  };
  // columns of this lane's accumulator registers: ncol(j) + 8 g + e
  auto ncol = [&](int j) { return (NT * wave + j) * 32 + 4 * (lane >> 5); };
  // LayerNorm over the 256 columns of every row of the tile set (NW wavefronts x 32 NT columns), in place; two-pass
  // statistics, exchanged through LDS (torch.nn.LayerNorm: biased variance, eps inside the square root)
  auto layernorm = [&](auto &c, const float *gamma, const float *beta, float eps) {
    float mean[MT], rstd[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      float sum = 0.f;
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) sum += c[i][j][r];
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
      mean[i] = t * (1.0f / kChainC);
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      float ss = 0.f;
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float d = c[i][j][r] - mean[i];
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
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      float t = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) t += stat[w * BM + i * 32 + (lane & 31)];
      rstd[i] = rsqrtf(fma_scalar(t, 1.0f / kChainC, eps));     // (scalar_ops.h: the two rows' pair is not formed with eps from a high half)
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int n = ncol(j) + 8 * g;
          const float4 ga = *reinterpret_cast<const float4 *>(gamma + n);
          const float4 be = *reinterpret_cast<const float4 *>(beta + n);
          c[i][j][4 * g] = (c[i][j][4 * g] - mean[i]) * rstd[i] * ga.x + be.x;
          c[i][j][4 * g + 1] = (c[i][j][4 * g + 1] - mean[i]) * rstd[i] * ga.y + be.y;
          c[i][j][4 * g + 2] = (c[i][j][4 * g + 2] - mean[i]) * rstd[i] * ga.z + be.z;
          c[i][j][4 * g + 3] = (c[i][j][4 * g + 3] - mean[i]) * rstd[i] * ga.w + be.w;
        }
    }
    __syncthreads();                           // `stat` may be written again
  };
  // c[i][j][4g .. 4g+3] += vec[ncol(j) + 8 g ..] (a per-column constant from LDS)
  auto add_cols = [&](auto &c, const float *vec, int col0) {
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const float4 v = *reinterpret_cast<const float4 *>(vec + col0 + ncol(j) + 8 * g);
          c[i][j][4 * g] += v.x; c[i][j][4 * g + 1] += v.y; c[i][j][4 * g + 2] += v.z; c[i][j][4 * g + 3] += v.w;
        }
  };

  const unsigned w0b = 8u * 16 * 2 * 1024, w2b = 8u * 32 * 2 * 1024;   // image bytes
  const unsigned w1b = MODE == 3 ? 8u * static_cast<unsigned>(a.N2 / 16) * 2 * 1024      // (W1^T: 256 rows, K = N2)
                                 : static_cast<unsigned>((nb1 + 63) / 64 * 2) * 16 * 2 * 1024;
  __amdgpu_buffer_rsrc_t r0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t *>(a.w0), 0, static_cast<int>(w0b), 0x00020000);
  __amdgpu_buffer_rsrc_t r1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t *>(a.w1), 0, static_cast<int>(w1b), 0x00020000);
  __amdgpu_buffer_rsrc_t r2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t *>(MODE != 1 ? a.w2 : a.w0), 0, static_cast<int>(MODE != 1 ? w2b : w0b), 0x00020000);
  long mrow[MT];
#pragma unroll
  for (int i = 0; i < MT; ++i) mrow[i] = m0 + i * 32 + (lane & 31);
  // the residual rows of stage 0 and its first weight fragments travel under the panel fetch
  float4 rs[MT][NT][4];
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    const float *rrow = a.res ? a.res + (mrow[i] < a.M ? mrow[i] : a.M - 1) * a.ld_res : nullptr;
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        rs[i][j][g] = rrow ? *reinterpret_cast<const float4 *>(rrow + ncol(j) + 8 * g) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  if constexpr (MODE < 2) wprefetch(r0, wave, 16, 0);

  auto store_tile = [&](const auto &c, float *out, long ld, int col0) {
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      if (mrow[i] >= a.M) continue;
      float *yrow = out + mrow[i] * ld + col0;
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g)
          *reinterpret_cast<float4 *>(yrow + j * 32 + 4 * (lane >> 5) + 8 * g) =
              make_float4(c[i][j][4 * g], c[i][j][4 * g + 1], c[i][j][4 * g + 2], c[i][j][4 * g + 3]);
    }
  };

  // c *= mask[row, col0 + my columns] (a dropout scale tensor; rows past M read row M - 1 and are never stored)
  auto scale_tile = [&](auto &c, const float *mask, long ld, int col0) {
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const float *mrowp = mask + (mrow[i] < a.M ? mrow[i] : a.M - 1) * ld + col0;
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const float4 mk = *reinterpret_cast<const float4 *>(mrowp + j * 32 + 4 * (lane >> 5) + 8 * g);
          c[i][j][4 * g] *= mk.x; c[i][j][4 * g + 1] *= mk.y; c[i][j][4 * g + 2] *= mk.z; c[i][j][4 * g + 3] *= mk.w;
        }
    }
  };

  if constexpr (MODE >= 2) {
    // ================================================================== MODE 2 / 3: the backwards of MODE 0 / 1 on the same machinery
    //   gz1 = LN1'(z1; gy)                         column sums -> grad gamma1 / beta1
    //   gh  = (gz1 W2) where h > 0                  (two halves of 256 hidden columns, as the forward)
    //   gx  = gh W1 + gz1                           (x feeds the FFN and, as its residual, LayerNorm1's input)
    //   gz0 = LN0'(z0; gx)                          column sums -> grad gamma0 / beta0
    //   din = gz0 W0
    // Rows live where the forward's accumulators had them (lane = row, 4 consecutive columns per register quad), so the
    // LayerNorm backward is elementwise + the forward LayerNorm's row-sum exchange, and every GEMM is the forward's.
    float *const stat2 = stat + NW * BM;
    const bool row_ok = mrow[0] < a.M;
    const long mclamp = row_ok ? mrow[0] : a.M - 1;
    auto load_tile = [&](auto &c, const float *src, long ld, int col0, bool zero_invalid) {
      const float *rowp = src + mclamp * ld + col0;
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          float4 v = *reinterpret_cast<const float4 *>(rowp + j * 32 + 4 * (lane >> 5) + 8 * g);
          if (zero_invalid && !row_ok) v = make_float4(0.f, 0.f, 0.f, 0.f);
          c[0][j][4 * g] = v.x; c[0][j][4 * g + 1] = v.y; c[0][j][4 * g + 2] = v.z; c[0][j][4 * g + 3] = v.w;
        }
    };
    // sum over the 32 rows (lanes with equal lane >> 5) of 32 per-lane values by recursive halving: lane l ends up with
    // the total of value (l & 31), which is added to out[column of that value] (31 exchanges instead of 160)
    auto colsum_add = [&](float (&v)[32], float *out) {
      auto level = [&](auto xtag) {          // X values stay per lane after the exchange with lane ^ X
        constexpr int X = decltype(xtag)::value;
        const bool up = (lane & X) != 0;
#pragma unroll
        for (int k = 0; k < X; ++k) {
          const float keep = up ? v[k + X] : v[k];
          const float send = up ? v[k] : v[k + X];
          v[k] = keep + __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(send), (X << 10) | 0x1f));
        }
      };
      level(std::integral_constant<int, 16>{});
      level(std::integral_constant<int, 8>{});
      level(std::integral_constant<int, 4>{});
      level(std::integral_constant<int, 2>{});
      level(std::integral_constant<int, 1>{});
      const int idx = lane & 31, j = idx >> 4, r = idx & 15;
      unsafeAtomicAdd(out + (NT * wave + j) * 32 + 4 * (lane >> 5) + 8 * (r >> 2) + (r & 3), v[0]);
    };
    // g <- the gradient of the LayerNorm's INPUT z, given g = the gradient of its output; z is overwritten (x hat)
    auto layernorm_bwd = [&](auto &z, auto &g, const float *gamma, float eps, float *dgb) {
      float sum = 0.f;
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) sum += z[0][j][r];
      sum += __shfl_xor(sum, 32, 64);
      if (lane < 32) stat[wave * BM + lane] = sum;
      __syncthreads();
      float mean = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) mean += stat[w * BM + (lane & 31)];
      mean *= (1.0f / kChainC);
      float ss = 0.f;
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float d = z[0][j][r] - mean;
          ss = fmaf(d, d, ss);
        }
      ss += __shfl_xor(ss, 32, 64);
      if (lane < 32) stat2[wave * BM + lane] = ss;
      __syncthreads();
      float var = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) var += stat2[w * BM + (lane & 31)];
      const float rstd = rsqrtf(var * (1.0f / kChainC) + eps);
      // x hat; the column sums of g * xhat and of g
      float tmp[32];
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          z[0][j][r] = (z[0][j][r] - mean) * rstd;
          tmp[j * 16 + r] = g[0][j][r] * z[0][j][r];
        }
      colsum_add(tmp, dgb);
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) tmp[j * 16 + r] = g[0][j][r];
      colsum_add(tmp, dgb + kChainC);
      // t = g * gamma; s1 = mean_c t, s2 = mean_c (t * xhat)
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float4 ga = *reinterpret_cast<const float4 *>(gamma + ncol(j) + 8 * q);
          g[0][j][4 * q] *= ga.x; g[0][j][4 * q + 1] *= ga.y; g[0][j][4 * q + 2] *= ga.z; g[0][j][4 * q + 3] *= ga.w;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            s1 += g[0][j][4 * q + e];
            s2 = fmaf(g[0][j][4 * q + e], z[0][j][4 * q + e], s2);
          }
        }
      s1 += __shfl_xor(s1, 32, 64);
      s2 += __shfl_xor(s2, 32, 64);
      __syncthreads();                         // (every wavefront has read the statistics above)
      if (lane < 32) {
        stat[wave * BM + lane] = s1;
        stat2[wave * BM + lane] = s2;
      }
      __syncthreads();
      s1 = 0.f; s2 = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) {
        s1 += stat[w * BM + (lane & 31)];
        s2 += stat2[w * BM + (lane & 31)];
      }
      s1 *= (1.0f / kChainC);
      s2 *= (1.0f / kChainC);
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) g[0][j][r] = rstd * (g[0][j][r] - s1 - z[0][j][r] * s2);
      __syncthreads();                         // `stat` / `stat2` may be written again
    };

    if constexpr (MODE == 3) {
      // ---------------------------------------------------------------- MODE 3: the backward of MODE 1
      //   gx  = gp W1 (+ the gradient that reached x directly)      gp (M, N2): N2 / 256 panel passes, K = N2
      //   gz0 = LN0'(z0; gx)                                        column sums -> grad gamma0 / beta0
      //   din = gz0 W0
      // bw_gy = gp (row stride bw_ld_gy), bw_z1 = the direct gradient of x (M, 256) or nullptr, w1 = image of W1^T
      auto fetch_plain = [&](const float *src, long ld, unsigned char *buf) {     // 32 rows x 256 floats -> planes
        const int d_rl = lane >> 3, d_cc = lane & 7;
        const int row = panel_row_of(wave * PPW / 4, d_rl);
        static_assert(PPW == 4, "one row block (4 line pairs) per wavefront");
        const int cx = d_cc ^ (row & 7);
        long gm = m0 + row;
        if (gm >= a.M) gm = a.M - 1;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
          const float *sp = src + gm * ld + (2 * p) * 32 + cx * 4;
          unsigned char *dst = buf + (wave * 4 + p) * 2048;
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(sp),
                                           (__attribute__((address_space(3))) void *)(dst), 16, 0, 0);
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(sp + 32),
                                           (__attribute__((address_space(3))) void *)(dst + 1024), 16, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int p = 0; p < 4; ++p) {
          unsigned char *slot = buf + (wave * 4 + p) * 2048 + lane * 16;
          const float4 va = *reinterpret_cast<const float4 *>(slot);
          const float4 vb = *reinterpret_cast<const float4 *>(slot + 1024);
          uint4 hi, lo;
          lin_split8<LO>(va, vb, hi, lo);
          *reinterpret_cast<uint4 *>(slot) = hi;
          if (LO) *reinterpret_cast<uint4 *>(slot + 1024) = lo;
        }
      };
      const int nck = a.N2 / 256, nstep1 = a.N2 / 16;
      zero(acc);
#pragma unroll 1
      for (int c = 0; c < nck; ++c) {
        unsigned char *buf = (c & 1) ? buf1 : buf0;
        wprefetch(r1, wave, nstep1, c * 16);
        fetch_plain(a.bw_gy + c * 256, a.bw_ld_gy, buf);
        __syncthreads();                       // the pass's planes are complete (and, c >= 2: the buffer was free)
        gemm16(acc, buf, r1, wave, nstep1, c * 16);
        if (c + 2 < nck) __syncthreads();      // pass c + 2 rewrites this buffer
      }
      if (a.bw_z1) {                           // the gradient x received as the next attention's residual
        load_tile(xk, a.bw_z1, kChainC, NT * wave * 32, true);
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[0][j][r] += xk[0][j][r];
      }
      if (!row_ok) {                           // rows past M: no contribution to the column sums
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[0][j][r] = 0.f;
      }
      load_tile(acc2, a.bw_z0, kChainC, NT * wave * 32, false);
      layernorm_bwd(acc2, acc, c_g0, a.eps0, a.bw_dgb0);
      wprefetch(r0, wave, 16, 0);
      store_tile(acc, a.bw_gz0, kChainC, NT * wave * 32);
      if (a.dk0) {                             // through the dropout on the projection's output
        scale_tile(acc, a.dk0, kChainC, NT * wave * 32);
        store_tile(acc, a.bw_gzp, kChainC, NT * wave * 32);
      }
      unsigned char *bufz = (nck & 1) ? buf1 : buf0;      // the buffer the last pass did not read
      to_planes(acc, bufz);
      __syncthreads();
      zero(acc2);
      gemm16(acc2, bufz, r0, wave, 16, 0);
      store_tile(acc2, a.bw_din, kChainC, NT * wave * 32);
      return;
    }
    load_tile(acc, a.bw_gy, a.bw_ld_gy, NT * wave * 32, true);
    load_tile(acc2, a.bw_z1, kChainC, NT * wave * 32, false);
    __syncthreads();                           // the per-column constants are in LDS
    layernorm_bwd(acc2, acc, c_g1, a.eps1, a.bw_dgb1);
    wprefetch(r1, wave, 16, 0);
#pragma unroll
    for (int j = 0; j < NT; ++j) xk[0][j] = acc[0][j];
    if (a.dk1) scale_tile(acc, a.dk1, kChainC, NT * wave * 32);     // the FFN's output dropout: df = gz1 * dk1 (x keeps gz1)
    store_tile(acc, a.bw_gz1, kChainC, NT * wave * 32);
    to_planes(acc, buf1);
    __syncthreads();
    zero(acc2);
#pragma unroll 1
    for (int half = 0; half < 2; ++half) {
      zero(acc);
      gemm16(acc, buf1, r1, half * NW + wave, 16, 0);
      wprefetch(r2, wave, 32, half * 16);
      {                                        // ReLU backward: the forward's hidden activations are the mask
        const float *hrow = a.bw_h + mclamp * kChainF + half * 256 + NT * wave * 32;
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const float4 hv = *reinterpret_cast<const float4 *>(hrow + j * 32 + 4 * (lane >> 5) + 8 * g);
            const float hs = a.bw_hscale;
            acc[0][j][4 * g] = hv.x > 0.f ? acc[0][j][4 * g] * hs : 0.f;
            acc[0][j][4 * g + 1] = hv.y > 0.f ? acc[0][j][4 * g + 1] * hs : 0.f;
            acc[0][j][4 * g + 2] = hv.z > 0.f ? acc[0][j][4 * g + 2] * hs : 0.f;
            acc[0][j][4 * g + 3] = hv.w > 0.f ? acc[0][j][4 * g + 3] * hs : 0.f;
          }
      }
      store_tile(acc, a.bw_gh, kChainF, half * 256 + NT * wave * 32);
      to_planes(acc, buf0);
      __syncthreads();
      gemm16(acc2, buf0, r2, wave, 32, half * 16);
      if (half == 0) wprefetch(r1, NW + wave, 16, 0);
      __syncthreads();
    }
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc2[0][j][r] += xk[0][j][r];
    load_tile(acc, a.bw_z0, kChainC, NT * wave * 32, false);
    layernorm_bwd(acc, acc2, c_g0, a.eps0, a.bw_dgb0);
    wprefetch(r0, wave, 16, 0);
    store_tile(acc2, a.bw_gz0, kChainC, NT * wave * 32);
    if (a.dk0) {                               // through the attention's dropout
      scale_tile(acc2, a.dk0, kChainC, NT * wave * 32);
      store_tile(acc2, a.bw_gzp, kChainC, NT * wave * 32);
    }
    to_planes(acc2, buf1);
    __syncthreads();
    zero(acc);
    gemm16(acc, buf1, r0, wave, 16, 0);
    store_tile(acc, a.bw_din, kChainC, NT * wave * 32);
    return;
  }
  // ------------------------------------------------------------------ stage 0: fetch + split the A panel (buffer 0)
  {
    const int d_rl = lane >> 3, d_cc = lane & 7;
    const int row = panel_row_of(wave * PPW / 4, d_rl);   // PPW / 4 row blocks per wavefront (here: one)
    static_assert(PPW == 4, "one row block (4 line pairs) per wavefront");
    const int cx = d_cc ^ (row & 7);
    long gm = m0 + row;
    if (gm >= a.M) gm = a.M - 1;               // clamped rows are computed and never stored
    long srow = gm, arow = gm;
    int g0 = 0, g1 = -1, g2 = -1;
    float gs = 1.f;
    const int gst = a.gstride > 2 ? a.gstride : 2;
    if (PRE == 2) {
      g0 = a.gidx[gm * gst];
      g1 = a.gidx[gm * gst + 1];
      if (gst > 2) g2 = a.gidx[gm * gst + 2];
      gs = a.gscale[gm];
      srow = g0 < 0 ? 0 : g0;
      arow = g1 < 0 ? 0 : g1;
    }
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const float *src = a.rows + srow * a.ld_rows + (2 * p) * 32 + cx * 4;
      unsigned char *dst = buf0 + (wave * 4 + p) * 2048;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src),
                                       (__attribute__((address_space(3))) void *)(dst), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + 32),
                                       (__attribute__((address_space(3))) void *)(dst + 1024), 16, 0, 0);
    }
    float4 ad[PRE == 2 ? 4 : 1][2];
    if (PRE == 2) {
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        const float *s1 = a.rows + arow * a.ld_rows + (2 * p) * 32 + cx * 4;
        ad[PRE == 2 ? p : 0][0] = *reinterpret_cast<const float4 *>(s1);
        ad[PRE == 2 ? p : 0][1] = *reinterpret_cast<const float4 *>(s1 + 32);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // my own DMA slots have landed
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      unsigned char *slot = buf0 + (wave * 4 + p) * 2048 + lane * 16;
      float4 va = *reinterpret_cast<const float4 *>(slot);
      float4 vb = *reinterpret_cast<const float4 *>(slot + 1024);
      if (PRE == 2) {
        if (g2 >= 0) {
          // rows of a third.. camera (rare): added to the first row's values in column order, as the stand-alone fold launch
          // (frame_plan.h: fold_extra_rows_kernel) adds them — the same sums, without that launch
          for (int j = 2; j < gst; ++j) {
            const int r = a.gidx[gm * gst + j];
            if (r < 0) break;
            const float *sx = a.rows + static_cast<long>(r) * a.ld_rows + (2 * p) * 32 + cx * 4;
            const float4 xa = *reinterpret_cast<const float4 *>(sx), xb = *reinterpret_cast<const float4 *>(sx + 32);
            va.x += xa.x; va.y += xa.y; va.z += xa.z; va.w += xa.w;
            vb.x += xb.x; vb.y += xb.y; vb.z += xb.z; vb.w += xb.w;
          }
        }
        va = panel_gsum(va, g0 >= 0, ad[PRE == 2 ? p : 0][0], g1 >= 0, gs);
        vb = panel_gsum(vb, g0 >= 0, ad[PRE == 2 ? p : 0][1], g1 >= 0, gs);
      }
      uint4 hi, lo;
      lin_split8<LO>(va, vb, hi, lo);
      *reinterpret_cast<uint4 *>(slot) = hi;
      if (LO) *reinterpret_cast<uint4 *>(slot + 1024) = lo;
    }
  }
  __syncthreads();
  CHAIN_STAMP(0);                              // panel fetch + split

  // ------------------------------------------------------------------ stage 0: x = LN0(A W0^T + b0 + res)
  zero(acc);
  gemm16(acc, buf0, r0, wave, 16, 0);
  CHAIN_STAMP(1);                              // GEMM 0
  wprefetch(r1, wave, 16, 0);
  add_cols(acc, c_b0, 0);
  if constexpr (DROP) { if (a.dk0) scale_tile(acc, a.dk0, kChainC, NT * wave * 32); }
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        acc[i][j][4 * g] += rs[i][j][g].x; acc[i][j][4 * g + 1] += rs[i][j][g].y;
        acc[i][j][4 * g + 2] += rs[i][j][g].z; acc[i][j][4 * g + 3] += rs[i][j][g].w;
      }
  if constexpr (SAVE) store_tile(acc, a.sv_z0, kChainC, NT * wave * 32);
  layernorm(acc, c_g0, c_be0, a.eps0);
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) xk[i][j] = acc[i][j];
  if constexpr (SAVE && MODE == 0) store_tile(xk, a.sv_x, kChainC, NT * wave * 32);
  to_planes(xk, buf1);
  __syncthreads();                             // x planes complete (and every wavefront is done with buffer 0)
  CHAIN_STAMP(2);                              // bias + residual + LayerNorm 0 + plane write

  if constexpr (MODE == 1) {
    // x is the next attention's residual: store it, then project it (from its planes) tile by tile
    store_tile(xk, a.y, a.ld_y, NT * wave * 32);
    const int ntile = a.N2 / (32 * NT);        // (N2 a multiple of 32 NT: checked by the launcher)
#pragma unroll 1
    for (int t = wave; t < ntile; t += NW) {
      zero(acc);
      gemm16(acc, buf1, r1, t, 16, 0);
      if (t + NW < ntile) wprefetch(r1, t + NW, 16, 0);
      // (bias of tile t: columns 32 NT t ..; add_cols indexes by this wavefront's own tile, so shift)
      add_cols(acc, c_b1, (t - wave) * 32 * NT);
      store_tile(acc, a.y2, a.ld_y2, t * 32 * NT);
    }
    return;
  } else {
    // ---------------------------------------------------------------- FFN, the hidden layer in two halves of 256
    zero(acc2);
#pragma unroll 1
    for (int half = 0; half < 2; ++half) {
      zero(acc);
      gemm16(acc, buf1, r1, half * NW + wave, 16, 0);
      CHAIN_STAMP(3 + 3 * half);               // GEMM 1 (half)
      wprefetch(r2, wave, 32, half * 16);
      add_cols(acc, c_b1, half * 256);
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[i][j][r] = acc[i][j][r] < 0.f ? 0.f : acc[i][j][r];     // NaN stays NaN, as torch.relu
      if constexpr (DROP) { if (a.dkh) scale_tile(acc, a.dkh, kChainF, half * 256 + NT * wave * 32); }
      if constexpr (SAVE) store_tile(acc, a.sv_h, kChainF, half * 256 + NT * wave * 32);
      to_planes(acc, buf0);
      __syncthreads();                         // this half of the hidden layer is complete
      CHAIN_STAMP(4 + 3 * half);               // bias + ReLU + plane write
      gemm16(acc2, buf0, r2, wave, 32, half * 16);
      if (half == 0) wprefetch(r1, NW + wave, 16, 0);
      __syncthreads();                         // ... and consumed: buffer 0 may be rewritten
      CHAIN_STAMP(5 + 3 * half);               // GEMM 2 (half)
    }

    // ---------------------------------------------------------------- TP: the `first` panel travels under LayerNorm1
    float4 ps[TP ? MT : 1][TP ? NT : 1][4];
    __amdgpu_buffer_rsrc_t r3 = r0;
    if constexpr (TP) {
      const unsigned w3b = static_cast<unsigned>((a.N3 + 63) / 64 * 2) * 32 * 2 * 1024;
      r3 = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t *>(a.w3), 0, static_cast<int>(w3b), 0x00020000);
      // (every wavefront is past both buffers: GEMM 1 of the second half read buffer 1, the barrier above closed GEMM 2)
      const int d_rl = lane >> 3, d_cc = lane & 7;
      const int row = panel_row_of(wave * PPW / 4, d_rl);
      const int cx = d_cc ^ (row & 7);
      long gm = m0 + row;
      if (gm >= a.M) gm = a.M - 1;
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        const float *src = a.tp_first + gm * a.ld_tp_first + (2 * p) * 32 + cx * 4;
        unsigned char *dst = buf1 + (wave * 4 + p) * 2048;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src),
                                         (__attribute__((address_space(3))) void *)(dst), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + 32),
                                         (__attribute__((address_space(3))) void *)(dst + 1024), 16, 0, 0);
      }
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        const float *prow = a.tp_pos ? a.tp_pos + (mrow[i] < a.M ? mrow[i] : a.M - 1) * a.ld_tp_pos : nullptr;
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
          for (int g = 0; g < 4; ++g)
            ps[i][j][g] = prow ? *reinterpret_cast<const float4 *>(prow + ncol(j) + 8 * g) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }

    // ---------------------------------------------------------------- y = LN1(x + ffn(x))
    add_cols(acc2, c_b2, 0);
    if constexpr (DROP) { if (a.dk1) scale_tile(acc2, a.dk1, kChainC, NT * wave * 32); }
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc2[i][j][r] += xk[i][j][r];
    if constexpr (SAVE) store_tile(acc2, a.sv_z1, kChainC, NT * wave * 32);
    layernorm(acc2, c_g1, c_be1, a.eps1);
    CHAIN_STAMP(9);                            // bias + residual + LayerNorm 1
    if constexpr (TP) {
      // ---------------------------------------------------------------- p = [first | y + pos] W3^T + b3
      const int ntile = a.N3 / (32 * NT);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // my own DMA slots (and the pos rows) have landed
#pragma unroll
      for (int p = 0; p < 4; ++p) {                         // split `first` in place
        unsigned char *slot = buf1 + (wave * 4 + p) * 2048 + lane * 16;
        const float4 va = *reinterpret_cast<const float4 *>(slot);
        const float4 vb = *reinterpret_cast<const float4 *>(slot + 1024);
        uint4 hi, lo;
        lin_split8<LO>(va, vb, hi, lo);
        *reinterpret_cast<uint4 *>(slot) = hi;
        if (LO) *reinterpret_cast<uint4 *>(slot + 1024) = lo;
      }
      // (the first weight fragments are requested BEFORE y's stores: the wavefront's memory counter retires in order)
      if (wave < ntile) wprefetch(r3, wave, 32, 0);
      store_tile(acc2, a.y, a.ld_y, NT * wave * 32);
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            acc2[i][j][4 * g] += ps[i][j][g].x; acc2[i][j][4 * g + 1] += ps[i][j][g].y;
            acc2[i][j][4 * g + 2] += ps[i][j][g].z; acc2[i][j][4 * g + 3] += ps[i][j][g].w;
          }
      to_planes(acc2, buf0);
      __syncthreads();                         // both halves of the projection's input are planes
      CHAIN_STAMP(10);
      // c += [planes(buf1) | planes(buf0)] x W3[column tile]^T over the 32 k16 steps of K = 512 (gemm16 twice over, one
      // weight ring: no L2 round trip between the halves)
      auto gemm32 = [&](auto &c, int tile) {
        lin_bf16x8 af[2][MT][NPL];
        auto aload = [&](int set, int s) {
          const unsigned char *buf = s < 16 ? buf1 : buf0;
          const unsigned base = f_addr[s & 3] + ((s & 15) >> 2) * 2048;
#pragma unroll
          for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl)
              af[set][i][pl] = *reinterpret_cast<const lin_bf16x8 *>(buf + base + i * (4 * 4 * 2048) + pl * 1024);
        };
        aload(0, 0);
#pragma unroll
        for (int s = 0; s < 32; ++s) {
          if (s + WD < 32) wload(r3, tile, 32, (s + WD) % (WD + 1), s + WD);
          if (s + 1 < 32) aload((s + 1) & 1, s + 1);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int i = 0; i < MT; ++i) {
              if constexpr (LO) {
                c[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[s % (WD + 1)][j][0], af[s & 1][i][1], c[i][j], 0, 0, 0);
                c[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[s % (WD + 1)][j][1], af[s & 1][i][0], c[i][j], 0, 0, 0);
              }
              c[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[s % (WD + 1)][j][0], af[s & 1][i][0], c[i][j], 0, 0, 0);
            }
          __builtin_amdgcn_sched_barrier(0);
        }
      };
#pragma unroll 1
      for (int t = wave; t < ntile; t += NW) {
        zero(acc);
        gemm32(acc, t);
        if (t + NW < ntile) wprefetch(r3, t + NW, 32, 0);
        add_cols(acc, c_b1 + kChainF, (t - wave) * 32 * NT);
        store_tile(acc, a.y3, a.ld_y3, t * 32 * NT);
      }
      return;
    }
    store_tile(acc2, a.y, a.ld_y, NT * wave * 32);
    CHAIN_STAMP(10);                           // stores issued
  }
}

}  // namespace bevmsda
