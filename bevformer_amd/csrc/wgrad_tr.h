// Weight / bias gradients, second generation (round 4): the TN GEMM  grad_W[n, k] += sum_m G[m, n] X[m, k]  with its
// operands split to bf16 planes ONCE per workgroup and the MFMA fragments fetched with gfx950's transposing LDS read.
//
// The MFMA wants, per lane, 8 consecutive elements ALONG THE REDUCTION (the rows m) of one column — a transposed access
// of the row-major operands.  The first kernel (wgrad_mfma.h) keeps the fp32 tiles in LDS and gathers every fragment
// with 8 ds_read_b32 + an in-register split: 64 LDS reads and ~150 VALU instructions per wavefront and 32-row chunk,
// each element split once per CONSUMING wavefront (twice) — 16-19 % MFMA-pipe utilisation (r2n / r3 counters).
// Here a thread loads its 16-byte pieces of both operand tiles from global memory (next chunk's loads in flight under
// this chunk's MFMAs), splits each value once into hi = bf16(x), lo = bf16(x - hi), and writes the [hi | lo] planes
// row-major into LDS; `ds_read_b64_tr_b16` then returns, for a 16-lane group whose lanes point at the sixteen 8-byte
// pieces of a 4-row x 16-column block, COLUMN i of that block to lane i — four consecutive rows of one column, i.e.
// half an MFMA fragment per instruction (semantics measured with tools/probes/tr_read_probe.hip: lane i of a group
// receives, as element j, element (i & 3) of the piece addressed by lane (i >> 2) + 4 j).  32 LDS reads per wavefront
// and chunk, no VALU in the fragment path.
//
// LDS image: planes [G hi | G lo | X hi | X lo], each 32 rows x 160 bf16 (128 columns + 32 pad): the 320-byte row stride
// is 80 banks = 16 (mod 64), which puts the 32 pieces of a half-wave's transposing read — 4 rows x (4 pieces x 2
// sixteen-column blocks) — on 32 distinct bank pairs (tests/test_linear_layout_model.py replays the arithmetic).
// One stage (40 KB): two workgroups per CU; the next TWO chunks travel in registers (2 x 8 float4 per thread), two
// LDS-only barriers per chunk.  Epilogue as the first kernel (fp32 atomics over row slices; the caller zeroes grad_W / grad_b).
#pragma once
#include "wgrad_mfma.h"

namespace bevmsda {

constexpr int kWtStride = 160;                       // bf16 elements per LDS row
constexpr int kWtRowBytes = kWtStride * 2;           // 320
constexpr int kWtPlane = 32 * kWtRowBytes;           // 10,240 bytes: one plane of a 32-row chunk
typedef short wt_i16x4 __attribute__((ext_vector_type(4)));
typedef short wt_i16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void wt_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// 8 consecutive reduction elements of one column as an MFMA operand: rows r .. r+3 and r+4 .. r+7 of the block at `p`
__device__ __forceinline__ lin_bf16x8 wt_fragment(const unsigned char *p) {
  const wt_i16x4 lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (__attribute__((address_space(3))) wt_i16x4 *)(const_cast<unsigned char *>(p)));
  const wt_i16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (__attribute__((address_space(3))) wt_i16x4 *)(const_cast<unsigned char *>(p + 4 * kWtRowBytes)));
  const wt_i16x8 v = __builtin_shufflevector(lo4, hi4, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(lin_bf16x8, v);
}

template <int NPROD>
__device__ __forceinline__ void wgrad_tile_tr(const WgradArgs &a, int tile, int slice, unsigned char *lds) {
  constexpr bool LO = NPROD == 3;
  constexpr int NPL = LO ? 2 : 1;
  unsigned char *const pg = lds, *const px = lds + NPL * kWtPlane;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wn = wave >> 1, wk = wave & 1;
  const int tn = tile / a.tiles_k, tk = tile - tn * a.tiles_k;
  const int n0 = tn * 128, k0 = tk * 128;
  const long m_begin = static_cast<long>(slice) * a.rows_per_block;
  if (m_begin >= a.M) return;
  const long m_end = m_begin + a.rows_per_block < a.M ? m_begin + a.rows_per_block : a.M;
  const int nchunks = static_cast<int>((m_end - m_begin + 31) / 32);

  // my pieces of a chunk tile (32 rows x 32 pieces of 4 floats): column piece tid & 31 of rows (tid >> 5) + 8 i
  const int prow0 = tid >> 5, c4 = (tid & 31) * 4;
  // columns past the matrix are clamped (their products land in rows / columns that are never stored)
  const int gcol = n0 + c4 < a.N ? n0 + c4 : (a.N - 4);
  const int xcol = k0 + c4 < a.K ? k0 + c4 : (a.K - 4);
  const bool want_bias = a.gb != nullptr && tk == 0;
  // two register sets: the loads of chunk c + 2 are issued while chunk c is being consumed — one chunk's MFMAs
  // (~0.3 us) are far shorter than a trip to L2 / HBM (~2 us), and with a single set every chunk waited for its loads
  // (measured: 2.3 us per chunk)
  float4 rgA[4], rxA[4], rgB[4], rxB[4];
  float4 csum = make_float4(0.f, 0.f, 0.f, 0.f);     // bias gradient: my four columns over my rows
  auto load = [&](int c, float4 (&rg)[4], float4 (&rx)[4]) {
    const long mb = m_begin + static_cast<long>(c) * 32;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      long m = mb + prow0 + 8 * i;
      if (m >= a.M) m = a.M - 1;                      // rows past the end: a valid row, zeroed at the split
      rg[i] = *reinterpret_cast<const float4 *>(a.g + m * a.ldg + gcol);
      rx[i] = *reinterpret_cast<const float4 *>(a.x + m * a.ldx + xcol);
    }
  };
  auto split_store = [&](unsigned char *plane, int row, float4 v) {
    uint2 hi;
    hi.x = lin_pack2(v.x, v.y);
    hi.y = lin_pack2(v.z, v.w);
    unsigned char *dst = plane + row * kWtRowBytes + c4 * 2;
    *reinterpret_cast<uint2 *>(dst) = hi;
    if (LO) {
      uint2 lo;
      lo.x = lin_pack2(v.x - __uint_as_float(hi.x << 16), v.y - __uint_as_float(hi.x & 0xffff0000u));
      lo.y = lin_pack2(v.z - __uint_as_float(hi.y << 16), v.w - __uint_as_float(hi.y & 0xffff0000u));
      *reinterpret_cast<uint2 *>(dst + kWtPlane) = lo;
    }
  };
  auto store = [&](int c, const float4 (&rg)[4], const float4 (&rx)[4]) {
    const long mrem = m_end - (m_begin + static_cast<long>(c) * 32);      // valid rows of this chunk (>= 1)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = prow0 + 8 * i;
      const bool ok = row < mrem;
      const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
      const float4 gv = ok ? rg[i] : z, xv = ok ? rx[i] : z;
      if (want_bias) csum = lin_add4(csum, gv);
      split_store(pg, row, gv);
      split_store(px, row, xv);
    }
  };

  // fragment addresses: lane L = (g1 = L >> 5, g0 = (L >> 4) & 1, s = L & 15) points at the piece (row 8 g1 + (s >> 2),
  // columns 16 g0 + 4 (s & 3) .. + 3) of the 32-column tile; the read returns rows 8 g1 + 0..3 of column L & 31
  const int s = lane & 15, g0 = (lane >> 4) & 1, g1 = lane >> 5;
  const int frow = 8 * g1 + (s >> 2), fcolb = (16 * g0 + 4 * (s & 3)) * 2;
  unsigned fa[2], fb[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    fa[t] = static_cast<unsigned>(frow * kWtRowBytes + (wn * 64 + t * 32) * 2 + fcolb);
    fb[t] = static_cast<unsigned>(frow * kWtRowBytes + (wk * 64 + t * 32) * 2 + fcolb);
  }

  lin_f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  auto mma = [&]() {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      lin_bf16x8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        ah[t] = wt_fragment(pg + fa[t] + ks * 16 * kWtRowBytes);
        bh[t] = wt_fragment(px + fb[t] + ks * 16 * kWtRowBytes);
        if (LO) {
          al[t] = wt_fragment(pg + kWtPlane + fa[t] + ks * 16 * kWtRowBytes);
          bl[t] = wt_fragment(px + kWtPlane + fb[t] + ks * 16 * kWtRowBytes);
        }
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
  };
  auto step = [&](int c, float4 (&rg)[4], float4 (&rx)[4]) {
    wt_lds_barrier();                 // every wavefront is done with the previous chunk's planes
    store(c, rg, rx);
    if (c + 2 < nchunks) load(c + 2, rg, rx);       // two chunks ahead, into the set just consumed
    wt_lds_barrier();                 // planes complete
    mma();
  };
  load(0, rgA, rxA);
  if (nchunks > 1) load(1, rgB, rxB);
  for (int c = 0; c < nchunks; c += 2) {
    step(c, rgA, rxA);
    if (c + 1 < nchunks) step(c + 1, rgB, rxB);
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
  if (want_bias) {
    // the 8 threads that share a column piece (tid & 31) combine through LDS; one atomic per column and workgroup
    wt_lds_barrier();
    float *red = reinterpret_cast<float *>(lds);                      // [8][128]
    *reinterpret_cast<float4 *>(red + prow0 * 128 + c4) = csum;
    wt_lds_barrier();
    if (tid < 128) {
      float sum = 0.f;
#pragma unroll
      for (int r = 0; r < 8; ++r) sum += red[r * 128 + tid];
      const int n = n0 + tid;
      if (n < a.N) unsafeAtomicAdd(a.gb + n, sum);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Third shape (round 6): 256 x 128 output tiles on EIGHT wavefronts with TWO LDS stages.
//
// The 128 x 128 / four-wavefront form above is issue-bound, not MFMA- or HBM-bound (27 % MFMA busy, 2.7 TB/s): per 32-row
// chunk a wavefront issues 24 MFMAs (~770 clk) against ~1,000 clk of VALU for the hi / lo split of its 32 elements per
// thread, and the chunk is bracketed by two barriers because the single LDS stage serialises "store planes" and "read
// fragments" inside a workgroup.  Here a workgroup's 512 threads stage a 32 x 256 strip of G and a 32 x 128 strip of X
// (24 elements per thread: the G strip is split once for two 128-column X tiles' worth of MFMAs), the planes of chunk
// c + 1 are written into the OTHER stage while chunk c's fragments are read — one barrier per chunk — and the loads run
// three chunks ahead.  One workgroup (112 KB of LDS) per CU = the same eight wavefronts per CU as two of the old ones.
// G rows keep the bank property of the 320-byte stride: 576 bytes = 144 banks = 16 (mod 64).
constexpr int kW2GRowBytes = 576;                    // (256 + 32 pad) bf16
constexpr int kW2GPlane = 32 * kW2GRowBytes;         // 18,432 bytes
constexpr int kW2XPlane = kWtPlane;                  // 10,240 bytes (320-byte rows as above)

template <int ROWBYTES>
__device__ __forceinline__ lin_bf16x8 wt_fragment_rb(const unsigned char *p) {
  const wt_i16x4 lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (__attribute__((address_space(3))) wt_i16x4 *)(const_cast<unsigned char *>(p)));
  const wt_i16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (__attribute__((address_space(3))) wt_i16x4 *)(const_cast<unsigned char *>(p + 4 * ROWBYTES)));
  const wt_i16x8 v = __builtin_shufflevector(lo4, hi4, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(lin_bf16x8, v);
}

template <int NPROD>
__device__ __forceinline__ void wgrad_tile_tr2(const WgradArgs &a, int tile, int slice, unsigned char *lds) {
  constexpr bool LO = NPROD == 3;
  constexpr int NPL = LO ? 2 : 1;
  constexpr int STAGE = NPL * (kW2GPlane + kW2XPlane);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wn = wave >> 1, wk = wave & 1;                  // 4 x 2 wavefronts of 64 x 64 over the 256 x 128 tile
  const int tiles_k = (a.K + 127) / 128;
  const int tn = tile / tiles_k, tk = tile - tn * tiles_k;
  const int n0 = tn * 256, k0 = tk * 128;
  const long m_begin = static_cast<long>(slice) * a.rows_per_block;
  if (m_begin >= a.M) return;
  const long m_end = m_begin + a.rows_per_block < a.M ? m_begin + a.rows_per_block : a.M;
  const int nchunks = static_cast<int>((m_end - m_begin + 31) / 32);

  // my pieces of a chunk: G column piece tid & 63 of rows (tid >> 6) + 8 i (i < 4); X column piece tid & 31 of rows (tid >> 5) + 16 i (i < 2)
  const int grow0 = tid >> 6, gc4 = (tid & 63) * 4;
  const int xrow0 = tid >> 5, xc4 = (tid & 31) * 4;
  const int gcol = n0 + gc4 < a.N ? n0 + gc4 : (a.N - 4);     // (columns past the matrix: clamped, never stored)
  const int xcol = k0 + xc4 < a.K ? k0 + xc4 : (a.K - 4);
  const bool want_bias = a.gb != nullptr && tk == 0;
  float4 rgA[4], rxA[2], rgB[4], rxB[2];
  float4 csum = make_float4(0.f, 0.f, 0.f, 0.f);
  auto load = [&](int c, float4 (&rg)[4], float4 (&rx)[2]) {
    const long mb = m_begin + static_cast<long>(c) * 32;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      long m = mb + grow0 + 8 * i;
      if (m >= a.M) m = a.M - 1;
      rg[i] = *reinterpret_cast<const float4 *>(a.g + m * a.ldg + gcol);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      long m = mb + xrow0 + 16 * i;
      if (m >= a.M) m = a.M - 1;
      rx[i] = *reinterpret_cast<const float4 *>(a.x + m * a.ldx + xcol);
    }
  };
  auto split_store = [&](unsigned char *dst, int plane_bytes, float4 v) {
    uint2 hi;
    hi.x = lin_pack2(v.x, v.y);
    hi.y = lin_pack2(v.z, v.w);
    *reinterpret_cast<uint2 *>(dst) = hi;
    if (LO) {
      uint2 lo;
      lo.x = lin_pack2(v.x - __uint_as_float(hi.x << 16), v.y - __uint_as_float(hi.x & 0xffff0000u));
      lo.y = lin_pack2(v.z - __uint_as_float(hi.y << 16), v.w - __uint_as_float(hi.y & 0xffff0000u));
      *reinterpret_cast<uint2 *>(dst + plane_bytes) = lo;
    }
  };
  auto store = [&](int c, const float4 (&rg)[4], const float4 (&rx)[2]) {
    unsigned char *const pg = lds + (c & 1) * STAGE, *const px = pg + NPL * kW2GPlane;
    const long mrem = m_end - (m_begin + static_cast<long>(c) * 32);      // valid rows of this chunk (>= 1)
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = grow0 + 8 * i;
      const float4 gv = row < mrem ? rg[i] : z;
      if (want_bias) csum = lin_add4(csum, gv);
      split_store(pg + row * kW2GRowBytes + gc4 * 2, kW2GPlane, gv);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int row = xrow0 + 16 * i;
      split_store(px + row * kWtRowBytes + xc4 * 2, kW2XPlane, row < mrem ? rx[i] : z);
    }
  };

  const int s = lane & 15, g0 = (lane >> 4) & 1, g1 = lane >> 5;
  const int frow = 8 * g1 + (s >> 2), fcolb = (16 * g0 + 4 * (s & 3)) * 2;
  unsigned fa[2], fb[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    fa[t] = static_cast<unsigned>(frow * kW2GRowBytes + (wn * 64 + t * 32) * 2 + fcolb);
    fb[t] = static_cast<unsigned>(frow * kWtRowBytes + (wk * 64 + t * 32) * 2 + fcolb);
  }

  lin_f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  auto mma = [&](int c) {
    const unsigned char *const pg = lds + (c & 1) * STAGE, *const px = pg + NPL * kW2GPlane;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      lin_bf16x8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        ah[t] = wt_fragment_rb<kW2GRowBytes>(pg + fa[t] + ks * 16 * kW2GRowBytes);
        bh[t] = wt_fragment_rb<kWtRowBytes>(px + fb[t] + ks * 16 * kWtRowBytes);
        if (LO) {
          al[t] = wt_fragment_rb<kW2GRowBytes>(pg + kW2GPlane + fa[t] + ks * 16 * kW2GRowBytes);
          bl[t] = wt_fragment_rb<kWtRowBytes>(px + kW2XPlane + fb[t] + ks * 16 * kWtRowBytes);
        }
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          if (LO) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
          }
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
        }
    }
  };
  // iteration c: the planes of chunk c + 1 go into the other stage (its last readers passed the barrier that ended
  // iteration c - 1), the loads of chunk c + 3 go into the register set just stored, chunk c is multiplied
  auto step = [&](int c, float4 (&rg)[4], float4 (&rx)[2]) {     // (rg, rx) hold chunk c + 1
    if (c + 1 < nchunks) store(c + 1, rg, rx);
    if (c + 3 < nchunks) load(c + 3, rg, rx);
    mma(c);
    wt_lds_barrier();
  };
  load(0, rgA, rxA);
  if (nchunks > 1) load(1, rgB, rxB);
  store(0, rgA, rxA);
  if (nchunks > 2) load(2, rgA, rxA);
  wt_lds_barrier();
  for (int c = 0; c < nchunks; c += 2) {
    step(c, rgB, rxB);
    if (c + 1 < nchunks) step(c + 1, rgA, rxA);
  }

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
  if (want_bias) {
    // the 8 threads that share a G column piece (tid & 63) combine through LDS; one atomic per column and workgroup
    float *red = reinterpret_cast<float *>(lds);                      // [8][256]  (the last barrier of the loop passed)
    *reinterpret_cast<float4 *>(red + grow0 * 256 + gc4) = csum;
    wt_lds_barrier();
    if (tid < 256) {
      float sum = 0.f;
#pragma unroll
      for (int r = 0; r < 8; ++r) sum += red[r * 256 + tid];
      const int n = n0 + tid;
      if (n < a.N) unsafeAtomicAdd(a.gb + n, sum);
    }
  }
}

template <int NPROD>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2)))
wgrad_tr2_multi_kernel(const WgradMultiArgs a) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * (NPROD == 3 ? 2 : 1) * (kW2GPlane + kW2XPlane)];
  const int ntile = a.tile0[a.nprob];
  const int xcd = blockIdx.x & 7, seq = blockIdx.x >> 3;
  const int tile = seq % ntile;
  const int slice = (seq / ntile) * 8 + xcd;
  WgradArgs w = a.p[0];
  int t0 = 0;
#pragma unroll
  for (int i = 1; i < kWgMaxProblems; ++i)
    if (i < a.nprob && tile >= a.tile0[i]) {
      w = a.p[i];
      t0 = a.tile0[i];
    }
  wgrad_tile_tr2<NPROD>(w, tile - t0, slice, lds);
}

template <int NPROD>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2)))
wgrad_tr_multi_kernel(const WgradMultiArgs a) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[4 * kWtPlane];
  const int ntile = a.tile0[a.nprob];
  const int xcd = blockIdx.x & 7, seq = blockIdx.x >> 3;
  const int tile = seq % ntile;
  const int slice = (seq / ntile) * 8 + xcd;
  WgradArgs w = a.p[0];
  int t0 = 0;
#pragma unroll
  for (int i = 1; i < kWgMaxProblems; ++i)
    if (i < a.nprob && tile >= a.tile0[i]) {
      w = a.p[i];
      t0 = a.tile0[i];
    }
  wgrad_tile_tr<NPROD>(w, tile - t0, slice, lds);
}

}  // namespace bevmsda
