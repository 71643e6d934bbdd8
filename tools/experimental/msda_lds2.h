// SpatialCrossAttention sampling, third form: the two COARSE feature levels of a (camera, head) patch
// served from LDS while the two fine levels stream through the vector-memory path — both data paths
// of a CU busy at once.
//
// Why: the fused kernel (msda_d32.h) requests 6.0 GB of 128-byte taps through the
// per-CU vector L1 (64 B / clk / CU): 238 us, 77 % of that path's peak, whatever HBM does.  Half of the
// taps go to levels 2 and 3, whose footprint for 256 image-ordered rows of one (camera, head) is small:
// all 375 pixels of level 3 and a ~320-pixel box of level 2 (measured on the base rig), ~90 KB in fp32.
// An LDS read (ds_read_b128) moves 4x the bytes per clock of the L1 path and runs beside it.
//
// Structure: ONE workgroup of 512 threads per CU (the tiles take 152 KB of its LDS): 8 wavefronts with 256
// VGPRs each — with so few wavefronts the memory-level parallelism has to come from inside each of them: a
// whole fine level (32 tap requests) is in flight per wavefront.  (16 wavefronts at 128 VGPRs, the plain
// kernel's shape, do not fit this kernel's state: hipcc spills ~2 KB per lane.)  The workgroup owns `rows_per_block` consecutive rows of ONE head, split at
// camera boundaries.  Per sub-range it (1) finds the bounding box of the level-2
// taps (ds_min / ds_max), (2) copies level 3 (whole, or its box when larger than the budget) and the
// level-2 box of that head's value slice into LDS with global_load_lds, (3) runs the rows: per (row,
// head) lane group the softmax / location prologue of the fused kernel, then
//     level 0 (memory) -> level 2 (LDS) -> level 1 (memory) -> level 3 (LDS).
// A point whose 2 x 2 footprint is not wholly inside a staged box takes the memory path (rare).
// fp32, D = 32, P = 8, L = 4, pillar-anchor references (SpatialCrossAttention); anything else stays on
// msda_fused_d32_kernel.  Results equal that kernel up to the order of the level sums.
//
// MEASURED (base frame, profiles/r2): 270 us against 238 us for the plain fused kernel — opt-in only
// (BEVMSDA_SCA_LDS2=1 / ops.set_sca_lds2).  The tiles leave room for one workgroup per CU, and 8 wavefronts
// with 16 requests each in flight do not keep the L1 path as busy as the plain kernel's 16 wavefronts do;
// every attempt to buy the parallelism back lost more: 16 or 12 wavefronts per workgroup (128 / 168 VGPRs)
// or a whole fine level in flight per wavefront make hipcc spill ~2 KB per lane (2.9 ms), half-size tiles
// with two workgroups per CU push a third of the coarse taps back to the memory path (320 us), and the
// first version's per-tap LDS min / max atomics for the boxes alone cost 320 us (same-address LDS atomics
// serialise; now one per wavefront).  What the experiment shows: with fp32 values the coarse levels'
// share of the L1 traffic can move to LDS, but the price is the occupancy the fine levels live on.
#pragma once
#include "msda_d32.h"

namespace bevmsda {

struct Lds2Args {
  FusedArgs f;          // as for msda_fused_d32_kernel (k.NQ = rows or their capacity, nrows / launch_rows for dynamic counts)
  int rows_per_block;   // multiple of 64
  int cap2, cap3;       // pixel capacity of the two LDS tiles (level L-2, level L-1)
};

constexpr int kLds2Threads = 512;

struct LdsPointParams {
  float k00, k01, k10, k11;   // coefficients for the LDS pass (all 0 when the point is not served from LDS)
  int li;                // float index of the top-left tap inside the tile (0 when not served from LDS)
  int dli;               // low 30 bits: float stride to the row below; bit 30: the x + 1 tap is one pixel to the right
  bool spill;            // inside the map but outside the staged box: the point takes the memory path
};

// Parameters of one point of a staged level: `pm` are its memory-path parameters (point_params), (bx0, by0,
// tw, th) the staged box.
__device__ __forceinline__ LdsPointParams lds_point_params(const PointParams &pm, float lx, float ly, int H, int W,
                                                           int bx0, int by0, int tw, int th) {
  LdsPointParams p;
  const float Wf = static_cast<float>(W), Hf = static_cast<float>(H);
  const float x = lx * Wf - 0.5f, y = ly * Hf - 0.5f;
  const bool inside = (x > -1.f) && (y > -1.f) && (x < Wf) && (y < Hf);
  const int x0 = static_cast<int>(floorf(x)), y0 = static_cast<int>(floorf(y));
  const bool x0ok = x0 >= 0, x1ok = x0 + 1 < W, y0ok = y0 >= 0, y1ok = y0 + 1 < H;
  // the 2 x 2 footprint clamped into the map must lie inside the box; an out-of-map tap has a zero
  // coefficient, so its clamped neighbour inside the box will do
  const int cx0 = x0 < 0 ? 0 : x0, cy0 = y0 < 0 ? 0 : y0;
  const int cx1 = x0 + 1 < W ? x0 + 1 : W - 1, cy1 = y0 + 1 < H ? y0 + 1 : H - 1;
  const bool in_box = inside && cx0 >= bx0 && cy0 >= by0 && cx1 < bx0 + tw && cy1 < by0 + th;
  p.k00 = in_box ? pm.k00 : 0.f;
  p.k01 = in_box ? pm.k01 : 0.f;
  p.k10 = in_box ? pm.k10 : 0.f;
  p.k11 = in_box ? pm.k11 : 0.f;
  p.li = in_box ? ((cy0 - by0) * tw + (cx0 - bx0)) * 32 : 0;
  p.dli = in_box ? (((y0ok && y1ok) ? tw * 32 : 0) | ((x0ok && x1ok) ? 1 << 30 : 0)) : 0;
  p.spill = inside && !in_box;
  return p;
}

// level served from LDS: broadcast point j's parameters, read its four taps, accumulate (branch-free: a
// point that is not in the box reads tile[0] with zero coefficients)
template <int J0, int j, int CNT>
struct Lds2Points {
  static __device__ __forceinline__ void run(const LdsPointParams &p, const float *tile, int lane4, f32x4 &acc) {
    constexpr int J = J0 + j;
    const int li = static_cast<int>(bcast8<J>(static_cast<uint32_t>(p.li)));
    const uint32_t dl = bcast8<J>(static_cast<uint32_t>(p.dli));
    const float k00 = bcast8<J>(p.k00), k01 = bcast8<J>(p.k01), k10 = bcast8<J>(p.k10), k11 = bcast8<J>(p.k11);
    const int dx = (dl >> 30) ? 32 : 0, dy = static_cast<int>(dl & 0x3fffffffu);
    const float *b = tile + li + lane4;
    const f32x4 v00 = *reinterpret_cast<const f32x4 *>(b);
    const f32x4 v01 = *reinterpret_cast<const f32x4 *>(b + dx);
    const f32x4 v10 = *reinterpret_cast<const f32x4 *>(b + dy);
    const f32x4 v11 = *reinterpret_cast<const f32x4 *>(b + dy + dx);
#pragma unroll
    for (int c = 0; c < 4; ++c)
      acc[c] = fmaf(k11, v11[c], fmaf(k10, v10[c], fmaf(k01, v01[c], fmaf(k00, v00[c], acc[c]))));
    if constexpr (j + 1 < CNT) Lds2Points<J0, j + 1, CNT>::run(p, tile, lane4, acc);
  }
};

// points of a staged level that fell outside its box (rare; wave-uniform decision): the memory path
__device__ __forceinline__ void spill_level(const PointParams &pm, const LdsPointParams &pl, __amdgpu_buffer_rsrc_t r,
                                            uint32_t lane_term, uint32_t dxb, uint32_t dyb, f32x4 &acc) {
  if (__ballot(pl.spill) == 0) return;
  PointParams q = pm;
  if (!pl.spill) { q.k00 = q.k01 = q.k10 = q.k11 = 0.f; q.off = kOobOffset; }
  sample_points<0, 4, float>(q, r, lane_term, dxb, dyb, acc);
  sample_points<4, 4, float>(q, r, lane_term, dxb, dyb, acc);
}

// memory-path level split in two halves: issue (broadcasts + 32 loads), consume (FMAs)
template <int J0, int CNT>
__device__ __forceinline__ void issue_level(const PointParams &p, __amdgpu_buffer_rsrc_t r, uint32_t lane_term,
                                            uint32_t dxb, uint32_t dyb, f32x4 (&v)[CNT][4], float (&k)[CNT][4]) {
  IssuePoints<J0, 0, CNT, float>::run(p, r, lane_term, dxb, dyb, v, k);
}
template <int CNT>
__device__ __forceinline__ void consume_level(const f32x4 (&v)[CNT][4], const float (&k)[CNT][4], f32x4 &acc) {
#pragma unroll
  for (int j = 0; j < CNT; ++j)
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      acc[0] = fmaf(k[j][t], v[j][t][0], acc[0]);
      acc[1] = fmaf(k[j][t], v[j][t][1], acc[1]);
      acc[2] = fmaf(k[j][t], v[j][t][2], acc[2]);
      acc[3] = fmaf(k[j][t], v[j][t][3], acc[3]);
    }
}

// copy the box (bx0, by0, tw, th) of level `lvl` of (entry n, head m) into `tile` ([pixel][32 floats])
__device__ __forceinline__ void stage_box(const KArgs &a, float *tile, long n, int m, int lvl, int bx0, int by0, int tw,
                                          int th) {
  const int W = static_cast<int>(a.shapes[2 * lvl + 1]);
  const float *src = static_cast<const float *>(a.value) + ((n * a.S + a.lstart[lvl]) * a.M + m) * 32;
  const int pieces = tw * th * 8;                       // 16-byte pieces
  for (int i0 = 0; i0 < pieces; i0 += kLds2Threads) {   // wave-uniform trip count and LDS base
    const int i = i0 + threadIdx.x;
    const int px = i >> 3, q4 = (i & 7) * 4;
    const int ty = px / tw, tx = px - ty * tw;
    if (i0 + (threadIdx.x & ~63) < pieces) {
      // lanes past the end read pixel 0 of the box again into the slack after the tile (allocated)
      const long gp = i < pieces ? static_cast<long>(by0 + ty) * W + bx0 + tx : static_cast<long>(by0) * W + bx0;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + gp * a.M * 32 + q4),
                                       (__attribute__((address_space(3))) void *)(tile + (i0 + (threadIdx.x & ~63)) * 4),
                                       16, 0, 0);
    }
  }
}

template <bool DYN>
__global__ void __launch_bounds__(kLds2Threads) __attribute__((amdgpu_waves_per_eu(2, 2)))
msda_fused_d32_lds2_kernel(const Lds2Args g) {
  constexpr int D = 32, PT = 8;
  extern __shared__ __attribute__((aligned(16))) float lds2[];
  const FusedArgs &f = g.f;
  const KArgs &a = f.k;
  float *tile3 = lds2;                                    // level L-1
  float *tile2 = lds2 + (static_cast<long>(g.cap3) + 64) * D;   // level L-2 (64 pixels of slack after each tile)
  int *box = reinterpret_cast<int *>(tile2 + (static_cast<long>(g.cap2) + 64) * D);     // 4 ints level 2, 4 ints level 3, 1 misc
  const int tid = threadIdx.x;
  const int lig = tid & 7;
  const int m = blockIdx.x % a.M;
  const int chunk = blockIdx.x / a.M;
  long NQ = a.NQ;
  if constexpr (DYN) {
    const int n = *f.nrows;
    const int cap = static_cast<int>(a.NQ);
    const int total = n < cap ? n : cap;
    NQ = total < f.launch_rows ? total : f.launch_rows;   // rows beyond the hint: the strided tail launch
  }
  const long c0 = static_cast<long>(chunk) * g.rows_per_block;
  if (c0 >= NQ) return;
  const long c1 = c0 + g.rows_per_block < NQ ? c0 + g.rows_per_block : NQ;
  constexpr int L2 = 2, L3 = 3;                          // L = 4 (launch condition): compile-time level indices
  const uint32_t pix_bytes = static_cast<uint32_t>(a.M) * D * sizeof(float);
  const uint32_t lane_term = lig * 16u;
  const uint32_t total_bytes = static_cast<uint32_t>(static_cast<unsigned long long>(a.N) * a.S * pix_bytes);
  __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(a.value), 0,
                                                                  static_cast<int>(total_bytes), 0x00020000);
  int Hs[4], Ws[4];
  uint32_t lb[4];
#pragma unroll
  for (int l = 0; l < 4; ++l) {
    Hs[l] = static_cast<int>(a.shapes[2 * l]);
    Ws[l] = static_cast<int>(a.shapes[2 * l + 1]);
    lb[l] = static_cast<uint32_t>(a.lstart[l]) * pix_bytes;
  }

  long r0 = c0;
  while (r0 < c1) {
    // ---- sub-range of rows that share a camera
    const long n0 = a.row_batch ? static_cast<long>(a.row_batch[r0]) : r0 / a.Q;
    __syncthreads();
    if (tid == 0) box[8] = static_cast<int>(c1 - r0);
    if (tid < 8) box[tid] = (tid & 2) ? INT_MIN : INT_MAX;
    __syncthreads();
    if (a.row_batch) {
      for (long r = r0 + tid; r < c1; r += kLds2Threads)
        if (a.row_batch[r] != n0) atomicMin(&box[8], static_cast<int>(r - r0));
    } else {
      const long e = (n0 + 1) * a.Q;
      if (tid == 0 && e < c1) box[8] = static_cast<int>(e - r0);
    }
    __syncthreads();
    const long r1 = r0 + box[8];
    const int nrows = static_cast<int>(r1 - r0);
    const uint32_t head_base = static_cast<uint32_t>((static_cast<unsigned long long>(n0) * a.S * a.M + m) * D * sizeof(float));

    // ---- (1) boxes of the level L-2 / L-1 taps of my rows: one thread per (row, point)
    {
      int mn[4] = {INT_MAX, INT_MAX, INT_MAX, INT_MAX}, mxv[4] = {INT_MIN, INT_MIN, INT_MIN, INT_MIN};
      for (int i = tid; i < nrows * PT; i += kLds2Threads) {
        const int rw = i >> 3, pj = i & 7;
        const long r = r0 + rw;
        const long rs = f.row_src ? static_cast<long>(f.row_src[r]) : r;
        const float2 rf = reinterpret_cast<const float2 *>(f.ref)[r * f.A + pj % f.A];
        const float2 *ofp = reinterpret_cast<const float2 *>(f.offs + rs * f.proj_row + m * f.off_head) + pj;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const int l = L2 + q;
          const float2 of = ofp[l * PT];
          const float x = (rf.x + of.x / static_cast<float>(Ws[l])) * Ws[l] - 0.5f;
          const float y = (rf.y + of.y / static_cast<float>(Hs[l])) * Hs[l] - 0.5f;
          if (!(x > -1.f && y > -1.f && x < Ws[l] && y < Hs[l])) continue;
          const int x0 = static_cast<int>(floorf(x)), y0 = static_cast<int>(floorf(y));
          const int xa = x0 > 0 ? x0 : 0, ya = y0 > 0 ? y0 : 0;
          const int xb = x0 + 1 < Ws[l] ? x0 + 1 : Ws[l] - 1, yb = y0 + 1 < Hs[l] ? y0 + 1 : Hs[l] - 1;
          mn[q * 2] = xa < mn[q * 2] ? xa : mn[q * 2];
          mn[q * 2 + 1] = ya < mn[q * 2 + 1] ? ya : mn[q * 2 + 1];
          mxv[q * 2] = xb > mxv[q * 2] ? xb : mxv[q * 2];
          mxv[q * 2 + 1] = yb > mxv[q * 2 + 1] ? yb : mxv[q * 2 + 1];
        }
      }
      // one LDS atomic per wavefront and value (not per tap: same-address LDS atomics serialise)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
          const int t0 = __shfl_xor(mn[q], o, 64), t1 = __shfl_xor(mxv[q], o, 64);
          mn[q] = t0 < mn[q] ? t0 : mn[q];
          mxv[q] = t1 > mxv[q] ? t1 : mxv[q];
        }
      }
      if ((tid & 63) == 0) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          if (mn[q * 2] <= mxv[q * 2]) {
            atomicMin(&box[q * 4 + 0], mn[q * 2]);
            atomicMin(&box[q * 4 + 1], mn[q * 2 + 1]);
            atomicMax(&box[q * 4 + 2], mxv[q * 2]);
            atomicMax(&box[q * 4 + 3], mxv[q * 2 + 1]);
          }
        }
      }
    }
    __syncthreads();
    int bx[2], by[2], tw[2], th[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int cap = q == 0 ? g.cap2 : g.cap3;
      const int x0 = box[q * 4], y0 = box[q * 4 + 1], x1 = box[q * 4 + 2], y1 = box[q * 4 + 3];
      const bool have = x0 <= x1;
      bx[q] = have ? x0 : 0;
      by[q] = have ? y0 : 0;
      const int w = have ? x1 - x0 + 1 : 0, h = have ? y1 - y0 + 1 : 0;
      tw[q] = w <= cap ? w : cap;                          // clip: whole box rows while they fit
      th[q] = (tw[q] > 0) ? ((h <= cap / tw[q]) ? h : cap / tw[q]) : 0;
    }
    // ---- (2) stage the two boxes
    stage_box(a, tile2, n0, m, L2, bx[0], by[0], tw[0], th[0]);
    stage_box(a, tile3, n0, m, L3, bx[1], by[1], tw[1], th[1]);
    __syncthreads();                                       // (the compiler drains the LDS-DMA before the barrier)

    // ---- (3) the rows: 64 (row, head) lane groups per pass
    for (int rw0 = 0; rw0 < nrows; rw0 += kLds2Threads / 8) {
      const int rw = rw0 + (tid >> 3);
      const bool active = rw < nrows;
      const long r = r0 + (active ? rw : nrows - 1);
      const long rs = f.row_src ? static_cast<long>(f.row_src[r]) : r;
      const float *__restrict__ lgp = f.logits + rs * f.proj_row + m * f.lg_head + lig;
      const float2 *__restrict__ ofp = reinterpret_cast<const float2 *>(f.offs + rs * f.proj_row + m * f.off_head) + lig;
      const float2 rf = reinterpret_cast<const float2 *>(f.ref)[r * f.A + lig % f.A];
      float e[4];
      float2 of[4];
#pragma unroll
      for (int l = 0; l < 4; ++l) {
        e[l] = lgp[l * PT];
        of[l] = ofp[l * PT];
      }
      const float mx = lanes_max<PT>(fmaxf(fmaxf(e[0], e[1]), fmaxf(e[2], e[3])));
#pragma unroll
      for (int l = 0; l < 4; ++l) e[l] = expf(e[l] - mx);
      const float sum = lanes_sum<PT>((e[0] + e[1]) + (e[2] + e[3]));
      float lx[4], ly[4], aw[4];
#pragma unroll
      for (int l = 0; l < 4; ++l) {
        lx[l] = rf.x + of[l].x / static_cast<float>(Ws[l]);
        ly[l] = rf.y + of[l].y / static_cast<float>(Hs[l]);
        aw[l] = active ? e[l] / sum : 0.f;
      }
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      // half a fine level (4 points = 16 requests, 16 KB per wavefront) in flight while half a coarse level is
      // summed out of LDS: what fits 256 VGPRs without spills (a whole level in flight spills 2 KB per lane)
      f32x4 v[4][4];
      float k[4][4];
      const PointParams p0 = point_params(lx[0], ly[0], aw[0], Hs[0], Ws[0], head_base + lb[0], pix_bytes);
      const PointParams p1 = point_params(lx[1], ly[1], aw[1], Hs[1], Ws[1], head_base + lb[1], pix_bytes);
      const PointParams m2 = point_params(lx[L2], ly[L2], aw[L2], Hs[L2], Ws[L2], head_base + lb[L2], pix_bytes);
      const PointParams m3 = point_params(lx[L3], ly[L3], aw[L3], Hs[L3], Ws[L3], head_base + lb[L3], pix_bytes);
      const LdsPointParams p2 = lds_point_params(m2, lx[L2], ly[L2], Hs[L2], Ws[L2], bx[0], by[0], tw[0], th[0]);
      const LdsPointParams p3 = lds_point_params(m3, lx[L3], ly[L3], Hs[L3], Ws[L3], bx[1], by[1], tw[1], th[1]);
      const uint32_t dy0 = static_cast<uint32_t>(Ws[0]) * pix_bytes, dy1 = static_cast<uint32_t>(Ws[1]) * pix_bytes;
      issue_level<0, 4>(p0, rsrc, lane_term, pix_bytes, dy0, v, k);
      Lds2Points<0, 0, 4>::run(p2, tile2, lig * 4, acc);
      consume_level<4>(v, k, acc);
      issue_level<4, 4>(p0, rsrc, lane_term, pix_bytes, dy0, v, k);
      Lds2Points<4, 0, 4>::run(p2, tile2, lig * 4, acc);
      consume_level<4>(v, k, acc);
      issue_level<0, 4>(p1, rsrc, lane_term, pix_bytes, dy1, v, k);
      Lds2Points<0, 0, 4>::run(p3, tile3, lig * 4, acc);
      consume_level<4>(v, k, acc);
      issue_level<4, 4>(p1, rsrc, lane_term, pix_bytes, dy1, v, k);
      Lds2Points<4, 0, 4>::run(p3, tile3, lig * 4, acc);
      consume_level<4>(v, k, acc);
      spill_level(m2, p2, rsrc, lane_term, pix_bytes, static_cast<uint32_t>(Ws[L2]) * pix_bytes, acc);
      spill_level(m3, p3, rsrc, lane_term, pix_bytes, static_cast<uint32_t>(Ws[L3]) * pix_bytes, acc);
      if (active) {
        float *op = static_cast<float *>(a.out) + (r * a.M + m) * D + lig * 4;
        *reinterpret_cast<float4 *>(op) = make_float4(acc[0], acc[1], acc[2], acc[3]);
      }
    }
    r0 = r1;
  }
}

}  // namespace bevmsda
