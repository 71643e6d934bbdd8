// Backward of multi-scale deformable attention, D = 32, second generation:
// grad_value by a per-workgroup counting sort + segmented reduction in LDS instead of one memory-side
// atomic per bilinear tap.
//
// Why: the first backward (msda_kernels.h, msda_bwd_d32_kernel) issues one 128-byte-line fp32 atomic
// per (row, head, level, point, tap): 47 M line operations per base SCA call, and the L2 atomic units
// retire ~10 G of them per second whatever their scope (tools/probes/atomic_probe.hip) — 4.7 ms of
// 6.5-8.6 ms, 1 % of the HBM roofline.  Neighbouring rows hit the same pixels: per 256-row block of one
// (camera, head) the taps of one (level, pillar anchor) land on 20-400 distinct pixels out of 2,000
// (measured on the base rig).  LDS float atomics do not help either (a first version
// accumulated in a dense LDS tile with ds_add_f32: 6.5 ms — the LDS retires float atomics far below its
// load / store rate).  So the taps are SORTED by pixel in LDS and each pixel's contributions are summed
// in registers:
//
//   * msda_bwd_d32_kernel<T, PT, false> (msda_kernels.h) keeps grad_loc / grad_attn (a gather with an
//     in-register channel reduction) and drops the scatter;
//   * msda_gradvalue_sort_kernel (here) owns the scatter.  A workgroup takes `rows_per_block`
//     consecutive rows of ONE head (split at value-batch-entry boundaries).  For every (level, point
//     group) — point group g = points p with p % G == g, i.e. the points that share a pillar anchor in
//     SpatialCrossAttention — one thread per (row, point) record computes the four taps; the workgroup
//     finds their bounding box, counts taps per pixel of the box (integer LDS atomics), scans the
//     counts, places (pixel, row, coefficient) entries in pixel order, and then every half-wave walks a
//     contiguous share of the sorted entries: lane c accumulates coefficient x grad_out[row, c] in a
//     register while the pixel stays the same and issues ONE memory-side atomic per (pixel, head) line
//     when it changes.  Memory-side atomics drop from 47 M to ~3.5 M per base SCA call.
//     (As built: ONE pass per level — all points of all rows, 2,048 records — and instead of a bounding
//     box the taps are counted into 4,096 buckets keyed by (point group, y mod 2^b, x mod 2^b): pixels of
//     one group's footprint fall into distinct buckets unless the footprint is wider than 2^b, and a
//     bucket that does mix pixels only costs an extra flush, never a wrong sum.  5 barriers per level.)
//
// Numerics: fp32 sums in a different (still unordered) association; parity tests use the same
// tolerances as for the first kernel.  grad_value is accumulated into (caller zeroes it).
#pragma once
#include "msda_kernels.h"

namespace bevmsda {

struct GradValueArgs {
  int dense_tiles;     // > 0 (dense single-level calls, rows_per_block = 64 / 128 / 256): the launch has this many row
                       // chunks per batch entry; when the level is an H0 x W0 grid with Q = H0 * W0 (read on
                       // the device: TemporalSelfAttention, whose rows are the BEV grid in raster order) and
                       // its 16 x (rows_per_block / 16) tiles fit that count, chunk t takes tile t of the grid instead of that many
                       // consecutive rows — any row order gives the same sums, this one makes the rows of a
                       // workgroup share taps
  KArgs k;             // value unused; loc, attn, grad_out, grad_value, row_batch, NQ, N, S, M, L, Q, P
  int rows_per_block;  // <= kGvMaxRows
  int gbits;           // log2 of the point groups per level G (G = 1, 2, 4: points p with equal p % G share a
                       // pillar anchor in SpatialCrossAttention and therefore a footprint)
  unsigned long long *prof;   // PROF kernels only: phase clock sums + [7] flush count
};

constexpr int kGvThreads = 1024;     // 32 half-waves
constexpr int kGvMaxRows = 256;      // rows of one head per workgroup
constexpr int kGvBuckets = 4096;     // counters of the sort
constexpr int kGvMaxLevels = 4;
// the walk of the sorted entries: 8 = 8-lane groups with 4 channels per lane and an LDS stage for finished runs
// (round 6), 32 = a half-wave per share, a channel per lane (round 2)
#ifndef BEVMSDA_GV_WALK
#define BEVMSDA_GV_WALK 8
#endif
// diagnostic builds only (-DBEVMSDA_GV_DIAG_NOATOMIC=1): the flush atomics are issued for one impossible sum only, everything
// before them stays — the kernel's time without its memory-side traffic (results are wrong by construction)
#ifndef BEVMSDA_GV_DIAG_NOATOMIC
#define BEVMSDA_GV_DIAG_NOATOMIC 0
#endif
__device__ __forceinline__ void gv_flush(float *p, float v) {
  if (BEVMSDA_GV_DIAG_NOATOMIC && v != 12345.678f) return;
  unsafeAtomicAdd(p, v);
}
#ifndef BEVMSDA_GV_WALK_U
#define BEVMSDA_GV_WALK_U 4
#endif
constexpr int kGvWalkU = BEVMSDA_GV_WALK_U;                               // entries a group reads ahead (and sentinel entries behind the last)
constexpr int kGvGlStride = 36;                           // words per grad_out row in LDS: 32 + 4, so that the 64- / 128-byte
                                                          // pieces different groups read spread over the 64 banks
// the stage of finished runs, per wavefront: S slots of 32 sums, then S pixel indices.  A step can park one run per
// group (8 or 16 groups per wavefront) on top of at most S - groups parked ones.
constexpr int kGvStageSlots = BEVMSDA_GV_WALK == 4 ? 24 : 15;
constexpr int kGvStageWords = BEVMSDA_GV_WALK == 4 ? 800 : 512;
// LDS carve, in 4-byte words: [cnt: kGvBuckets][stage overflow][entries: (rows * P * 4 + U) x 2][grad_out rows: rows x 36]
// [wsum 16][misc 4].  The stage starts at cnt (the counters are dead during the walk) and runs into the overflow words.
constexpr int gv_stage_overflow_words(int threads) {
  return BEVMSDA_GV_WALK == 32 || (threads / 64) * kGvStageWords <= kGvBuckets ? 0 : (threads / 64) * kGvStageWords - kGvBuckets;
}
constexpr size_t gv_lds_bytes(int threads, int rows, int P) {
  return (static_cast<size_t>(kGvBuckets) + gv_stage_overflow_words(threads) + (static_cast<size_t>(rows) * P * 4 + kGvWalkU) * 2 +
          static_cast<size_t>(rows) * kGvGlStride + kGvThreads / 64 + 4) * 4;
}

// The kernel's operands are read once (records, grad_out rows) while the lines its flush atomics touch are revisited by the
// workgroups of neighbouring rows: -DBEVMSDA_GV_STREAM_NT=1 marks the reads non-temporal so that they do not push those
// lines out of the L2 (round 6 A/B, DESIGN K2).
#ifndef BEVMSDA_GV_STREAM_NT
#define BEVMSDA_GV_STREAM_NT 0
#endif
typedef float gv_f32x4 __attribute__((ext_vector_type(4)));
typedef float gv_f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned gv_u32x2 __attribute__((ext_vector_type(2)));
template <typename V> __device__ __forceinline__ V gv_stream_load(const V *p) {
#if BEVMSDA_GV_STREAM_NT
  return __builtin_nontemporal_load(p);
#else
  return *p;
#endif
}
template <typename T> __device__ __forceinline__ float4 load_gout4(const T *p);
template <> __device__ __forceinline__ float4 load_gout4<float>(const float *p) {
  const gv_f32x4 t = gv_stream_load(reinterpret_cast<const gv_f32x4 *>(p));
  return make_float4(t[0], t[1], t[2], t[3]);
}
template <> __device__ __forceinline__ float4 load_gout4<bf16_t>(const bf16_t *p) {
  const gv_u32x2 t = gv_stream_load(reinterpret_cast<const gv_u32x2 *>(p));
  return make_float4(bf16_lo(t[0]), bf16_hi(t[0]), bf16_lo(t[1]), bf16_hi(t[1]));
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() carries a workgroup-scope fence, which on
// gfx9 is `s_waitcnt vmcnt(0)` before the s_barrier: every wavefront would first wait for its outstanding
// memory-side flush atomics (vmcnt counts them until the L2 has retired them), i.e. the ~10 G line-atomics / s
// of the L2 would sit on the critical path of the NEXT level's sort instead of running under it (measured:
// the phase after a level's flush took 59 % of the kernel).  The sort only communicates through LDS, and the
// atomics are fire-and-forget (no return value, nothing reads grad_value here), so lgkmcnt(0) + s_barrier is
// all the ordering it needs.
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// In-place exclusive scan of cnt[0 .. kGvBuckets] (kGvBuckets + 1 entries, the last one = total) by the
// whole workgroup: 4 counters per thread.  Two barriers.
template <int THREADS>
__device__ __forceinline__ int block_exclusive_scan(int *cnt, int *wsum /* THREADS / 64 ints */) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  constexpr int PER = kGvBuckets / THREADS;          // counters per thread: 4, 8 or 16
  static_assert(PER == 4 || PER == 8 || PER == 16, "4, 8 or 16 counters per thread");
  constexpr int NV = PER / 4;
  int4 v[NV];
  int part[NV];
  int s = 0;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    v[i] = reinterpret_cast<int4 *>(cnt)[tid * NV + i];
    part[i] = (v[i].x + v[i].y) + (v[i].z + v[i].w);
    s += part[i];
  }
  int inc = s;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int t = __shfl_up(inc, o, 64);
    if (lane >= o) inc += t;
  }
  if (lane == 63) wsum[wave] = inc;
  lds_barrier();
  int base = inc - s, total = 0;
#pragma unroll
  for (int w = 0; w < THREADS / 64; ++w) {
    const int t = wsum[w];
    if (w < wave) base += t;
    total += t;
  }
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    reinterpret_cast<int4 *>(cnt)[tid * NV + i] =
        make_int4(base, base + v[i].x, base + v[i].x + v[i].y, base + v[i].x + v[i].y + v[i].z);
    base += part[i];
  }
  lds_barrier();
  return total;
}

#define GV_TICK(slot)                                                                          \
  if constexpr (PROF) {                                                                        \
    const unsigned long long t_ = __builtin_readcyclecounter();                                \
    if (tid == 0) atomicAdd(&s.prof[slot], t_ - t_prev);                                       \
    t_prev = t_;                                                                               \
  }

// LDS carve: gv_lds_bytes() above.
// RPT = (row, point) records per thread and level = ceil(rows_per_block * P / THREADS).
// THREADS = 1024 with 256 rows per workgroup (112 KB of LDS: one workgroup per CU) or 512 with 128 rows (64 KB: two
// workgroups per CU, one sorting while the other's flushes drain, at the price of a smaller footprint per flush).
template <typename T, int RPT, bool PROF = false, int THREADS = kGvThreads>
__global__ void __launch_bounds__(THREADS) msda_gradvalue_sort_kernel(const GradValueArgs s) {
  unsigned long long t_prev = 0;
  if constexpr (PROF) t_prev = __builtin_readcyclecounter();
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const KArgs &a = s.k;
  constexpr int D = 32;
  const int L = a.L, P = a.P;
  int *cnt = reinterpret_cast<int *>(lds);
  int2 *ent = reinterpret_cast<int2 *>(cnt + kGvBuckets + gv_stage_overflow_words(THREADS));
  float *gl = reinterpret_cast<float *>(ent + s.rows_per_block * P * 4 + kGvWalkU);
  int *wsum = reinterpret_cast<int *>(gl + static_cast<long>(s.rows_per_block) * kGvGlStride);
  int *misc = wsum + kGvThreads / 64;
  [[maybe_unused]] float *const stage = reinterpret_cast<float *>(cnt);
  const int tid = threadIdx.x;
  const int m = blockIdx.x % a.M;
  const int chunk = blockIdx.x / a.M;
  // virtual row -> row of the operands (-1: a tile cell outside the grid)
  const int W0 = static_cast<int>(a.shapes[1]), H0 = static_cast<int>(a.shapes[0]);
  const int th = s.rows_per_block >> 4;                                      // tiles are 16 cells wide, rows_per_block / 16 high
  const int tile_w = (W0 + 15) >> 4, tile_h = (H0 + th - 1) / th;
  const bool tiled = s.dense_tiles > 0 && static_cast<long>(H0) * W0 == a.Q && tile_w * tile_h <= s.dense_tiles;
  // (row counts fit 31 bits: checked by the launcher)
  const int vq = tiled ? s.dense_tiles * s.rows_per_block : a.Q;                           // virtual rows per batch entry
  const int vrows = tiled ? vq * a.N : static_cast<int>(effective_rows(a));    // (device-side row count: ragged calls)
  const int c0 = chunk * s.rows_per_block;
  if (c0 >= vrows) return;
  const int c1 = c0 + s.rows_per_block < vrows ? c0 + s.rows_per_block : vrows;
  const int my_tile = tiled ? chunk % s.dense_tiles : 0;                      // tiled: one chunk = one tile
  if (tiled && my_tile >= tile_w * tile_h) return;
  const int tile_y0 = tiled ? (my_tile / tile_w) * th : 0, tile_x0 = tiled ? (my_tile % tile_w) * 16 : 0;
  const int tile_n = tiled ? chunk / s.dense_tiles : 0;
  auto phys = [&](int vr) -> int {
    if (!tiled) return vr;
    const int w = vr - c0;                                                    // 0 .. rows_per_block - 1 inside my tile
    const int y = tile_y0 + (w >> 4), x = tile_x0 + (w & 15);
    return (y < H0 && x < W0) ? tile_n * a.Q + y * W0 + x : -1;
  };
  const long pix_stride = a.gv_stride > 0 ? a.gv_stride : static_cast<long>(a.M * D);
  [[maybe_unused]] const int half = tid >> 5;          // 32 half-waves per workgroup
  const int c = tid & 31;             // my channel
  // bucket of a tap: [point group | y mod 2^yb | x mod 2^xb], 12 bits
  const int gbits = s.gbits, gmask = (1 << gbits) - 1;
  const int xb = (12 - gbits + 1) >> 1, yb = 12 - gbits - xb;
  const int xm = (1 << xb) - 1, ym = (1 << yb) - 1;

  // grad_out of the chunk's rows and my head -> LDS
  for (int i = tid; i < (c1 - c0) * 8; i += THREADS) {
    const int row = i >> 3, q4 = (i & 7) * 4;
    const int pr = phys(c0 + row);
    float4 gq = make_float4(0.f, 0.f, 0.f, 0.f);
    if (pr >= 0) {
      const long grow = a.gout_rows > 0 ? static_cast<long>(pr) % a.gout_rows : static_cast<long>(pr);
      gq = load_gout4<T>(static_cast<const T *>(a.grad_out) + (grow * a.M + m) * D + q4);
      if (a.gout_rows > 0) { gq.x *= a.gout_scale; gq.y *= a.gout_scale; gq.z *= a.gout_scale; gq.w *= a.gout_scale; }
    }
    *reinterpret_cast<float4 *>(gl + row * kGvGlStride + q4) = gq;
  }

  // sub-ranges of rows that share a value batch entry (rows are grouped by camera: normally one)
  int r0 = c0;
  while (r0 < c1) {
    const long n0 = a.row_batch ? static_cast<long>(a.row_batch[r0]) : r0 / vq;
    int r1;
    if (a.row_batch) {
      __syncthreads();
      if (tid == 0) misc[0] = c1 - r0;
      __syncthreads();
      for (int r = r0 + tid; r < c1; r += THREADS)
        if (a.row_batch[r] != n0) atomicMin(&misc[0], r - r0);
      __syncthreads();
      r1 = r0 + misc[0];
    } else {
      const long e = (n0 + 1) * vq;
      r1 = e < c1 ? static_cast<int>(e) : c1;
    }
    const int nrows = r1 - r0;
    const int grow0 = r0 - c0;                         // first row of the sub-range inside gl
    const int nrec = nrows * P;                        // records per level

    // level geometry up front as well: a load inside the level loop would wait on vmcnt and with it on every
    // flush atomic still in flight (the counter retires in order)
    int Hs[kGvMaxLevels], Ws[kGvMaxLevels];
    long ls[kGvMaxLevels];
#pragma unroll
    for (int l = 0; l < kGvMaxLevels; ++l) {
      const int ll = l < L ? l : 0;
      // (readfirstlane pins the loads here — the scheduler would otherwise sink them back into the loop —
      // and keeps the uniform values in SGPRs)
      Hs[l] = __builtin_amdgcn_readfirstlane(static_cast<int>(a.shapes[2 * ll]));
      Ws[l] = __builtin_amdgcn_readfirstlane(static_cast<int>(a.shapes[2 * ll + 1]));
      ls[l] = __builtin_amdgcn_readfirstlane(static_cast<int>(a.lstart[ll]));
    }
    // every record of every level of my rows, loaded up front (the only trips to memory on the critical
    // path; a later wait for loads would also drain the wave's outstanding flush atomics)
    float2 xy[kGvMaxLevels][RPT];
    float aw[kGvMaxLevels][RPT];
#pragma unroll
    for (int l = 0; l < kGvMaxLevels; ++l)
#pragma unroll
      for (int j = 0; j < RPT; ++j) {
        xy[l][j] = make_float2(-8.f, -8.f);             // outside every map: no tap
        aw[l][j] = 0.f;
        const int i = j * THREADS + tid;
        if (l < L && i < nrec) {
          const int rw = i / P, p = i - rw * P;
          const int pr = phys(r0 + rw);
          if (pr >= 0) {
            const long pi = ((static_cast<long>(pr) * a.M + m) * L + l) * P + p;
            const gv_f32x2 t2 = gv_stream_load(reinterpret_cast<const gv_f32x2 *>(loc_records(a, pr, m)) + l * P + p);
            // (KArgs::loc == nullptr: t2 is the raw offset; the location is the forward's reference + offset / (W_l, H_l))
            xy[l][j] = loc_of_record(a, make_float2(t2[0], t2[1]), loc_reference(a, pr, p), Hs[l], Ws[l]);
            aw[l][j] = gv_stream_load(a.attn + pi);
          }
        }
      }
    GV_TICK(0)

#pragma unroll
    for (int l = 0; l < kGvMaxLevels; ++l) {
      if (l >= L) break;
      const int H = Hs[l], W = Ws[l];
      float *gv = a.grad_value + (n0 * a.S + ls[l]) * pix_stride + m * D + c;
      // ---- (1) zero the counters
      for (int z = tid; z < kGvBuckets / 4; z += THREADS) reinterpret_cast<int4 *>(cnt)[z] = make_int4(0, 0, 0, 0);
      lds_barrier();
      // ---- (2) my records: four taps each, counted into their buckets
      int pix[RPT][4], key[RPT][4], rowj[RPT];
      float k[RPT][4];
      bool ok[RPT][4];
#pragma unroll
      for (int j = 0; j < RPT; ++j) {
        const int i = j * THREADS + tid;
        const int rw = i / P, p = i - rw * P;
        rowj[j] = rw;
        const float x = xy[l][j].x * W - 0.5f, y = xy[l][j].y * H - 0.5f;
        const bool inside = i < nrec && x > -1.f && y > -1.f && x < W && y < H;
        const float xf = floorf(x), yf = floorf(y);
        const int x0 = static_cast<int>(xf), y0 = static_cast<int>(yf);
        const float fx = x - xf, fy = y - yf;
        const float gx = 1.f - fx, gy = 1.f - fy;
        const int gkey = (p & gmask) << (xb + yb);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const int xx = x0 + (t & 1), yy = y0 + (t >> 1);
          k[j][t] = ((t >> 1) ? fy : gy) * ((t & 1) ? fx : gx) * aw[l][j];
          ok[j][t] = inside && xx >= 0 && yy >= 0 && xx < W && yy < H && k[j][t] != 0.f;
          pix[j][t] = yy * W + xx;
          key[j][t] = gkey | ((yy & ym) << xb) | (xx & xm);
          if (ok[j][t]) atomicAdd(&cnt[key[j][t]], 1);
        }
      }
      lds_barrier();
      GV_TICK(1)
      // ---- (3) scan, place
      const int total = block_exclusive_scan<THREADS>(cnt, wsum);
      GV_TICK(2)
      if (tid < kGvWalkU) ent[total + tid] = make_int2(0, 0);   // what the walk reads past the last entry: coefficient 0
#pragma unroll
      for (int j = 0; j < RPT; ++j)
#pragma unroll
        for (int t = 0; t < 4; ++t)
          if (ok[j][t]) {
            const int pos = atomicAdd(&cnt[key[j][t]], 1);
            ent[pos] = make_int2(pix[j][t] | ((grow0 + rowj[j]) << 23), __float_as_int(k[j][t]));
          }
      lds_barrier();
      GV_TICK(3)
      // ---- (4) segmented reduction, one memory-side atomic per pixel run
#if BEVMSDA_GV_WALK == 8 || BEVMSDA_GV_WALK == 4
      // Round 6.  The walk is where this kernel's instructions go (round 5: 309 M wavefront instructions per base SCA call,
      // most of them here: a half-wave per entry = 17 issue slots per TWO entries).  Now a group of LG = 8 (4) lanes owns
      // a share of the sorted entries and a lane carries 4 (8) channels: one wavefront instruction serves 8 (16) entries,
      // and the run logic (pixel changed?) is one compare + a wavefront-uniform branch.  A finished run is LG lanes x
      // float4 — not the shape of a line atomic — so it is parked in a per-wavefront LDS stage (the counters' space: they
      // are dead during the walk) and drained by the whole wavefront, half-wave per run: still ONE 128-byte-line atomic
      // per run.  Shares are cut at pixel changes (searched from the nominal cut, LG candidates per step), so a share
      // boundary does not split a run and costs no extra flush.
      {
        constexpr int LG = BEVMSDA_GV_WALK, LGS = LG == 8 ? 3 : 2;
        constexpr int NV = D / LG / 4;                 // float4 per lane: 1 or 2
        constexpr int NGW = 64 / LG;                   // groups per wavefront
        constexpr int NG = THREADS / LG;
        constexpr int U = kGvWalkU;
        const int lane = tid & 63, gw = lane >> LGS, j = lane & (LG - 1), grp = tid >> LGS;
        float *const stg = stage + (tid >> 6) * kGvStageWords;
        int *const spx = reinterpret_cast<int *>(stg + kGvStageSlots * D);
        // first e >= nom where a pixel run starts (or 0, or total); gives up after 32 entries (a split run costs one
        // more atomic, never a wrong sum)
        auto run_start = [&](int nom) -> int {
          int res = -1;
#pragma unroll 1
          for (int step = 0; step < 32 / LG; ++step) {
            const int e = nom + j;
            bool b = e >= total || e <= 0;
            if (!b) b = ((ent[e].x ^ ent[e - 1].x) & 0x7fffff) != 0;
            const unsigned long long mk = __ballot(b);
            const unsigned mine = static_cast<unsigned>(mk >> (gw * LG)) & ((1u << LG) - 1u);
            if (res < 0 && mine) res = nom + __ffs(mine) - 1;
            if (res < 0) nom += LG;
            if (__ballot(res < 0) == 0) break;
          }
          if (res < 0) res = nom;
          return res < total ? res : total;
        };
        int e = run_start(static_cast<int>(static_cast<long>(total) * grp / NG));
        const int e1 = grp == NG - 1 ? total : run_start(static_cast<int>(static_cast<long>(total) * (grp + 1) / NG));
        GV_TICK(5)
        int nrem = e1 - e;
        int cur = nrem > 0 ? (ent[e].x & 0x7fffff) : -1;   // my first run is open from the start: every change parks a run
        float4 acc[NV];
#pragma unroll
        for (int v = 0; v < NV; ++v) acc[v] = make_float4(0.f, 0.f, 0.f, 0.f);
        int nst = 0;                                   // parked runs of my wavefront (uniform)
        const int half32 = lane >> 5, c32 = lane & 31;
        auto drain = [&](int n) {
          for (int sl = half32; sl < n; sl += 2) {
            const float v = stg[sl * D + c32];
            const int p = spx[sl];
            gv_flush(gv + p * pix_stride, v);
            if constexpr (PROF) { if (c32 == 0) atomicAdd(&s.prof[7], 1ULL); }
          }
        };
        auto park = [&](bool fl, unsigned long long fm) {   // fm = ballot(fl), not 0; called by the whole wavefront
          const int below = __builtin_amdgcn_mbcnt_hi(static_cast<unsigned>(fm >> 32),
                                                      __builtin_amdgcn_mbcnt_lo(static_cast<unsigned>(fm), 0u));
          const int slot = nst + (below >> LGS);
          if (fl) {
#pragma unroll
            for (int v = 0; v < NV; ++v) *reinterpret_cast<float4 *>(stg + slot * D + v * (LG * 4) + j * 4) = acc[v];
            spx[slot] = cur;                           // (the same word from every lane of the group)
          }
          nst += __popcll(fm) >> LGS;
          if (nst > kGvStageSlots - NGW) { drain(nst); nst = 0; }
        };
        const float *const glj = gl + j * 4;
        const int2 *ep = ent + e;
        while (__ballot(nrem > 0)) {
          int2 en[U];
          float4 gg[U][NV];
#pragma unroll
          for (int u = 0; u < U; ++u) en[u] = ep[u];    // (past my share: the next group's entries or the sentinels)
#pragma unroll
          for (int u = 0; u < U; ++u)
#pragma unroll
            for (int v = 0; v < NV; ++v)
              gg[u][v] = *reinterpret_cast<const float4 *>(glj + (static_cast<unsigned>(en[u].x) >> 23) * kGvGlStride + v * (LG * 4));
#pragma unroll
          for (int u = 0; u < U; ++u) {
            const bool act = u < nrem;
            const int px = act ? (en[u].x & 0x7fffff) : cur;
            const float kk = act ? __int_as_float(en[u].y) : 0.f;
            const bool chg = px != cur;
            const unsigned long long cm = __ballot(chg);
            if (cm) {
              park(chg, cm);
              if (chg) {
#pragma unroll
                for (int v = 0; v < NV; ++v) acc[v] = make_float4(0.f, 0.f, 0.f, 0.f);
                cur = px;
              }
            }
#pragma unroll
            for (int v = 0; v < NV; ++v) {
              acc[v].x = fmaf(kk, gg[u][v].x, acc[v].x);
              acc[v].y = fmaf(kk, gg[u][v].y, acc[v].y);
              acc[v].z = fmaf(kk, gg[u][v].z, acc[v].z);
              acc[v].w = fmaf(kk, gg[u][v].w, acc[v].w);
            }
          }
          ep += nrem > U ? U : 0;                      // (a finished group keeps re-reading entries that exist: what follows
          nrem -= U;                                   // the sentinels is not an entry, and 0 x garbage may be a NaN)
        }
        {
          const bool fl = cur >= 0;
          const unsigned long long fm = __ballot(fl);
          if (fm) park(fl, fm);
        }
        drain(nst);
        lds_barrier();                                 // the stage is the counters' space: the next level zeroes it
      }
#else
      // (round 2: a half-wave per share, lane c = channel c)
      {
        const int e0 = static_cast<int>(static_cast<long>(total) * half / (THREADS / 32));
        const int e1 = static_cast<int>(static_cast<long>(total) * (half + 1) / (THREADS / 32));
        int cur = -1;
        float acc = 0.f;
        // 8 entries at a time: their LDS reads (entry, then grad_out of its row) are issued together,
        // the run logic walks them in order (entries past the end repeat the last one with a zero
        // coefficient: same pixel, no extra flush)
        for (int e = e0; e < e1; e += 8) {
          int2 en[8];
          float gg[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const bool in = e + j < e1;
            en[j] = ent[in ? e + j : e1 - 1];
            if (!in) en[j].y = 0;
          }
#pragma unroll
          for (int j = 0; j < 8; ++j) gg[j] = gl[(en[j].x >> 23) * kGvGlStride + c];
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const int px = en[j].x & 0x7fffff;
            if (px != cur) {
              if (cur >= 0) {
                gv_flush(gv + cur * pix_stride, acc);
                if constexpr (PROF) { if (c == 0) atomicAdd(&s.prof[7], 1ULL); }
              }
              cur = px;
              acc = 0.f;
            }
            acc = fmaf(__int_as_float(en[j].y), gg[j], acc);
          }
        }
        if (cur >= 0) gv_flush(gv + cur * pix_stride, acc);
      }
#endif
      GV_TICK(4)
      // (the next level's placement is separated from these reads by the barriers of its steps 1-3)
    }
    r0 = r1;
  }
  GV_TICK(6)
}
#undef GV_TICK

}  // namespace bevmsda
