// Per-frame geometry of the BEV encoder, on the device, with no host synchronisation.
//
// What the reference does on the host side of every SpatialCrossAttention call and once per
// encoder call:
//   * BEVFormerEncoder.point_sampling (bevformer/modules/encoder.py:88-149): project the D pillar
//     anchors of every BEV query into every camera, visibility mask;
//   * SpatialCrossAttention.forward (spatial_cross_attention.py:136-153): per camera
//     `mask_per_img[0].sum(-1).nonzero()` (a host sync per camera and per layer), padded rebatch;
//   * ibid. :169-172: number of cameras that see a query, clamped to >= 1.
// Here: three small launches per frame.  plan_project_kernel does the projection for one BEV query
// and all cameras per thread (in the SAME fp32 operation order as geometry.point_sampling: every
// product and sum rounded separately, no fused multiply-add), writes reference_points_cam,
// bev_mask, 1 / camera count, a per-(camera, position) visibility byte and per-block visible
// counts; plan_scan_kernel + plan_rows_kernel turn those into the ragged row list — row_query,
// row_batch, row_ref, the per-query row table — by a prefix sum per camera, and leave the row
// COUNT in device memory: the sampling kernel reads it there (FusedArgs::nrows), so a frame with new
// camera matrices needs no `.item()` / `nonzero()` and a captured HIP graph of the step stays
// valid when the number of rows changes (everything is sized by `row_capacity`).
//
// Row order inside a camera = the caller's `order` (position -> query).  The package passes a
// STATIC polar Z-order of the BEV grid (azimuth / inverse range around the ego origin, where the
// cameras sit): neighbouring positions project to neighbouring pixels in whichever camera sees
// them, so the order needs no per-frame sort (modules/geometry.py, polar_order).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace bevmsda {

struct PlanArgs {
  const float *l2i;       // (B, Nc, 4, 4) lidar2img, fp32
  const float *ref3d;     // (B, D, Q, 3) normalised pillar anchors (encoder.py:62-71)
  const int32_t *order;   // (Q,) position -> query
  float *ref_cam;         // (Nc, B, Q, D, 2)
  uint8_t *bev_mask;      // (Nc, B, Q, D)
  float *inv_count;       // (B, Q)
  uint8_t *slot;          // (Nc, Q) by POSITION: 1 + (number of lower cameras that see the query), 0 = not seen
  int32_t *row_query;     // (cap,)  j * Qt + (q - q_lo), Qt = q_hi - q_lo (tile-local slot index)
  int32_t *row_batch;     // (cap,)  j * Nc + cam
  float *row_ref;         // (cap, D, 2)
  int32_t *q_rows;        // (B * Qt, Nc) rows of every tile-local slot, increasing, -1 padded
  int32_t *q_rows2;       // (B * Qt, 2)  first two entries of the above (A-load of the output projection)
  int32_t *block_counts;  // (Nc, nblk) visible positions of every block of 256 positions (scratch)
  int32_t *block_base;    // (Nc, nblk) first row (batch element 0) of every block (scratch)
  int32_t *counters;      // [0] rows R, [1] rows dropped (capacity), [2] (j, q) with more than two cameras,
                          // [3] rows of one batch element, [4 .. 4 + B * Nc] first row of every (j, cam) run
  float sx, ox, sy, oy, sz, oz;   // de-normalisation: x = p * sx + ox (pc_range, encoder.py:102-107)
  float img_w, img_h;
  int B, Nc, Q, D;
  int q_lo, q_hi;         // only queries in [q_lo, q_hi) make rows (BEV tiling); ref_cam / bev_mask cover all
  int cap;
};

constexpr int kPlanMaxCams = 16;
constexpr int kPlanMaxAnchors = 8;

__device__ __forceinline__ void plan_project_one(const PlanArgs &a, int i, int q, int *cnt) {
#pragma clang fp contract(off)
  const bool mine = q >= a.q_lo && q < a.q_hi;
  const float eps = 1e-5f;
  for (int j = 0; j < a.B; ++j) {
    float px[kPlanMaxAnchors], py[kPlanMaxAnchors], pz[kPlanMaxAnchors];
    for (int d = 0; d < a.D; ++d) {
      const float *p = a.ref3d + (static_cast<long>(j * a.D + d) * a.Q + q) * 3;
      // plain operators under `fp contract(off)`: every product and sum is rounded on its own (the
      // __fmul_rn / __fadd_rn wrappers are inlined WITH their own contract flags and fuse into FMAs)
      const float tx = p[0] * a.sx, ty = p[1] * a.sy, tz = p[2] * a.sz;
      px[d] = tx + a.ox;
      py[d] = ty + a.oy;
      pz[d] = tz + a.oz;
    }
    int seen = 0;
    for (int cam = 0; cam < a.Nc; ++cam) {
      const float *m = a.l2i + static_cast<long>(j * a.Nc + cam) * 16;
      bool any = false;
      float *rc = a.ref_cam + ((static_cast<long>(cam) * a.B + j) * a.Q + q) * a.D * 2;
      uint8_t *bm = a.bev_mask + ((static_cast<long>(cam) * a.B + j) * a.Q + q) * a.D;
      for (int d = 0; d < a.D; ++d) {
        float c[3];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
          const float t0 = m[r * 4 + 0] * px[d], t1 = m[r * 4 + 1] * py[d], t2 = m[r * 4 + 2] * pz[d];
          float s = t0 + t1;
          s = s + t2;
          c[r] = s + m[r * 4 + 3];
        }
        const float depth = c[2];
        const float den = (depth > eps || depth != depth) ? depth : eps;   // torch.clamp(min=eps), NaN kept
        const float u = (c[0] / den) / a.img_w;       // IEEE division (hipcc's default: correctly rounded)
        const float v = (c[1] / den) / a.img_h;
        const bool ok = depth > eps && v > 0.f && v < 1.f && u < 1.f && u > 0.f;
        rc[2 * d] = u;
        rc[2 * d + 1] = v;
        bm[d] = ok ? 1 : 0;
        any |= ok;
      }
      if (j == 0) {       // the visible set of a camera comes from batch element 0 (spatial_cross_attention.py:139)
        a.slot[static_cast<long>(cam) * a.Q + i] = (any && mine) ? static_cast<uint8_t>(1 + seen) : 0;
        if (any && mine) atomicAdd(&cnt[cam], 1);
      }
      seen += any ? 1 : 0;
    }
    a.inv_count[static_cast<long>(j) * a.Q + q] = 1.0f / static_cast<float>(seen > 1 ? seen : 1);
    if (mine) {
      const long sl = static_cast<long>(j) * (a.q_hi - a.q_lo) + (q - a.q_lo);
      int32_t *qr = a.q_rows + sl * a.Nc;
      for (int s = 0; s < a.Nc; ++s) qr[s] = -1;
      a.q_rows2[sl * 2] = -1;
      a.q_rows2[sl * 2 + 1] = -1;
    }
  }
}

__global__ void __launch_bounds__(256) plan_project_kernel(const PlanArgs a) {
#pragma clang fp contract(off)
  __shared__ int cnt[kPlanMaxCams];
  if (threadIdx.x < kPlanMaxCams) cnt[threadIdx.x] = 0;
  __syncthreads();
  const int i = blockIdx.x * 256 + threadIdx.x;
  const bool live = i < a.Q;
  const int q = live ? a.order[i] : 0;
  if (live) plan_project_one(a, i, q, cnt);
  __syncthreads();
  if (threadIdx.x < a.Nc) a.block_counts[threadIdx.x * gridDim.x + blockIdx.x] = cnt[threadIdx.x];
}

// plan_project_kernel also leaves, per camera and per block of 256 positions, the number of visible
// queries (block_counts[cam * nblk + blk]); plan_scan_kernel (one small workgroup) turns them into the
// first row of every (camera, block) and the counters; plan_rows_kernel (one thread per camera and
// position) writes the rows: position i of camera c goes to row block_base[c, i / 256] + (visible
// positions before i inside its block), times the batch entries.
__global__ void __launch_bounds__(256) plan_scan_kernel(const PlanArgs a, int nblk) {
  __shared__ int lds[256];
  __shared__ int cam_first[kPlanMaxCams + 1];
  const int tid = threadIdx.x;
  int run = 0;
  for (int cam = 0; cam < a.Nc; ++cam) {
    if (tid == 0) cam_first[cam] = run;
    // exclusive scan of this camera's nblk block counts, 256 at a time
    for (int b0 = 0; b0 < nblk; b0 += 256) {
      const int b = b0 + tid;
      const int v = b < nblk ? a.block_counts[cam * nblk + b] : 0;
      __syncthreads();
      lds[tid] = v;
      __syncthreads();
      for (int o = 1; o < 256; o <<= 1) {
        const int t = tid >= o ? lds[tid - o] : 0;
        __syncthreads();
        lds[tid] += t;
        __syncthreads();
      }
      if (b < nblk) a.block_base[cam * nblk + b] = run + lds[tid] - v;
      run += lds[255];
    }
  }
  const int R0 = run;
  if (tid == 0) {
    cam_first[a.Nc] = R0;
    const long Rl = static_cast<long>(R0) * a.B;
    a.counters[0] = Rl < a.cap ? static_cast<int>(Rl) : a.cap;
    a.counters[1] = Rl > a.cap ? static_cast<int>(Rl - a.cap) : 0;
    a.counters[3] = R0;
  }
  __syncthreads();
  if (tid <= a.Nc) {
    for (int j = 0; j < a.B; ++j)
      if (tid < a.Nc || j == a.B - 1) {
        const long v = static_cast<long>(j) * R0 + cam_first[tid];
        a.counters[4 + j * a.Nc + tid] = v < a.cap ? static_cast<int>(v) : a.cap;
      }
  }
}

__global__ void __launch_bounds__(256) plan_rows_kernel(const PlanArgs a, int nblk) {
  __shared__ int wave_cnt[4];
  const int cam = blockIdx.x / nblk, blk = blockIdx.x - cam * nblk;
  const int i = blk * 256 + threadIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int s = i < a.Q ? a.slot[static_cast<long>(cam) * a.Q + i] : 0;
  const unsigned long long bal = __ballot(s != 0);
  const int before = __popcll(bal & ((1ULL << lane) - 1ULL));
  if (lane == 0) wave_cnt[wave] = __popcll(bal);
  __syncthreads();
  int off = before;
  for (int w = 0; w < wave; ++w) off += wave_cnt[w];
  if (!s) return;
  const int R0 = a.counters[3];
  const int r = a.block_base[cam * nblk + blk] + off;
  const int q = a.order[i];
  for (int j = 0; j < a.B; ++j) {
    const long rr = static_cast<long>(j) * R0 + r;
    if (rr >= a.cap) continue;
    const long sl = static_cast<long>(j) * (a.q_hi - a.q_lo) + (q - a.q_lo);
    a.row_query[rr] = static_cast<int>(sl);
    a.row_batch[rr] = j * a.Nc + cam;
    const float *src = a.ref_cam + ((static_cast<long>(cam) * a.B + j) * a.Q + q) * a.D * 2;
    float *dst = a.row_ref + rr * a.D * 2;
    for (int d = 0; d < a.D * 2; ++d) dst[d] = src[d];
    a.q_rows[sl * a.Nc + (s - 1)] = static_cast<int>(rr);
    if (s <= 2) a.q_rows2[sl * 2 + (s - 1)] = static_cast<int>(rr);
    if (s == 3) atomicAdd(&a.counters[2], 1);
  }
}

// rows[q_rows[s, 0]] += sum_{j >= 2} rows[q_rows[s, j]] for the (rare) slots seen by more than two
// cameras, so that the two-row gather of the output projection (linear_mfma.h, gather mode) still
// yields the sum over ALL cameras (spatial_cross_attention.py:165-167).  `n_extra` (device) = number
// of such slots: the launch exits at once when it is 0.
__global__ void __launch_bounds__(256) fold_extra_rows_kernel(float *rows, long ld, const int32_t *q_rows, long slots,
                                                              int J, int C, const int32_t *n_extra) {
  if (*n_extra == 0) return;
  const long t = static_cast<long>(blockIdx.x) * 256 + threadIdx.x;
  const int c4 = C / 4;
  const long s = t / c4;
  if (s >= slots) return;
  const int32_t *qr = q_rows + s * J;
  if (qr[2] < 0) return;
  const int col = static_cast<int>(t - s * c4) * 4;
  float4 acc = *reinterpret_cast<const float4 *>(rows + qr[0] * ld + col);
  for (int j = 2; j < J; ++j) {
    const int r = qr[j];
    if (r < 0) break;
    const float4 v = *reinterpret_cast<const float4 *>(rows + r * ld + col);
    acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
  }
  *reinterpret_cast<float4 *>(rows + qr[0] * ld + col) = acc;
}

}  // namespace bevmsda
