// Kernels of the encoder's caller, PerceptionTransformer.get_bev_features
// (projects/mmdet3d_plugin/bevformer/modules/transformer.py:104-200),
// for CDNA4 (gfx950).  Both are HBM-bound row movers over the same (pixels, 256) grids the
// encoder works on.
//
//   rotate_bev:     prev_bev rotated about rotate_center by the ego yaw delta with nearest
//       sampling and zero fill — torchvision.transforms.functional.rotate as called at
//       transformer.py:146-156 (affine grid from pixel centres, grid_sample(nearest, zeros,
//       align_corners=False): un-normalise with ((g + 1) * size - 1) / 2, round half to even).
//       One wavefront per output pixel: the source index is wave-uniform, the 1 KB row is one
//       16-byte access per lane.  1 read + 1 write of the grid.
//   flatten_feats:  camera feature level (bs, Nc, C, h*w) -> rows of feat_flatten
//       (Nc, S, bs, C) with "+ cams_embeds[cam] + level_embeds[lvl]" (transformer.py:165-184):
//       the reference makes four passes (permute view + two adds + cat); here one LDS-tiled
//       transpose: 64-pixel x 64-channel tiles, 256-byte coalesced reads along pixels, 16-byte
//       stores along channels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace bevmsda {

struct RotateArgs {
  const float *src;   // row p of the source grid at src + p * ld_src
  float *dst;
  long ld_src, ld_dst;
  int H, W, C;        // C = 256 * VPL floats per row
  float t00, t01, t02, t10, t11, t12;   // inverse affine matrix rows, already divided by (W/2, H/2)
  const float *theta_dev;               // when set: the six values above are read from DEVICE memory instead (a
                                        // captured launch then follows the pose of every replayed frame)
};

template <int VPL>
__global__ void __launch_bounds__(256) rotate_bev_kernel(const RotateArgs a) {
  const int lane = threadIdx.x & 63;
  const long p = static_cast<long>(blockIdx.x) * 4 + (threadIdx.x >> 6);
  if (p >= static_cast<long>(a.H) * a.W) return;
  const int oy = static_cast<int>(p / a.W), ox = static_cast<int>(p % a.W);
  // base grid of pixel centres: linspace(-W/2 + 0.5, W/2 - 0.5, W) (exact: step 1)
  const float bx = static_cast<float>(ox) + (0.5f - 0.5f * static_cast<float>(a.W));
  const float by = static_cast<float>(oy) + (0.5f - 0.5f * static_cast<float>(a.H));
  const float t00 = a.theta_dev ? a.theta_dev[0] : a.t00, t01 = a.theta_dev ? a.theta_dev[1] : a.t01;
  const float t02 = a.theta_dev ? a.theta_dev[2] : a.t02, t10 = a.theta_dev ? a.theta_dev[3] : a.t10;
  const float t11 = a.theta_dev ? a.theta_dev[4] : a.t11, t12 = a.theta_dev ? a.theta_dev[5] : a.t12;
  // (bx, by, 1) . rescaled_theta, products and sums rounded separately (no contraction)
  const float gx = __fadd_rn(__fadd_rn(__fmul_rn(bx, t00), __fmul_rn(by, t01)), t02);
  const float gy = __fadd_rn(__fadd_rn(__fmul_rn(bx, t10), __fmul_rn(by, t11)), t12);
  const float fx = __fdiv_rn(__fadd_rn(__fmul_rn(__fadd_rn(gx, 1.f), static_cast<float>(a.W)), -1.f), 2.f);
  const float fy = __fdiv_rn(__fadd_rn(__fmul_rn(__fadd_rn(gy, 1.f), static_cast<float>(a.H)), -1.f), 2.f);
  const float rx = rintf(fx), ry = rintf(fy);          // round half to even (nearbyint)
  const bool ok = rx >= 0.f && rx <= static_cast<float>(a.W - 1) && ry >= 0.f && ry <= static_cast<float>(a.H - 1);
  float4 *out = reinterpret_cast<float4 *>(a.dst + p * a.ld_dst);
  if (ok) {
    const long q = static_cast<long>(ry) * a.W + static_cast<long>(rx);
    const float4 *in = reinterpret_cast<const float4 *>(a.src + q * a.ld_src);
#pragma unroll
    for (int i = 0; i < VPL; ++i) out[lane + 64 * i] = in[lane + 64 * i];
  } else {
#pragma unroll
    for (int i = 0; i < VPL; ++i) out[lane + 64 * i] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
}

struct FlattenArgs {
  const float *feat;        // (bs, Nc, C, hw) contiguous
  const float *cams_embeds; // (Nc, C) or nullptr
  const float *level_embed; // (C) or nullptr
  float *out;               // (Nc, S, bs, C)
  int bs, Nc, C, hw, S, s0; // s0: first row of this level inside S
};

// grid: (ceil(hw / 64), C / 64, bs * Nc); 256 threads.  VEC: 16-byte loads along the pixel axis
// (hw % 4 == 0 and a 16-byte aligned level: the two fine levels, 94 % of the bytes).
template <bool VEC>
__global__ void __launch_bounds__(256) flatten_feats_kernel(const FlattenArgs a) {
  __shared__ float tile[64][65];          // [channel][pixel], +1 pad: conflict-free transposed reads
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int p0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
  const int b = blockIdx.z / a.Nc, cam = blockIdx.z % a.Nc;
  const float *src = a.feat + (static_cast<long>(blockIdx.z) * a.C + c0) * a.hw;
  if (VEC) {
    const int pq = (tid & 15) * 4;        // 16 lanes x float4 = one 256-byte row piece
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = (tid >> 4) + 16 * i;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (p0 + pq < a.hw) v = *reinterpret_cast<const float4 *>(src + static_cast<long>(c) * a.hw + p0 + pq);
      tile[c][pq] = v.x; tile[c][pq + 1] = v.y; tile[c][pq + 2] = v.z; tile[c][pq + 3] = v.w;
    }
  } else {
    const int p = p0 + lane;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int c = wave * 16 + i;
      tile[c][lane] = p < a.hw ? src[static_cast<long>(c) * a.hw + p] : 0.f;
    }
  }
  __syncthreads();
  const int c4 = (tid & 15) * 4;
  float4 e = make_float4(0.f, 0.f, 0.f, 0.f);
  if (a.cams_embeds) {
    const float4 v = *reinterpret_cast<const float4 *>(a.cams_embeds + static_cast<long>(cam) * a.C + c0 + c4);
    e = v;
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int pp = (tid >> 4) + 16 * j;
    if (p0 + pp >= a.hw) continue;
    // the reference adds the camera embedding first, then the level embedding
    // (transformer.py:170-173): keep the two roundings
    float4 v = make_float4(tile[c4][pp] + e.x, tile[c4 + 1][pp] + e.y, tile[c4 + 2][pp] + e.z,
                           tile[c4 + 3][pp] + e.w);
    if (a.level_embed) {
      const float4 l = *reinterpret_cast<const float4 *>(a.level_embed + c0 + c4);
      v.x += l.x; v.y += l.y; v.z += l.z; v.w += l.w;
    }
    float *dst = a.out + ((static_cast<long>(cam) * a.S + a.s0 + p0 + pp) * a.bs + b) * a.C + c0 + c4;
    *reinterpret_cast<float4 *>(dst) = v;
  }
}

}  // namespace bevmsda
