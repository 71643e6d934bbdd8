// Row-wise HBM-bound helpers of the encoder layer for CDNA4 (gfx950).
//
//   add_layernorm:  y = LayerNorm(x + res) * gamma + beta      the "attention/FFN output
//       + identity, then norm" pair of every BEVFormerLayer step (encoder.py:360-404 run
//       'self_attn','norm','cross_attn','norm','ffn','norm'); in the reference two
//       launches that each stream the (Q, C) grid — here one pass: 2 reads + 1 write.
//   gather_mean:    out[q] = scale[q] * sum_j rows[idx[q, j]]  the per-camera
//       scatter-add + division of SpatialCrossAttention (spatial_cross_attention.py:165-172)
//       turned into a gather over the <= J cameras that see query q (no atomics, no
//       zero-fill of the slots tensor).
//
// One wavefront per row, 16-byte accesses, reductions with DPP-free xor swizzles /
// bpermute across the 64 lanes; 4 rows per 256-thread block.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "scalar_ops.h"

namespace bevmsda {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int s = 32; s >= 1; s >>= 1) v += __shfl_xor(v, s, 64);
  return v;
}

// C = 4 * 64 * VPL floats per row (VPL float4 per lane); rows handled by waves.
template <int VPL>
__global__ void __launch_bounds__(256) add_layernorm_kernel(const float *__restrict__ x,
                                                           const float *__restrict__ res,
                                                           const float *__restrict__ gamma,
                                                           const float *__restrict__ beta, float eps,
                                                           long rows, float *__restrict__ out) {
  constexpr int C = 256 * VPL;
  const int lane = threadIdx.x & 63;
  const long row = static_cast<long>(blockIdx.x) * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float4 *xp = reinterpret_cast<const float4 *>(x + row * C);
  const float4 *rp = res ? reinterpret_cast<const float4 *>(res + row * C) : nullptr;
  float4 v[VPL];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    v[i] = xp[lane + 64 * i];
    if (rp) {
      const float4 t = rp[lane + 64 * i];
      v[i].x += t.x; v[i].y += t.y; v[i].z += t.z; v[i].w += t.w;
    }
    s += add_scalar(v[i].x, v[i].y) + add_scalar(v[i].z, v[i].w);     // (scalar_ops.h: no packed add of swapped halves)
  }
  const float mean = wave_sum(s) * (1.0f / C);
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    const float a = v[i].x - mean, b = v[i].y - mean, c = v[i].z - mean, d = v[i].w - mean;
    ss += add_scalar(a * a + b * b, c * c + d * d);
  }
  const float rstd = rsqrtf(wave_sum(ss) * (1.0f / C) + eps);
  const float4 *gp = reinterpret_cast<const float4 *>(gamma);
  const float4 *bp = reinterpret_cast<const float4 *>(beta);
  float4 *op = reinterpret_cast<float4 *>(out + row * C);
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    const float4 g = gp[lane + 64 * i], b = bp[lane + 64 * i];
    float4 o;
    o.x = (v[i].x - mean) * rstd * g.x + b.x;
    o.y = (v[i].y - mean) * rstd * g.y + b.y;
    o.z = (v[i].z - mean) * rstd * g.z + b.z;
    o.w = (v[i].w - mean) * rstd * g.w + b.w;
    op[lane + 64 * i] = o;
  }
}

// Backward of add_layernorm: z = x + res is recomputed (x and res are kept by the caller; nothing else is saved
// by the forward), grad_x = rstd * (g*gamma - mean(g*gamma) - xhat * mean(g*gamma*xhat)) is the gradient of BOTH
// x and res; the column sums sum_rows g * xhat and sum_rows g are written as per-workgroup partial rows
// of a (gridDim.x, 2 C) scratch matrix [gamma sums | beta sums] (`ggamma`; `gbeta` unused), reduced by
// colsum_partials_kernel.
// One wavefront walks rows row0, row0 + W, ... with its column sums in registers; the 4 wavefronts of a block
// combine them through LDS.  3 reads + 1 write of the grid.
template <int VPL>
__global__ void __launch_bounds__(256) add_layernorm_bwd_kernel(const float *__restrict__ x,
                                                               const float *__restrict__ res,
                                                               const float *__restrict__ gamma,
                                                               const float *__restrict__ gout, float eps, long rows,
                                                               float *__restrict__ gx, float *__restrict__ ggamma,
                                                               float *__restrict__ gbeta,
                                                               const float *__restrict__ gout2 = nullptr) {
  constexpr int C = 256 * VPL;
  __shared__ float part[2][4][C];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long nwaves = static_cast<long>(gridDim.x) * 4;
  const float4 *gp = reinterpret_cast<const float4 *>(gamma);
  float4 gam[VPL], dg[VPL], db[VPL];
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    gam[i] = gp[lane + 64 * i];
    dg[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    db[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  for (long row = static_cast<long>(blockIdx.x) * 4 + wave; row < rows; row += nwaves) {
    const float4 *xp = reinterpret_cast<const float4 *>(x + row * C);
    const float4 *rp = res ? reinterpret_cast<const float4 *>(res + row * C) : nullptr;
    const float4 *yp = reinterpret_cast<const float4 *>(gout + row * C);
    const float4 *yp2 = gout2 ? reinterpret_cast<const float4 *>(gout2 + row * C) : nullptr;
    float4 v[VPL], gy[VPL];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
      v[i] = xp[lane + 64 * i];
      gy[i] = yp[lane + 64 * i];
      if (yp2) {                      // the gradient arrives as two addends (a residual branch and a projection's)
        const float4 t = yp2[lane + 64 * i];
        gy[i].x += t.x; gy[i].y += t.y; gy[i].z += t.z; gy[i].w += t.w;
      }
      if (rp) {
        const float4 t = rp[lane + 64 * i];
        v[i].x += t.x; v[i].y += t.y; v[i].z += t.z; v[i].w += t.w;
      }
      s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    }
    const float mean = wave_sum(s) * (1.0f / C);
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
      v[i].x -= mean; v[i].y -= mean; v[i].z -= mean; v[i].w -= mean;
      ss += (v[i].x * v[i].x + v[i].y * v[i].y) + (v[i].z * v[i].z + v[i].w * v[i].w);
    }
    const float rstd = rsqrtf(wave_sum(ss) * (1.0f / C) + eps);
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
      v[i].x *= rstd; v[i].y *= rstd; v[i].z *= rstd; v[i].w *= rstd;           // xhat
      dg[i].x += gy[i].x * v[i].x; dg[i].y += gy[i].y * v[i].y; dg[i].z += gy[i].z * v[i].z; dg[i].w += gy[i].w * v[i].w;
      db[i].x += gy[i].x; db[i].y += gy[i].y; db[i].z += gy[i].z; db[i].w += gy[i].w;
      gy[i].x *= gam[i].x; gy[i].y *= gam[i].y; gy[i].z *= gam[i].z; gy[i].w *= gam[i].w;   // g * gamma
      s1 += (gy[i].x + gy[i].y) + (gy[i].z + gy[i].w);
      s2 += (gy[i].x * v[i].x + gy[i].y * v[i].y) + (gy[i].z * v[i].z + gy[i].w * v[i].w);
    }
    s1 = wave_sum(s1) * (1.0f / C);
    s2 = wave_sum(s2) * (1.0f / C);
    float4 *op = reinterpret_cast<float4 *>(gx + row * C);
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
      float4 o;
      o.x = rstd * (gy[i].x - s1 - v[i].x * s2);
      o.y = rstd * (gy[i].y - s1 - v[i].y * s2);
      o.z = rstd * (gy[i].z - s1 - v[i].z * s2);
      o.w = rstd * (gy[i].w - s1 - v[i].w * s2);
      op[lane + 64 * i] = o;
    }
  }
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    reinterpret_cast<float4 *>(part[0][wave])[lane + 64 * i] = dg[i];
    reinterpret_cast<float4 *>(part[1][wave])[lane + 64 * i] = db[i];
  }
  __syncthreads();
  // per-block partial sums: row blockIdx.x of ggamma / gbeta (gridDim.x, C); the caller adds the rows up
  // (1,024 atomics per address were a third of the kernel's time)
  for (int c = threadIdx.x; c < C; c += 256) {
    const float a = (part[0][0][c] + part[0][1][c]) + (part[0][2][c] + part[0][3][c]);
    const float b = (part[1][0][c] + part[1][1][c]) + (part[1][2][c] + part[1][3][c]);
    ggamma[static_cast<long>(blockIdx.x) * (2 * C) + c] = a;          // (P, 2 C): [gamma sums | beta sums]
    ggamma[static_cast<long>(blockIdx.x) * (2 * C) + C + c] = b;
  }
}

// Second stage of the column sums: out[c] += sum_p parts[p, c] for a (P, C2) matrix of per-workgroup partials
// (C2 = both vectors side by side; `out` zeroed by the launcher).  Block (x, y): 64 columns, rows y*4 + rg,
// + 4 * gridDim.y, ...; the 4 row groups of a block combine through LDS, one atomic per column and block
// (gridDim.y = 16 atomics per address).
__global__ void __launch_bounds__(256) colsum_partials_kernel(const float *__restrict__ parts, long P, int C2,
                                                              float *__restrict__ out) {
  __shared__ float red[4][64];
  const int col = blockIdx.x * 64 + (threadIdx.x & 63), rg = threadIdx.x >> 6;
  float s = 0.f;
  if (col < C2)
    for (long p = blockIdx.y * 4 + rg; p < P; p += 4 * gridDim.y) s += parts[p * C2 + col];
  red[rg][threadIdx.x & 63] = s;
  __syncthreads();
  if (rg == 0 && col < C2)
    unsafeAtomicAdd(out + col, (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]));
}

// rows (R, C); idx (Q, J) int32 row ids, -1 = empty; scale (Q,); out (Q, C); C % 4 == 0.
__global__ void __launch_bounds__(256) gather_mean_kernel(const float *__restrict__ rows,
                                                         const int32_t *__restrict__ idx,
                                                         const float *__restrict__ scale, long Q, int J,
                                                         int C, float *__restrict__ out) {
  const int c4 = C >> 2;                               // float4 per row
  const long t = static_cast<long>(blockIdx.x) * 256 + threadIdx.x;
  if (t >= Q * c4) return;
  const long q = t / c4;
  const int c = static_cast<int>(t - q * c4);
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int j = 0; j < J; ++j) {
    const int r = idx[q * J + j];
    if (r >= 0) {
      const float4 v = reinterpret_cast<const float4 *>(rows + static_cast<long>(r) * C)[c];
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
  }
  const float s = scale[q];
  reinterpret_cast<float4 *>(out + q * C)[c] = make_float4(acc.x * s, acc.y * s, acc.z * s, acc.w * s);
}

// rows[r] = scale[slot] * slots[slot], slot = row_slot[r], for r < the row count (`nrows_dev`: read on the device, else R):
// the backward of the camera mean (gather_mean: every (camera, query) row of a BEV query takes the query's gradient
// times 1 / cameras).  One float4 per thread; rows beyond the count are not touched.
__global__ void __launch_bounds__(256) rows_from_slots_kernel(const float *__restrict__ slots, long ld_slots,
                                                             const float *__restrict__ scale,
                                                             const int32_t *__restrict__ row_slot,
                                                             const int32_t *__restrict__ nrows_dev, long R, int C,
                                                             float *__restrict__ rows) {
  const int c4 = C >> 2;
  const long t = static_cast<long>(blockIdx.x) * 256 + threadIdx.x;
  long n = R;
  if (nrows_dev) {
    const long d = static_cast<long>(*nrows_dev);
    n = d < n ? (d < 0 ? 0 : d) : n;
  }
  if (t >= n * c4) return;
  const long r = t / c4;
  const int c = static_cast<int>(t - r * c4);
  const long sl = row_slot[r];
  const float s = scale[sl];
  const float4 v = reinterpret_cast<const float4 *>(slots + sl * ld_slots)[c];
  reinterpret_cast<float4 *>(rows + r * C)[c] = make_float4(v.x * s, v.y * s, v.z * s, v.w * s);
}

// out[r, :] = bf16(scale * in[r, :]) for r < the row count (device-side `nrows_dev`, else R): the gradient rows of the
// sampling backward in the storage type of its kernels, touching only the rows that exist.  8 floats per thread.
__global__ void __launch_bounds__(256) cast_rows_bf16_kernel(const float *__restrict__ in, const int32_t *__restrict__ nrows_dev,
                                                            long R, int C, float scale, uint16_t *__restrict__ out) {
  const int c8 = C >> 3;
  const long t = static_cast<long>(blockIdx.x) * 256 + threadIdx.x;
  long n = R;
  if (nrows_dev) {
    const long d = static_cast<long>(*nrows_dev);
    n = d < n ? (d < 0 ? 0 : d) : n;
  }
  if (t >= n * c8) return;
  const float4 a = reinterpret_cast<const float4 *>(in)[2 * t], b = reinterpret_cast<const float4 *>(in)[2 * t + 1];
  auto rn = [](float f) -> uint32_t {      // round-to-nearest-even fp32 -> bf16 (NaN kept quiet)
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (u >> 16) | 0x40u;
    return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
  };
  uint4 o;
  o.x = rn(a.x * scale) | (rn(a.y * scale) << 16);
  o.y = rn(a.z * scale) | (rn(a.w * scale) << 16);
  o.z = rn(b.x * scale) | (rn(b.y * scale) << 16);
  o.w = rn(b.z * scale) | (rn(b.w * scale) << 16);
  reinterpret_cast<uint4 *>(out)[t] = o;
}

}  // namespace bevmsda
