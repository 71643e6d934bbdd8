// RETIRED (round 3): cut out of bevformer_amd/csrc/msda_d32.h; last built at commit 936a008 (round 2).
// Measured 259.8 us against 237.8 us for the default kernel (profiles/r1/r1o_*): record only, not compiled.
// ------------------------------------------------------------------ coarse level from LDS
// SpatialCrossAttention variant of the fused kernel with the LAST (coarsest) feature level of
// one (camera, head) staged in LDS (north-star design point "feature maps staged through LDS
// tiles"): a block owns a run of rows of ONE camera and ONE head, copies that head's slice of
// the level (H_l * W_l pixels x 128 bytes: 48 KB at base, level 3 = 15 x 25) into LDS once and
// serves every bilinear tap of that level with ds_read_b128 — a 4x higher address rate than the
// vector-memory path that bounds the kernel — while the finer levels, which do
// not fit (185 KB for level 2), keep the buffer-load path.  Rows must be grouped by camera
// (`row_batch` non-decreasing: the frame plan's row list is); `cam_start` (N + 1) gives the
// row range of every camera.  fp32, PT = 8, one queue entry, pillar-anchor references.
// Broadcast point j's parameters inside the lane group and read its four taps from the staged
// level (compile-time j for the swizzle patterns).  A tap outside the map has coefficient 0:
// any staged pixel will do (clamped index); p.off == kOobOffset (point outside) clamps as well,
// all four coefficients are 0 then.
template <int j, int CNT>
struct LdsPoints {
  static __device__ __forceinline__ void run(const PointParams &p, const float *lds_level, int lig, int WL,
                                             int last_px, f32x4 (&v)[CNT][4], float (&k)[CNT][4]) {
    const int i00 = static_cast<int>(bcast8<j>(p.off));     // PIXEL index of the top-left tap
    k[j][0] = bcast8<j>(p.k00); k[j][1] = bcast8<j>(p.k01);
    k[j][2] = bcast8<j>(p.k10); k[j][3] = bcast8<j>(p.k11);
    const int c00 = min(max(i00, 0), last_px), c01 = min(max(i00 + 1, 0), last_px);
    const int c10 = min(max(i00 + WL, 0), last_px), c11 = min(max(i00 + WL + 1, 0), last_px);
    v[j][0] = *reinterpret_cast<const f32x4 *>(&lds_level[c00 * 32 + lig * 4]);
    v[j][1] = *reinterpret_cast<const f32x4 *>(&lds_level[c01 * 32 + lig * 4]);
    v[j][2] = *reinterpret_cast<const f32x4 *>(&lds_level[c10 * 32 + lig * 4]);
    v[j][3] = *reinterpret_cast<const f32x4 *>(&lds_level[c11 * 32 + lig * 4]);
    if constexpr (j + 1 < CNT) LdsPoints<j + 1, CNT>::run(p, lds_level, lig, WL, last_px, v, k);
  }
};

struct FusedLdsArgs {
  FusedArgs f;
  const int32_t *cam_start;   // (N + 1) first row of every camera's run
  int rows_per_block;         // multiple of 32
  int chunks;                 // blocks per (camera, head): ceil(max camera rows / rows_per_block)
  int lds_pixels;             // H * W of the last level
};

template <int WPE>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WPE, WPE)))
msda_fused_d32_ldslevel_kernel(const FusedLdsArgs g) {
  constexpr int D = 32, PT = 8;
  extern __shared__ __attribute__((aligned(16))) float lds_level[];   // [pixel][32 channels] of (camera, head)
  const FusedArgs &f = g.f;
  const KArgs &a = f.k;
  const int lig = threadIdx.x & 7;
  const int b = blockIdx.x;
  const int ch = b % g.chunks;
  const int m = (b / g.chunks) % a.M;
  const int cam = b / (g.chunks * a.M);
  const long r0 = static_cast<long>(g.cam_start[cam]) + static_cast<long>(ch) * g.rows_per_block;
  const long rend = g.cam_start[cam + 1];
  if (r0 >= rend) return;                                   // whole block: before any barrier
  const long r1 = r0 + g.rows_per_block < rend ? r0 + g.rows_per_block : rend;
  const int L = a.L, LL = L - 1;
  const uint32_t pix_bytes = static_cast<uint32_t>(a.M) * D * sizeof(float);

  // stage value[cam, lstart[LL] + p, m, :] for every pixel p of the last level
  {
    const float *src = static_cast<const float *>(a.value) +
                       ((static_cast<long>(cam) * a.S + a.lstart[LL]) * a.M + m) * D;
    for (int i = threadIdx.x; i < g.lds_pixels * 8; i += 256) {
      const int px = i >> 3, q4 = (i & 7) * 4;
      *reinterpret_cast<float4 *>(&lds_level[px * D + q4]) =
          *reinterpret_cast<const float4 *>(src + static_cast<long>(px) * a.M * D + q4);
    }
  }
  __syncthreads();

  const uint32_t lane_term = lig * 4 * static_cast<uint32_t>(sizeof(float));
  const uint32_t total_bytes = static_cast<uint32_t>(static_cast<unsigned long long>(a.N) * a.S * pix_bytes);
  __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(a.value), 0,
                                                                  static_cast<int>(total_bytes), 0x00020000);
  const uint32_t head_base = static_cast<uint32_t>((static_cast<unsigned long long>(cam) * a.S * a.M + m) * D * sizeof(float));
  const int HL = static_cast<int>(a.shapes[2 * LL]), WL = static_cast<int>(a.shapes[2 * LL + 1]);
  const int last_px = HL * WL - 1;

  for (long rr = r0 + (threadIdx.x >> 3); rr - (threadIdx.x >> 3) < r1; rr += 32) {   // uniform trip count
    const bool active = rr < r1;
    const long r = active ? rr : r1 - 1;                 // whole groups stay alive for the swizzles
    const long rs = f.row_src ? static_cast<long>(f.row_src[r]) : r;
    const float *__restrict__ lgp = f.logits + rs * f.proj_row + m * f.lg_head + lig;
    const float2 *__restrict__ ofp = reinterpret_cast<const float2 *>(f.offs + rs * f.proj_row + m * f.off_head) + lig;
    const float2 *__restrict__ rfp = reinterpret_cast<const float2 *>(f.ref) + r * f.A;
    float e0 = lgp[0];
    float e1 = L > 1 ? lgp[PT] : -INFINITY;
    float e2 = L > 2 ? lgp[2 * PT] : -INFINITY;
    float e3 = L > 3 ? lgp[3 * PT] : -INFINITY;
    float2 of = ofp[0];
    const float2 rf = rfp[lig % f.A];
    const float mx = lanes_max<PT>(fmaxf(fmaxf(e0, e1), fmaxf(e2, e3)));
    e0 = expf(e0 - mx); e1 = expf(e1 - mx); e2 = expf(e2 - mx); e3 = expf(e3 - mx);
    const float sum = lanes_sum<PT>((e0 + e1) + (e2 + e3));

    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int l = 0; l < LL; ++l) {                       // finer levels: vector-memory path
      const int H = static_cast<int>(a.shapes[2 * l]), W = static_cast<int>(a.shapes[2 * l + 1]);
      const uint32_t lbytes = static_cast<uint32_t>(a.lstart[l]) * pix_bytes;
      const float lx = rf.x + of.x / static_cast<float>(W);
      const float ly = rf.y + of.y / static_cast<float>(H);
      const float e = l == 0 ? e0 : (l == 1 ? e1 : e2);
      const PointParams p = point_params(lx, ly, active ? e / sum : 0.f, H, W, head_base + lbytes, pix_bytes);
      of = ofp[(l + 1) * PT];
      sample_points<0, PT, float>(p, rsrc, lane_term, pix_bytes, static_cast<uint32_t>(W) * pix_bytes, acc);
    }
    {                                                    // last level: taps from LDS
      const float lx = rf.x + of.x / static_cast<float>(WL);
      const float ly = rf.y + of.y / static_cast<float>(HL);
      const float e = LL == 0 ? e0 : (LL == 1 ? e1 : (LL == 2 ? e2 : e3));
      // same coefficients as the global path; `off` here is the PIXEL index of the top-left tap
      const PointParams p = point_params(lx, ly, active ? e / sum : 0.f, HL, WL, 0u, 1u);
      f32x4 v[PT][4];
      float k[PT][4];
      LdsPoints<0, PT>::run(p, lds_level, lig, WL, last_px, v, k);
#pragma unroll
      for (int j = 0; j < PT; ++j)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          acc[0] = fmaf(k[j][t], v[j][t][0], acc[0]);
          acc[1] = fmaf(k[j][t], v[j][t][1], acc[1]);
          acc[2] = fmaf(k[j][t], v[j][t][2], acc[2]);
          acc[3] = fmaf(k[j][t], v[j][t][3], acc[3]);
        }
    }
    if (active) {
      float *op = static_cast<float *>(a.out) + (r * a.M + m) * D + lig * 4;
      *reinterpret_cast<float4 *>(op) = make_float4(acc[0], acc[1], acc[2], acc[3]);
    }
  }
}

