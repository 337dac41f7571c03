// Fused image-to-voxel unprojection (projection + nearest gather + view mean + zero fill + valid
// mask) for a whole batch in one launch.  Replaces mmdet3d/models/detectors/imvoxelnet.py:58-76,
// :132-141 (get_points) and :145-160 (backproject); see include/imvoxel.h.
//
// HBM-bound: the volume [B,X,Y,Z,C] is written exactly once (coalesced: a voxel's C channels are
// contiguous and consecutive voxels are consecutive in memory), the feature maps are read through
// L2 / Infinity Cache (neighbouring voxels hit neighbouring pixels), nothing else is materialised
// (the reference materialises [V,C,N] and makes four more passes over it).
//
// Work split: a group of LPV lanes (power of two, >= ceil(C/VEC) up to 64) owns one voxel; each
// lane owns VEC consecutive channels (float4 when C % 4 == 0).  The per-view projection is spread
// over the group's lanes -- lane g projects view (chunk*LPV + g) -- and the resulting pixel
// offsets are broadcast with wavefront shuffles, so a 50-view scene costs one projection per
// lane per LPV views instead of one per lane per view.
//
// Arithmetic parity (measured against the imported reference, tests/golden/backproject_cases.npz):
//   point   = float(idx) * voxel_size + new_origin                (fp32 mul, fp32 add, no FMA)
//   (u,v,w) = fma(P3,1, fma(P2,z, fma(P1,y, P0*x)))               (what torch.bmm does on CPU)
//   x = rint(u / w), y = rint(v / w)                               (IEEE divide, round-half-even)
//   valid   = 0 <= x < w_crop  &&  0 <= y < h_crop  &&  w > 0      (tested on the rounded floats:
//             NaN/inf fail, as do the INT64_MIN values the reference's .long() produces for them)
//   mean    = (sum over valid views in view order) / float(count)  (IEEE divide), 0 if count == 0
// This file is compiled with -ffp-contract=off.
#include "ivx_common.h"
#include <stdlib.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct BpParams {
  const float *feat;        // [B*V, FH, FW, C]
  const float *proj;        // [B, V, 12]
  const float *new_origin;  // [B, 3]
  const int *crop_hw;       // [B, 2]
  float *volume;            // [B, N, C]
  uint8_t *valid;           // [B, N]
  int *count;               // [B, N]  (sum mode: number of views that saw the voxel)
  float vs0, vs1, vs2;
  int V, FH, FW, C;
  int X, Y, Z;
  int N;        // X*Y*Z
  int nchunk;   // ceil(C / VEC)
  int lpv_log2; // lanes per voxel = 1 << lpv_log2
  float *pmax;  // single-view lift only: per-workgroup max |volume| goes to pmax[blockIdx.y * gridDim.x + blockIdx.x], or NULL
  int nblk, q;  // multi-view kernel: voxel blocks per sample and per XCD (grid.x = 8 * q, see the kernel's block order); q = 0: plain order (A/B)
};

// MEAN = true: the reference's view mean + valid mask.  MEAN = false (view-sharded multi-GPU mode): the raw sum over
// this rank's views and the per-voxel view count, to be all-reduced and normalised by volume_normalize_kernel.
// T = float (the reference's precision) or __bf16 (optional storage mode: features / volume stored as bf16, the view sum
// and the division in fp32, one rounding at the store; VEC 4 only).
template <int VEC, bool MEAN = true, typename T = float>
__global__ __launch_bounds__(256) void backproject_mean_kernel(const BpParams p) {
  typedef T tv4 __attribute__((ext_vector_type(4)));
  const int b = blockIdx.y;
  const int lpv = 1 << p.lpv_log2;
  const int lane = threadIdx.x & 63;
  const int g = lane & (lpv - 1);        // lane inside the voxel group
  const int gbase = lane & ~(lpv - 1);   // first lane of the group inside the wave
  const int vox_per_block = 256 >> p.lpv_log2;
  // Optional XCD-contiguous block order (IVX_BP_ORDER=1; round 5, the round-4 verdict's item 5): workgroup w runs on XCD w % 8 (observed
  // dispatch rule) and grid.x = 8 * q, so XCD x gets the x-th contiguous eighth of a sample's voxel blocks -- an x-slab of the volume.
  // MEASURED AND NOT ADOPTED (profiles/r05_unprojection_block_order.md): ScanNet, 50 views, 80x80x32: 0.300 -> 0.370 ms per 2 scenes,
  // FETCH_SIZE 871 -> 928 MB per scene, L2 hits -29 %.  An x-slab is seen by all 50 views (a band of ~1 MB of each 4.9 MB map: 50 MB per
  // XCD against 4 MB of L2), neighbouring voxels land 4-5 feature pixels apart, and the reuse that exists -- voxels along one camera ray --
  // is far apart in any voxel order; dealt round-robin the eight XCDs at least walk the same region at the same time, so the Infinity
  // Cache serves their common lines once.  The values do not depend on the order (bit-exact tests run both).
  const int vb = p.q ? (int)(blockIdx.x & 7) * p.q + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
  if (vb >= p.nblk) return;              // padding of the last XCD's range (whole workgroups; no barrier in this kernel)
  const int n = vb * vox_per_block + (threadIdx.x >> p.lpv_log2);
  const bool active = n < p.N;
  const int nn = active ? n : p.N - 1;

  // voxel coordinates: n = (i*Y + j)*Z + k
  const int k = nn % p.Z;
  const int t = nn / p.Z;
  const int j = t % p.Y;
  const int i = t / p.Y;
  const float *no = p.new_origin + b * 3;
  const float px = __fadd_rn(__fmul_rn((float)i, p.vs0), no[0]);
  const float py = __fadd_rn(__fmul_rn((float)j, p.vs1), no[1]);
  const float pz = __fadd_rn(__fmul_rn((float)k, p.vs2), no[2]);
  const int hc = min(p.crop_hw[b * 2 + 0], p.FH), wc = min(p.crop_hw[b * 2 + 1], p.FW);   // slice semantics (detectors/imvoxelnet.py:69): clamped to the map

  constexpr int MAXCH = 4;  // channel chunks per lane when ceil(C/VEC) > 64 lanes (C up to 1024 for VEC 4)
  float acc[MAXCH][VEC];
#pragma unroll
  for (int q = 0; q < MAXCH; ++q)
#pragma unroll
    for (int e = 0; e < VEC; ++e) acc[q][e] = 0.f;
  int cnt = 0;

  for (int v0 = 0; v0 < p.V; v0 += lpv) {
    // lane g projects view v0 + g
    const int v = v0 + g;
    int off = -1;
    if (v < p.V) {
      const float *P = p.proj + ((size_t)b * p.V + v) * 12;
      float u = __fmul_rn(P[0], px);
      u = __fmaf_rn(P[1], py, u);
      u = __fmaf_rn(P[2], pz, u);
      u = __fmaf_rn(P[3], 1.0f, u);
      float w_ = __fmul_rn(P[4], px);
      w_ = __fmaf_rn(P[5], py, w_);
      w_ = __fmaf_rn(P[6], pz, w_);
      w_ = __fmaf_rn(P[7], 1.0f, w_);
      float d = __fmul_rn(P[8], px);
      d = __fmaf_rn(P[9], py, d);
      d = __fmaf_rn(P[10], pz, d);
      d = __fmaf_rn(P[11], 1.0f, d);
      const float xr = rintf(__fdiv_rn(u, d));
      const float yr = rintf(__fdiv_rn(w_, d));
      const bool ok = (xr >= 0.f) && (yr >= 0.f) && (xr < (float)wc) && (yr < (float)hc) && (d > 0.f);
      if (ok) off = (((b * p.V + v) * p.FH + (int)yr) * p.FW + (int)xr);
    }
    const int nv = (p.V - v0) < lpv ? (p.V - v0) : lpv;
    for (int s = 0; s < nv; ++s) {
      const int o = __shfl(off, gbase + s, 64);
      if (o >= 0) {
        ++cnt;
        const T *src = reinterpret_cast<const T *>(p.feat) + (size_t)o * p.C;
#pragma unroll
        for (int q = 0; q < MAXCH; ++q) {
          const int ch = g + q * lpv;
          if (ch < p.nchunk) {
            if constexpr (VEC == 4) {
              const tv4 xr = *reinterpret_cast<const tv4 *>(src + ch * 4);
              const f32x4 x = {(float)xr[0], (float)xr[1], (float)xr[2], (float)xr[3]};
#pragma unroll
              for (int e = 0; e < 4; ++e) acc[q][e] = __fadd_rn(acc[q][e], x[e]);
            } else {
              acc[q][0] = __fadd_rn(acc[q][0], (float)src[ch]);
            }
          }
        }
      }
    }
  }

  if (!active) return;
  const float dn = (float)cnt;
  T *dst = reinterpret_cast<T *>(p.volume) + ((size_t)b * p.N + n) * p.C;
#pragma unroll
  for (int q = 0; q < MAXCH; ++q) {
    const int ch = g + q * lpv;
    if (ch < p.nchunk) {
      if constexpr (VEC == 4) {
        tv4 y;
#pragma unroll
        for (int e = 0; e < 4; ++e) y[e] = (T)(MEAN ? (cnt ? __fdiv_rn(acc[q][e], dn) : 0.f) : acc[q][e]);
        *reinterpret_cast<tv4 *>(dst + ch * 4) = y;
      } else {
        dst[ch] = (T)(MEAN ? (cnt ? __fdiv_rn(acc[q][0], dn) : 0.f) : acc[q][0]);
      }
    }
  }
  if (g == 0) {
    if (MEAN)
      p.valid[(size_t)b * p.N + n] = cnt > 0 ? 1 : 0;
    else
      p.count[(size_t)b * p.N + n] = cnt;
  }
}

// volume[b,n,:] = count ? sum / count : 0 (in place), valid = count > 0 (detectors/imvoxelnet.py:70-74 after the
// all-reduce of the per-rank partial sums).  One float4 per thread.
__global__ __launch_bounds__(256) void volume_normalize_kernel(float *volume, const int *count, uint8_t *valid, long long total4, int C4) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (long long)gridDim.x * blockDim.x) {
    const long long vox = i / C4;
    const int cnt = count[vox];
    f32x4 x = *reinterpret_cast<f32x4 *>(volume + i * 4);
    const float dn = (float)cnt;
#pragma unroll
    for (int e = 0; e < 4; ++e) x[e] = cnt ? __fdiv_rn(x[e], dn) : 0.f;
    *reinterpret_cast<f32x4 *>(volume + i * 4) = x;
    if (i % C4 == 0) valid[vox] = cnt > 0 ? 1 : 0;
  }
}

// Single-view specialisation (KITTI, SUN RGB-D: V == 1).  With one view the mean is the gathered value itself, so no
// accumulation is needed and the roles flip: a group of LPV lanes owns LPV consecutive voxels, lane g projects voxel
// n0 + g (one projection per lane instead of one per group), then the group walks its voxels and every lane
// copies its VEC channels of voxel n0 + s from the pixel that voxel hit (offset broadcast with __shfl).  Per
// 1 KiB of volume written a wave issues a handful of instructions, so the kernel is bound by the volume write.
template <int VEC>
__global__ __launch_bounds__(256) void backproject_single_view_kernel(const BpParams p) {
  const int b = blockIdx.y;
  const int lpv = 1 << p.lpv_log2;
  const int lane = threadIdx.x & 63;
  const int g = lane & (lpv - 1);
  const int gbase = lane & ~(lpv - 1);
  const int group = (blockIdx.x * 256 + threadIdx.x) >> p.lpv_log2;   // global group index
  const long long n0 = (long long)group * lpv;
  const long long n = n0 + g;
  int off = -1;
  if (n < p.N) {
    const int k = (int)(n % p.Z);
    const int t = (int)(n / p.Z);
    const int j = t % p.Y;
    const int i = t / p.Y;
    const float *no = p.new_origin + b * 3;
    const float px = __fadd_rn(__fmul_rn((float)i, p.vs0), no[0]);
    const float py = __fadd_rn(__fmul_rn((float)j, p.vs1), no[1]);
    const float pz = __fadd_rn(__fmul_rn((float)k, p.vs2), no[2]);
    const int hc = min(p.crop_hw[b * 2 + 0], p.FH), wc = min(p.crop_hw[b * 2 + 1], p.FW);   // slice semantics (detectors/imvoxelnet.py:69): clamped to the map
    const float *P = p.proj + (size_t)b * 12;
    float u = __fmul_rn(P[0], px);
    u = __fmaf_rn(P[1], py, u);
    u = __fmaf_rn(P[2], pz, u);
    u = __fmaf_rn(P[3], 1.0f, u);
    float w_ = __fmul_rn(P[4], px);
    w_ = __fmaf_rn(P[5], py, w_);
    w_ = __fmaf_rn(P[6], pz, w_);
    w_ = __fmaf_rn(P[7], 1.0f, w_);
    float d = __fmul_rn(P[8], px);
    d = __fmaf_rn(P[9], py, d);
    d = __fmaf_rn(P[10], pz, d);
    d = __fmaf_rn(P[11], 1.0f, d);
    const float xr = rintf(__fdiv_rn(u, d));
    const float yr = rintf(__fdiv_rn(w_, d));
    const bool ok = (xr >= 0.f) && (yr >= 0.f) && (xr < (float)wc) && (yr < (float)hc) && (d > 0.f);
    if (ok) off = ((b * p.FH + (int)yr) * p.FW + (int)xr);
    p.valid[(size_t)b * p.N + n] = ok ? 1 : 0;
  }
  const int nvox = (p.N - n0) < lpv ? (int)(p.N - n0) : lpv;   // group-uniform; <= 0 for groups past the end
  float *dst0 = p.volume + ((size_t)b * p.N + n0) * p.C;
  float vmax = 0.f;      // max |value stored| (ivx_backproject_mean_fwd_amax: the first neck layer's operand scale needs max |volume|)
  for (int s = 0; s < lpv; ++s) {
    const int o = __shfl(off, gbase + s, 64);
    if (s >= nvox) continue;
    const float *src = p.feat + (size_t)(o < 0 ? 0 : o) * p.C;
    for (int ch = g; ch < p.nchunk; ch += lpv) {
      if constexpr (VEC == 4) {
        f32x4 x = {0.f, 0.f, 0.f, 0.f};
        if (o >= 0) x = *reinterpret_cast<const f32x4 *>(src + ch * 4);
        *reinterpret_cast<f32x4 *>(dst0 + (size_t)s * p.C + ch * 4) = x;
        vmax = fmaxf(fmaxf(vmax, fmaxf(fabsf(x[0]), fabsf(x[1]))), fmaxf(fabsf(x[2]), fabsf(x[3])));
      } else {
        const float x = o >= 0 ? src[ch] : 0.f;
        dst0[(size_t)s * p.C + ch] = x;
        vmax = fmaxf(vmax, fabsf(x));
      }
    }
  }
  if (p.pmax) {          // (uniform) every lane of the workgroup gets here
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, o));
    __shared__ float wmax[4];
    if (lane == 0) wmax[threadIdx.x >> 6] = vmax;
    __syncthreads();
    if (threadIdx.x == 0) p.pmax[(size_t)blockIdx.y * gridDim.x + blockIdx.x] = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
  }
}

// A/B knob of the block order (default 0: workgroups in plain voxel order; IVX_BP_ORDER=1: XCD-contiguous eighths, see the kernel)
static int bp_q(int nblk) {
  static const int xcd_order = getenv("IVX_BP_ORDER") ? atoi(getenv("IVX_BP_ORDER")) : 0;
  return xcd_order ? (nblk + 7) / 8 : 0;
}
static unsigned bp_grid(const BpParams &p) { return p.q ? 8u * (unsigned)p.q : (unsigned)p.nblk; }

static int backproject_launch(const float *feat, int32_t B, int32_t V, int32_t FH, int32_t FW, int32_t C, const float *proj,
                              const float *new_origin, const int32_t *crop_hw, const float *voxel_size, int32_t X, int32_t Y,
                              int32_t Z, float *volume, uint8_t *valid, int32_t *count, ivx_stream_t stream, float *partials = nullptr) {
  const bool mean = count == nullptr;
  IVX_REQUIRE(feat && proj && new_origin && crop_hw && voxel_size && volume && (valid || count), "ivx_backproject_mean_fwd: null argument");
  IVX_REQUIRE(B > 0 && V > 0 && FH > 0 && FW > 0 && C > 0 && X > 0 && Y > 0 && Z > 0, "ivx_backproject_mean_fwd: non-positive dims");
  IVX_REQUIRE((int64_t)X * Y * Z < (1LL << 31), "ivx_backproject_mean_fwd: voxel grid too large");
  IVX_REQUIRE((int64_t)B * V * FH * FW < (1LL << 31), "ivx_backproject_mean_fwd: feature maps too large");
  IVX_REQUIRE(B <= 65535, "ivx_backproject_mean_fwd: batch too large");
  BpParams p;
  p.feat = feat; p.proj = proj; p.new_origin = new_origin; p.crop_hw = crop_hw; p.volume = volume; p.valid = valid;
  p.count = count;
  p.pmax = (mean && V == 1) ? partials : nullptr;
  p.nblk = 0; p.q = 0;
  p.vs0 = voxel_size[0]; p.vs1 = voxel_size[1]; p.vs2 = voxel_size[2];
  p.V = V; p.FH = FH; p.FW = FW; p.C = C; p.X = X; p.Y = Y; p.Z = Z; p.N = X * Y * Z;
  const int vec = (C % 4 == 0) ? 4 : 1;
  p.nchunk = (C + vec - 1) / vec;
  IVX_REQUIRE(p.nchunk <= 64 * 4, "ivx_backproject_mean_fwd: C=%d too large (max %d)", C, 256 * vec);
  int lg = 0;
  while ((1 << lg) < p.nchunk && lg < 6) ++lg;
  p.lpv_log2 = lg;
  if (!mean) {
    IVX_REQUIRE(vec == 4, "ivx_backproject_sum_fwd: C %% 4 must be 0");
    const int vpb = 256 >> lg;
    p.nblk = (p.N + vpb - 1) / vpb; p.q = bp_q(p.nblk);
    hipLaunchKernelGGL((backproject_mean_kernel<4, false>), dim3(bp_grid(p), B), dim3(256), 0, (hipStream_t)stream, p);
    IVX_CHECK_LAUNCH("ivx_backproject_sum_fwd");
    return IVX_OK;
  }
  if (V == 1) {
    // 256 voxels per workgroup regardless of the group width (each lane projects one voxel)
    dim3 g1((p.N + 255) / 256, B);
    if (vec == 4)
      hipLaunchKernelGGL(backproject_single_view_kernel<4>, g1, dim3(256), 0, (hipStream_t)stream, p);
    else
      hipLaunchKernelGGL(backproject_single_view_kernel<1>, g1, dim3(256), 0, (hipStream_t)stream, p);
    IVX_CHECK_LAUNCH("ivx_backproject_mean_fwd");
    return IVX_OK;
  }
  const int vox_per_block = 256 >> lg;
  p.nblk = (p.N + vox_per_block - 1) / vox_per_block; p.q = bp_q(p.nblk);
  dim3 grid(bp_grid(p), B);
  if (vec == 4)
    hipLaunchKernelGGL(backproject_mean_kernel<4>, grid, dim3(256), 0, (hipStream_t)stream, p);
  else
    hipLaunchKernelGGL(backproject_mean_kernel<1>, grid, dim3(256), 0, (hipStream_t)stream, p);
  IVX_CHECK_LAUNCH("ivx_backproject_mean_fwd");
  return IVX_OK;
}

extern "C" int ivx_backproject_mean_fwd(const float *feat, int32_t B, int32_t V, int32_t FH, int32_t FW, int32_t C,
                                        const float *proj, const float *new_origin, const int32_t *crop_hw,
                                        const float *voxel_size, int32_t X, int32_t Y, int32_t Z, float *volume,
                                        uint8_t *valid, ivx_stream_t stream) {
  IVX_REQUIRE(valid, "ivx_backproject_mean_fwd: null argument");
  return backproject_launch(feat, B, V, FH, FW, C, proj, new_origin, crop_hw, voxel_size, X, Y, Z, volume, valid, nullptr, stream);
}

// Single-view lift that also leaves one max |volume| per workgroup: ivx_backproject_amax_blocks(B, V, X, Y, Z) floats (0: this shape
// takes the multi-view kernel, which does not -- the consumer reduces the volume itself), for ivx_conv_winograd_input_amax.
extern "C" int32_t ivx_backproject_amax_blocks(int32_t B, int32_t V, int32_t X, int32_t Y, int32_t Z) {
  if (B <= 0 || V != 1 || X <= 0 || Y <= 0 || Z <= 0 || (int64_t)X * Y * Z >= (1LL << 31)) return 0;
  return (int32_t)(((int64_t)X * Y * Z + 255) / 256 * B);
}

extern "C" int ivx_backproject_mean_fwd_amax(const float *feat, int32_t B, int32_t V, int32_t FH, int32_t FW, int32_t C, const float *proj,
                                             const float *new_origin, const int32_t *crop_hw, const float *voxel_size, int32_t X, int32_t Y,
                                             int32_t Z, float *volume, uint8_t *valid, float *partials, ivx_stream_t stream) {
  IVX_REQUIRE(valid, "ivx_backproject_mean_fwd_amax: null argument");
  IVX_REQUIRE(!partials || V == 1, "ivx_backproject_mean_fwd_amax: partial maxima come from the single-view kernel only");
  return backproject_launch(feat, B, V, FH, FW, C, proj, new_origin, crop_hw, voxel_size, X, Y, Z, volume, valid, nullptr, stream, partials);
}

extern "C" int ivx_backproject_sum_fwd(const float *feat, int32_t B, int32_t V, int32_t FH, int32_t FW, int32_t C,
                                       const float *proj, const float *new_origin, const int32_t *crop_hw,
                                       const float *voxel_size, int32_t X, int32_t Y, int32_t Z, float *volume_sum,
                                       int32_t *count, ivx_stream_t stream) {
  IVX_REQUIRE(count, "ivx_backproject_sum_fwd: null argument");
  return backproject_launch(feat, B, V, FH, FW, C, proj, new_origin, crop_hw, voxel_size, X, Y, Z, volume_sum, nullptr, count, stream);
}

extern "C" int ivx_volume_normalize_fwd(float *volume, const int32_t *count, int64_t n_voxels, int32_t C, uint8_t *valid,
                                        ivx_stream_t stream) {
  IVX_REQUIRE(volume && count && valid, "ivx_volume_normalize_fwd: null argument");
  IVX_REQUIRE(n_voxels > 0 && C > 0 && C % 4 == 0, "ivx_volume_normalize_fwd: bad dims (C %% 4 must be 0)");
  const long long total4 = (long long)n_voxels * (C / 4);
  long long blocks = (total4 + 255) / 256;
  if (blocks > 256 * 64) blocks = 256 * 64;
  hipLaunchKernelGGL(volume_normalize_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, volume, count, valid, total4, C / 4);
  IVX_CHECK_LAUNCH("ivx_volume_normalize_fwd");
  return IVX_OK;
}

// bf16 storage (optional reduced-precision mode): feat / volume are bf16, everything else as ivx_backproject_mean_fwd.
// Multi-view launches only (with one view the lift is a byte copy: pass the bf16 map as C/2 32-bit words to
// ivx_backproject_mean_fwd).  C % 4 == 0.
extern "C" int ivx_backproject_mean_fwd_bf16(const void *feat, int32_t B, int32_t V, int32_t FH, int32_t FW, int32_t C,
                                             const float *proj, const float *new_origin, const int32_t *crop_hw,
                                             const float *voxel_size, int32_t X, int32_t Y, int32_t Z, void *volume,
                                             uint8_t *valid, ivx_stream_t stream) {
  IVX_REQUIRE(feat && proj && new_origin && crop_hw && voxel_size && volume && valid, "ivx_backproject_mean_fwd_bf16: null argument");
  IVX_REQUIRE(B > 0 && V > 0 && FH > 0 && FW > 0 && C > 0 && C % 4 == 0 && X > 0 && Y > 0 && Z > 0, "ivx_backproject_mean_fwd_bf16: bad dims (C %% 4 must be 0)");
  IVX_REQUIRE((int64_t)X * Y * Z < (1LL << 31) && (int64_t)B * V * FH * FW < (1LL << 31) && B <= 65535, "ivx_backproject_mean_fwd_bf16: problem too large");
  BpParams p;
  p.feat = (const float *)feat; p.proj = proj; p.new_origin = new_origin; p.crop_hw = crop_hw; p.volume = (float *)volume; p.valid = valid;
  p.count = nullptr; p.pmax = nullptr;
  p.vs0 = voxel_size[0]; p.vs1 = voxel_size[1]; p.vs2 = voxel_size[2];
  p.V = V; p.FH = FH; p.FW = FW; p.C = C; p.X = X; p.Y = Y; p.Z = Z; p.N = X * Y * Z;
  p.nchunk = C / 4;
  IVX_REQUIRE(p.nchunk <= 64 * 4, "ivx_backproject_mean_fwd_bf16: C=%d too large (max 1024)", C);
  int lg = 0;
  while ((1 << lg) < p.nchunk && lg < 6) ++lg;
  p.lpv_log2 = lg;
  const int vpb = 256 >> lg;
  p.nblk = (p.N + vpb - 1) / vpb; p.q = bp_q(p.nblk);
  hipLaunchKernelGGL((backproject_mean_kernel<4, true, __bf16>), dim3(bp_grid(p), B), dim3(256), 0, (hipStream_t)stream, p);
  IVX_CHECK_LAUNCH("ivx_backproject_mean_fwd_bf16");
  return IVX_OK;
}
