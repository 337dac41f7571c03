// Small memory-bound helpers around the conv stack: max-pool (ResNet stem), layout changes at the
// boundary (the Python surface keeps the reference's NCHW/NCDHW tensors; the kernels run
// channels-last).  See include/imvoxel.h.
#include "ivx_common.h"

#include <float.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));

// nn.MaxPool2d(k, s, p) on NHWC, C % 4 == 0: one thread per (output pixel, 4 channels).
__global__ __launch_bounds__(256) void maxpool2d_nhwc_kernel(const float *in, int B, int H, int W, int C, int k, int s,
                                                             int pd, int Ho, int Wo, float *out) {
  const int C4 = C >> 2;
  const size_t total = (size_t)B * Ho * Wo * C4;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const int c4 = (int)(idx % C4);
    size_t t = idx / C4;
    const int ow = (int)(t % Wo);
    t /= Wo;
    const int oh = (int)(t % Ho);
    const int b = (int)(t / Ho);
    f32x4 m = {-FLT_MAX, -FLT_MAX, -FLT_MAX, -FLT_MAX};
    // -inf padding semantics of torch: start from -inf; use -FLT_MAX then fix below if no tap (never: k > p)
    bool any = false;
    for (int e = 0; e < k; ++e) {
      const int ih = oh * s - pd + e;
      if ((unsigned)ih >= (unsigned)H) continue;
      for (int f = 0; f < k; ++f) {
        const int iw = ow * s - pd + f;
        if ((unsigned)iw >= (unsigned)W) continue;
        const f32x4 v = *reinterpret_cast<const f32x4 *>(in + (((size_t)b * H + ih) * W + iw) * C + c4 * 4);
#pragma unroll
        for (int q = 0; q < 4; ++q) m[q] = (v[q] > m[q] || v[q] != v[q]) ? v[q] : m[q];
        any = true;
      }
    }
    if (!any) m = f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    *reinterpret_cast<f32x4 *>(out + idx * 4) = m;
  }
}

extern "C" int ivx_maxpool2d_fwd(const float *in, int32_t B, int32_t H, int32_t W, int32_t C, int32_t k, int32_t s,
                                 int32_t p, float *out, ivx_stream_t stream) {
  IVX_REQUIRE(in && out, "ivx_maxpool2d_fwd: null argument");
  IVX_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, "ivx_maxpool2d_fwd: bad dims (C %% 4 must be 0)");
  IVX_REQUIRE(k > 0 && s > 0 && p >= 0 && 2 * p <= k, "ivx_maxpool2d_fwd: bad window");
  const int Ho = (H + 2 * p - k) / s + 1, Wo = (W + 2 * p - k) / s + 1;
  IVX_REQUIRE(Ho > 0 && Wo > 0, "ivx_maxpool2d_fwd: empty output");
  const size_t total = (size_t)B * Ho * Wo * (C / 4);
  size_t blocks = (total + 255) / 256;
  if (blocks > 256 * 32) blocks = 256 * 32;
  hipLaunchKernelGGL(maxpool2d_nhwc_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, in, B, H, W, C, k, s, p,
                     Ho, Wo, out);
  IVX_CHECK_LAUNCH("ivx_maxpool2d_fwd");
  return IVX_OK;
}

// [B,C,S] -> [B,S,Cpad] through a 32x33 LDS tile so both sides are coalesced.
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const float *in, int C, long long S, int Cpad, float *out) {
  __shared__ float tile[32][33];
  const int b = blockIdx.z;
  const long long s0 = (long long)blockIdx.x * 32;
  const int c0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  for (int r = ty; r < 32; r += 8) {
    const int c = c0 + r;
    const long long s = s0 + tx;
    tile[r][tx] = (c < C && s < S) ? in[((size_t)b * C + c) * S + s] : 0.f;
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    const long long s = s0 + r;
    const int c = c0 + tx;
    if (s < S && c < Cpad) out[((size_t)b * S + s) * Cpad + c] = tile[tx][r];
  }
}

extern "C" int ivx_nchw_to_nhwc(const float *in, int32_t B, int32_t C, int64_t S, int32_t Cpad, float *out,
                                ivx_stream_t stream) {
  IVX_REQUIRE(in && out && B > 0 && C > 0 && S > 0 && Cpad >= C, "ivx_nchw_to_nhwc: bad argument");
  IVX_REQUIRE(B <= 65535 && (Cpad + 31) / 32 <= 65535, "ivx_nchw_to_nhwc: dims too large");
  dim3 grid((unsigned)((S + 31) / 32), (Cpad + 31) / 32, B);
  hipLaunchKernelGGL(nchw_to_nhwc_kernel, grid, dim3(256), 0, (hipStream_t)stream, in, C, (long long)S, Cpad, out);
  IVX_CHECK_LAUNCH("ivx_nchw_to_nhwc");
  return IVX_OK;
}

// [B,S,C] -> [B,C,S]
__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(const float *in, long long S, int C, float *out) {
  __shared__ float tile[32][33];
  const int b = blockIdx.z;
  const long long s0 = (long long)blockIdx.x * 32;
  const int c0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int r = ty; r < 32; r += 8) {
    const long long s = s0 + r;
    const int c = c0 + tx;
    tile[r][tx] = (s < S && c < C) ? in[((size_t)b * S + s) * C + c] : 0.f;
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    const int c = c0 + r;
    const long long s = s0 + tx;
    if (c < C && s < S) out[((size_t)b * C + c) * S + s] = tile[tx][r];
  }
}

extern "C" int ivx_nhwc_to_nchw(const float *in, int32_t B, int64_t S, int32_t C, float *out, ivx_stream_t stream) {
  IVX_REQUIRE(in && out && B > 0 && C > 0 && S > 0, "ivx_nhwc_to_nchw: bad argument");
  IVX_REQUIRE(B <= 65535 && (C + 31) / 32 <= 65535, "ivx_nhwc_to_nchw: dims too large");
  dim3 grid((unsigned)((S + 31) / 32), (C + 31) / 32, B);
  hipLaunchKernelGGL(nhwc_to_nchw_kernel, grid, dim3(256), 0, (hipStream_t)stream, in, (long long)S, C, out);
  IVX_CHECK_LAUNCH("ivx_nhwc_to_nchw");
  return IVX_OK;
}
