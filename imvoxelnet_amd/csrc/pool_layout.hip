// Small memory-bound helpers around the conv stack: max-pool (ResNet stem), layout changes at the
// boundary (the Python surface keeps the reference's NCHW/NCDHW tensors; the kernels run
// channels-last).  See include/imvoxel.h.
#include "ivx_common.h"

#include <float.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));

// nn.MaxPool2d(k, s, p) on NHWC, C % 4 == 0: one thread per (output pixel, 4 channels).  T = float or __bf16 (the max
// of bf16 values is exact, so the bf16 instantiation is the same op on the reduced-precision storage).
template <typename T>
__global__ __launch_bounds__(256) void maxpool2d_nhwc_kernel(const T *in, int B, int H, int W, int C, int k, int s,
                                                             int pd, int Ho, int Wo, T *out) {
  typedef T tx4 __attribute__((ext_vector_type(4)));
  const int C4 = C >> 2;
  const size_t total = (size_t)B * Ho * Wo * C4;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const int c4 = (int)(idx % C4);
    size_t t = idx / C4;
    const int ow = (int)(t % Wo);
    t /= Wo;
    const int oh = (int)(t % Ho);
    const int b = (int)(t / Ho);
    // -inf padding semantics of torch (a window always holds at least one tap because 2p <= k)
    float m[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    for (int e = 0; e < k; ++e) {
      const int ih = oh * s - pd + e;
      if ((unsigned)ih >= (unsigned)H) continue;
      for (int f = 0; f < k; ++f) {
        const int iw = ow * s - pd + f;
        if ((unsigned)iw >= (unsigned)W) continue;
        const tx4 v = *reinterpret_cast<const tx4 *>(in + (((size_t)b * H + ih) * W + iw) * C + c4 * 4);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float x = (float)v[q];
          m[q] = (x > m[q] || x != x) ? x : m[q];
        }
      }
    }
    tx4 o;
#pragma unroll
    for (int q = 0; q < 4; ++q) o[q] = (T)m[q];
    *reinterpret_cast<tx4 *>(out + idx * 4) = o;
  }
}

template <typename T>
static int maxpool2d_launch(const T *in, int32_t B, int32_t H, int32_t W, int32_t C, int32_t k, int32_t s, int32_t p, T *out,
                            ivx_stream_t stream) {
  IVX_REQUIRE(in && out, "ivx_maxpool2d_fwd: null argument");
  IVX_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, "ivx_maxpool2d_fwd: bad dims (C %% 4 must be 0)");
  IVX_REQUIRE(k > 0 && s > 0 && p >= 0 && 2 * p <= k, "ivx_maxpool2d_fwd: bad window");
  const int Ho = (H + 2 * p - k) / s + 1, Wo = (W + 2 * p - k) / s + 1;
  IVX_REQUIRE(Ho > 0 && Wo > 0, "ivx_maxpool2d_fwd: empty output");
  const size_t total = (size_t)B * Ho * Wo * (C / 4);
  size_t blocks = (total + 255) / 256;
  if (blocks > 256 * 32) blocks = 256 * 32;
  hipLaunchKernelGGL(maxpool2d_nhwc_kernel<T>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, in, B, H, W, C, k, s, p,
                     Ho, Wo, out);
  IVX_CHECK_LAUNCH("ivx_maxpool2d_fwd");
  return IVX_OK;
}

extern "C" int ivx_maxpool2d_fwd(const float *in, int32_t B, int32_t H, int32_t W, int32_t C, int32_t k, int32_t s,
                                 int32_t p, float *out, ivx_stream_t stream) {
  return maxpool2d_launch<float>(in, B, H, W, C, k, s, p, out, stream);
}

extern "C" int ivx_maxpool2d_fwd_bf16(const void *in, int32_t B, int32_t H, int32_t W, int32_t C, int32_t k, int32_t s,
                                      int32_t p, void *out, ivx_stream_t stream) {
  return maxpool2d_launch<__bf16>((const __bf16 *)in, B, H, W, C, k, s, p, (__bf16 *)out, stream);
}

// The same pool on e4m3 bytes (C % 16 == 0): one thread per (output pixel, 16 channels), one 16-byte load per tap.  The
// maximum of representable values is representable, so decode -> max -> encode is exact; the per-tensor scale is unchanged.
__global__ __launch_bounds__(256) void maxpool2d_nhwc_fp8_kernel(const unsigned char *in, int B, int H, int W, int C, int k, int s,
                                                                 int pd, int Ho, int Wo, unsigned char *out) {
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  const int C16 = C >> 4;
  const size_t total = (size_t)B * Ho * Wo * C16;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const int c16 = (int)(idx % C16);
    size_t t = idx / C16;
    const int ow = (int)(t % Wo);
    t /= Wo;
    const int oh = (int)(t % Ho);
    const int b = (int)(t / Ho);
    float m[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) m[q] = -INFINITY;
    for (int e = 0; e < k; ++e) {
      const int ih = oh * s - pd + e;
      if ((unsigned)ih >= (unsigned)H) continue;
      for (int f = 0; f < k; ++f) {
        const int iw = ow * s - pd + f;
        if ((unsigned)iw >= (unsigned)W) continue;
        const u32x4 v = *reinterpret_cast<const u32x4 *>(in + (((size_t)b * H + ih) * W + iw) * C + c16 * 16);
#pragma unroll
        for (int w = 0; w < 4; ++w) {
          const f32x2 lo = __builtin_amdgcn_cvt_pk_f32_fp8((int)v[w], false), hi = __builtin_amdgcn_cvt_pk_f32_fp8((int)v[w], true);
          const float x[4] = {lo[0], lo[1], hi[0], hi[1]};
#pragma unroll
          for (int q = 0; q < 4; ++q) m[4 * w + q] = (x[q] > m[4 * w + q] || x[q] != x[q]) ? x[q] : m[4 * w + q];
        }
      }
    }
    u32x4 o;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      int pk = __builtin_amdgcn_cvt_pk_fp8_f32(m[4 * w], m[4 * w + 1], 0, false);
      pk = __builtin_amdgcn_cvt_pk_fp8_f32(m[4 * w + 2], m[4 * w + 3], pk, true);
      o[w] = (unsigned int)pk;
    }
    *reinterpret_cast<u32x4 *>(out + idx * 16) = o;
  }
}

extern "C" int ivx_maxpool2d_fwd_fp8(const void *in, int32_t B, int32_t H, int32_t W, int32_t C, int32_t k, int32_t s,
                                     int32_t p, void *out, ivx_stream_t stream) {
  IVX_REQUIRE(in && out, "ivx_maxpool2d_fwd_fp8: null argument");
  IVX_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0 && C % 16 == 0, "ivx_maxpool2d_fwd_fp8: bad dims (C %% 16 must be 0)");
  IVX_REQUIRE(k > 0 && s > 0 && p >= 0 && 2 * p <= k, "ivx_maxpool2d_fwd_fp8: bad window");
  const int Ho = (H + 2 * p - k) / s + 1, Wo = (W + 2 * p - k) / s + 1;
  IVX_REQUIRE(Ho > 0 && Wo > 0, "ivx_maxpool2d_fwd_fp8: empty output");
  const size_t total = (size_t)B * Ho * Wo * (C / 16);
  size_t blocks = (total + 255) / 256;
  if (blocks > 256 * 32) blocks = 256 * 32;
  hipLaunchKernelGGL(maxpool2d_nhwc_fp8_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const unsigned char *)in, B, H, W,
                     C, k, s, p, Ho, Wo, (unsigned char *)out);
  IVX_CHECK_LAUNCH("ivx_maxpool2d_fwd_fp8");
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

// The image case -- C <= 4 planes into 4-channel pixels, S % 4 == 0 -- as a streaming kernel: a thread reads 16 bytes of every plane (4
// pixels) and writes the 4 pixels as one 16-byte word each.  The 32 x 32 tile kernel above moves 900 bytes per workgroup there (3 of its 32
// channel rows are real): 0.8 TB/s, i.e. 0.53 ms for the 50 views of a ScanNet scene (round-4 kernel trace) against 0.1 at streaming rate.
// AMAX: also accumulate max |in| into the caller's slots (ivx_nchw_to_nhwc_amax).
template <int AMAX>
__global__ __launch_bounds__(256) void nchw_to_nhwc4_kernel(const float *in, int C, long long S4, float *out, unsigned *amax, long long total) {
  typedef float f32x4p __attribute__((ext_vector_type(4)));
  float m = 0.f;
  for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long long)gridDim.x * 256) {
    const long long b = t / S4, s4 = t - b * S4;
    f32x4p v[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      v[c] = f32x4p{0.f, 0.f, 0.f, 0.f};
      if (c < C) v[c] = *reinterpret_cast<const f32x4p *>(in + ((size_t)b * C + c) * (size_t)(4 * S4) + (size_t)s4 * 4);
      if (AMAX) m = fmaxf(m, fmaxf(fmaxf(fabsf(v[c][0]), fabsf(v[c][1])), fmaxf(fabsf(v[c][2]), fabsf(v[c][3]))));
    }
    f32x4p *o = reinterpret_cast<f32x4p *>(out) + (size_t)t * 4;
#pragma unroll
    for (int q = 0; q < 4; ++q) o[q] = f32x4p{v[0][q], v[1][q], v[2][q], v[3][q]};
  }
  if (AMAX) {
    __shared__ float red[4];
    ivx_amax_commit_wg(amax, m, red, (int)blockIdx.x);
  }
}
static bool nchw4_applicable(const float *in, const float *out, int C, long long S, int Cpad) {
  return C <= 4 && Cpad == 4 && S % 4 == 0 && ((uintptr_t)in & 15) == 0 && ((uintptr_t)out & 15) == 0;
}

extern "C" int ivx_nchw_to_nhwc(const float *in, int32_t B, int32_t C, int64_t S, int32_t Cpad, float *out,
                                ivx_stream_t stream) {
  IVX_REQUIRE(in && out && B > 0 && C > 0 && S > 0 && Cpad >= C, "ivx_nchw_to_nhwc: bad argument");
  IVX_REQUIRE(B <= 65535 && (Cpad + 31) / 32 <= 65535, "ivx_nchw_to_nhwc: dims too large");
  if (nchw4_applicable(in, out, C, S, Cpad)) {
    const long long total = (long long)B * (S / 4), nb = (total + 255) / 256;
    hipLaunchKernelGGL(nchw_to_nhwc4_kernel<0>, dim3((unsigned)(nb > 65536 ? 65536 : nb)), dim3(256), 0, (hipStream_t)stream, in, C, (long long)(S / 4), out,
                       (unsigned *)nullptr, total);
    IVX_CHECK_LAUNCH("ivx_nchw_to_nhwc");
    return IVX_OK;
  }
  dim3 grid((unsigned)((S + 31) / 32), (Cpad + 31) / 32, B);
  hipLaunchKernelGGL(nchw_to_nhwc_kernel, grid, dim3(256), 0, (hipStream_t)stream, in, C, (long long)S, Cpad, out);
  IVX_CHECK_LAUNCH("ivx_nchw_to_nhwc");
  return IVX_OK;
}

// Image -> space-to-depth blocks for the bf16 stem (optional reduced-precision mode).  The 7x7 stride-2 pad-3 stem reads, for
// output (oh, ow), the pixels 2*oh-3 .. 2*oh+3: exactly the four 2x2 blocks oh-1 .. oh+2 of a block grid whose block p holds
// the pixels (2p-1, 2p).  So the stem is a 4x4 stride-1 pad-1 convolution over out[b][ph][pw][(a*2+e)*3 + c] =
// img[b][c][2*ph-1+a][2*pw-1+e] (zero outside the image; channels 12..15 zero): K = 4*4*16 = 256 bf16 values per output, of
// which 147 are real taps, instead of 7*7*8 = 392 with the 3 channels padded to one 16-byte chunk -- and no strided gather.
__global__ __launch_bounds__(256) void image_s2d_bf16_kernel(const float *img, int H, int W, int PH, int PW, __bf16 *out) {
  const int b = blockIdx.z;
  const int pw = blockIdx.x * 64 + (threadIdx.x & 63);
  const int ph = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (pw >= PW || ph >= PH) return;
  typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
  bf16x8_t lo, hi;
  const float *base = img + (size_t)b * 3 * H * W;
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int y = 2 * ph - 1 + a, x = 2 * pw - 1 + e;
      const bool in = (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float v = in ? base[((size_t)c * H + y) * W + x] : 0.f;
        const int k = (a * 2 + e) * 3 + c;
        if (k < 8) lo[k] = (__bf16)v; else hi[k - 8] = (__bf16)v;
      }
    }
#pragma unroll
  for (int k = 4; k < 8; ++k) hi[k] = (__bf16)0.f;
  bf16x8_t *o = reinterpret_cast<bf16x8_t *>(out + (((size_t)b * PH + ph) * PW + pw) * 16);
  o[0] = lo;
  o[1] = hi;
}

extern "C" int ivx_image_s2d_bf16(const float *img, int32_t B, int32_t H, int32_t W, void *out, ivx_stream_t stream) {
  IVX_REQUIRE(img && out && B > 0 && H > 0 && W > 0, "ivx_image_s2d_bf16: bad argument");
  IVX_REQUIRE(H % 2 == 0 && W % 2 == 0, "ivx_image_s2d_bf16: H and W must be even (got %d x %d)", H, W);
  IVX_REQUIRE(B <= 65535, "ivx_image_s2d_bf16: batch too large");
  const int PH = H / 2 + 1, PW = W / 2 + 1;
  dim3 grid((PW + 63) / 64, (PH + 3) / 4, B);
  hipLaunchKernelGGL(image_s2d_bf16_kernel, grid, dim3(256), 0, (hipStream_t)stream, img, H, W, PH, PW, (__bf16 *)out);
  IVX_CHECK_LAUNCH("ivx_image_s2d_bf16");
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

// F.interpolate(scale_factor=2, mode='trilinear', align_corners=False) on NDHWC (the Atlas decoder,
// necks/imvoxelnet.py:359).  Source coordinate per axis: max(0.5*(o + 0.5) - 0.5, 0); i1 = min(i0 + 1, n - 1).
// The 8-corner blend follows ATen's CPU order: d outermost, then h, then w.
template <typename T>
__global__ __launch_bounds__(256) void upsample_trilinear2x_kernel(const T *in, int B, int D, int H, int W, int C, T *out) {
  typedef T tx4 __attribute__((ext_vector_type(4)));
  const int C4 = C >> 2;
  const int Do = 2 * D, Ho = 2 * H, Wo = 2 * W;
  const size_t total = (size_t)B * Do * Ho * Wo * C4;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const int c4 = (int)(idx % C4);
    size_t t = idx / C4;
    const int ow = (int)(t % Wo); t /= Wo;
    const int oh = (int)(t % Ho); t /= Ho;
    const int od = (int)(t % Do);
    const int b = (int)(t / Do);
    float sd = 0.5f * (od + 0.5f) - 0.5f, sh = 0.5f * (oh + 0.5f) - 0.5f, sw = 0.5f * (ow + 0.5f) - 0.5f;
    sd = sd < 0.f ? 0.f : sd; sh = sh < 0.f ? 0.f : sh; sw = sw < 0.f ? 0.f : sw;
    const int d0 = (int)sd, h0 = (int)sh, w0 = (int)sw;
    const int d1 = d0 + (d0 < D - 1 ? 1 : 0), h1 = h0 + (h0 < H - 1 ? 1 : 0), w1 = w0 + (w0 < W - 1 ? 1 : 0);
    const float ld1 = sd - d0, lh1 = sh - h0, lw1 = sw - w0;
    const float ld0 = 1.f - ld1, lh0 = 1.f - lh1, lw0 = 1.f - lw1;
    auto at = [&](int d, int h, int w) {
      const tx4 v = *reinterpret_cast<const tx4 *>(in + ((((size_t)b * D + d) * H + h) * W + w) * C + c4 * 4);
      return f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
    };
    const f32x4 v000 = at(d0, h0, w0), v001 = at(d0, h0, w1), v010 = at(d0, h1, w0), v011 = at(d0, h1, w1);
    const f32x4 v100 = at(d1, h0, w0), v101 = at(d1, h0, w1), v110 = at(d1, h1, w0), v111 = at(d1, h1, w1);
    tx4 o;
#pragma unroll
    for (int q = 0; q < 4; ++q)
      o[q] = (T)(ld0 * (lh0 * (lw0 * v000[q] + lw1 * v001[q]) + lh1 * (lw0 * v010[q] + lw1 * v011[q])) +
                 ld1 * (lh0 * (lw0 * v100[q] + lw1 * v101[q]) + lh1 * (lw0 * v110[q] + lw1 * v111[q])));
    *reinterpret_cast<tx4 *>(out + idx * 4) = o;
  }
}

template <typename T>
static int upsample_launch(const T *in, int32_t B, int32_t D, int32_t H, int32_t W, int32_t C, T *out, ivx_stream_t stream) {
  IVX_REQUIRE(in && out, "ivx_upsample_trilinear2x_fwd: null argument");
  IVX_REQUIRE(B > 0 && D > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, "ivx_upsample_trilinear2x_fwd: bad dims (C %% 4 must be 0)");
  const size_t total = (size_t)B * 8 * D * H * W * (C / 4);
  size_t blocks = (total + 255) / 256;
  if (blocks > 256 * 32) blocks = 256 * 32;
  hipLaunchKernelGGL(upsample_trilinear2x_kernel<T>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, in, B, D, H, W, C, out);
  IVX_CHECK_LAUNCH("ivx_upsample_trilinear2x_fwd");
  return IVX_OK;
}

extern "C" int ivx_upsample_trilinear2x_fwd(const float *in, int32_t B, int32_t D, int32_t H, int32_t W, int32_t C, float *out,
                                            ivx_stream_t stream) {
  return upsample_launch<float>(in, B, D, H, W, C, out, stream);
}

extern "C" int ivx_upsample_trilinear2x_fwd_bf16(const void *in, int32_t B, int32_t D, int32_t H, int32_t W, int32_t C, void *out,
                                                 ivx_stream_t stream) {
  return upsample_launch<__bf16>((const __bf16 *)in, B, D, H, W, C, (__bf16 *)out, stream);
}

// Global average pool over the spatial positions of a channels-last map: in [B,S,C] -> out [B,C]
// (`x.mean(dim=(2, 3))` of LayoutHead.forward, mmdet3d/models/dense_heads/layout_head.py:42).  One workgroup per
// (64-channel group, sample): 4 spatial phases x 64 channels, each lane walks its phase with coalesced 256-byte rows,
// then the four partial sums are combined in a fixed order (deterministic).
__global__ __launch_bounds__(256) void global_avgpool_kernel(const float *in, int S, int C, float *out) {
  __shared__ float part[4][64];
  const int b = blockIdx.y;
  const int c = blockIdx.x * 64 + (threadIdx.x & 63);
  const int ph = threadIdx.x >> 6;
  float acc = 0.f;
  if (c < C)
    for (int s = ph; s < S; s += 4) acc += in[((size_t)b * S + s) * C + c];
  part[ph][threadIdx.x & 63] = acc;
  __syncthreads();
  if (ph == 0 && c < C) out[(size_t)b * C + c] = (((part[0][threadIdx.x] + part[1][threadIdx.x]) + part[2][threadIdx.x]) + part[3][threadIdx.x]) / (float)S;
}

extern "C" int ivx_global_avgpool_fwd(const float *in, int32_t B, int64_t S, int32_t C, float *out, ivx_stream_t stream) {
  IVX_REQUIRE(in && out, "ivx_global_avgpool_fwd: null argument");
  IVX_REQUIRE(B > 0 && B <= 65535 && S > 0 && S < (1LL << 31) && C > 0, "ivx_global_avgpool_fwd: bad dims");
  hipLaunchKernelGGL(global_avgpool_kernel, dim3((C + 63) / 64, B), dim3(256), 0, (hipStream_t)stream, in, (int)S, C, out);
  IVX_CHECK_LAUNCH("ivx_global_avgpool_fwd");
  return IVX_OK;
}

// fp32 -> (hi, lo) pairs in the IVX_BF16_PAIR / IVX_F16_PAIR order (include/imvoxel.h): per 16 values [hi x16 | lo x16].  One lane
// converts 8 values: 32 contiguous bytes in, 16 bytes of hi and 16 bytes of lo out; a pair of lanes fills one 64-byte group.
typedef float pf32x4 __attribute__((ext_vector_type(4)));
template <typename H> struct PairIsF16 { static constexpr bool value = false; };
template <> struct PairIsF16<_Float16> { static constexpr bool value = true; };
template <typename H>
__global__ __launch_bounds__(256) void pair_split_kernel(const float *__restrict__ in, size_t n8, float scale, H *__restrict__ out) {
  typedef H hx8 __attribute__((ext_vector_type(8)));
  constexpr bool F16 = PairIsF16<H>::value;
  for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < n8; t += (size_t)gridDim.x * blockDim.x) {
    const pf32x4 a = __builtin_nontemporal_load(reinterpret_cast<const pf32x4 *>(in) + 2 * t);
    const pf32x4 b = __builtin_nontemporal_load(reinterpret_cast<const pf32x4 *>(in) + 2 * t + 1);
    hx8 hi, lo;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float x = e < 4 ? a[e] : b[e - 4];
      if (F16) x = __builtin_fminf(__builtin_fmaxf(x * scale, -65504.f), 65504.f);
      hi[e] = (H)x;
      lo[e] = (H)(x - (float)hi[e]);
    }
    H *o = out + (t >> 1) * 32 + (t & 1) * 8;
    *reinterpret_cast<hx8 *>(o) = hi;
    *reinterpret_cast<hx8 *>(o + 16) = lo;
  }
}

template <typename H>
static int pair_split_launch(const float *in, int64_t n, float scale, void *out, ivx_stream_t stream, const char *who) {
  IVX_REQUIRE(in && out && n >= 0 && n % 16 == 0, "%s: null argument or n not a multiple of 16", who);
  if (n == 0) return IVX_OK;
  const size_t n8 = (size_t)n / 8;
  size_t blocks = (n8 + 255) / 256;
  if (blocks > 256 * 32) blocks = 256 * 32;
  hipLaunchKernelGGL(pair_split_kernel<H>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, in, n8, scale, (H *)out);
  IVX_CHECK_LAUNCH(who);
  return IVX_OK;
}

extern "C" int ivx_bf16_pair_split(const float *in, int64_t n, void *out, ivx_stream_t stream) {
  return pair_split_launch<__bf16>(in, n, 1.0f, out, stream, "ivx_bf16_pair_split");
}

extern "C" int ivx_f16_pair_split(const float *in, int64_t n, float scale, void *out, ivx_stream_t stream) {
  IVX_REQUIRE(scale > 0.f, "ivx_f16_pair_split: the scale must be positive");
  return pair_split_launch<_Float16>(in, n, scale, out, stream, "ivx_f16_pair_split");
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Head of a chain of fp16-pair activations (include/imvoxel.h, "Chained fp16-pair activations").
// ivx_nchw_to_nhwc that also accumulates max |in| (the image) into the caller's amax slots.
__global__ __launch_bounds__(256) void nchw_to_nhwc_amax_kernel(const float *in, int C, long long S, int Cpad, float *out, unsigned *amax) {
  __shared__ float tile[32][33];
  const int b = blockIdx.z;
  const long long s0 = (long long)blockIdx.x * 32;
  const int c0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  float m = 0.f;
  for (int r = ty; r < 32; r += 8) {
    const int c = c0 + r;
    const long long s = s0 + tx;
    const float v = (c < C && s < S) ? in[((size_t)b * C + c) * S + s] : 0.f;
    tile[r][tx] = v;
    m = fmaxf(m, fabsf(v));
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    const long long s = s0 + r;
    const int c = c0 + tx;
    if (s < S && c < Cpad) out[((size_t)b * S + s) * Cpad + c] = tile[tx][r];
  }
  __shared__ float red[4];
  ivx_amax_commit_wg(amax, m, red, (int)(blockIdx.x + blockIdx.z));
}

extern "C" int ivx_nchw_to_nhwc_amax(const float *in, int32_t B, int32_t C, int64_t S, int32_t Cpad, float *out, uint32_t *amax,
                                     ivx_stream_t stream) {
  IVX_REQUIRE(in && out && amax && B > 0 && C > 0 && S > 0 && Cpad >= C, "ivx_nchw_to_nhwc_amax: bad argument");
  IVX_REQUIRE(B <= 65535 && (Cpad + 31) / 32 <= 65535, "ivx_nchw_to_nhwc_amax: dims too large");
  if (nchw4_applicable(in, out, C, S, Cpad)) {
    const long long total = (long long)B * (S / 4), nb = (total + 255) / 256;
    hipLaunchKernelGGL(nchw_to_nhwc4_kernel<1>, dim3((unsigned)(nb > 65536 ? 65536 : nb)), dim3(256), 0, (hipStream_t)stream, in, C, (long long)(S / 4), out,
                       amax, total);
    IVX_CHECK_LAUNCH("ivx_nchw_to_nhwc_amax");
    return IVX_OK;
  }
  dim3 grid((unsigned)((S + 31) / 32), (Cpad + 31) / 32, B);
  hipLaunchKernelGGL(nchw_to_nhwc_amax_kernel, grid, dim3(256), 0, (hipStream_t)stream, in, C, (long long)S, Cpad, out, amax);
  IVX_CHECK_LAUNCH("ivx_nchw_to_nhwc_amax");
  return IVX_OK;
}

// nn.MaxPool2d(k, s, p) on an fp32 NHWC map -> IVX_F16_PAIR tensor.  One thread per (output pixel, 4 channels): 8 bytes of hi halves
// and, 32 bytes further, 8 bytes of lo halves.  The scale comes from the bound amax_in * wbound + sbound of the INPUT map (the stem's
// output as a function of the image's maximum): a window maximum cannot exceed it.  -inf padding semantics of torch as in
// maxpool2d_nhwc_kernel; the pool is exact (a maximum of fp32 values), only the final split rounds (to 22 bits).
__global__ __launch_bounds__(256) void maxpool2d_nhwc_pair_kernel(const float *in, int B, int H, int W, int C, int k, int s, int pd, int Ho,
                                                                  int Wo, _Float16 *out, const unsigned *amax_in, float wbound, float sbound,
                                                                  float *out_scale, unsigned *amax_out) {
  typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
  const float bound = (ivx_amax_read(amax_in) * wbound + sbound) * 1.001f;
  const bool sat = !(bound < 3.0e38f);                      // non-finite image: fixed scale, saturating split (as conv_pair_io)
  const float sc = sat ? 0.00390625f : ivx_pow2_scale(bound);
  if (blockIdx.x == 0 && threadIdx.x == 0) *out_scale = sc;
  const int C4 = C >> 2;
  const size_t total = (size_t)B * Ho * Wo * C4;
  float omax = 0.f;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const int c4 = (int)(idx % C4);
    size_t t = idx / C4;
    const int ow = (int)(t % Wo);
    t /= Wo;
    const int oh = (int)(t % Ho);
    const int b = (int)(t / Ho);
    float m[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    for (int e = 0; e < k; ++e) {
      const int ih = oh * s - pd + e;
      if ((unsigned)ih >= (unsigned)H) continue;
      for (int f = 0; f < k; ++f) {
        const int iw = ow * s - pd + f;
        if ((unsigned)iw >= (unsigned)W) continue;
        const f32x4 v = *reinterpret_cast<const f32x4 *>(in + (((size_t)b * H + ih) * W + iw) * C + c4 * 4);
#pragma unroll
        for (int q = 0; q < 4; ++q) m[q] = (v[q] > m[q] || v[q] != v[q]) ? v[q] : m[q];
      }
    }
    f16x4 hi, lo;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      omax = fmaxf(omax, fabsf(m[q]));
      float y = m[q] * sc;
      if (sat) y = (y > 65504.f && y < __builtin_inff()) ? 65504.f : ((y < -65504.f && y > -__builtin_inff()) ? -65504.f : y);
      hi[q] = (_Float16)y;
      lo[q] = (_Float16)(y - (float)hi[q]);
    }
    const int n = c4 * 4;
    _Float16 *o = out + (idx / C4) * (size_t)(2 * C) + (size_t)((n >> 4) * 32 + (n & 15));
    *reinterpret_cast<f16x4 *>(o) = hi;
    *reinterpret_cast<f16x4 *>(o + 16) = lo;
  }
  if (amax_out) {        // (uniform)
    __shared__ float red[4];
    ivx_amax_commit_wg(amax_out, omax, red, (int)blockIdx.x);
  }
}

extern "C" int ivx_maxpool2d_fwd_pair(const float *in, int32_t B, int32_t H, int32_t W, int32_t C, int32_t k, int32_t s, int32_t p, void *out,
                                      const uint32_t *amax_in, float wbound, float sbound, float *out_scale, uint32_t *amax_out,
                                      ivx_stream_t stream) {
  IVX_REQUIRE(in && out && amax_in && out_scale, "ivx_maxpool2d_fwd_pair: null argument");
  IVX_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0 && C % 16 == 0, "ivx_maxpool2d_fwd_pair: bad dims (C %% 16 must be 0)");
  IVX_REQUIRE(k > 0 && s > 0 && p >= 0 && 2 * p <= k, "ivx_maxpool2d_fwd_pair: bad window");
  IVX_REQUIRE(wbound >= 0.f && sbound >= 0.f, "ivx_maxpool2d_fwd_pair: negative bound terms");
  const int Ho = (H + 2 * p - k) / s + 1, Wo = (W + 2 * p - k) / s + 1;
  IVX_REQUIRE(Ho > 0 && Wo > 0, "ivx_maxpool2d_fwd_pair: empty output");
  const size_t total = (size_t)B * Ho * Wo * (C / 4);
  size_t blocks = (total + 255) / 256;
  if (blocks > 256 * 32) blocks = 256 * 32;
  hipLaunchKernelGGL(maxpool2d_nhwc_pair_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, in, B, H, W, C, k, s, p, Ho, Wo,
                     (_Float16 *)out, amax_in, wbound, sbound, out_scale, amax_out);
  IVX_CHECK_LAUNCH("ivx_maxpool2d_fwd_pair");
  return IVX_OK;
}

// IVX_F16_PAIR [n] -> fp32 [n]: x = (hi + lo) / scale (exact: hi + lo has at most 22 significant bits, the scale is a power of two)
__global__ __launch_bounds__(256) void pair_merge_kernel(const _Float16 *in, size_t n16, const float *scale, float *out) {
  const float inv = scale ? 1.0f / *scale : 1.0f;
  for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < n16; t += (size_t)gridDim.x * blockDim.x) {
    const _Float16 *g = in + t * 32;
#pragma unroll
    for (int e = 0; e < 16; ++e) out[t * 16 + e] = ((float)g[e] + (float)g[16 + e]) * inv;
  }
}

extern "C" int ivx_f16_pair_merge(const void *in, int64_t n, const float *scale_dev, float *out, ivx_stream_t stream) {
  IVX_REQUIRE(in && out && n >= 0 && n % 16 == 0, "ivx_f16_pair_merge: null argument or n not a multiple of 16");
  if (n == 0) return IVX_OK;
  const size_t n16 = (size_t)n / 16;
  size_t blocks = (n16 + 255) / 256;
  if (blocks > 256 * 32) blocks = 256 * 32;
  hipLaunchKernelGGL(pair_merge_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const _Float16 *)in, n16, scale_dev, out);
  IVX_CHECK_LAUNCH("ivx_f16_pair_merge");
  return IVX_OK;
}


// max |x| over a bf16 tensor into *out (atomic max on the bits of a non-negative float; the caller zeroes it): the calibration pass of the
// e4m3 trunk (ivx_model_calibrate_fp8) records the maximum of every tensor it is going to store as e4m3.
__global__ __launch_bounds__(256) void amax_bf16_kernel(const __bf16 *x, size_t n, unsigned *out) {
  float m = 0.f;
  for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (size_t)gridDim.x * blockDim.x) m = fmaxf(m, fabsf((float)x[t]));
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  if ((threadIdx.x & 63) == 0 && m > 0.f) atomicMax(out, __float_as_uint(m));
}

extern "C" int ivx_amax_bf16(const void *x, int64_t n, float *out, ivx_stream_t stream) {
  IVX_REQUIRE(x && out && n >= 0, "ivx_amax_bf16: bad argument");
  if (n == 0) return IVX_OK;
  size_t blocks = ((size_t)n + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(amax_bf16_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const __bf16 *)x, (size_t)n, (unsigned *)out);
  IVX_CHECK_LAUNCH("ivx_amax_bf16");
  return IVX_OK;
}
