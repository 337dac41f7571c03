// Modulated deformable convolution (DCNv2) -- the nuScenes reference backbone uses it in ResNet stages 3-4
// (configs/imvoxelnet/imvoxelnet_nuscenes.py:13-14; the op itself lives in mmcv-full 1.2.7, which is not in the
// reference tree: restated from its published algorithm, parity UNPINNED).
//
//   out[b,co,h,w] = sum_{k=(i,j)} sum_c  W[co,c,i,j] * mask_k(h,w) * bilinear(x[b,c], h*s - p + i*d + dh_k(h,w), w*s - p + j*d + dw_k(h,w))
//
// Split in two: this kernel builds the modulated, bilinearly sampled columns  col[m][k][c]  (channels-last, so a
// sample is four contiguous C-vectors blended with scalar weights); the contraction over (k, c) is then an ordinary
// 1x1 convolution with K = kh*kw*C on the MFMA kernel (ivx_conv_fwd), with BN + ReLU fused in its epilogue.
// Offsets/mask come from the companion conv (ModulatedDeformConv2dPack.conv_offset) as raw channels
// [dh_0, dw_0, dh_1, dw_1, ..., dh_{K-1}, dw_{K-1}, m_0 .. m_{K-1}]  (chunk(3) + cat(o1, o2) of mmcv == raw order), mask = sigmoid.
#include "ivx_common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void dcn_im2col_kernel(const float *x, const float *om, int B, int H, int W, int C, int kh,
                                                         int kw, int stride, int pad, int dil, int Ho, int Wo, int OMC, float *col, int xcd_order) {
  const int C4 = C >> 2;
  const int KK = kh * kw;
  const size_t total = (size_t)B * Ho * Wo * KK * C4;
  size_t first = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (xcd_order) {     // one pass, grid rounded up to 8 * per blocks: XCD x takes the x-th contiguous eighth (see dcn_im2col_pair_kernel)
    const size_t per = gridDim.x >> 3;
    first = ((size_t)(blockIdx.x & 7) * per + (blockIdx.x >> 3)) * blockDim.x + threadIdx.x;
  }
  for (size_t idx = first; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const int c4 = (int)(idx % C4);
    size_t t = idx / C4;
    const int k = (int)(t % KK);
    t /= KK;
    const int wo = (int)(t % Wo);
    t /= Wo;
    const int ho = (int)(t % Ho);
    const int b = (int)(t / Ho);
    const float *o = om + (((size_t)b * Ho + ho) * Wo + wo) * OMC;
    const float dh = o[2 * k], dw = o[2 * k + 1];
    const float mk = 1.0f / (1.0f + expf(-o[2 * KK + k]));
    const int i = k / kw, j = k - i * kw;
    const float h_im = (float)(ho * stride - pad + i * dil) + dh;
    const float w_im = (float)(wo * stride - pad + j * dil) + dw;
    f32x4 val = {0.f, 0.f, 0.f, 0.f};
    if (h_im > -1.f && w_im > -1.f && h_im < (float)H && w_im < (float)W) {
      const int h_low = (int)floorf(h_im), w_low = (int)floorf(w_im);
      const int h_high = h_low + 1, w_high = w_low + 1;
      const float lh = h_im - h_low, lw = w_im - w_low, hh = 1.f - lh, hw = 1.f - lw;
      const float *xb = x + (size_t)b * H * W * C + c4 * 4;
      auto at = [&](int hy, int wx) { return *reinterpret_cast<const f32x4 *>(xb + ((size_t)hy * W + wx) * C); };
      const f32x4 z = {0.f, 0.f, 0.f, 0.f};
      const f32x4 v1 = (h_low >= 0 && w_low >= 0) ? at(h_low, w_low) : z;
      const f32x4 v2 = (h_low >= 0 && w_high <= W - 1) ? at(h_low, w_high) : z;
      const f32x4 v3 = (h_high <= H - 1 && w_low >= 0) ? at(h_high, w_low) : z;
      const f32x4 v4 = (h_high <= H - 1 && w_high <= W - 1) ? at(h_high, w_high) : z;
      const float w1 = hh * hw, w2 = hh * lw, w3 = lh * hw, w4 = lh * lw;
#pragma unroll
      for (int q = 0; q < 4; ++q) val[q] = (w1 * v1[q] + w2 * v2[q] + w3 * v3[q] + w4 * v4[q]) * mk;
    }
    *reinterpret_cast<f32x4 *>(col + idx * 4) = val;
  }
}

extern "C" int ivx_dcn_im2col_fwd(const float *x, const float *offset_mask, int32_t B, int32_t H, int32_t W, int32_t C, int32_t kh,
                                  int32_t kw, int32_t stride, int32_t pad, int32_t dil, int32_t om_channels, float *col,
                                  ivx_stream_t stream) {
  IVX_REQUIRE(x && offset_mask && col, "ivx_dcn_im2col_fwd: null argument");
  IVX_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, "ivx_dcn_im2col_fwd: bad dims (C %% 4 must be 0)");
  IVX_REQUIRE(kh > 0 && kw > 0 && stride > 0 && pad >= 0 && dil > 0, "ivx_dcn_im2col_fwd: bad window");
  IVX_REQUIRE(om_channels >= 3 * kh * kw, "ivx_dcn_im2col_fwd: offset/mask map needs 3*kh*kw channels (deform_groups = 1)");
  const int Ho = (H + 2 * pad - (dil * (kh - 1) + 1)) / stride + 1;
  const int Wo = (W + 2 * pad - (dil * (kw - 1) + 1)) / stride + 1;
  IVX_REQUIRE(Ho > 0 && Wo > 0, "ivx_dcn_im2col_fwd: empty output");
  const size_t total = (size_t)B * Ho * Wo * kh * kw * (C / 4);
  size_t blocks = (total + 255) / 256;
  int xcd_order = 0;
  if (blocks <= 256 * 256 - 8) { blocks = (blocks + 7) / 8 * 8; xcd_order = 1; }
  else blocks = 256 * 64;
  hipLaunchKernelGGL(dcn_im2col_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, offset_mask, B, H, W, C, kh, kw,
                     stride, pad, dil, Ho, Wo, om_channels, col, xcd_order);
  IVX_CHECK_LAUNCH("ivx_dcn_im2col_fwd");
  return IVX_OK;
}

// The same columns inside a chain of fp16-pair activations (include/imvoxel.h, "Chained fp16-pair activations"; round 4): x is an
// IVX_F16_PAIR map, col an IVX_F16_PAIR tensor with 9 * C channels -- so that the contraction over (k, c) runs on the 16-bit matrix cores
// (ivx_conv_fwd_pio) instead of the fp32 MFMA (K = 2304 / 4608: the DCNv2 stages were 8 ms of a 20 ms nuScenes step).  The columns keep
// the SCALE of x: every column is a convex combination of four values of x times a mask in (0, 1), so |col| <= max |x|, and since the
// scale is a power of two, blending the scaled values equals scaling the blend bit for bit -- the kernel never multiplies by the scale; it
// copies it to the column tensor's scalar block and records max |col| (true units) for the bound of the next layer.
__global__ __launch_bounds__(256) void dcn_im2col_pair_kernel(const _Float16 *x, const float *x_scale, const float *om, int B, int H, int W, int C,
                                                              int kh, int kw, int stride, int pad, int dil, int Ho, int Wo, int OMC, _Float16 *col,
                                                              float *col_scale, unsigned *amax_out, int xcd_order) {
  // one thread: 8 channels of one pixel, the kh * kw taps one after the other -- 16 bytes of hi halves and, 32 bytes further, 16 bytes of lo
  // halves per corner and for the column.  The corners of a pixel's taps overlap (offsets of a trained net are a few pixels), so walking
  // the taps in one thread turns most of the 4 x 9 corner reads into L1 hits; with one thread per (pixel, tap) the taps of a pixel sit in
  // different workgroups and every corner comes from L2.
  typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
  const int C8 = C >> 3;
  const int KK = kh * kw;
  const size_t total = (size_t)B * Ho * Wo * C8;
  const float sx = *x_scale;
  if (blockIdx.x == 0 && threadIdx.x == 0) *col_scale = sx;
  float omax = 0.f;
  // Workgroups are dealt to the 8 XCDs round-robin; each XCD has its own L2.  In launch order every XCD would touch the WHOLE map (PMC,
  // nuScenes stage 3: 650 MB fetched per launch for a 36 MB map, L2 hit rate 0.56, 6.3 TB/s of fabric traffic = the bound).  When one
  // pass covers the problem, XCD x takes the x-th contiguous eighth of the pixel range instead: its L2 then holds an eighth of the map.
  size_t first = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (xcd_order) {                           // the launcher rounded the grid up to 8 * per blocks and one pass covers the problem
    const size_t per = gridDim.x >> 3;
    first = ((size_t)(blockIdx.x & 7) * per + (blockIdx.x >> 3)) * blockDim.x + threadIdx.x;
  }
  for (size_t idx = first; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const int c8 = (int)(idx % C8);
    size_t t = idx / C8;            // t = output pixel m
    const size_t m = t;
    const int wo = (int)(t % Wo);
    t /= Wo;
    const int ho = (int)(t % Ho);
    const int b = (int)(t / Ho);
    const float *o = om + m * OMC;
    const int n = c8 * 8;
    const _Float16 *xb = x + (size_t)b * H * W * (2 * C) + (size_t)((n >> 4) * 32 + (n & 15));
    for (int k = 0; k < KK; ++k) {
      const float dh = o[2 * k], dw = o[2 * k + 1];
      const float mk = 1.0f / (1.0f + expf(-o[2 * KK + k]));
      const int i = k / kw, j = k - i * kw;
      const float h_im = (float)(ho * stride - pad + i * dil) + dh;
      const float w_im = (float)(wo * stride - pad + j * dil) + dw;
      float val[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) val[q] = 0.f;
      if (h_im > -1.f && w_im > -1.f && h_im < (float)H && w_im < (float)W) {
        const int h_low = (int)floorf(h_im), w_low = (int)floorf(w_im);
        const int h_high = h_low + 1, w_high = w_low + 1;
        const float lh = h_im - h_low, lw = w_im - w_low, hh = 1.f - lh, hw = 1.f - lw;
        const float w1 = hh * hw, w2 = hh * lw, w3 = lh * hw, w4 = lh * lw;
        const bool ok1 = h_low >= 0 && w_low >= 0, ok2 = h_low >= 0 && w_high <= W - 1, ok3 = h_high <= H - 1 && w_low >= 0,
                   ok4 = h_high <= H - 1 && w_high <= W - 1;
        const f16x8 z8 = {0, 0, 0, 0, 0, 0, 0, 0};
        auto ld = [&](bool ok, int hy, int wx, int off) {
          return ok ? *reinterpret_cast<const f16x8 *>(xb + ((size_t)hy * W + wx) * (size_t)(2 * C) + off) : z8;
        };
        const f16x8 h1 = ld(ok1, h_low, w_low, 0), l1 = ld(ok1, h_low, w_low, 16), h2 = ld(ok2, h_low, w_high, 0), l2 = ld(ok2, h_low, w_high, 16);
        const f16x8 h3 = ld(ok3, h_high, w_low, 0), l3 = ld(ok3, h_high, w_low, 16), h4 = ld(ok4, h_high, w_high, 0), l4 = ld(ok4, h_high, w_high, 16);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const float v1 = (float)h1[q] + (float)l1[q], v2 = (float)h2[q] + (float)l2[q], v3 = (float)h3[q] + (float)l3[q], v4 = (float)h4[q] + (float)l4[q];
          val[q] = (w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4) * mk;
        }
      }
      f16x8 hi, lo;
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        omax = fmaxf(omax, fabsf(val[q]));
        hi[q] = (_Float16)val[q];
        lo[q] = (_Float16)(val[q] - (float)hi[q]);
      }
      const int nc = k * C + n;          // channel of the 9 * C columns
      _Float16 *op = col + m * (size_t)(2 * KK * C) + (size_t)((nc >> 4) * 32 + (nc & 15));
      *reinterpret_cast<f16x8 *>(op) = hi;
      *reinterpret_cast<f16x8 *>(op + 16) = lo;
    }
  }
  if (amax_out) {        // (uniform)
    __shared__ float red[16];
    ivx_amax_commit_wg(amax_out, omax / sx, red, (int)blockIdx.x);
  }
}

extern "C" int ivx_dcn_im2col_fwd_pair(const void *x, const float *x_scale, const float *offset_mask, int32_t B, int32_t H, int32_t W, int32_t C,
                                       int32_t kh, int32_t kw, int32_t stride, int32_t pad, int32_t dil, int32_t om_channels, void *col,
                                       float *col_scale, uint32_t *col_amax, ivx_stream_t stream) {
  IVX_REQUIRE(x && x_scale && offset_mask && col && col_scale, "ivx_dcn_im2col_fwd_pair: null argument");
  IVX_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0 && C % 16 == 0, "ivx_dcn_im2col_fwd_pair: bad dims (C %% 16 must be 0)");
  IVX_REQUIRE(kh > 0 && kw > 0 && stride > 0 && pad >= 0 && dil > 0, "ivx_dcn_im2col_fwd_pair: bad window");
  IVX_REQUIRE(om_channels >= 3 * kh * kw, "ivx_dcn_im2col_fwd_pair: offset/mask map needs 3*kh*kw channels (deform_groups = 1)");
  const int Ho = (H + 2 * pad - (dil * (kh - 1) + 1)) / stride + 1;
  const int Wo = (W + 2 * pad - (dil * (kw - 1) + 1)) / stride + 1;
  IVX_REQUIRE(Ho > 0 && Wo > 0, "ivx_dcn_im2col_fwd_pair: empty output");
  const size_t total = (size_t)B * Ho * Wo * (C / 8);
  size_t blocks = (total + 255) / 256;
  int xcd_order = 0;
  if (blocks <= 256 * 64 - 8) { blocks = (blocks + 7) / 8 * 8; xcd_order = 1; }      // one pass: XCD x takes the x-th eighth of the pixels
  else blocks = 256 * 64;
  hipLaunchKernelGGL(dcn_im2col_pair_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const _Float16 *)x, x_scale, offset_mask, B, H, W,
                     C, kh, kw, stride, pad, dil, Ho, Wo, om_channels, (_Float16 *)col, col_scale, col_amax, xcd_order);
  IVX_CHECK_LAUNCH("ivx_dcn_im2col_fwd_pair");
  return IVX_OK;
}

