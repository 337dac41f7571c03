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
                                                         int kw, int stride, int pad, int dil, int Ho, int Wo, int OMC, float *col) {
  const int C4 = C >> 2;
  const int KK = kh * kw;
  const size_t total = (size_t)B * Ho * Wo * KK * C4;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
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
  if (blocks > 256 * 64) blocks = 256 * 64;
  hipLaunchKernelGGL(dcn_im2col_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, offset_mask, B, H, W, C, kh, kw,
                     stride, pad, dil, Ho, Wo, om_channels, col);
  IVX_CHECK_LAUNCH("ivx_dcn_im2col_fwd");
  return IVX_OK;
}
