// TEST INFRASTRUCTURE (oracle/).  A CPU restatement of the op-level entry points of include/imvoxel.h -- the SAME C-ABI
// (names, argument meaning, status codes) served from host memory, so that csrc/model.cpp (the model-level handle: layer graph,
// weight packing, planning) and tests/c/e2e_small.c can be built and run WITHOUT a GPU: SURVEY 8d "same ABI served by the CPU
// restatement".  Built by oracle/cpu_abi/build.py into oracle/_cpuabi/libimvoxel_cpu.so; only tests/ load it.  The product
// (imvoxelnet_amd/_lib.py) loads csrc/libimvoxel_hip.so and nothing else -- there is no CPU fallback.
//
// Each function restates the HIP kernel of the same name (file cited per function), which in turn cites the reference lines it
// replaces.  fp32 only; plain loops + OpenMP, separate multiply and add (-ffp-contract=off), so results differ from the MFMA
// kernels by summation order only; the geometry (unprojection, decoding, rotated NMS) uses the same operation order as the HIP
// kernels and the C oracle (oracle/ivx_oracle.c, linked in for the rotated-box geometry).
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "../../include/imvoxel.h"

void ivx_set_error(const char *fmt, ...);          // csrc/api_common.cpp

extern "C" {                                      // oracle/ivx_oracle.c
int ivxo_nms_rotated_sorted(const float *boxes, int n, float thr, int64_t *keep);
int ivxo_nms_normal_sorted(const float *boxes, int n, float thr, int64_t *keep);
void ivxo_boxes_overlap_bev(const float *a, int na, const float *b, int nb, float *out);
void ivxo_boxes_iou_bev(const float *a, int na, const float *b, int nb, float *out);
int ivxo_aligned_3d_nms(const float *boxes, const float *scores, const int64_t *classes, const int64_t *order, int n, float thresh,
                        int64_t *pick);
}

#define C_REQUIRE(cond, ...)      \
  do {                            \
    if (!(cond)) {                \
      ivx_set_error(__VA_ARGS__); \
      return IVX_ERR_INVALID_ARG; \
    }                             \
  } while (0)

// ---------------------------------------------------------------------------------------------- convolution (csrc/conv_igemm.hip)
extern "C" int ivx_conv_out_dims(const ivx_conv_desc *d, int32_t *Do, int32_t *Ho, int32_t *Wo) {
  C_REQUIRE(d && Do && Ho && Wo, "ivx_conv_out_dims: null argument");
  const int od = (d->D + 2 * d->pd - d->KD) / d->sd + 1, oh = (d->H + 2 * d->ph - d->KH) / d->sh + 1, ow = (d->W + 2 * d->pw - d->KW) / d->sw + 1;
  C_REQUIRE(d->D + 2 * d->pd >= d->KD && d->H + 2 * d->ph >= d->KH && d->W + 2 * d->pw >= d->KW && od > 0 && oh > 0 && ow > 0,
            "ivx_conv_out_dims: kernel larger than padded input");
  *Do = od; *Ho = oh; *Wo = ow;
  return IVX_OK;
}

extern "C" int64_t ivx_conv_workspace_bytes(const ivx_conv_desc *d) { return d ? 0 : -1; }

// nearest-upsampled residual row (res_mode 2), as res2_row_base of the HIP kernel
static size_t res2_row(int m, int Ho, int Wo, int rH, int rW, int Cout) {
  const int ow = m % Wo, t = m / Wo, oh = t % Ho, b = t / Ho;
  int sh_ = (Ho == rH) ? oh : (Ho == 2 * rH ? (oh >> 1) : (int)floorf(oh * ((float)rH / Ho)));
  int sw_ = (Wo == rW) ? ow : (Wo == 2 * rW ? (ow >> 1) : (int)floorf(ow * ((float)rW / Wo)));
  sh_ = sh_ < rH - 1 ? sh_ : rH - 1;
  sw_ = sw_ < rW - 1 ? sw_ : rW - 1;
  return (((size_t)b * rH + sh_) * rW + sw_) * Cout;
}

extern "C" int ivx_conv_fwd_ws(const ivx_conv_desc *d, const void *in_, const void *wgt_, const float *scale, const float *shift,
                               const void *res_, void *out_, void *workspace, int64_t workspace_bytes, ivx_stream_t stream) {
  (void)workspace; (void)workspace_bytes; (void)stream;
  C_REQUIRE(d && in_ && wgt_ && out_, "ivx_conv_fwd: null argument");
  C_REQUIRE(d->in_dtype == IVX_F32 && d->out_dtype == IVX_F32, "ivx_conv_fwd (CPU restatement): fp32 only");
  C_REQUIRE(d->Cin % 4 == 0, "ivx_conv_fwd: Cin (%d) must be a multiple of 4 (pad the input channels)", d->Cin);
  C_REQUIRE(d->wgt_layout == 0 || (d->wgt_layout == 1 && d->Cin % 32 == 0), "ivx_conv_fwd: wgt_layout 1 needs Cin %% 32 == 0");
  C_REQUIRE(d->res_mode >= 0 && d->res_mode <= 2 && (d->res_mode == 0 || res_), "ivx_conv_fwd: bad res_mode / null res");
  int32_t Do, Ho, Wo;
  if (ivx_conv_out_dims(d, &Do, &Ho, &Wo) != IVX_OK) return IVX_ERR_INVALID_ARG;
  const float *in = (const float *)in_, *wgt = (const float *)wgt_, *res = (const float *)res_;
  float *out = (float *)out_;
  const int Cin = d->Cin, Cout = d->Cout, ntap = d->KD * d->KH * d->KW;
  const int64_t K = (int64_t)ntap * Cin, M = (int64_t)d->B * Do * Ho * Wo;
  const int Cr = d->out_mode == 1 ? Cout / 8 : Cout;
  const float post = d->post_scale == 0.f ? 1.0f : d->post_scale, rs = d->res_scale == 0.f ? 1.0f : d->res_scale;
  // filters re-laid [tap][ci][co] once per call: the inner loop then runs over the output channels (contiguous, no reduction
  // dependency -> vectorises); the sum over (tap, ci) of every output element keeps the tap-major, channel-ascending order
  std::vector<float> wt((size_t)ntap * Cin * Cout);
#pragma omp parallel for schedule(static)
  for (int n = 0; n < Cout; ++n)
    for (int tap = 0; tap < ntap; ++tap)
      for (int c = 0; c < Cin; ++c) {
        const size_t src = d->wgt_layout == 1 ? ((size_t)(c / 32) * ntap + tap) * 32 + (c % 32) : (size_t)tap * Cin + c;
        wt[((size_t)tap * Cin + c) * Cout + n] = wgt[(size_t)n * K + src];
      }
#pragma omp parallel for schedule(static)
  for (int64_t m = 0; m < M; ++m) {
    const int ow = (int)(m % Wo);
    int64_t t = m / Wo;
    const int oh = (int)(t % Ho);
    t /= Ho;
    const int od = (int)(t % Do), b = (int)(t / Do);
    std::vector<float> acc(Cout, 0.f);
    float *ap = acc.data();
    for (int a = 0; a < d->KD; ++a) {
      const int id = od * d->sd - d->pd + a;
      if ((unsigned)id >= (unsigned)d->D) continue;
      for (int e = 0; e < d->KH; ++e) {
        const int ih = oh * d->sh - d->ph + e;
        if ((unsigned)ih >= (unsigned)d->H) continue;
        for (int f = 0; f < d->KW; ++f) {
          const int iw = ow * d->sw - d->pw + f;
          if ((unsigned)iw >= (unsigned)d->W) continue;
          const float *x = in + ((((size_t)b * d->D + id) * d->H + ih) * d->W + iw) * Cin;
          const float *wtap = wt.data() + (size_t)((a * d->KH + e) * d->KW + f) * Cin * Cout;
          for (int c = 0; c < Cin; ++c) {
            const float xc = x[c];
            const float *wr = wtap + (size_t)c * Cout;
#pragma omp simd
            for (int n = 0; n < Cout; ++n) ap[n] += xc * wr[n];
          }
        }
      }
    }
    for (int n = 0; n < Cout; ++n) {          // conv_finish / conv_store_one of the HIP kernel
      size_t o = (size_t)m * Cout + n, ridx = o;
      int ch = n;
      if (d->out_mode == 1) {                 // ConvTranspose3d(k2, s2): column n = ((a*2+e)*2+f)*Cr + co
        const int tapn = n / Cr, a2 = tapn >> 2, e2 = (tapn >> 1) & 1, f2 = tapn & 1;
        ch = n - tapn * Cr;
        const int w_ = (int)(m % d->W);
        int64_t q = m / d->W;
        const int h_ = (int)(q % d->H);
        q /= d->H;
        const int d_ = (int)(q % d->D), b_ = (int)(q / d->D);
        o = ridx = ((((size_t)b_ * 2 * d->D + 2 * d_ + a2) * 2 * d->H + 2 * h_ + e2) * 2 * d->W + 2 * w_ + f2) * Cr + ch;
      } else if (d->res_mode == 2) {
        ridx = res2_row((int)m, Ho, Wo, d->res_h, d->res_w, Cout) + n;
      }
      float v = acc[n] * (scale ? scale[ch] : 1.0f) + (shift ? shift[ch] : 0.0f);
      if (d->res_mode && !d->res_after_act) v = res[ridx] * rs + v;
      if (d->relu) v = v > 0.f ? v : 0.f;
      if (d->res_mode && d->res_after_act) v = res[ridx] * rs + v;
      out[o] = v * post;
    }
  }
  return IVX_OK;
}

extern "C" int ivx_conv_fwd(const ivx_conv_desc *d, const void *in, const void *wgt, const float *scale, const float *shift, const void *res,
                            void *out, ivx_stream_t stream) {
  return ivx_conv_fwd_ws(d, in, wgt, scale, shift, res, out, nullptr, 0, stream);
}

// ---------------------------------------------------------------------------------------------- chained fp16-pair activations
// (include/imvoxel.h "Chained fp16-pair activations"; csrc/conv_igemm.hip conv_epilogue_wide_pio, csrc/pool_layout.hip).  The CPU
// restatement decodes the pairs to the fp32 values they stand for, runs the fp32 convolution above and encodes the result with the
// scale rule of the device (bound of the output from the measured maxima of the operands): the handle's planning of the chain -- which
// tensors are pairs, which scalar block feeds which layer -- runs unchanged on top.  Values agree with the device to fp32 rounding
// (the device drops the lo*lo term of every product, 2^-22 of it).
static uint16_t c_f32_to_f16(float f) {
  uint32_t x;
  memcpy(&x, &f, 4);
  const uint32_t sign = (x >> 16) & 0x8000u, ax = x & 0x7fffffffu;
  if (ax >= 0x7f800000u) return (uint16_t)(sign | (ax > 0x7f800000u ? 0x7e00u : 0x7c00u));
  if (ax < 0x38800000u) {
    float a;
    memcpy(&a, &ax, 4);
    return (uint16_t)(sign | (uint32_t)lrintf(a * 16777216.0f));
  }
  const uint32_t mant = ax & 0x7fffffu, exp = (ax >> 23) - 127 + 15;
  uint32_t h = (exp << 10) | (mant >> 13);
  const uint32_t rem = mant & 0x1fffu;
  if (rem > 0x1000u || (rem == 0x1000u && (h & 1u))) ++h;
  return (uint16_t)(sign | h);
}
static float c_f16_to_f32(uint16_t h) {
  const uint32_t sign = (uint32_t)(h & 0x8000u) << 16, e = (h >> 10) & 31u, m = h & 0x3ffu;
  float f;
  if (e == 0) {
    f = (float)m * 5.9604644775390625e-08f;
    return sign ? -f : f;
  }
  const uint32_t x = sign | (e == 31 ? 0x7f800000u | (m << 13) : ((e - 15 + 127) << 23) | (m << 13));
  memcpy(&f, &x, 4);
  return f;
}
static float c_pow2_scale(float amax) {          // ivx_pow2_scale of csrc/ivx_common.h
  if (!(amax > 0.f) || !(amax < 3.0e38f)) return 1.0f;
  int e;
  (void)frexpf(amax, &e);
  int k = 15 - e;
  k = k < -120 ? -120 : (k > 120 ? 120 : k);
  return ldexpf(1.0f, k);
}
static float c_amax_read(const uint32_t *slots) {
  float a = 0.f;
  for (int i = 0; i < IVX_AMAX_SLOTS; ++i) { float v; memcpy(&v, slots + i, 4); a = v > a ? v : a; }
  return a;
}
static void c_amax_commit(uint32_t *slots, float m) {
  float cur;
  memcpy(&cur, slots, 4);
  if (m > cur) memcpy(slots, &m, 4);
}
// pair tensor [rows, 2C] halves -> fp32 [rows, C] of (hi + lo) * inv
static void c_pair_decode(const uint16_t *p, int64_t rows, int C, float inv, float *out) {
#pragma omp parallel for schedule(static)
  for (int64_t r = 0; r < rows; ++r)
    for (int n = 0; n < C; ++n) {
      const uint16_t *g = p + r * 2 * C + (n >> 4) * 32 + (n & 15);
      out[r * C + n] = (c_f16_to_f32(g[0]) + c_f16_to_f32(g[16])) * inv;
    }
}
static void c_pair_encode(const float *x, int64_t rows, int C, float s, uint16_t *p) {
#pragma omp parallel for schedule(static)
  for (int64_t r = 0; r < rows; ++r)
    for (int n = 0; n < C; ++n) {
      const float y = x[r * C + n] * s;
      const uint16_t h = c_f32_to_f16(y);
      uint16_t *g = p + r * 2 * C + (n >> 4) * 32 + (n & 15);
      g[0] = h;
      g[16] = c_f32_to_f16(y - c_f16_to_f32(h));
    }
}

extern "C" int64_t ivx_conv_pio_workspace_bytes(const ivx_conv_desc *d, const ivx_pair_io *io) { return d && io ? 0 : -1; }

extern "C" int ivx_conv_fwd_pio(const ivx_conv_desc *d, const ivx_pair_io *io, const void *in_, const void *wgt_, const float *scale, const float *shift,
                                const void *res_, void *out_, void *, int64_t, ivx_stream_t stream) {
  C_REQUIRE(d && io && in_ && wgt_ && out_, "ivx_conv_fwd_pio: null argument");
  C_REQUIRE(d->in_dtype == IVX_F16_PAIR && d->out_mode == 0 && d->Cout % 4 == 0 && d->Cin % 32 == 0 && d->wgt_layout == 1 &&
                (d->out_dtype == IVX_F32 || d->out_dtype == IVX_F16_PAIR),
            "ivx_conv_fwd_pio (CPU restatement): IVX_F16_PAIR input, chunk-major pair filters, out_mode 0, Cout %% 4 == 0");
  const bool out_pair = d->out_dtype == IVX_F16_PAIR, res_pair = d->res_mode && io->res_dtype == IVX_F16_PAIR;
  C_REQUIRE(!out_pair || (d->Cout % 16 == 0 && io->out_scale && io->amax_in && (d->res_mode == 0 || io->amax_res)),
            "ivx_conv_fwd_pio: a pair output needs Cout %% 16 == 0, out_scale, amax_in (and amax_res with a residual)");
  C_REQUIRE(!res_pair || (d->Cout % 16 == 0 && io->res_scale), "ivx_conv_fwd_pio: a pair residual needs Cout %% 16 == 0 and res_scale");
  int32_t Do, Ho, Wo;
  if (ivx_conv_out_dims(d, &Do, &Ho, &Wo) != IVX_OK) return IVX_ERR_INVALID_ARG;
  const int64_t rows_in = (int64_t)d->B * d->D * d->H * d->W, M = (int64_t)d->B * Do * Ho * Wo;
  const int ntap = d->KD * d->KH * d->KW;
  std::vector<float> x((size_t)rows_in * d->Cin), w((size_t)d->Cout * ntap * d->Cin), r, o;
  c_pair_decode((const uint16_t *)in_, rows_in, d->Cin, io->in_scale ? 1.0f / *io->in_scale : 1.0f, x.data());
  // pair filters [Cout][Cin/32][taps][hi16 lo16 hi16 lo16] -> fp32 chunk-major [Cout][Cin/32][taps][32] of s_w * w (scale[] carries 1 / s_w)
  c_pair_decode((const uint16_t *)wgt_, (int64_t)d->Cout * (d->Cin / 32) * ntap, 32, 1.0f, w.data());
  ivx_conv_desc f = *d;
  f.in_dtype = IVX_F32; f.out_dtype = IVX_F32; f.res_scale = 1.0f;
  const float *resf = (const float *)res_;
  if (res_pair) {
    const int64_t rrows = d->res_mode == 2 ? (int64_t)d->B * d->res_h * d->res_w : M;
    r.resize((size_t)rrows * d->Cout);
    c_pair_decode((const uint16_t *)res_, rrows, d->Cout, 1.0f / *io->res_scale, r.data());
    resf = r.data();
  }
  float *outf = (float *)out_;
  if (out_pair) { o.resize((size_t)M * d->Cout); outf = o.data(); }
  int rc = ivx_conv_fwd_ws(&f, x.data(), w.data(), scale, shift, resf, outf, nullptr, 0, stream);
  if (rc != IVX_OK) return rc;
  float omax = 0.f;
  for (int64_t i = 0; i < M * d->Cout; ++i) omax = fabsf(outf[i]) > omax ? fabsf(outf[i]) : omax;
  if (io->amax_out) c_amax_commit(io->amax_out, omax);
  if (out_pair) {
    const float a = c_amax_read(io->amax_in), rr = (d->res_mode && io->amax_res) ? c_amax_read(io->amax_res) : 0.f;
    const float post = d->post_scale == 0.f ? 1.0f : d->post_scale;
    const float bound = (a * io->wbound + io->sbound + rr) * fabsf(post) * 1.001f;
    const float s = !(bound < 3.0e38f) ? 0.00390625f : c_pow2_scale(bound);      // (non-finite bound: fixed scale; the device also saturates)
    *io->out_scale = s;
    c_pair_encode(outf, M, d->Cout, s, (uint16_t *)out_);
  }
  return IVX_OK;
}

// csrc/bottleneck.hip: one identity bottleneck in one launch.  The CPU restatement runs the three pair convolutions above with the scales of the
// device's BOUND chain for the two intermediates (slots handed to the layers hold the bound, so that each layer's own rule reproduces it).
extern "C" int ivx_bottleneck_supported(const ivx_bottleneck_desc *d) {
  if (!d || d->B <= 0 || d->H <= 0 || d->W <= 0) return 0;
  if (d->P != 64 && d->P != 128) return 0;
  return (int64_t)d->B * d->H * d->W * d->P * 16 < (1LL << 31);
}
extern "C" int ivx_bottleneck_fwd_pio(const ivx_bottleneck_desc *d, const ivx_bottleneck_io *io, const void *in, const void *w1, const float *scale1,
                                      const float *shift1, const void *w2, const float *scale2, const float *shift2, const void *w3,
                                      const float *scale3, const float *shift3, void *out, ivx_stream_t stream) {
  C_REQUIRE(d && io && in && w1 && w2 && w3 && out && io->in_scale && io->amax_in && io->out_scale, "ivx_bottleneck_fwd_pio: null argument");
  C_REQUIRE(ivx_bottleneck_supported(d), "ivx_bottleneck_fwd_pio: planes must be 64 or 128 and the tensor below 2 GiB");
  const int P = d->P, C4 = 4 * P;
  const int64_t rows = (int64_t)d->B * d->H * d->W;
  const float a = c_amax_read(io->amax_in);
  const float b1 = (a * io->wbound[0] + io->sbound[0]) * 1.001f, b2 = (b1 * io->wbound[1] + io->sbound[1]) * 1.001f;
  const float b3 = (b2 * io->wbound[2] + io->sbound[2] + a) * 1.001f;
  const bool sat = !(b3 < 3.0e38f);
  std::vector<uint16_t> m1((size_t)rows * 2 * P), m2((size_t)rows * 2 * P);
  uint32_t sl1[IVX_AMAX_SLOTS + 1] = {0}, sl2[IVX_AMAX_SLOTS + 1] = {0}, scratch[IVX_AMAX_SLOTS] = {0};
  ivx_conv_desc c;
  memset(&c, 0, sizeof(c));
  c.B = d->B; c.D = 1; c.H = d->H; c.W = d->W; c.Cin = C4; c.Cout = P; c.KD = c.KH = c.KW = 1; c.sd = c.sh = c.sw = 1;
  c.relu = 1; c.wgt_layout = 1; c.post_scale = 1.0f; c.in_dtype = IVX_F16_PAIR; c.out_dtype = IVX_F32; c.res_scale = 1.0f;
  std::vector<float> t((size_t)rows * C4);
  // conv1 -> fp32, then encoded with the bound's scale (not the layer rule's: identical here, written out so that the chain is explicit)
  ivx_pair_io p1;
  memset(&p1, 0, sizeof(p1));
  p1.in_scale = io->in_scale; p1.amax_in = io->amax_in; p1.amax_out = scratch;
  int rc = ivx_conv_fwd_pio(&c, &p1, in, w1, scale1, shift1, nullptr, t.data(), nullptr, 0, stream);
  if (rc != IVX_OK) return rc;
  float s1 = sat ? 0.00390625f : c_pow2_scale(b1), s2 = sat ? 0.00390625f : c_pow2_scale(b2), s3 = sat ? 0.00390625f : c_pow2_scale(b3);
  c_pair_encode(t.data(), rows, P, s1, m1.data());
  memcpy(&sl1[IVX_AMAX_SLOTS], &s1, 4);
  c.Cin = P; c.KH = c.KW = 3; c.ph = c.pw = 1;
  ivx_pair_io p2;
  memset(&p2, 0, sizeof(p2));
  p2.in_scale = (const float *)&sl1[IVX_AMAX_SLOTS]; p2.amax_in = sl1; p2.amax_out = scratch;
  rc = ivx_conv_fwd_pio(&c, &p2, m1.data(), w2, scale2, shift2, nullptr, t.data(), nullptr, 0, stream);
  if (rc != IVX_OK) return rc;
  c_pair_encode(t.data(), rows, P, s2, m2.data());
  memcpy(&sl2[IVX_AMAX_SLOTS], &s2, 4);
  c.KH = c.KW = 1; c.ph = c.pw = 0; c.Cout = C4; c.res_mode = 1;
  ivx_pair_io p3;
  memset(&p3, 0, sizeof(p3));
  p3.in_scale = (const float *)&sl2[IVX_AMAX_SLOTS]; p3.amax_in = sl2; p3.res_dtype = IVX_F16_PAIR; p3.res_scale = io->in_scale; p3.amax_res = io->amax_in;
  p3.amax_out = io->amax_out ? io->amax_out : scratch;
  rc = ivx_conv_fwd_pio(&c, &p3, m2.data(), w3, scale3, shift3, in, t.data(), nullptr, 0, stream);
  if (rc != IVX_OK) return rc;
  *io->out_scale = s3;
  c_pair_encode(t.data(), rows, C4, s3, (uint16_t *)out);
  return IVX_OK;
}

// csrc/bottleneck.hip, projection form: the first block of stage 1 in one launch.  CPU restatement: conv1 and conv2 as above (scales of the bound chain),
// then ONE 1x1 convolution over the concatenation [x | mid2] with the joint filter bank (BN scales folded in, scale3d = 1 / s_w, shift3d = shift3 + shiftd).
extern "C" int ivx_bottleneck_proj_supported(const ivx_bottleneck_desc *d, int32_t Cin) {
  if (!d || d->B <= 0 || d->H <= 0 || d->W <= 0) return 0;
  if (d->P != 64 || Cin != 64) return 0;
  return (int64_t)d->B * d->H * d->W * d->P * 16 < (1LL << 31);
}
extern "C" int ivx_bottleneck_proj_fwd_pio(const ivx_bottleneck_desc *d, int32_t Cin, const ivx_bottleneck_io *io, float wbound_shortcut, const void *in,
                                           const void *w1, const float *scale1, const float *shift1, const void *w2, const float *scale2,
                                           const float *shift2, const void *w3d, const float *scale3d, const float *shift3d, void *out,
                                           ivx_stream_t stream) {
  C_REQUIRE(d && io && in && w1 && w2 && w3d && out && io->in_scale && io->amax_in && io->out_scale, "ivx_bottleneck_proj_fwd_pio: null argument");
  C_REQUIRE(ivx_bottleneck_proj_supported(d, Cin), "ivx_bottleneck_proj_fwd_pio: built for planes = Cin = 64 and the output below 2 GiB");
  const int P = d->P, C4 = 4 * P, K = Cin + P;
  const int64_t rows = (int64_t)d->B * d->H * d->W;
  const float a = c_amax_read(io->amax_in);
  const float b1 = (a * io->wbound[0] + io->sbound[0]) * 1.001f, b2 = (b1 * io->wbound[1] + io->sbound[1]) * 1.001f;
  const float b3 = (b2 * io->wbound[2] + io->sbound[2] + a * wbound_shortcut) * 1.001f;
  const bool sat = !(b3 < 3.0e38f);
  const float s1 = sat ? 0.00390625f : c_pow2_scale(b1), s2 = sat ? 0.00390625f : c_pow2_scale(b2), s3 = sat ? 0.00390625f : c_pow2_scale(b3);
  std::vector<uint16_t> m1((size_t)rows * 2 * P), m2((size_t)rows * 2 * P);
  uint32_t sl1[IVX_AMAX_SLOTS + 1] = {0}, scratch[IVX_AMAX_SLOTS] = {0};
  ivx_conv_desc c;
  memset(&c, 0, sizeof(c));
  c.B = d->B; c.D = 1; c.H = d->H; c.W = d->W; c.Cin = Cin; c.Cout = P; c.KD = c.KH = c.KW = 1; c.sd = c.sh = c.sw = 1;
  c.relu = 1; c.wgt_layout = 1; c.post_scale = 1.0f; c.in_dtype = IVX_F16_PAIR; c.out_dtype = IVX_F32; c.res_scale = 1.0f;
  std::vector<float> t((size_t)rows * C4);
  ivx_pair_io p1;
  memset(&p1, 0, sizeof(p1));
  p1.in_scale = io->in_scale; p1.amax_in = io->amax_in; p1.amax_out = scratch;
  int rc = ivx_conv_fwd_pio(&c, &p1, in, w1, scale1, shift1, nullptr, t.data(), nullptr, 0, stream);
  if (rc != IVX_OK) return rc;
  c_pair_encode(t.data(), rows, P, s1, m1.data());
  memcpy(&sl1[IVX_AMAX_SLOTS], &s1, 4);
  c.Cin = P; c.KH = c.KW = 3; c.ph = c.pw = 1;
  ivx_pair_io p2;
  memset(&p2, 0, sizeof(p2));
  p2.in_scale = (const float *)&sl1[IVX_AMAX_SLOTS]; p2.amax_in = sl1; p2.amax_out = scratch;
  rc = ivx_conv_fwd_pio(&c, &p2, m1.data(), w2, scale2, shift2, nullptr, t.data(), nullptr, 0, stream);
  if (rc != IVX_OK) return rc;
  c_pair_encode(t.data(), rows, P, s2, m2.data());
  // [x | mid2] as fp32 values, the joint bank as fp32 chunk-major filters of s_w * w'
  std::vector<float> xin((size_t)rows * Cin), mid((size_t)rows * P), cat((size_t)rows * K), w((size_t)C4 * K);
  c_pair_decode((const uint16_t *)in, rows, Cin, 1.0f / *io->in_scale, xin.data());
  c_pair_decode(m2.data(), rows, P, 1.0f / s2, mid.data());
#pragma omp parallel for schedule(static)
  for (int64_t r = 0; r < rows; ++r) {
    memcpy(&cat[(size_t)r * K], &xin[(size_t)r * Cin], (size_t)Cin * 4);
    memcpy(&cat[(size_t)r * K + Cin], &mid[(size_t)r * P], (size_t)P * 4);
  }
  c_pair_decode((const uint16_t *)w3d, (int64_t)C4 * (K / 32), 32, 1.0f, w.data());
  ivx_conv_desc f;
  memset(&f, 0, sizeof(f));
  f.B = d->B; f.D = 1; f.H = d->H; f.W = d->W; f.Cin = K; f.Cout = C4; f.KD = f.KH = f.KW = 1; f.sd = f.sh = f.sw = 1;
  f.relu = 1; f.wgt_layout = 1; f.post_scale = 1.0f; f.in_dtype = IVX_F32; f.out_dtype = IVX_F32; f.res_scale = 1.0f;
  rc = ivx_conv_fwd_ws(&f, cat.data(), w.data(), scale3d, shift3d, nullptr, t.data(), nullptr, 0, stream);
  if (rc != IVX_OK) return rc;
  float omax = 0.f;
  for (int64_t i = 0; i < rows * C4; ++i) omax = fabsf(t[i]) > omax ? fabsf(t[i]) : omax;
  if (io->amax_out) c_amax_commit(io->amax_out, omax);
  *io->out_scale = s3;
  c_pair_encode(t.data(), rows, C4, s3, (uint16_t *)out);
  return IVX_OK;
}

extern "C" int ivx_conv_fwd_pio_naive(const ivx_conv_desc *d, const ivx_pair_io *io, const void *in, const void *wgt, const float *scale, const float *shift,
                                      const void *res, void *out, ivx_stream_t stream) {
  return ivx_conv_fwd_pio(d, io, in, wgt, scale, shift, res, out, nullptr, 0, stream);
}

extern "C" int ivx_f16_pair_merge(const void *in, int64_t n, const float *scale_dev, float *out, ivx_stream_t) {
  C_REQUIRE(in && out && n >= 0 && n % 16 == 0, "ivx_f16_pair_merge: null argument or n not a multiple of 16");
  c_pair_decode((const uint16_t *)in, n / 16, 16, scale_dev ? 1.0f / *scale_dev : 1.0f, out);
  return IVX_OK;
}

// csrc/stem.hip: layout change + 7x7 stem + BN + ReLU + max-pool in one launch.  CPU restatement: the image's maximum, then the fp32 direct
// convolution on the filters the packed fragments stand for (s_w * w; scale_p carries 1 / s_w), the exact pool, the pair encoding.
extern "C" int ivx_amax_f32(const float *x, int64_t n, uint32_t *amax, ivx_stream_t) {
  C_REQUIRE(x && amax && n >= 0, "ivx_amax_f32: bad argument");
  float m = 0.f;
  for (int64_t i = 0; i < n; ++i) { const float a = x[i] != x[i] ? INFINITY : fabsf(x[i]); m = a > m ? a : m; }
  c_amax_commit(amax, m);
  return IVX_OK;
}
extern "C" int ivx_stem_pool_out_dims(int32_t H, int32_t W, int32_t *Hp, int32_t *Wp) {
  C_REQUIRE(Hp && Wp && H > 0 && W > 0, "ivx_stem_pool_out_dims: bad argument");
  const int Hc = (H + 6 - 7) / 2 + 1, Wc = (W + 6 - 7) / 2 + 1;
  *Hp = (Hc + 2 - 3) / 2 + 1;
  *Wp = (Wc + 2 - 3) / 2 + 1;
  return IVX_OK;
}
extern "C" int ivx_maxpool2d_fwd_pair(const float *in, int32_t B, int32_t H, int32_t W, int32_t C, int32_t k, int32_t s, int32_t p, void *out,
                                      const uint32_t *amax_in, float wbound, float sbound, float *out_scale, uint32_t *amax_out, ivx_stream_t stream);
extern "C" int ivx_stem_pool_fwd_pair(const float *img, int32_t B, int32_t H, int32_t W, const void *wfrag, const float *scale_p, const float *shift,
                                      float wbound, float sbound, const uint32_t *amax_img, void *out, float *out_scale, uint32_t *amax_out,
                                      ivx_stream_t stream) {
  C_REQUIRE(img && wfrag && scale_p && shift && amax_img && out && out_scale && B > 0 && H >= 7 && W >= 7, "ivx_stem_pool_fwd_pair: bad argument");
  // fragments [nt 2][step 11][hi, lo][lane 64][8] -> tap-major fp32 filters [64][7][7][4] (channel 3: zero)
  std::vector<float> wt((size_t)64 * 49 * 4, 0.f);
  const uint16_t *f = (const uint16_t *)wfrag;
  for (int nt = 0; nt < 2; ++nt)
    for (int kk = 0; kk < 11; ++kk)
      for (int lane = 0; lane < 64; ++lane)
        for (int e = 0; e < 7; ++e) {
          const int n = nt * 32 + (lane & 31), pq = 2 * kk + (lane >> 5);
          if (pq >= 21) continue;
          const size_t o = ((((size_t)nt * 11 + kk) * 2 + 0) * 64 + lane) * 8 + e;
          wt[(((size_t)n * 7 + pq % 7) * 7 + e) * 4 + pq / 7] = c_f16_to_f32(f[o]) + c_f16_to_f32(f[o + 64 * 8]);
        }
  const int64_t S = (int64_t)H * W;
  std::vector<float> x4((size_t)B * S * 4);
  int rc = ivx_nchw_to_nhwc(img, B, 3, S, 4, x4.data(), stream);
  if (rc != IVX_OK) return rc;
  ivx_conv_desc c;
  memset(&c, 0, sizeof(c));
  c.B = B; c.D = 1; c.H = H; c.W = W; c.Cin = 4; c.Cout = 64; c.KD = 1; c.KH = c.KW = 7; c.sd = 1; c.sh = c.sw = 2; c.ph = c.pw = 3;
  c.relu = 1; c.post_scale = 1.0f; c.res_scale = 1.0f;
  const int Hc = (H + 6 - 7) / 2 + 1, Wc = (W + 6 - 7) / 2 + 1;
  std::vector<float> y((size_t)B * Hc * Wc * 64);
  rc = ivx_conv_fwd_ws(&c, x4.data(), wt.data(), scale_p, shift, nullptr, y.data(), nullptr, 0, stream);
  if (rc != IVX_OK) return rc;
  return ivx_maxpool2d_fwd_pair(y.data(), B, Hc, Wc, 64, 3, 2, 1, out, amax_img, wbound, sbound, out_scale, amax_out, stream);
}

// The minimal-filtering form is a device-side optimisation: the CPU restatement reports "not supported" and the handle
// (csrc/model.cpp plan_conv) falls back to the direct convolution, as it does for any layer the Winograd entry points refuse.
extern "C" int ivx_conv_winograd_supported(const ivx_conv_desc *, int32_t) { return 0; }
extern "C" float ivx_conv_winograd_issued_fraction(const ivx_conv_desc *) { return 1.0f; }
extern "C" int64_t ivx_conv_winograd_weight_elems(const ivx_conv_desc *, int32_t) { return -1; }
extern "C" int64_t ivx_conv_winograd_workspace_bytes(const ivx_conv_desc *, int32_t) { return -1; }
// The split-operand (IVX_BF16_PAIR) form is a device-side optimisation too: "not supported" keeps the handle's 3x3x3 layers on the direct fp32 convolution.
extern "C" int ivx_conv_pair_supported(const ivx_conv_desc *) { return 0; }
extern "C" int ivx_bf16_pair_split(const float *, int64_t, void *, ivx_stream_t) {
  ivx_set_error("ivx_bf16_pair_split: the CPU restatement has no split-operand form (ivx_conv_pair_supported returns 0)");
  return IVX_ERR_UNSUPPORTED;
}
static int no_wino(const char *who) {
  ivx_set_error("%s: the CPU restatement has no Winograd form (ivx_conv_winograd_supported returns 0)", who);
  return IVX_ERR_UNSUPPORTED;
}
extern "C" int ivx_conv_winograd_weights(const ivx_conv_desc *, int32_t, const float *, float *, ivx_stream_t) { return no_wino("ivx_conv_winograd_weights"); }
extern "C" int ivx_conv_winograd_input(const ivx_conv_desc *, int32_t, const void *, void *, int64_t, ivx_stream_t) { return no_wino("ivx_conv_winograd_input"); }
extern "C" int ivx_conv_winograd_gemm(const ivx_conv_desc *, int32_t, const float *, void *, int64_t, ivx_stream_t) { return no_wino("ivx_conv_winograd_gemm"); }
extern "C" int ivx_conv_winograd_output(const ivx_conv_desc *, int32_t, const float *, const float *, const void *, void *, void *, int64_t,
                                        ivx_stream_t) { return no_wino("ivx_conv_winograd_output"); }
extern "C" int32_t ivx_conv_winograd_output_blocks(const ivx_conv_desc *, int32_t) { return -1; }
extern "C" int ivx_conv_winograd_output_amax(const ivx_conv_desc *, int32_t, const float *, const float *, const void *, void *, void *, int64_t, float *,
                                             ivx_stream_t) { return no_wino("ivx_conv_winograd_output_amax"); }
extern "C" int ivx_conv_winograd_input_amax(const ivx_conv_desc *, int32_t, const void *, void *, int64_t, const float *, int32_t, ivx_stream_t) {
  return no_wino("ivx_conv_winograd_input_amax");
}
extern "C" int ivx_conv_winograd_fwd(const ivx_conv_desc *, int32_t, const void *, const float *, const float *, const float *, const void *,
                                     void *, void *, int64_t, ivx_stream_t) { return no_wino("ivx_conv_winograd_fwd"); }

// bf16 storage (ivx_model_cfg.storage = IVX_BF16) is a device-side mode: the CPU restatement is fp32 only
static int no_bf16(const char *who) {
  ivx_set_error("%s: the CPU restatement has no bf16 storage mode", who);
  return IVX_ERR_UNSUPPORTED;
}
extern "C" int ivx_amax_bf16(const void *, int64_t, float *, ivx_stream_t) { return no_bf16("ivx_amax_bf16"); }
extern "C" int ivx_image_s2d_bf16(const float *, int32_t, int32_t, int32_t, void *, ivx_stream_t) { return no_bf16("ivx_image_s2d_bf16"); }
extern "C" int ivx_maxpool2d_fwd_bf16(const void *, int32_t, int32_t, int32_t, int32_t, int32_t, int32_t, int32_t, void *, ivx_stream_t) {
  return no_bf16("ivx_maxpool2d_fwd_bf16");
}
extern "C" int ivx_upsample_trilinear2x_fwd_bf16(const void *, int32_t, int32_t, int32_t, int32_t, int32_t, void *, ivx_stream_t) {
  return no_bf16("ivx_upsample_trilinear2x_fwd_bf16");
}
extern "C" int ivx_backproject_mean_fwd_bf16(const void *, int32_t, int32_t, int32_t, int32_t, int32_t, const float *, const float *, const int32_t *,
                                             const float *, int32_t, int32_t, int32_t, void *, uint8_t *, ivx_stream_t) {
  return no_bf16("ivx_backproject_mean_fwd_bf16");
}

// ---------------------------------------------------------------------------------------------- pool / layout (csrc/pool_layout.hip)
extern "C" int ivx_maxpool2d_fwd(const float *in, int32_t B, int32_t H, int32_t W, int32_t C, int32_t k, int32_t s, int32_t p, float *out,
                                 ivx_stream_t) {
  C_REQUIRE(in && out && B > 0 && H > 0 && W > 0 && C > 0 && k > 0 && s > 0 && p >= 0 && 2 * p <= k, "ivx_maxpool2d_fwd: bad argument");
  const int Ho = (H + 2 * p - k) / s + 1, Wo = (W + 2 * p - k) / s + 1;
#pragma omp parallel for collapse(2) schedule(static)
  for (int b = 0; b < B; ++b)
    for (int oh = 0; oh < Ho; ++oh)
      for (int ow = 0; ow < Wo; ++ow)
        for (int c = 0; c < C; ++c) {
          float m = -INFINITY;                       // -inf padding semantics of torch
          for (int e = 0; e < k; ++e) {
            const int ih = oh * s - p + e;
            if ((unsigned)ih >= (unsigned)H) continue;
            for (int f = 0; f < k; ++f) {
              const int iw = ow * s - p + f;
              if ((unsigned)iw >= (unsigned)W) continue;
              const float x = in[(((size_t)b * H + ih) * W + iw) * C + c];
              m = (x > m || x != x) ? x : m;
            }
          }
          out[(((size_t)b * Ho + oh) * Wo + ow) * C + c] = m;
        }
  return IVX_OK;
}

extern "C" int ivx_maxpool2d_fwd_pair(const float *in, int32_t B, int32_t H, int32_t W, int32_t C, int32_t k, int32_t s, int32_t p, void *out,
                                      const uint32_t *amax_in, float wbound, float sbound, float *out_scale, uint32_t *amax_out, ivx_stream_t st) {
  C_REQUIRE(in && out && amax_in && out_scale && C % 16 == 0, "ivx_maxpool2d_fwd_pair: bad argument");
  const int Ho = (H + 2 * p - k) / s + 1, Wo = (W + 2 * p - k) / s + 1;
  std::vector<float> o((size_t)B * Ho * Wo * C);
  int rc = ivx_maxpool2d_fwd(in, B, H, W, C, k, s, p, o.data(), st);
  if (rc != IVX_OK) return rc;
  const float sc = c_pow2_scale((c_amax_read(amax_in) * wbound + sbound) * 1.001f);
  *out_scale = sc;
  float omax = 0.f;
  for (float v : o) omax = fabsf(v) > omax ? fabsf(v) : omax;
  if (amax_out) c_amax_commit(amax_out, omax);
  c_pair_encode(o.data(), (int64_t)B * Ho * Wo, C, sc, (uint16_t *)out);
  return IVX_OK;
}

extern "C" int ivx_nchw_to_nhwc(const float *in, int32_t B, int32_t C, int64_t S, int32_t Cpad, float *out, ivx_stream_t);
extern "C" int ivx_nchw_to_nhwc_amax(const float *in, int32_t B, int32_t C, int64_t S, int32_t Cpad, float *out, uint32_t *amax, ivx_stream_t st) {
  C_REQUIRE(amax, "ivx_nchw_to_nhwc_amax: null amax");
  int rc = ivx_nchw_to_nhwc(in, B, C, S, Cpad, out, st);
  if (rc != IVX_OK) return rc;
  float m = 0.f;
  for (int64_t i = 0; i < (int64_t)B * C * S; ++i) m = fabsf(in[i]) > m ? fabsf(in[i]) : m;
  c_amax_commit(amax, m);
  return IVX_OK;
}

extern "C" int ivx_nchw_to_nhwc(const float *in, int32_t B, int32_t C, int64_t S, int32_t Cpad, float *out, ivx_stream_t) {
  C_REQUIRE(in && out && B > 0 && C > 0 && S > 0 && Cpad >= C, "ivx_nchw_to_nhwc: bad argument");
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < (int64_t)B * S; ++i) {
    const int64_t b = i / S, s = i % S;
    for (int c = 0; c < Cpad; ++c) out[i * Cpad + c] = c < C ? in[((size_t)b * C + c) * S + s] : 0.f;
  }
  return IVX_OK;
}

extern "C" int ivx_nhwc_to_nchw(const float *in, int32_t B, int64_t S, int32_t C, float *out, ivx_stream_t) {
  C_REQUIRE(in && out && B > 0 && C > 0 && S > 0, "ivx_nhwc_to_nchw: bad argument");
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < (int64_t)B * S; ++i) {
    const int64_t b = i / S, s = i % S;
    for (int c = 0; c < C; ++c) out[((size_t)b * C + c) * S + s] = in[i * C + c];
  }
  return IVX_OK;
}

// F.interpolate(scale_factor=2, mode='trilinear', align_corners=False), the blend order of upsample_trilinear2x_kernel
extern "C" int ivx_upsample_trilinear2x_fwd(const float *in, int32_t B, int32_t D, int32_t H, int32_t W, int32_t C, float *out, ivx_stream_t) {
  C_REQUIRE(in && out && B > 0 && D > 0 && H > 0 && W > 0 && C > 0, "ivx_upsample_trilinear2x_fwd: bad argument");
  const int Do = 2 * D, Ho = 2 * H, Wo = 2 * W;
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < (int64_t)B * Do * Ho * Wo; ++i) {
    const int ow = (int)(i % Wo);
    int64_t t = i / Wo;
    const int oh = (int)(t % Ho);
    t /= Ho;
    const int od = (int)(t % Do), b = (int)(t / Do);
    float sd = 0.5f * (od + 0.5f) - 0.5f, sh = 0.5f * (oh + 0.5f) - 0.5f, sw = 0.5f * (ow + 0.5f) - 0.5f;
    sd = sd < 0.f ? 0.f : sd; sh = sh < 0.f ? 0.f : sh; sw = sw < 0.f ? 0.f : sw;
    const int d0 = (int)sd, h0 = (int)sh, w0 = (int)sw;
    const int d1 = d0 + (d0 < D - 1 ? 1 : 0), h1 = h0 + (h0 < H - 1 ? 1 : 0), w1 = w0 + (w0 < W - 1 ? 1 : 0);
    const float ld1 = sd - d0, lh1 = sh - h0, lw1 = sw - w0, ld0 = 1.f - ld1, lh0 = 1.f - lh1, lw0 = 1.f - lw1;
    auto at = [&](int dd, int hh, int ww) { return in + ((((size_t)b * D + dd) * H + hh) * W + ww) * C; };
    const float *v000 = at(d0, h0, w0), *v001 = at(d0, h0, w1), *v010 = at(d0, h1, w0), *v011 = at(d0, h1, w1);
    const float *v100 = at(d1, h0, w0), *v101 = at(d1, h0, w1), *v110 = at(d1, h1, w0), *v111 = at(d1, h1, w1);
    for (int c = 0; c < C; ++c)
      out[i * C + c] = ld0 * (lh0 * (lw0 * v000[c] + lw1 * v001[c]) + lh1 * (lw0 * v010[c] + lw1 * v011[c])) +
                       ld1 * (lh0 * (lw0 * v100[c] + lw1 * v101[c]) + lh1 * (lw0 * v110[c] + lw1 * v111[c]));
  }
  return IVX_OK;
}

// ---------------------------------------------------------------------------------------------- unprojection (csrc/backproject.hip)
// detectors/imvoxelnet.py:58-76, 132-160: voxel centre -> P @ [x, y, z, 1] -> round(u / d), round(v / d) -> in-crop and d > 0
// -> mean of the hit views; the operation order of backproject_mean_kernel (fmul, 3 fma, IEEE divide, rintf).
extern "C" int ivx_backproject_mean_fwd(const float *feat, int32_t B, int32_t V, int32_t FH, int32_t FW, int32_t C, const float *proj,
                                        const float *new_origin, const int32_t *crop_hw, const float *voxel_size, int32_t X, int32_t Y,
                                        int32_t Z, float *volume, uint8_t *valid, ivx_stream_t) {
  C_REQUIRE(feat && proj && new_origin && crop_hw && voxel_size && volume && valid, "ivx_backproject_mean_fwd: null argument");
  C_REQUIRE(B > 0 && V > 0 && FH > 0 && FW > 0 && C > 0 && X > 0 && Y > 0 && Z > 0, "ivx_backproject_mean_fwd: non-positive dims");
  const int64_t N = (int64_t)X * Y * Z;
#pragma omp parallel for schedule(static)
  for (int64_t bn = 0; bn < (int64_t)B * N; ++bn) {
    const int b = (int)(bn / N);
    const int64_t n = bn % N;
    const int k = (int)(n % Z), j = (int)((n / Z) % Y), i = (int)(n / ((int64_t)Z * Y));
    const float *no = new_origin + b * 3;
    const float px = (float)i * voxel_size[0] + no[0], py = (float)j * voxel_size[1] + no[1], pz = (float)k * voxel_size[2] + no[2];
    const int hc = std::min(crop_hw[b * 2], FH), wc = std::min(crop_hw[b * 2 + 1], FW);
    float *dst = volume + bn * C;
    for (int c = 0; c < C; ++c) dst[c] = 0.f;
    int cnt = 0;
    for (int v = 0; v < V; ++v) {
      const float *P = proj + ((size_t)b * V + v) * 12;
      float u = P[0] * px; u = fmaf(P[1], py, u); u = fmaf(P[2], pz, u); u = fmaf(P[3], 1.0f, u);
      float w = P[4] * px; w = fmaf(P[5], py, w); w = fmaf(P[6], pz, w); w = fmaf(P[7], 1.0f, w);
      float dd = P[8] * px; dd = fmaf(P[9], py, dd); dd = fmaf(P[10], pz, dd); dd = fmaf(P[11], 1.0f, dd);
      const float xr = rintf(u / dd), yr = rintf(w / dd);
      if (!((xr >= 0.f) && (yr >= 0.f) && (xr < (float)wc) && (yr < (float)hc) && (dd > 0.f))) continue;
      ++cnt;
      const float *src = feat + ((((size_t)b * V + v) * FH + (int)yr) * FW + (int)xr) * C;
      for (int c = 0; c < C; ++c) dst[c] = dst[c] + src[c];
    }
    const float dn = (float)cnt;
    for (int c = 0; c < C; ++c) dst[c] = cnt ? dst[c] / dn : 0.f;
    valid[bn] = cnt > 0 ? 1 : 0;
  }
  return IVX_OK;
}

// ---------------------------------------------------------------------------------------------- anchor tail (csrc/anchor_tail.hip)
static int next_pow2(int v) { int r = 1; while (r < v) r <<= 1; return r; }

extern "C" int32_t ivx_backproject_amax_blocks(int32_t, int32_t, int32_t, int32_t, int32_t) { return 0; }   // no partial maxima on the CPU restatement
extern "C" int ivx_backproject_mean_fwd_amax(const float *feat, int32_t B, int32_t V, int32_t FH, int32_t FW, int32_t C, const float *proj,
                                             const float *new_origin, const int32_t *crop_hw, const float *voxel_size, int32_t X, int32_t Y, int32_t Z,
                                             float *volume, uint8_t *valid, float *, ivx_stream_t st) {
  return ivx_backproject_mean_fwd(feat, B, V, FH, FW, C, proj, new_origin, crop_hw, voxel_size, X, Y, Z, volume, valid, st);
}
extern "C" int64_t ivx_anchor_head_workspace_bytes(const ivx_anchor_head_desc *d) { return d ? 256 : -1; }

// Anchor3DHead.get_bboxes_single (anchor3d_head.py:420-520): sigmoid scores, top nms_pre (ties -> lower index), box decoding,
// direction argmax, BEV boxes, rotated / axis-aligned NMS over the candidates above score_thr, first max_num, yaw fix-up.
extern "C" int ivx_anchor_head_get_bboxes(const ivx_anchor_head_desc *d, const float *head_out, const float *anchors, void *workspace,
                                          int64_t workspace_bytes, float *out_boxes, float *out_scores, int64_t *out_labels, int32_t *out_count,
                                          int64_t *cand_idx, float *cand_boxes, float *cand_scores, ivx_stream_t) {
  (void)workspace; (void)workspace_bytes;
  C_REQUIRE(d && head_out && anchors && out_boxes && out_scores && out_labels && out_count, "ivx_anchor_head_get_bboxes: null argument");
  C_REQUIRE(d->num_classes == 1, "ivx_anchor_head: only num_classes == 1 is built; got %d", d->num_classes);
  const int HW = d->H * d->W, A = d->num_anchors, n = HW * A;
  const int k = (d->nms_pre > 0 && d->nms_pre < n) ? d->nms_pre : n;
  C_REQUIRE(k <= 65536 && d->max_num > 0 && d->max_num <= 65536 && d->nms_pre > 0, "ivx_anchor_head: nms_pre / max_num out of range");
  (void)next_pow2;
  const float PI = 3.14159265358979323846f;
  for (int b = 0; b < d->B; ++b) {
    auto mem = [&](int hw) { if (!d->hw_transposed) return hw; const int y = hw / d->W, x = hw - y * d->W; return x * d->H + y; };
    std::vector<float> key(n);
    for (int i = 0; i < n; ++i) {
      const float *src = head_out + ((size_t)b * HW + mem(i / A)) * d->CH + d->cls_off + (i % A) * d->num_classes;
      key[i] = 1.0f / (1.0f + expf(-src[0]));
    }
    std::vector<int> order(n);
    for (int i = 0; i < n; ++i) order[i] = i;
    std::partial_sort(order.begin(), order.begin() + k, order.end(), [&](int a, int c) { return key[a] > key[c] || (key[a] == key[c] && a < c); });
    std::vector<float> boxes((size_t)k * 7), bev((size_t)k * 5), sc(k);
    std::vector<int> dir(k);
    int n1 = 0;
    for (int j = 0; j < k; ++j) {
      const int idx = order[j], hw = idx / A, a = idx % A;
      const float *row = head_out + ((size_t)b * HW + mem(hw)) * d->CH, *an = anchors + (size_t)idx * 7, *dl = row + d->reg_off + a * 7;
      const float xa = an[0], ya = an[1], wa = an[3], la = an[4], ha = an[5], ra = an[6];
      float za = an[2];
      za = za + ha / 2;
      const float diag = sqrtf(la * la + wa * wa);
      const float xg = dl[0] * diag + xa, yg = dl[1] * diag + ya;
      float zg = dl[2] * ha + za;
      const float lg = expf(dl[4]) * la, wg = expf(dl[3]) * wa, hg = expf(dl[5]) * ha, rg = dl[6] + ra;
      zg = zg - hg / 2;
      float *ob = &boxes[(size_t)j * 7];
      ob[0] = xg; ob[1] = yg; ob[2] = zg; ob[3] = wg; ob[4] = lg; ob[5] = hg; ob[6] = rg;
      const float hwid = wg / 2, hlen = lg / 2;
      float *bv = &bev[(size_t)j * 5];
      bv[0] = xg - hwid; bv[1] = yg - hlen; bv[2] = xg + hwid; bv[3] = yg + hlen; bv[4] = rg;
      dir[j] = row[d->dir_off + a * 2 + 1] > row[d->dir_off + a * 2] ? 1 : 0;
      sc[j] = key[idx];
      if (sc[j] > d->score_thr) ++n1;          // a prefix: the candidates are sorted
    }
    std::vector<int64_t> keep(n1 > 0 ? n1 : 1);
    int nk = 0;
    if (n1 > 0) nk = d->use_rotate_nms ? ivxo_nms_rotated_sorted(bev.data(), n1, d->nms_thr, keep.data()) : ivxo_nms_normal_sorted(bev.data(), n1, d->nms_thr, keep.data());
    if (nk > d->max_num) nk = d->max_num;
    for (int j = 0; j < d->max_num; ++j) {
      float *ob = out_boxes + ((size_t)b * d->max_num + j) * 7;
      if (j < nk) {
        const int i = (int)keep[j];
        for (int q = 0; q < 6; ++q) ob[q] = boxes[(size_t)i * 7 + q];
        const float val = boxes[(size_t)i * 7 + 6] - d->dir_offset;
        const float t = floorf(val / PI + d->dir_limit_offset);
        const float dir_rot = val - t * PI;
        ob[6] = (dir_rot + d->dir_offset) + PI * (float)dir[i];
        out_scores[(size_t)b * d->max_num + j] = sc[i];
      } else {
        for (int q = 0; q < 7; ++q) ob[q] = 0.f;
        out_scores[(size_t)b * d->max_num + j] = 0.f;
      }
      out_labels[(size_t)b * d->max_num + j] = 0;
    }
    out_count[b] = nk;
    for (int j = 0; j < d->nms_pre; ++j) {
      const bool in = j < k;
      if (cand_idx) cand_idx[(size_t)b * d->nms_pre + j] = in ? order[j] : -1;
      if (cand_boxes) for (int q = 0; q < 7; ++q) cand_boxes[((size_t)b * d->nms_pre + j) * 7 + q] = in ? boxes[(size_t)j * 7 + q] : 0.f;
      if (cand_scores) cand_scores[(size_t)b * d->nms_pre + j] = in ? sc[j] : 0.f;
    }
  }
  return IVX_OK;
}

// ---------------------------------------------------------------------------------------------- NMS entry points (via oracle/ivx_oracle.c)
extern "C" int64_t ivx_nms_workspace_bytes(int32_t n) { return n < 0 || n > 65536 ? -1 : 256; }

extern "C" int ivx_nms_bev(const float *boxes_sorted, int32_t n, float thresh, int32_t rotated, void *, int64_t, int64_t *keep, int32_t *num_out,
                           ivx_stream_t) {
  C_REQUIRE(n >= 0 && n <= 65536 && keep && num_out && (n == 0 || boxes_sorted), "ivx_nms_bev: bad argument");
  *num_out = n == 0 ? 0 : (rotated ? ivxo_nms_rotated_sorted(boxes_sorted, n, thresh, keep) : ivxo_nms_normal_sorted(boxes_sorted, n, thresh, keep));
  return IVX_OK;
}

extern "C" int ivx_boxes_overlap_bev(const float *a, int32_t na, const float *b, int32_t nb, int32_t iou, float *out, ivx_stream_t) {
  C_REQUIRE(na >= 0 && nb >= 0, "ivx_boxes_overlap_bev: bad sizes");
  if (na == 0 || nb == 0) return IVX_OK;
  C_REQUIRE(a && b && out, "ivx_boxes_overlap_bev: null argument");
  if (iou) ivxo_boxes_iou_bev(a, na, b, nb, out); else ivxo_boxes_overlap_bev(a, na, b, nb, out);
  return IVX_OK;
}

extern "C" int ivx_aligned_3d_nms(const float *boxes, const float *scores, const int64_t *classes, int32_t n, float thresh, int64_t *pick,
                                  int32_t *num_out, ivx_stream_t) {
  C_REQUIRE(n >= 0 && n <= 4096 && pick && num_out && (n == 0 || (boxes && scores && classes)), "ivx_aligned_3d_nms: bad argument");
  std::vector<int64_t> order(n);
  for (int i = 0; i < n; ++i) order[i] = i;
  // ascending score for the oracle's scan from the back; among equal scores the LOWER index is processed first (the device's key)
  std::sort(order.begin(), order.end(), [&](int64_t a, int64_t c) { return scores[a] < scores[c] || (scores[a] == scores[c] && a > c); });
  *num_out = n == 0 ? 0 : ivxo_aligned_3d_nms(boxes, scores, classes, order.data(), n, thresh, pick);
  return IVX_OK;
}

// ---------------------------------------------------------------------------------------------- general-N and fused NMS forms
extern "C" int64_t ivx_aligned_3d_nms_workspace_bytes(int32_t n) { return n < 0 || n > 65536 ? -1 : 256; }

extern "C" int ivx_aligned_3d_nms_ws(const float *boxes, const float *scores, const int64_t *classes, int32_t n, float thresh, void *, int64_t,
                                     int64_t *pick, int32_t *num_out, ivx_stream_t) {
  C_REQUIRE(n >= 0 && n <= 65536 && pick && num_out && (n == 0 || (boxes && scores && classes)), "ivx_aligned_3d_nms_ws: bad argument");
  std::vector<int64_t> order(n);
  for (int i = 0; i < n; ++i) order[i] = i;
  std::sort(order.begin(), order.end(), [&](int64_t a, int64_t c) { return scores[a] < scores[c] || (scores[a] == scores[c] && a > c); });
  *num_out = n == 0 ? 0 : ivxo_aligned_3d_nms(boxes, scores, classes, order.data(), n, thresh, pick);
  return IVX_OK;
}

extern "C" int64_t ivx_multiclass_nms_workspace_bytes(int32_t n, int32_t num_classes) {
  return (n < 0 || n > 65536 || num_classes < 1 || num_classes > 64) ? -1 : 256;
}

// box3d_multiclass_nms (core/post_processing/box3d_nms.py:8-88) as csrc/anchor_tail.hip ivx_multiclass_nms_bev orders it: per class
// the candidates above score_thr by descending score (ties: lower index), greedy NMS, class-major concatenation, or -- beyond max_num --
// the max_num best by score (ties: lower class, then earlier position).
extern "C" int ivx_multiclass_nms_bev(const float *boxes, const float *scores, int32_t n, int32_t score_stride, int32_t num_classes, float score_thr,
                                      float nms_thr, int32_t rotated, int32_t max_num, void *, int64_t, int64_t *out_idx, int64_t *out_label,
                                      int32_t *out_count, ivx_stream_t) {
  C_REQUIRE(n >= 0 && n <= 65536 && num_classes >= 1 && num_classes <= 64 && out_idx && out_label && out_count && max_num > 0 &&
                score_stride >= num_classes,
            "ivx_multiclass_nms_bev: bad argument");
  struct Det { float s; int c, pos, idx; };
  std::vector<Det> all;
  for (int c = 0; c < num_classes; ++c) {
    std::vector<int> cand;
    for (int i = 0; i < n; ++i)
      if (scores[(size_t)i * score_stride + c] > score_thr) cand.push_back(i);
    std::sort(cand.begin(), cand.end(), [&](int a, int b) {
      const float sa = scores[(size_t)a * score_stride + c], sb = scores[(size_t)b * score_stride + c];
      return sa > sb || (sa == sb && a < b);
    });
    const int m = (int)cand.size();
    if (!m) continue;
    std::vector<float> sorted((size_t)m * 5);
    for (int j = 0; j < m; ++j) memcpy(&sorted[(size_t)j * 5], boxes + (size_t)cand[j] * 5, 5 * sizeof(float));
    std::vector<int64_t> keep(m);
    const int nk = rotated ? ivxo_nms_rotated_sorted(sorted.data(), m, nms_thr, keep.data()) : ivxo_nms_normal_sorted(sorted.data(), m, nms_thr, keep.data());
    for (int j = 0; j < nk; ++j) all.push_back({scores[(size_t)cand[keep[j]] * score_stride + c], c, j, cand[keep[j]]});
  }
  if ((int)all.size() > max_num) {
    std::stable_sort(all.begin(), all.end(), [](const Det &a, const Det &b) { return a.s > b.s; });   // class-major input: ties keep lower class first
    all.resize(max_num);
  }
  for (size_t j = 0; j < all.size(); ++j) { out_idx[j] = all[j].idx; out_label[j] = all[j].c; }
  *out_count = (int32_t)all.size();
  return IVX_OK;
}

// ---------------------------------------------------------------------------------------------- indoor heads (csrc/anchor_tail.hip)
extern "C" int64_t ivx_fcos_head_workspace_bytes(int32_t B, int32_t n, int32_t nms_pre) {
  if (B <= 0 || n <= 0) return -1;
  const int k = (nms_pre > 0 && nms_pre < n) ? nms_pre : n;
  return k > 65536 ? -1 : 256;
}

static inline float sigmoid_ref(float x) { return 1.0f / (1.0f + expf(-x)); }

// ImVoxelHeadV2._get_bboxes_single per level (imvoxel_head_v2.py:245-277): as fcos_scores_kernel / topk / fcos_decode_kernel.
extern "C" int ivx_fcos_head_level_candidates(const float *head_out, const uint8_t *valid0, const float *level_vs, const float *level_new_origin,
                                              float scale, int32_t B, int32_t nx, int32_t ny, int32_t nz, int32_t CH, int32_t n_classes, int32_t n_reg,
                                              int32_t level, int32_t X, int32_t Y, int32_t Z, int32_t nms_pre, void *, int64_t, float *cand_boxes,
                                              float *cand_scores, int32_t *cand_count, ivx_stream_t) {
  C_REQUIRE(head_out && valid0 && level_vs && level_new_origin && cand_boxes && cand_scores && cand_count, "ivx_fcos_head_level_candidates: null argument");
  C_REQUIRE(B > 0 && nx > 0 && ny > 0 && nz > 0 && n_classes > 0 && (n_reg == 6 || n_reg == 7) && CH >= 1 + n_reg + n_classes &&
                level >= 0 && level < 8 && (nx << level) == X && (ny << level) == Y && (nz << level) == Z,
            "ivx_fcos_head_level_candidates: bad dims");
  const int n = nx * ny * nz, k = (nms_pre > 0 && nms_pre < n) ? nms_pre : n;
  C_REQUIRE(k <= 65536, "ivx_fcos_head_level_candidates: at most 65536 candidates per level (got %d)", k);
  auto validf = [&](int b, int i) -> float {
    const int iz = i % nz, t = i / nz, iy = t % ny, ix = t / ny;
    const uint8_t *v = valid0 + (size_t)b * X * Y * Z;
    if (level == 0) return v[((size_t)ix * Y + iy) * Z + iz] ? 1.0f : 0.0f;
    const int h = (1 << (level - 1)) - 1, x0 = (ix << level) + h, y0 = (iy << level) + h, z0 = (iz << level) + h;
    int cnt = 0;
    for (int a = 0; a < 2; ++a)
      for (int e = 0; e < 2; ++e)
        for (int f = 0; f < 2; ++f) cnt += v[((size_t)(x0 + a) * Y + (y0 + e)) * Z + (z0 + f)] ? 1 : 0;
    return cnt >= 5 ? 1.0f : 0.0f;
  };
  for (int b = 0; b < B; ++b) {
    std::vector<float> key(n);
#pragma omp parallel for
    for (int i = 0; i < n; ++i) {
      const float *row = head_out + ((size_t)b * n + i) * CH;
      const float ctr = sigmoid_ref(row[0]), vf = validf(b, i);
      float m = -1.0f;
      for (int c = 0; c < n_classes; ++c) {
        const float s = (sigmoid_ref(row[1 + n_reg + c]) * ctr) * vf;
        m = s > m ? s : m;
      }
      key[i] = m;
    }
    std::vector<int> order(n);
    for (int i = 0; i < n; ++i) order[i] = i;
    std::partial_sort(order.begin(), order.begin() + k, order.end(), [&](int a, int c) { return key[a] > key[c] || (key[a] == key[c] && a < c); });
    const float *vs = level_vs + b * 3, *no = level_new_origin + b * 3;
    for (int j = 0; j < k; ++j) {
      const int i = order[j], iz = i % nz, t = i / nz, iy = t % ny, ix = t / ny;
      const float px = (float)ix * vs[0] + no[0], py = (float)iy * vs[1] + no[1], pz = (float)iz * vs[2] + no[2];
      const float *row = head_out + ((size_t)b * n + i) * CH;
      float d[6];
      for (int q = 0; q < 6; ++q) d[q] = expf(row[1 + q] * scale);
      float *ob = cand_boxes + ((size_t)b * k + j) * n_reg, *os = cand_scores + ((size_t)b * k + j) * n_classes;
      if (n_reg == 6) {
        ob[0] = px - d[0]; ob[1] = py - d[2]; ob[2] = pz - d[4];
        ob[3] = px + d[1]; ob[4] = py + d[3]; ob[5] = pz + d[5];
      } else {
        const float alpha = row[1 + 6];
        const float sx = (d[1] - d[0]) / 2, sy = (d[3] - d[2]) / 2, sz = (d[5] - d[4]) / 2;
        const float c = cosf(alpha), s = sinf(alpha);
        ob[0] = px + (sx * c + sy * s);
        ob[1] = py + (-sx * s + sy * c);
        ob[2] = pz + sz;
        ob[3] = d[0] + d[1]; ob[4] = d[2] + d[3]; ob[5] = d[4] + d[5];
        ob[6] = alpha;
      }
      const float ctr = sigmoid_ref(row[0]), vf = validf(b, i);
      for (int c = 0; c < n_classes; ++c) os[c] = (sigmoid_ref(row[1 + n_reg + c]) * ctr) * vf;
    }
    cand_count[b] = k;
  }
  return IVX_OK;
}

extern "C" int64_t ivx_indoor_tail_workspace_bytes(const ivx_indoor_tail_desc *d) { return d ? 256 : -1; }

// Cross-level tail (csrc/anchor_tail.hip ivx_indoor_tail_get_bboxes; imvoxel_head_v2.py:258-277, :528-545, :397-417).
extern "C" int ivx_indoor_tail_get_bboxes(const ivx_indoor_tail_desc *d, const float *const *cand_boxes, const float *const *cand_scores, void *, int64_t,
                                          float *out_boxes, float *out_scores, int64_t *out_labels, int32_t *out_count, ivx_stream_t st) {
  C_REQUIRE(d && cand_boxes && cand_scores && out_boxes && out_scores && out_labels && out_count, "ivx_indoor_tail_get_bboxes: null argument");
  C_REQUIRE(d->B > 0 && d->n_levels >= 1 && d->n_levels <= 4 && d->n_classes >= 1 && (d->n_reg == 6 || d->n_reg == 7) && d->max_num > 0,
            "ivx_indoor_tail: bad dims");
  int K = 0;
  for (int l = 0; l < d->n_levels; ++l) K += d->k[l];
  const int R = d->n_reg, nc = d->n_classes, M = d->max_num;
  for (int b = 0; b < d->B; ++b) {
    std::vector<float> boxes((size_t)K * R), sc((size_t)K * nc);
    int o = 0;
    for (int l = 0; l < d->n_levels; ++l) {
      memcpy(&boxes[(size_t)o * R], cand_boxes[l] + (size_t)b * d->k[l] * R, (size_t)d->k[l] * R * sizeof(float));
      memcpy(&sc[(size_t)o * nc], cand_scores[l] + (size_t)b * d->k[l] * nc, (size_t)d->k[l] * nc * sizeof(float));
      o += d->k[l];
    }
    float *ob = out_boxes + (size_t)b * M * 7, *os = out_scores + (size_t)b * M;
    int64_t *ol = out_labels + (size_t)b * M;
    memset(ob, 0, (size_t)M * 7 * sizeof(float));
    memset(os, 0, (size_t)M * sizeof(float));
    memset(ol, 0, (size_t)M * sizeof(int64_t));
    int cnt = 0;
    if (R == 6) {
      std::vector<float> best(K), masked(K);
      std::vector<int64_t> lab(K), pick(K);
      for (int j = 0; j < K; ++j) {
        float m = sc[(size_t)j * nc];
        int c0 = 0;
        for (int c = 1; c < nc; ++c)
          if (sc[(size_t)j * nc + c] > m) { m = sc[(size_t)j * nc + c]; c0 = c; }
        best[j] = m; lab[j] = c0;
        masked[j] = m > d->score_thr ? m : -INFINITY;
      }
      int32_t np = 0;
      int rc = ivx_aligned_3d_nms_ws(boxes.data(), masked.data(), lab.data(), K, d->nms_thr, nullptr, 0, pick.data(), &np, st);
      if (rc != IVX_OK) return rc;
      for (int j = 0; j < np && cnt < M; ++j) {
        const int i = (int)pick[j];
        if (!(best[i] > d->score_thr)) break;          // the real picks are a prefix
        const float *c = &boxes[(size_t)i * 6];
        const float dz = c[5] - c[2];
        float *r = ob + (size_t)cnt * 7;
        r[0] = (c[0] + c[3]) / 2.f; r[1] = (c[1] + c[4]) / 2.f; r[2] = (c[2] + c[5]) / 2.f + dz * -0.5f;
        r[3] = c[3] - c[0]; r[4] = c[4] - c[1]; r[5] = dz; r[6] = 0.f;
        os[cnt] = best[i]; ol[cnt] = lab[i];
        ++cnt;
      }
    } else {
      std::vector<float> bev((size_t)K * 5);
      for (int j = 0; j < K; ++j) {
        const float *s = &boxes[(size_t)j * 7];
        float *v = &bev[(size_t)j * 5];
        v[0] = s[0] - s[3] / 2.f; v[1] = s[1] - s[4] / 2.f; v[2] = s[0] + s[3] / 2.f; v[3] = s[1] + s[4] / 2.f; v[4] = s[6];
      }
      std::vector<int64_t> idx(M), lab(M);
      int32_t n = 0;
      int rc = ivx_multiclass_nms_bev(bev.data(), sc.data(), K, nc, nc, d->score_thr, d->nms_thr, d->use_rotate_nms, M, nullptr, 0, idx.data(), lab.data(), &n, st);
      if (rc != IVX_OK) return rc;
      for (int j = 0; j < n; ++j) {
        const float *s = &boxes[(size_t)idx[j] * 7];
        float *r = ob + (size_t)j * 7;
        r[0] = s[0]; r[1] = s[1]; r[2] = s[2] + s[5] * -0.5f; r[3] = s[3]; r[4] = s[4]; r[5] = s[5]; r[6] = s[6];
        os[j] = sc[(size_t)idx[j] * nc + lab[j]]; ol[j] = lab[j];
      }
      cnt = n;
    }
    out_count[b] = cnt;
  }
  return IVX_OK;
}

// ---------------------------------------------------------------------------------------------- DCNv2 columns, global pool
// csrc/dcn.hip dcn_im2col_kernel (mmcv ModulatedDeformConv2d, deform_groups 1): col[b,ho,wo,k,c] = sigmoid(m_k) * bilinear(x[..., c])
extern "C" int ivx_dcn_im2col_fwd(const float *x, const float *offset_mask, int32_t B, int32_t H, int32_t W, int32_t C, int32_t kh, int32_t kw,
                                  int32_t stride, int32_t pad, int32_t dil, int32_t om_channels, float *col, ivx_stream_t) {
  C_REQUIRE(x && offset_mask && col && B > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0 && kh > 0 && kw > 0 && stride > 0 && om_channels >= 3 * kh * kw,
            "ivx_dcn_im2col_fwd: bad argument");
  const int K = kh * kw, Ho = (H + 2 * pad - dil * (kh - 1) - 1) / stride + 1, Wo = (W + 2 * pad - dil * (kw - 1) - 1) / stride + 1;
#pragma omp parallel for collapse(2)
  for (int b = 0; b < B; ++b)
    for (int ho = 0; ho < Ho; ++ho)
      for (int wo = 0; wo < Wo; ++wo) {
        const float *om = offset_mask + (((size_t)b * Ho + ho) * Wo + wo) * om_channels;
        for (int k = 0; k < K; ++k) {
          const int i = k / kw, j = k % kw;
          const float hh = (float)(ho * stride - pad + i * dil) + om[2 * k], ww = (float)(wo * stride - pad + j * dil) + om[2 * k + 1];
          const float m = 1.0f / (1.0f + expf(-om[2 * K + k]));
          float *dst = col + ((((size_t)b * Ho + ho) * Wo + wo) * K + k) * C;
          if (!(hh > -1.0f && ww > -1.0f && hh < (float)H && ww < (float)W)) {
            for (int c = 0; c < C; ++c) dst[c] = 0.f;
            continue;
          }
          const int h0 = (int)floorf(hh), w0 = (int)floorf(ww), h1 = h0 + 1, w1 = w0 + 1;
          const float lh = hh - (float)h0, lw = ww - (float)w0, uh = 1.0f - lh, uw = 1.0f - lw;
          const float w00 = uh * uw, w01 = uh * lw, w10 = lh * uw, w11 = lh * lw;
          const float *xb = x + (size_t)b * H * W * C;
          for (int c = 0; c < C; ++c) {
            const float v00 = (h0 >= 0 && w0 >= 0) ? xb[((size_t)h0 * W + w0) * C + c] : 0.f;
            const float v01 = (h0 >= 0 && w1 <= W - 1) ? xb[((size_t)h0 * W + w1) * C + c] : 0.f;
            const float v10 = (h1 <= H - 1 && w0 >= 0) ? xb[((size_t)h1 * W + w0) * C + c] : 0.f;
            const float v11 = (h1 <= H - 1 && w1 <= W - 1) ? xb[((size_t)h1 * W + w1) * C + c] : 0.f;
            dst[c] = (((w00 * v00 + w01 * v01) + w10 * v10) + w11 * v11) * m;
          }
        }
      }
  return IVX_OK;
}

// csrc/dcn.hip dcn_im2col_pair_kernel: the columns inside the pair chain -- decode the map, the fp32 columns above, encode with the MAP's scale
extern "C" int ivx_dcn_im2col_fwd_pair(const void *x, const float *x_scale, const float *offset_mask, int32_t B, int32_t H, int32_t W, int32_t C,
                                       int32_t kh, int32_t kw, int32_t stride, int32_t pad, int32_t dil, int32_t om_channels, void *col,
                                       float *col_scale, uint32_t *col_amax, ivx_stream_t st) {
  C_REQUIRE(x && x_scale && col && col_scale && C % 16 == 0, "ivx_dcn_im2col_fwd_pair: bad argument");
  const int K = kh * kw, Ho = (H + 2 * pad - dil * (kh - 1) - 1) / stride + 1, Wo = (W + 2 * pad - dil * (kw - 1) - 1) / stride + 1;
  std::vector<float> xf((size_t)B * H * W * C), cf((size_t)B * Ho * Wo * K * C);
  c_pair_decode((const uint16_t *)x, (int64_t)B * H * W, C, 1.0f / *x_scale, xf.data());
  int rc = ivx_dcn_im2col_fwd(xf.data(), offset_mask, B, H, W, C, kh, kw, stride, pad, dil, om_channels, cf.data(), st);
  if (rc != IVX_OK) return rc;
  *col_scale = *x_scale;
  float omax = 0.f;
  for (float v : cf) omax = fabsf(v) > omax ? fabsf(v) : omax;
  if (col_amax) c_amax_commit(col_amax, omax);
  c_pair_encode(cf.data(), (int64_t)B * Ho * Wo, K * C, *x_scale, (uint16_t *)col);
  return IVX_OK;
}

// csrc/pool_layout.hip global_avgpool: in [B,S,C] -> out [B,C] (LayoutHead, layout_head.py:42)
extern "C" int ivx_global_avgpool_fwd(const float *in, int32_t B, int64_t S, int32_t C, float *out, ivx_stream_t) {
  C_REQUIRE(in && out && B > 0 && S > 0 && C > 0, "ivx_global_avgpool_fwd: bad argument");
  for (int b = 0; b < B; ++b)
    for (int c = 0; c < C; ++c) {
      double acc = 0.0;
      for (int64_t s = 0; s < S; ++s) acc += in[((size_t)b * S + s) * C + c];
      out[(size_t)b * C + c] = (float)(acc / (double)S);
    }
  return IVX_OK;
}
