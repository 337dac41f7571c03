/*
 * imvoxel.h -- C-ABI of libimvoxel_hip.so, the MI355X (gfx950) implementation of the
 * ImVoxelNet forward hot path.
 *
 * Boundary convention.  The reference reaches native code on this
 * path in exactly two ways:
 *   (1) torch.nn ops (Conv2d/Conv3d/BatchNorm/ReLU/max_pool/interpolate/bmm/index/topk)
 *       that dispatch into cuDNN/cuBLAS, and
 *   (2) its own pybind module iou3d_cuda:  int nms_gpu(Tensor boxes, Tensor keep,
 *       float thr, int device_id)   (mmdet3d/ops/iou3d/src/iou3d.cpp:95-147,203-208).
 * Every entry point below replaces one of those call sites and keeps convention (2):
 * plain pointers + sizes, caller-owned buffers, int status return (0 = ok, <0 = error;
 * the library never exit()s, unlike gpuAssert at iou3d.cpp:27-36), no torch types, no
 * hidden allocation, asynchronous on the HIP stream handed in (hipStream_t as void*;
 * NULL = the default stream).  All pointers are DEVICE pointers unless marked "host".
 *
 * Layouts: activations are channels-last fp32 -- NHWC for 2-D, NDHWC for 3-D (a 2-D
 * tensor is the D == 1 case).  For the feature volume the reference's (X, Y, Z) axes are
 * (D, H, W), so a voxel's C channels are contiguous and z is the fastest spatial axis,
 * matching the reference's flat voxel index n = (i*Y + j)*Z + k
 * (mmdet3d/models/detectors/imvoxelnet.py:148).
 */
#ifndef IMVOXEL_H_
#define IMVOXEL_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define IVX_OK 0
#define IVX_ERR_INVALID_ARG (-1)
#define IVX_ERR_HIP (-2)
#define IVX_ERR_UNSUPPORTED (-3)
#define IVX_ERR_WORKSPACE (-4)

typedef void *ivx_stream_t; /* hipStream_t */

/* Library version (major*10000 + minor*100 + patch; 400 = 0.4.0, the struct layouts of this header) and the message of
 * the last failing call on this thread (never NULL). */
int ivx_version(void);
const char *ivx_last_error(void);

/* ---------------------------------------------------------------------------------------
 * Channels-last convolution with fused epilogue -- replaces nn.Conv2d / nn.Conv3d followed
 * by eval-mode BatchNorm, residual add and ReLU:
 *   ResNet-50 / FPN call sites  mmdet3d/models/detectors/imvoxelnet.py:48,50
 *   3-D necks                   mmdet3d/models/necks/imvoxelnet.py:46-60,99-113,181-230
 *   head convs                  mmdet3d/models/dense_heads/anchor3d_head.py:122-153
 *
 *   out[b,do,ho,wo,co] = act( (sum_{kd,kh,kw,ci} in[b, do*sd-pd+kd, ho*sh-ph+kh, wo*sw-pw+kw, ci]
 *                              * wgt[co,kd,kh,kw,ci]) * scale[co] + shift[co] + res[...] )
 *
 * in    [B,D,H,W,Cin]   Cin % 4 == 0 (pad the image to 4 channels with ivx_nchw_to_nhwc)
 * wgt   wgt_layout 0: [Cout,KD,KH,KW,Cin]   (k = tap*Cin + ci)
 *       wgt_layout 1: [Cout,Cin/32,KD,KH,KW,32]  (k = (ci/32)*taps*32 + tap*32 + ci%32; needs Cin % 32 == 0).
 *       Layout 1 walks all taps of one 32-channel chunk back to back, so the 27 shifted re-reads of an input row
 *       are one K-slab apart and hit L1/L2 instead of going back to the fabric.
 * scale,shift [Cout] or NULL (1 / 0).  Conv bias and BN fold into them on the host.
 * Epilogue order: v = acc*scale + shift; [v += res]; [ReLU]; [v += res if res_after_act]; v *= post_scale.
 * res   NULL, or res_mode 1: same shape as out; res_mode 2: [B,1,res_h,res_w,Cout] read with
 *       nearest-neighbour up-sampling to (Ho,Wo) (FPN top-down path, F.interpolate 'nearest').
 * Arithmetic: fp32 inputs, fp32 MFMA (v_mfma_f32_32x32x2_f32), fp32 accumulate -- the reference's precision and
 * the default.  Optional reduced-precision storage (the reference has no such mode; BASELINE config 5):
 *   in_dtype  IVX_BF16: in and wgt are bf16 (v_mfma_f32_32x32x16_bf16, fp32 accumulate); needs Cin % 8 == 0,
 *             wgt_layout 1 then uses 64-channel chunks [Cout,Cin/64,KD,KH,KW,64]; scale/shift stay fp32.
 *   out_dtype IVX_BF16: out and res are bf16 (epilogue in fp32, one round-to-nearest-even at the store).  */
#define IVX_F32 0
#define IVX_BF16 1
#define IVX_FP8 2      /* OCP e4m3 bytes (gfx950) with a per-tensor scale kept by the caller: see res_scale below */
/* in_dtype only.  Split-operand fp32: every fp32 value x of `in` and `wgt` is stored as the bf16 pair hi = bf16(x),
 * lo = bf16(x - hi) (round to nearest even; hi + lo carries 16 significant bits of x), channels in groups of 16 as
 * [hi c0..c15 | lo c0..c15] (4 bytes per value, like fp32).  The contraction issues hi*hi + hi*lo + lo*hi on the bf16 matrix
 * cores (v_mfma_f32_32x32x16_bf16, 16x the fp32 MFMA rate) with fp32 accumulation: products are exact in fp32, the dropped lo*lo
 * term and the pair rounding are each <= 2^-17 of the product -- 64x finer than the TF32 operands the reference's cuDNN
 * convolutions use by default on its published GPUs, and below the rounding of the F(6x6,3x3) form of the fp32 path (measured in
 * DESIGN.md 4.1e).  Cin (real channels) % 16 == 0; wgt_layout 1 needs Cin % 32 == 0 (64 stored elements per 128-byte chunk);
 * out_mode 0; out / res / scale / shift as for IVX_F32 input.  ivx_bf16_pair_split makes the activation operand, the caller packs
 * the filters the same way once (imvoxelnet_amd/conv.py: pack_pair_weights). */
#define IVX_BF16_PAIR 3
/* The same with fp16 halves: hi = fp16(s*x), lo = fp16(s*x - hi) for a power-of-two tensor scale s kept by the caller (folded into
 * scale[] like the fp8 scales).  hi + lo carries 22 significant bits (pair rounding <= 2^-23, dropped lo*lo <= 2^-22 of a product:
 * fp32-level), at the price of fp16's range: |s*x| must stay below 65504 (ivx_f16_pair_split saturates) and values below
 * 2^-14 * 2^11 = 0.125 lose relative (not absolute: fp16 subnormals, spacing 2^-24) precision in lo.  Used by the Winograd-domain
 * GEMMs (ivx_conv_desc.wino_operands), where operand precision is amplified by the output transform and bf16 pairs are not enough. */
#define IVX_F16_PAIR 4
typedef struct ivx_conv_desc {
  int32_t B, D, H, W, Cin;
  int32_t Cout, KD, KH, KW;
  int32_t sd, sh, sw;
  int32_t pd, ph, pw;
  int32_t relu;
  int32_t res_mode, res_h, res_w;
  int32_t wgt_layout;
  int32_t out_mode;      /* 0: normal.  1: nn.ConvTranspose3d(kernel 2, stride 2) as a 1x1x1 GEMM with
                            Cout = 8*C columns n = ((a*2+e)*2+f)*C + co, scattered to out[b, 2d+a, 2h+e, 2w+f, co]
                            (out is [B,2D,2H,2W,C]; scale/shift have C entries; res, if any, has the out shape) */
  int32_t res_after_act; /* 1: the residual is added after the ReLU (skip adds of the U-shaped necks) */
  float post_scale;      /* final multiplier, 0 or 1 = none (Atlas neck: (x + y) / 2) */
  int32_t in_dtype;      /* IVX_F32 (default), IVX_BF16, IVX_FP8 or IVX_BF16_PAIR: element type of in and wgt */
  int32_t out_dtype;     /* IVX_F32 (default), IVX_BF16 or IVX_FP8: element type of out and res */
  float res_scale;       /* multiplier of the residual before it is added, 0 or 1 = none.  IVX_FP8 tensors are e4m3 bytes with a
                            per-tensor scale kept by the caller (x = byte_value * s_x): the caller folds s_in * s_wgt[co] / s_out
                            into scale[], 1 / s_out into shift[] and passes res_scale = s_res / s_out; the store saturates at
                            +-448 and rounds to nearest even.  fp8 input needs Cin % 16 == 0 (wgt_layout 1: Cin % 128 == 0). */
  int32_t wino_operands; /* ivx_conv_winograd_* only (in / out stay fp32): operand type of the transformed-domain GEMMs.  IVX_F32 (0,
                            default): fp32 MFMA, exact fp32 arithmetic.  IVX_F16_PAIR: the input / filter transforms write V and U as
                            fp16 (hi, lo) pairs (power-of-two scales taken from max |in| / max |U| on the device, undone exactly by the
                            output transform; the workspace's last 256 bytes carry them between the stages) and the GEMMs issue three
                            fp16 MFMA products per pair -- same bytes, ~3.3x the GEMM rate, error at the level of the fp32 form's own
                            rounding.  Needs tile 4 or 6 and Cin % 16 == 0 (wgt_layout 1: Cin % 32 == 0).  Filters
                            made by ivx_conv_winograd_weights carry the operand type they were made with
                            (ivx_conv_winograd_weight_elems counts one more plane for the pair form: it holds the filter scale). */
} ivx_conv_desc;

int ivx_conv_out_dims(const ivx_conv_desc *d, int32_t *Do, int32_t *Ho, int32_t *Wo);
int ivx_conv_fwd(const ivx_conv_desc *d, const void *in, const void *wgt, const float *scale,
                 const float *shift, const void *res, void *out, ivx_stream_t stream);
/* Same as ivx_conv_fwd with a caller-owned workspace of ivx_conv_workspace_bytes(d) bytes (0 for most layers): lets
 * the library split K across workgroups for layers whose output is too small to fill the chip (ResNet stage 4, FPN
 * laterals on C5, coarse levels of the indoor necks); slices are summed in a fixed order (deterministic).        */
int64_t ivx_conv_workspace_bytes(const ivx_conv_desc *d);
int ivx_conv_fwd_ws(const ivx_conv_desc *d, const void *in, const void *wgt, const float *scale, const float *shift,
                    const void *res, void *out, void *workspace, int64_t workspace_bytes, ivx_stream_t stream);

/* fp32 [n] -> bf16 pairs [2n] in the IVX_BF16_PAIR order (n % 16 == 0; channels-last tensors with C % 16 == 0 keep their shape
 * with 2C stored elements per voxel).  A streaming kernel: 4 bytes read + 4 written per value.
 * ivx_conv_pair_supported: 1 when ivx_conv_fwd_ws can run the fp32 convolution `d` (in_dtype / out_dtype IVX_F32) in the pair form,
 * i.e. with in_dtype = IVX_BF16_PAIR on operands converted as above (channel multiples, 31-bit operand offsets, out_mode 0). */
int ivx_bf16_pair_split(const float *in, int64_t n, void *out, ivx_stream_t stream);
int ivx_f16_pair_split(const float *in, int64_t n, float scale, void *out, ivx_stream_t stream);   /* IVX_F16_PAIR of scale * in, saturating */
int ivx_conv_pair_supported(const ivx_conv_desc *d);

/* ---------------------------------------------------------------------------------------
 * Chained fp16-pair activations (0.4.0): the 2-D trunk on the 16-bit matrix cores without split passes.
 * Replaces the same call sites as ivx_conv_fwd (ResNet-50 / FPN: mmdet3d/models/detectors/imvoxelnet.py:48,50) -- same arithmetic
 * contract, fp32 values, fp32 accumulation -- with the ACTIVATIONS between the layers stored as IVX_F16_PAIR tensors [B,D,H,W,2C]
 * (hi = fp16(s x), lo = fp16(s x - hi), 16-channel groups [hi x16 | lo x16], 4 bytes per value like fp32; 22 significant bits) so
 * that every convolution issues hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_f16 (16x the fp32 MFMA rate, three products per
 * multiply-add) and the epilogue of the PRODUCING layer writes the operand the consuming layer reads: no conversion pass.
 * The power-of-two scale s of a tensor lives in device memory and is chosen on the device, without a pass over the tensor, from a
 * bound of the output:  |out| <= max|in| * wbound + sbound + max|res|,  wbound = max_co |scale[co]| * sum_k |w[co][k]|,
 * sbound = max_co |shift[co]|, with the MEASURED maxima of the operands: every epilogue accumulates max |out| (true values) into
 * IVX_AMAX_SLOTS words of the tensor (atomic max on the bits; the caller zeroes them before the producer runs).  The bound
 * puts the largest value below 2^15 (fp16 holds 65504); a bound that is loose by a factor L costs accuracy only beyond L = 2^18
 * (error relative to the tensor's maximum: max(2^-22, 2^-40 L)).
 *   in      IVX_F16_PAIR (d->in_dtype), written with *in_scale (NULL: 1); wgt: IVX_F16_PAIR filters (ivx_conv_desc), their
 *           power-of-two scale folded into scale[] by the caller (scale[co] = bn_scale[co] / s_w: exact)
 *   out     d->out_dtype IVX_F32 or IVX_F16_PAIR (then *out_scale receives its scale; needs amax_in, and amax_res with a residual)
 *   res     res_dtype IVX_F32 or IVX_F16_PAIR (*res_scale), independent of the output's type
 *   amax_out  slots that receive max |out| (or NULL)
 * Restrictions: out_mode 0, Cout % 4 == 0 (pair output / residual: Cout % 16 == 0), tensors below 2 GiB. */
#define IVX_AMAX_SLOTS 64
typedef struct ivx_pair_io {
  const float *in_scale;       /* device [1] or NULL */
  const float *res_scale;      /* device [1]; res_dtype IVX_F16_PAIR */
  int32_t res_dtype;           /* IVX_F32 | IVX_F16_PAIR */
  float *out_scale;            /* device [1]; out_dtype IVX_F16_PAIR */
  const uint32_t *amax_in;     /* device [IVX_AMAX_SLOTS]: bits of max |in| */
  const uint32_t *amax_res;    /* device [IVX_AMAX_SLOTS]: bits of max |res| */
  uint32_t *amax_out;          /* device [IVX_AMAX_SLOTS] or NULL */
  float wbound, sbound;
} ivx_pair_io;
int64_t ivx_conv_pio_workspace_bytes(const ivx_conv_desc *d, const ivx_pair_io *io);
int ivx_conv_fwd_pio(const ivx_conv_desc *d, const ivx_pair_io *io, const void *in, const void *wgt, const float *scale, const float *shift,
                     const void *res, void *out, void *workspace, int64_t workspace_bytes, ivx_stream_t stream);
/* validation kernel of the same contract (tests only): plain fp32 products of the values the pairs stand for */
int ivx_conv_fwd_pio_naive(const ivx_conv_desc *d, const ivx_pair_io *io, const void *in, const void *wgt, const float *scale, const float *shift,
                           const void *res, void *out, ivx_stream_t stream);
/* Host-only (both hosts of the library call it, so they hand the kernels identical bits): fp32 filters w [Cout][taps][Cin] (tap-major,
 * Cin % 32 == 0) -> IVX_F16_PAIR filters `packed` (2 * Cout * taps * Cin halves, chunk-major K: [Cout][Cin/32][taps][hi16|lo16|hi16|lo16])
 * of s_w * w with s_w the power of two that puts max |w| into [2^14, 2^15); scale_out[co] = scale[co] / s_w (scale NULL: 1 / s_w);
 * *wbound = max_co |scale[co]| * sum_k |w[co][k]|, *sbound = max_co |shift[co]| (shift NULL: 0) -- the terms of ivx_pair_io's bound. */
int ivx_pair_pack_filters(const float *w, int32_t Cout, int32_t taps, int32_t Cin, const float *scale, const float *shift, void *packed,
                          float *scale_out, float *wbound, float *sbound);
/* The head of such a chain.  ivx_nchw_to_nhwc that also accumulates max |in| into amax (the image);  nn.MaxPool2d on an fp32 map
 * (the stem's output) that writes an IVX_F16_PAIR tensor with the scale of the bound amax_in * wbound + sbound (the stem as a function
 * of the image: a maximum over a window cannot exceed it), leaves that scale in *out_scale and max |out| in amax_out.  C % 16 == 0. */
int ivx_nchw_to_nhwc_amax(const float *in, int32_t B, int32_t C, int64_t S, int32_t Cpad, float *out, uint32_t *amax, ivx_stream_t stream);
int ivx_maxpool2d_fwd_pair(const float *in, int32_t B, int32_t H, int32_t W, int32_t C, int32_t k, int32_t s, int32_t p, void *out,
                           const uint32_t *amax_in, float wbound, float sbound, float *out_scale, uint32_t *amax_out, ivx_stream_t stream);
/* IVX_F16_PAIR [n] (scale *scale_dev or 1) -> fp32 [n] (tests / hosts that want to look at an intermediate tensor) */
int ivx_f16_pair_merge(const void *in, int64_t n, const float *scale_dev, float *out, ivx_stream_t stream);

/* The head of the 2-D trunk in ONE launch (csrc/stem.hip), inside the pair chain: conv 7x7 stride 2 pad 3 (3 -> 64) + BatchNorm + ReLU +
 * MaxPool2d(3, 2, 1) from the fp32 NCHW image [B,3,H,W] to the IVX_F16_PAIR map [B,1,Hp,Wp,64] (ivx_stem_pool_out_dims) -- mmdet ResNet's
 * conv1 / bn1 / relu / maxpool (mmdet3d/models/detectors/imvoxelnet.py:22,48; configs/imvoxelnet/imvoxelnet_kitti.py:4-12).  The conv runs on
 * fp16 (hi, lo) pair operands (image values split in registers with the power-of-two scale of amax_img; three MFMA products per multiply-add),
 * the pool is exact, the output scale is the one of ivx_maxpool2d_fwd_pair (bound amax_img * wbound + sbound).
 *   ivx_amax_f32                 max |x| of an fp32 buffer into IVX_AMAX_SLOTS zeroed words (the image's maximum; a NaN counts as Inf)
 *   ivx_stem_pool_pack_filters   host-only: [64,3,7,7] fp32 filters + BN scale -> the kernel's fragment-ordered pair filters
 *                                (ivx_stem_pool_filter_bytes() bytes) and scale / s_w
 *   ivx_stem_pool_fwd_pair       wfrag / scale_p / shift: device copies of those and of the BN shift; out_scale / amax_out as ivx_pair_io */
int ivx_amax_f32(const float *x, int64_t n, uint32_t *amax, ivx_stream_t stream);
int64_t ivx_stem_pool_filter_bytes(void);
int ivx_stem_pool_pack_filters(const float *w, const float *scale, void *packed, float *scale_out);
int ivx_stem_pool_out_dims(int32_t H, int32_t W, int32_t *Hp, int32_t *Wp);
int ivx_stem_pool_fwd_pair(const float *img, int32_t B, int32_t H, int32_t W, const void *wfrag, const float *scale_p, const float *shift,
                           float wbound, float sbound, const uint32_t *amax_img, void *out, float *out_scale, uint32_t *amax_out,
                           ivx_stream_t stream);

/* One ResNet bottleneck with an identity shortcut in ONE launch (csrc/bottleneck.hip), inside the pair chain:
 *   out = relu(bn3(conv3(relu(bn2(conv2(relu(bn1(conv1(in)))))))) + in),  conv1 1x1 4P -> P, conv2 3x3 pad 1 P -> P, conv3 1x1 P -> 4P, stride 1
 * -- the identity blocks of mmdet's ResNet(depth=50, style='pytorch') the reference builds at mmdet3d/models/detectors/imvoxelnet.py:22 from
 * configs/imvoxelnet/imvoxelnet_kitti.py:4-12 and runs at :48 (`self.backbone(img)`); stages 1 and 2 (P = 64, 128).  The P-channel
 * intermediates stay in LDS as pair tiles.  in / out: IVX_F16_PAIR [B,1,H,W,4P]; w1 / w2 / w3, scale* / shift*: the pair filters, scale / s_w
 * and shift of ivx_pair_pack_filters for the three layers (w1 [P][4P/32][1][64], w2 [P][P/32][9][64], w3 [4P][P/32][1][64] halves).
 * Scales: powers of two from the BOUND chain b1 = max|in| * wbound[0] + sbound[0], b2 = b1 * wbound[1] + sbound[1],
 * b3 = b2 * wbound[2] + sbound[2] + max|in| (max|in| read from amax_in); *out_scale receives the output's, amax_out max |out|.
 * Differs from three ivx_conv_fwd_pio calls only by the scales of the two intermediates (a-priori bound instead of the measured maximum):
 * fp32-level rounding.  ivx_bottleneck_supported: P in {64, 128} and the tensor below 2 GiB. */
typedef struct ivx_bottleneck_desc {
  int32_t B, H, W, P;
} ivx_bottleneck_desc;
typedef struct ivx_bottleneck_io {
  const float *in_scale;       /* device [1] */
  const uint32_t *amax_in;     /* device [IVX_AMAX_SLOTS] */
  float *out_scale;            /* device [1] */
  uint32_t *amax_out;          /* device [IVX_AMAX_SLOTS] or NULL */
  float wbound[3], sbound[3];
} ivx_bottleneck_io;
int ivx_bottleneck_supported(const ivx_bottleneck_desc *d);
int ivx_bottleneck_fwd_pio(const ivx_bottleneck_desc *d, const ivx_bottleneck_io *io, const void *in, const void *w1, const float *scale1,
                           const float *shift1, const void *w2, const float *scale2, const float *shift2, const void *w3, const float *scale3,
                           const float *shift3, void *out, ivx_stream_t stream);
/* The FIRST block of ResNet stage 1 in one launch (layer1.0 of mmdet's ResNet-50: stride 1, shortcut = 1x1 conv + BN of the block's input; same
 * reference lines as above):  out = relu(bn3(conv3(relu(bn2(conv2(relu(bn1(conv1(in)))))))) + bnd(convd(in))),  in: IVX_F16_PAIR [B,1,H,W,Cin],
 * out: [B,1,H,W,4P]; built for Cin = P = 64 (ivx_bottleneck_proj_supported).  conv3 and the shortcut conv run as ONE GEMM over K = Cin + P:
 *   ivx_bottleneck_proj_pack   host-only (both hosts call it): w3 [4P][P], wd [4P][Cin] fp32 + the two BatchNorms' scale / shift -> the joint pair
 *                              filter bank `packed` [4P][(Cin + P) / 32][1][64] halves of s_w * [scaled * wd | scale3 * w3] (the BN scales folded
 *                              into the filters), scale_out[n] = 1 / s_w, shift_out = shift3 + shiftd; *wbound3 = max_n sum_k |scale3 w3|,
 *                              *wboundd = max_n sum_k |scaled wd|, *sbound = max_n |shift_out|
 *   ivx_bottleneck_proj_fwd_pio  w1 / w2 and their vectors as ivx_bottleneck_fwd_pio; w3d / scale3d / shift3d: the packed bank and its vectors;
 *                              io->wbound[2] = wbound3, io->sbound[2] = sbound, wbound_shortcut = wboundd: b3 = b2 * wbound3 + sbound + max|in| * wboundd.
 * Differs from the four-launch chain by fp32-level rounding (scales from bounds; BN scales folded into the filters; one accumulation). */
int ivx_bottleneck_proj_supported(const ivx_bottleneck_desc *d, int32_t Cin);
int ivx_bottleneck_proj_pack(const float *w3, const float *scale3, const float *shift3, const float *wd, const float *scaled, const float *shiftd,
                             int32_t P, int32_t Cin, void *packed, float *scale_out, float *shift_out, float *wbound3, float *sbound, float *wboundd);
int ivx_bottleneck_proj_fwd_pio(const ivx_bottleneck_desc *d, int32_t Cin, const ivx_bottleneck_io *io, float wbound_shortcut, const void *in,
                                const void *w1, const float *scale1, const float *shift1, const void *w2, const float *scale2, const float *shift2,
                                const void *w3d, const float *scale3d, const float *shift3d, void *out, ivx_stream_t stream);

/* Validation kernel: same contract, one thread per output element, plain FMA loop. Used by the
 * GPU tests to cross-check the MFMA kernel at full size; never called by the product path.   */
int ivx_conv_fwd_naive(const ivx_conv_desc *d, const void *in, const void *wgt, const float *scale,
                       const float *shift, const void *res, void *out, ivx_stream_t stream);

/* Fraction of the 3 x slices tap-slices the Winograd-domain GEMM of this layer issues under the library's default kernel rule (FLOP
 * accounting of bench.py / the stage trace): 7/9 for stride-1, pad-1 layers on 3-slice columns with fp16 pair operands and more than 64
 * output channels (their z-blocked tile skips the taps outside the column), else 1.  d: the layer descriptor (W = slices). */
float ivx_conv_winograd_issued_fraction(const ivx_conv_desc *d);
/* Minimal-filtering (Winograd F(m x m, 3x3), tile m = 2, 4 or 6) form of the same convolution for 3x3xKW kernels with
 * stride 1 on the first two spatial axes (D, H) -- the 3x3x3 layers of the KITTI / nuScenes necks
 * (mmdet3d/models/necks/imvoxelnet.py:99-113,181-230), which are bound by the fp32 MFMA rate: (m+2)^2 instead of 9*m*m
 * multiplications per m x m output tile and W-tap (16 vs 36, 36 vs 144, 64 vs 324).  Same contract, epilogue and fp32
 * arithmetic as ivx_conv_fwd; the result differs by fp32 rounding only (max deviation from the direct kernel on a
 * 256-channel layer, as a fraction of the output range: tile 2 4e-6, tile 4 2e-5, tile 6 4e-5).  The W axis keeps its kernel
 * extent, stride and padding as a direct convolution.
 * Restrictions: KD = KH = 3, sd = sh = 1, fp32, out_mode 0, res_mode 0/1, Cin % 4 == 0, Cout % 4 == 0, and one transformed
 * plane [B, ceil(Do/m), ceil(Ho/m), W, Cin] below 2 GiB (ivx_conv_winograd_supported returns 1 when all hold).
 *   u          transformed filters, ivx_conv_winograd_weight_elems(d, tile) floats: [(m+2)^2][Cout][KW*Cin] with the K order
 *              of d->wgt_layout; made once per layer by ivx_conv_winograd_weights from wgt in layout 0 [Cout,3,3,KW,Cin].
 *   workspace  ivx_conv_winograd_workspace_bytes(d, tile) bytes (the transformed input and the (m+2)^2 partial outputs).
 * ivx_conv_winograd_fwd = _input (transform the input into the workspace) + _gemm ((m+2)^2 independent 1x1xKW
 * convolutions, one grouped launch of the implicit-GEMM kernel) + _output (inverse transform + epilogue); the stages are
 * exported separately so that a caller can time them. */
int ivx_conv_winograd_supported(const ivx_conv_desc *d, int32_t tile);
int64_t ivx_conv_winograd_weight_elems(const ivx_conv_desc *d, int32_t tile);
int ivx_conv_winograd_weights(const ivx_conv_desc *d, int32_t tile, const float *wgt, float *u, ivx_stream_t stream);
int64_t ivx_conv_winograd_workspace_bytes(const ivx_conv_desc *d, int32_t tile);
int ivx_conv_winograd_input(const ivx_conv_desc *d, int32_t tile, const void *in, void *workspace, int64_t workspace_bytes,
                            ivx_stream_t stream);
int ivx_conv_winograd_gemm(const ivx_conv_desc *d, int32_t tile, const float *u, void *workspace, int64_t workspace_bytes,
                           ivx_stream_t stream);
int ivx_conv_winograd_output(const ivx_conv_desc *d, int32_t tile, const float *scale, const float *shift, const void *res,
                             void *out, void *workspace, int64_t workspace_bytes, ivx_stream_t stream);
int ivx_conv_winograd_fwd(const ivx_conv_desc *d, int32_t tile, const void *in, const float *u, const float *scale,
                          const float *shift, const void *res, void *out, void *workspace, int64_t workspace_bytes,
                          ivx_stream_t stream);
/* Chained Winograd layers with fp16-pair operands: the scale of V needs max |in|, which ivx_conv_winograd_input takes with a pass over
 * the tensor.  When `in` is the output of another Winograd layer, that layer's output transform can leave one maximum per workgroup
 * (`partials`, ivx_conv_winograd_output_blocks(d, tile) floats, caller-owned device memory) and the consumer's input stage reduces those
 * few KB instead: ivx_conv_winograd_output_amax / ivx_conv_winograd_input_amax (partials NULL = the plain entry points).  The maxima are
 * those of the stored tensor (after the epilogue), so both routes give the same scale, bit for bit. */
int32_t ivx_conv_winograd_output_blocks(const ivx_conv_desc *d, int32_t tile);
int ivx_conv_winograd_output_amax(const ivx_conv_desc *d, int32_t tile, const float *scale, const float *shift, const void *res, void *out,
                                  void *workspace, int64_t workspace_bytes, float *partials, ivx_stream_t stream);
int ivx_conv_winograd_input_amax(const ivx_conv_desc *d, int32_t tile, const void *in, void *workspace, int64_t workspace_bytes,
                                 const float *partials, int32_t n_partials, ivx_stream_t stream);
/* F(4x4,3x3) on IVX_F16_PAIR operands with the GEMM stage and the output stage fused into ONE launch (round 5): the workgroup that
 * multiplies a block of tile rows walks all 36 frequency points, folds every partial product block into the output domain on chip
 * (out = At M A accumulated in registers) and applies the epilogue -- M is never written to the workspace.  Replaces the pair
 * ivx_conv_winograd_gemm + ivx_conv_winograd_output[_amax] after ivx_conv_winograd_input[_amax] on the SAME workspace; applies to the
 * stride-1, pad-1, 3-tap z kernels of the ResModules (mmdet3d/models/necks/imvoxelnet.py:94-123) with wgt_layout 1 and Cin % 32 == 0
 * while the 36 planes of V stay below 2 GiB (ivx_conv_winograd_fused_supported).  Results equal the three-stage form up to fp32
 * rounding (another summation order of the same products).  partials (or NULL): ivx_conv_winograd_fused_blocks(d, tile) floats, one
 * max |out| per workgroup, for the consumer's ivx_conv_winograd_input_amax. */
int ivx_conv_winograd_fused_supported(const ivx_conv_desc *d, int32_t tile);
int32_t ivx_conv_winograd_fused_blocks(const ivx_conv_desc *d, int32_t tile);
int ivx_conv_winograd_gemm_output_amax(const ivx_conv_desc *d, int32_t tile, const float *u, const float *scale, const float *shift,
                                       const void *res, void *out, void *workspace, int64_t workspace_bytes, float *partials,
                                       ivx_stream_t stream);


/* Modulated deformable convolution (DCNv2; mmcv ModulatedDeformConv2dPack, deform_groups = 1) -- nuScenes reference
 * backbone, configs/imvoxelnet/imvoxelnet_nuscenes.py:13-14.  Builds the modulated, bilinearly sampled columns
 *   col[b,ho,wo,k,c] = sigmoid(m_k) * bilinear(x[b,:,:,c], ho*s - p + i*d + dh_k, wo*s - p + j*d + dw_k)     (k = i*kw + j)
 * from x [B,H,W,C] and the raw output offset_mask [B,Ho,Wo,om_channels] of the companion conv_offset (channels
 * dh_0,dw_0,...,dh_{K-1},dw_{K-1},m_0..m_{K-1}).  The contraction over (k,c) is ivx_conv_fwd with a 1x1 kernel and
 * Cin = kh*kw*C on col viewed as [B,1,Ho,Wo,kh*kw*C].  C % 4 == 0. */
int ivx_dcn_im2col_fwd(const float *x, const float *offset_mask, int32_t B, int32_t H, int32_t W, int32_t C, int32_t kh,
                       int32_t kw, int32_t stride, int32_t pad, int32_t dil, int32_t om_channels, float *col,
                       ivx_stream_t stream);
/* The same inside a chain of fp16-pair activations (0.4.0): x is an IVX_F16_PAIR map [B,H,W,2C] with the scale *x_scale (device), col an
 * IVX_F16_PAIR tensor [B,Ho,Wo,2*kh*kw*C]; the columns keep x's scale (every column is a convex combination of values of x times a mask
 * in (0, 1)): *col_scale = *x_scale, and max |col| goes to the amax slots col_amax (may be NULL).  The decoded columns are the fp32
 * columns of the decoded x rounded to the pair format.  C % 16 == 0.  The contraction is ivx_conv_fwd_pio with Cin = kh*kw*C. */
int ivx_dcn_im2col_fwd_pair(const void *x, const float *x_scale, const float *offset_mask, int32_t B, int32_t H, int32_t W, int32_t C,
                            int32_t kh, int32_t kw, int32_t stride, int32_t pad, int32_t dil, int32_t om_channels, void *col,
                            float *col_scale, uint32_t *col_amax, ivx_stream_t stream);

/* nn.MaxPool2d(kernel, stride, padding) on NHWC (ResNet stem: 3, 2, 1). */
int ivx_maxpool2d_fwd(const float *in, int32_t B, int32_t H, int32_t W, int32_t C, int32_t k,
                      int32_t s, int32_t p, float *out, ivx_stream_t stream);
/* Same on bf16 storage (optional reduced-precision mode; a max of bf16 values is exact). */
int ivx_maxpool2d_fwd_bf16(const void *in, int32_t B, int32_t H, int32_t W, int32_t C, int32_t k,
                           int32_t s, int32_t p, void *out, ivx_stream_t stream);
/* The same pool on e4m3 bytes (IVX_FP8 storage; C % 16 == 0): exact, the tensor's scale is unchanged. */
int ivx_maxpool2d_fwd_fp8(const void *in, int32_t B, int32_t H, int32_t W, int32_t C, int32_t k, int32_t s, int32_t p,
                          void *out, ivx_stream_t stream);

/* F.interpolate(scale_factor=2, mode='trilinear', align_corners=False) on NDHWC [B,D,H,W,C] -> [B,2D,2H,2W,C]
 * (Atlas decoder of ImVoxelNeck, mmdet3d/models/necks/imvoxelnet.py:359).  C % 4 == 0. */
int ivx_upsample_trilinear2x_fwd(const float *in, int32_t B, int32_t D, int32_t H, int32_t W, int32_t C, float *out,
                                 ivx_stream_t stream);

/* NCHW [B,C,H,W] -> NHWC [B,H,W,Cpad] (channels >= C zero filled), and NHWC/NDHWC -> NCHW/NCDHW
 * ([B,S,C] -> [B,C,S] with S = product of the spatial dims). */
int ivx_nchw_to_nhwc(const float *in, int32_t B, int32_t C, int64_t S, int32_t Cpad, float *out,
                     ivx_stream_t stream);
/* bf16 mode only: image [B,3,H,W] fp32 NCHW (H, W even) -> 2x2 space-to-depth blocks [B, H/2+1, W/2+1, 16] bf16 with
 * out[b][ph][pw][(a*2+e)*3+c] = img[b][c][2ph-1+a][2pw-1+e] (zero outside, channels 12..15 zero): the 7x7 stride-2 pad-3 stem
 * (mmdet ResNet.conv1) becomes a 4x4 stride-1 pad-1 convolution over it (weights re-indexed kh = 2*th + a, kw = 2*tw + e). */
int ivx_image_s2d_bf16(const float *img, int32_t B, int32_t H, int32_t W, void *out, ivx_stream_t stream);
int ivx_nhwc_to_nchw(const float *in, int32_t B, int64_t S, int32_t C, float *out, ivx_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * Fused image-to-voxel unprojection -- replaces the per-sample Python loop
 * mmdet3d/models/detectors/imvoxelnet.py:58-76 (get_points :132-141, backproject :145-160,
 * view sum / count / divide / zero-fill :70-74) for a whole batch in one launch.
 *
 * feat        [B*V, FH, FW, C]  FPN level-0 features, NHWC
 * proj        [B, V, 3, 4]      per-view K' @ E[:3] (built on the host exactly as :114-129)
 * new_origin  [B, 3]            origin - n_voxels/2 * voxel_size (fp32, :139)
 * crop_hw     [B, 2] int32      img_shape // stride (h, w): the crop at :67-69
 * voxel_size  host float[3]
 * volume      [B, X, Y, Z, C]   mean over valid views, 0 where no view sees the voxel
 * valid       [B, X, Y, Z] u8   1 where >= 1 view sees the voxel (the reference's `valids`)
 * Projection arithmetic is the reference's bit for bit: fp32 FMA chain in k order (what
 * torch.bmm does), IEEE division, round-half-even, nearest-pixel gather.                    */
int ivx_backproject_mean_fwd(const float *feat, int32_t B, int32_t V, int32_t FH, int32_t FW, int32_t C,
                             const float *proj, const float *new_origin, const int32_t *crop_hw,
                             const float *voxel_size /*host*/, int32_t X, int32_t Y, int32_t Z,
                             float *volume, uint8_t *valid, ivx_stream_t stream);

/* bf16 storage variants (optional reduced-precision mode; sums, divisions and interpolation weights in fp32, one
 * rounding at the store): multi-view lift and the Atlas decoder's trilinear x2 up-sampling.  A single-view bf16 lift
 * is a byte copy: pass the map to ivx_backproject_mean_fwd as C/2 32-bit words.                                 */
int ivx_backproject_mean_fwd_bf16(const void *feat, int32_t B, int32_t V, int32_t FH, int32_t FW, int32_t C,
                                  const float *proj, const float *new_origin, const int32_t *crop_hw,
                                  const float *voxel_size, int32_t X, int32_t Y, int32_t Z, void *volume,
                                  uint8_t *valid, ivx_stream_t stream);
int ivx_upsample_trilinear2x_fwd_bf16(const void *in, int32_t B, int32_t D, int32_t H, int32_t W, int32_t C, void *out,
                                      ivx_stream_t stream);

/* View-sharded multi-GPU mode (second mode: one exchange step): every rank lifts ITS views with
 * ivx_backproject_sum_fwd -- same geometry, but the raw sum over the rank's views (volume_sum [B,X,Y,Z,C]) and the
 * per-voxel number of views that saw the voxel (count [B,X,Y,Z] int32) instead of the mean --, the two tensors are
 * all-reduced over the ranks (RCCL), and ivx_volume_normalize_fwd turns the totals into the reference's result in
 * place: volume = count ? sum / count : 0, valid = count > 0 (detectors/imvoxelnet.py:70-74).  C % 4 == 0.
 * The view sum is then ordered rank by rank instead of strictly by view: equal to the single-GPU result to fp32
 * rounding of the additions (the valid mask is exact).                                                         */
/* ivx_backproject_mean_fwd that also leaves one max |volume| per workgroup (single view only: ivx_backproject_amax_blocks floats, 0 when the
 * shape does not take that kernel) for the operand scale of the first neck layer (ivx_conv_winograd_input_amax). */
int32_t ivx_backproject_amax_blocks(int32_t B, int32_t V, int32_t X, int32_t Y, int32_t Z);
int ivx_backproject_mean_fwd_amax(const float *feat, int32_t B, int32_t V, int32_t FH, int32_t FW, int32_t C, const float *proj,
                                  const float *new_origin, const int32_t *crop_hw, const float *voxel_size, int32_t X, int32_t Y,
                                  int32_t Z, float *volume, uint8_t *valid, float *partials, ivx_stream_t stream);
int ivx_backproject_sum_fwd(const float *feat, int32_t B, int32_t V, int32_t FH, int32_t FW, int32_t C,
                            const float *proj, const float *new_origin, const int32_t *crop_hw,
                            const float *voxel_size, int32_t X, int32_t Y, int32_t Z, float *volume_sum,
                            int32_t *count, ivx_stream_t stream);
int ivx_volume_normalize_fwd(float *volume, const int32_t *count, int64_t n_voxels, int32_t C, uint8_t *valid,
                             ivx_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * Anchor3DHead tail -- replaces Anchor3DHead.get_bboxes_single
 * (mmdet3d/models/dense_heads/anchor3d_head.py:428-517) for a batch, single feature level,
 * sigmoid classification: permute to HWC, sigmoid, direction argmax, top-k(nms_pre),
 * DeltaXYZWLHRBBoxCoder.decode (coders/delta_xyzwhlr_bbox_coder.py:56-90), BEV boxes
 * (lidar_box3d.py:86-90 + structures/utils.py:64-82), box3d_multiclass_nms
 * (post_processing/box3d_nms.py:8-88) with rotated (iou3d_kernel.cu:284-333) or axis-aligned
 * (:345-396) NMS and the greedy scan of iou3d.cpp:127-143 done on the device, top max_num,
 * yaw fix-up with limit_period (anchor3d_head.py:510-515).
 *
 * head_out   [B, H, W, CH]  NHWC output of the fused 1x1 head conv; channel blocks
 *                           [cls: A*ncls | reg: A*7 | dir: A*2] starting at cls_off/reg_off/dir_off
 * anchors    [H*W*A, 7]     Anchor3DRangeGenerator.grid_anchors output (flat order y, x, size, rot)
 * workspace  ivx_anchor_head_workspace_bytes() bytes, 256-byte aligned
 * out_boxes  [B, max_num, 7], out_scores [B, max_num], out_labels [B, max_num] int64,
 * out_count  [B] int32.  Rows >= count are zero.
 * Optional debug outputs (NULL to skip): cand_idx [B,nms_pre] int64 top-k anchor indices in
 * descending-score order (-1 padded), cand_boxes [B,nms_pre,7], cand_scores [B,nms_pre].     */
typedef struct ivx_anchor_head_desc {
  int32_t B, H, W, CH;
  int32_t num_anchors; /* A: base anchors per location (sizes * rotations) */
  int32_t num_classes;
  int32_t cls_off, reg_off, dir_off;
  int32_t nms_pre, max_num;
  int32_t use_rotate_nms;
  int32_t hw_transposed; /* 1: head_out is stored [B, W, H, CH] (x-major), as the Kitti/NuScenes neck
                            leaves it before its .transpose(-1, -2) (necks/imvoxelnet.py:120) */
  float score_thr, nms_thr;
  float dir_offset, dir_limit_offset;
} ivx_anchor_head_desc;

int64_t ivx_anchor_head_workspace_bytes(const ivx_anchor_head_desc *d);
int ivx_anchor_head_get_bboxes(const ivx_anchor_head_desc *d, const float *head_out, const float *anchors,
                               void *workspace, int64_t workspace_bytes, float *out_boxes,
                               float *out_scores, int64_t *out_labels, int32_t *out_count,
                               int64_t *cand_idx, float *cand_boxes, float *cand_scores,
                               ivx_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * Indoor (anchor-free) head tail, one feature level -- replaces the per-level body of
 * ImVoxelHeadV2._get_bboxes_single (mmdet3d/models/dense_heads/imvoxel_head_v2.py:245-277 with :224-226, :206-214,
 * ScanNet decode :547-555 / SUN RGB-D decode :419-438 and the exp(Scale(.)) of forward_single :305-313, :444-449):
 * valid mask resized to the level (trilinear + round == ">= 5 of the 8 contributing level-0 voxels"), scores =
 * sigmoid(cls) * sigmoid(centerness) * valid, top-k(nms_pre) by class maximum, level points, distance decoding.
 *
 * head_out [B, nx, ny, nz, CH]  fused 3x3x3 head conv: channel 0 centerness, 1..n_reg raw regression, then classes
 * valid0   [B, X, Y, Z] u8      level-0 mask from ivx_backproject_mean_fwd; (nx,ny,nz) * 2^level == (X,Y,Z)
 * level_vs, level_new_origin [B,3]  voxel_size * 2^level and origin - n_level/2 * that (host-built fp32, as get_points)
 * n_reg 6: boxes are corners (x1,y1,z1,x2,y2,z2); n_reg 7: (cx,cy,cz,w,l,h,alpha)
 * cand_boxes [B, k, n_reg], cand_scores [B, k, n_classes], cand_count [B] with k = min(nms_pre, nx*ny*nz) <= 65536 (beyond 4096: rank sort of the candidates, same order),
 * candidates in descending class-maximum score.  The (cross-level) NMS is ivx_aligned_3d_nms / ivx_nms_bev.     */
int64_t ivx_fcos_head_workspace_bytes(int32_t B, int32_t n, int32_t nms_pre);
int ivx_fcos_head_level_candidates(const float *head_out, const uint8_t *valid0, const float *level_vs,
                                   const float *level_new_origin, float scale, int32_t B, int32_t nx, int32_t ny,
                                   int32_t nz, int32_t CH, int32_t n_classes, int32_t n_reg, int32_t level, int32_t X,
                                   int32_t Y, int32_t Z, int32_t nms_pre, void *workspace, int64_t workspace_bytes,
                                   float *cand_boxes, float *cand_scores, int32_t *cand_count, ivx_stream_t stream);

/* Cross-level tail of the anchor-free indoor heads: replaces the torch.cat over levels and `_nms` of
 * ImVoxelHeadV2._get_bboxes_single (mmdet3d/models/dense_heads/imvoxel_head_v2.py:258-277; ScanNet _nms :528-545 = class
 * maximum, score threshold, class-aware aligned_3d_nms, corners -> centre / size; SUN RGB-D _nms :397-417 = BEV boxes +
 * box3d_multiclass_nms with max_num = nms_pre) for a batch, without a host round trip.
 *   cand_boxes[l] [B, k[l], n_reg], cand_scores[l] [B, k[l], n_classes]   outputs of ivx_fcos_head_level_candidates per level
 *   n_reg 6 (ScanNet): nms_thr = test_cfg.iou_thr, max_num >= sum(k) (nothing is cut); n_reg 7 (SUN RGB-D): nms_thr / use_rotate_nms
 *   of test_cfg, max_num = test_cfg.nms_pre.
 *   out_boxes [B, max_num, 7] = the rows of the returned box object's tensor (x, y, z of the BOTTOM face, dx, dy, dz, yaw; yaw 0 for
 *   ScanNet) -- BaseInstance3DBoxes(origin=(.5,.5,.5)) applied (core/bbox/structures/base_box3d.py:63-66); out_scores, out_labels
 *   int64 [B, max_num], out_count [B]; rows >= count are zero. */
typedef struct ivx_indoor_tail_desc {
  int32_t B, n_levels;
  int32_t k[4];              /* candidates per level (min(nms_pre, level voxels)) */
  int32_t n_classes, n_reg;
  int32_t use_rotate_nms, max_num;
  float score_thr, nms_thr;
} ivx_indoor_tail_desc;
int64_t ivx_indoor_tail_workspace_bytes(const ivx_indoor_tail_desc *d);
int ivx_indoor_tail_get_bboxes(const ivx_indoor_tail_desc *d, const float *const *cand_boxes, const float *const *cand_scores,
                               void *workspace, int64_t workspace_bytes, float *out_boxes, float *out_scores, int64_t *out_labels,
                               int32_t *out_count, ivx_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * BEV NMS -- device-side replacement of iou3d_cuda.nms_gpu / nms_normal_gpu
 * (mmdet3d/ops/iou3d/src/iou3d.cpp:95-201, iou3d_kernel.cu:284-396).  Same contract as the
 * reference op except that nothing is copied to the host: boxes [n,5] (x1,y1,x2,y2,ry) must
 * already be sorted by descending score (as iou3d_utils.py:39-47 does before the call);
 * keep [n] int64 receives indices into that order, num_out [1] int32 their count.
 * workspace >= ivx_nms_workspace_bytes(n) (the n x ceil(n/64) mask words, as the reference's cudaMalloc at iou3d.cpp:110).
 * Any n up to 65536, like the reference op (col_blocks = DIVUP(N, 64)): up to 4096 boxes the scanning wave keeps one removal
 * word per lane, beyond that the words live in its LDS -- same kept sequence.                  */
int64_t ivx_nms_workspace_bytes(int32_t n);
int ivx_nms_bev(const float *boxes_sorted, int32_t n, float thresh, int32_t rotated, void *workspace,
                int64_t workspace_bytes, int64_t *keep, int32_t *num_out, ivx_stream_t stream);
/* Pairwise rotated BEV overlap area / IoU (boxes_overlap_bev_gpu / boxes_iou_bev_gpu,
 * iou3d.cpp:38-93): a [na,5], b [nb,5] -> out [na,nb]. */
int ivx_boxes_overlap_bev(const float *a, int32_t na, const float *b, int32_t nb, int32_t iou, float *out,
                          ivx_stream_t stream);

/* aligned_3d_nms (mmdet3d/core/post_processing/box3d_nms.py:91-138) on the device.
 * boxes [n,6] corners, scores [n], classes [n] int64; pick [n] int64 receives the kept box
 * indices in descending score order, num_out [1] int32.  One workgroup, no workspace: n <= 4096 (use the _ws form beyond). */
int ivx_aligned_3d_nms(const float *boxes, const float *scores, const int64_t *classes, int32_t n,
                       float thresh, int64_t *pick, int32_t *num_out, ivx_stream_t stream);
/* The same result with a workspace, n <= 65536.  Up to 4096 boxes the greedy chain runs per class on 64 workgroups in
 * parallel (boxes of different classes never suppress each other); inputs with degenerate boxes (non-finite corners, an extent
 * outside (0, 1e6)), where the reference's NaN IoU suppresses ACROSS classes, take the one-workgroup form inside the call.
 * Beyond 4096: rank sort by score, a suppression mask over the sorted boxes with the reference's own predicate
 * NOT(iou * same_class <= thresh), greedy scan with the removal bits in LDS -- same picks, any input. */
int64_t ivx_aligned_3d_nms_workspace_bytes(int32_t n);
int ivx_aligned_3d_nms_ws(const float *boxes, const float *scores, const int64_t *classes, int32_t n, float thresh,
                          void *workspace, int64_t workspace_bytes, int64_t *pick, int32_t *num_out, ivx_stream_t stream);

/* Global average pool of a channels-last map: in [B,S,C] -> out [B,C]; `x.mean(dim=(2,3))` of LayoutHead.forward
 * (mmdet3d/models/dense_heads/layout_head.py:42, SUN RGB-D Total configs).                                        */
int ivx_global_avgpool_fwd(const float *in, int32_t B, int64_t S, int32_t C, float *out, ivx_stream_t stream);

/* Fused multi-class BEV NMS -- replaces box3d_multiclass_nms (mmdet3d/core/post_processing/box3d_nms.py:8-88), i.e.
 * the host loop over classes around nms_gpu / nms_normal_gpu with its per-class D2H, for n <= 65536 candidates and
 * num_classes <= 64 (beyond 4096 candidates: per-class rank sort, one shared hit matrix, removal bits in LDS; same output).  boxes [n,5] (x1,y1,x2,y2,ry); scores [n,score_stride], class c in column c.  Per class the
 * candidates with score > score_thr are sorted by score (descending, ties -> lower index) and greedily suppressed at
 * IoU > nms_thr; the survivors are concatenated class-major, or, when more than max_num survive, cut to the max_num
 * best by score (descending; ties -> lower class).  out_idx (index into the n candidates) and out_label hold
 * min(max_num, n * num_classes) int64 entries; *out_count receives the number written.                            */
int64_t ivx_multiclass_nms_workspace_bytes(int32_t n, int32_t num_classes);
int ivx_multiclass_nms_bev(const float *boxes, const float *scores, int32_t n, int32_t score_stride, int32_t num_classes,
                           float score_thr, float nms_thr, int32_t rotated, int32_t max_num, void *workspace,
                           int64_t workspace_bytes, int64_t *out_idx, int64_t *out_label, int32_t *out_count,
                           ivx_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * KITTI AP evaluation, host side: the per-image matching loops the reference compiles with numba.jit.
 * HOST pointers, row-major double matrices, no device work, no stream -- the rotated BEV / 3-D overlaps that feed
 * them come from ivx_boxes_overlap_bev.  Python driver: imvoxelnet_amd/kitti_ap.py.
 *   ivx_kitti_image_box_overlap   mmdet3d/core/evaluation/kitti_utils/eval.py:83-112 (criterion -1 IoU, 0 / area(box),
 *                                 1 / area(query), other: intersection area)
 *   ivx_kitti_compute_statistics  eval.py:161-279 for one image.  overlaps[j*ld + i] = detection j vs ground truth i;
 *                                 gt rows (x1,y1,x2,y2,alpha), dt rows (x1,y1,x2,y2,alpha,score); ignored_* in
 *                                 {0 counted, 1 ignored, -1 other class}; stats4 = tp, fp, fn, similarity (-1 = none);
 *                                 thresholds (n_gt doubles) receives the scores of the matched detections.
 *   ivx_kitti_collect_scores      eval.py:500-514: the compute_fp = false pass over a list of images (rows of all
 *                                 images concatenated; ov_ptrs[i] = image i's [dt_nums[i], gt_nums[i]] matrix).
 *   ivx_kitti_fused_statistics    eval.py:291-338: pr[t*4 + {tp,fp,fn,similarity}] accumulated over the images for
 *                                 every score threshold t.                                                      */
int ivx_kitti_image_box_overlap(const double *boxes, int32_t n, const double *query, int32_t k, int32_t criterion,
                                double *out);
int ivx_kitti_compute_statistics(const double *overlaps, int32_t ld, const double *gt_datas, int32_t n_gt,
                                 const double *dt_datas, int32_t n_dt, const int64_t *ignored_gt,
                                 const int64_t *ignored_det, const double *dc_bboxes, int32_t n_dc, int32_t metric,
                                 double min_overlap, double thresh, int32_t compute_fp, int32_t compute_aos,
                                 double *stats4, double *thresholds, int32_t *n_thresholds);
int ivx_kitti_collect_scores(const double *const *ov_ptrs, int32_t n_img, const int32_t *gt_nums, const int32_t *dt_nums,
                             const double *gt_datas, const double *dt_datas, const int64_t *ignored_gts,
                             const int64_t *ignored_dets, int32_t metric, double min_overlap, double *scores_out,
                             int64_t *n_out);
int ivx_kitti_fused_statistics(const double *const *ov_ptrs, int32_t n_img, const int32_t *gt_nums, const int32_t *dt_nums,
                               const int32_t *dc_nums, const double *gt_datas, const double *dt_datas,
                               const double *dontcares, const int64_t *ignored_gts, const int64_t *ignored_dets,
                               int32_t metric, double min_overlap, const double *thresholds, int32_t n_thr,
                               int32_t compute_aos, double *pr);

/* ---------------------------------------------------------------------------------------
 * Model handle -- the anchor-head ImVoxelNet forward path as one native object
 * (ImVoxelNet.simple_test, mmdet3d/models/detectors/imvoxelnet.py:45-106, for the KITTI / nuScenes families:
 * ResNet-50 -> FPN level 0 -> unprojection -> KittiImVoxelNeck | NuScenesImVoxelNeck (necks/imvoxelnet.py:94-154) ->
 * Anchor3DHead forward (anchor3d_head.py:122-153) -> get_bboxes (:375-517)).  The layer sequence, weight packing
 * (conv + eval BatchNorm -> scale/shift epilogue, chunk-major K, Winograd-domain filters), Winograd / tile selection and
 * workspace planning live in the library; the caller owns the workspace and all input / output buffers.  The
 * reference's only native precedent is the pybind op at ops/iou3d/src/iou3d.cpp:95-147,203-208 (caller-allocated
 * outputs, int return); this keeps that convention.
 *
 *   ivx_create(&cfg, &m);
 *   for every tensor of the reference state dict: ivx_weights_load(m, key, host_ptr, shape, ndim);
 *       keys as in a released checkpoint: backbone.*, neck.lateral_convs.i.conv.*, neck.fpn_convs.0.conv.*,
 *       neck_3d.model.{0,2,4}.{conv1,bn1,conv2,bn2}.*, neck_3d.model.{1,3,5}.{0,1}.*, bbox_head.conv_{cls,reg,dir_cls}.*
 *       (a leading "module." is dropped; unused keys -- fpn_convs.1..3, num_batches_tracked -- are ignored);
 *       optional pseudo-key "anchors" [H*W*A, 7]: the anchor grid to use instead of the built-in generator
 *   ivx_weights_finalize(m, stream);                       packs and uploads; lists missing keys on error
 *   n = ivx_model_workspace_bytes(m, B, V, H, W);          plans this shape (allocates transformed filters once)
 *   ivx_model_forward(m, img, B, V, H, W, proj, new_origin, crop_hw, ws, n, boxes, scores, labels, count, valid, stream);
 *   ivx_destroy(m);
 * One handle per device and per host thread; calls are asynchronous on `stream`.                                  */
#define IVX_NECK_KITTI 0
#define IVX_NECK_NUSCENES 1
#define IVX_NECK_FAST 2      /* FastIndoorImVoxelNeck (necks/imvoxelnet.py:8-67): the handle ends at the 3 neck levels */
#define IVX_NECK_UNET 3      /* ImVoxelNeck = Atlas EncoderDecoder + conv blocks (necks/imvoxelnet.py:70-91, 297-372) */
#define IVX_HEAD_NONE 0
#define IVX_HEAD_SCANNET 1
#define IVX_HEAD_SUNRGBD 2
typedef struct ivx_model ivx_model;
typedef struct ivx_model_cfg {
  int32_t neck_type;          /* IVX_NECK_KITTI | IVX_NECK_NUSCENES | IVX_NECK_FAST | IVX_NECK_UNET */
  int32_t with_trunk;         /* 1: ResNet-50 + FPN inside the handle (input = image); 0: input = FPN level-0 map */
  int32_t fpn_channels;       /* FPN out_channels = neck_3d in_channels (64 in the reference configs) */
  int32_t neck_out_channels;  /* neck_3d out_channels = head in_channels (256) */
  int32_t n_voxels[3];
  float voxel_size[3];
  int32_t num_classes;
  int32_t n_sizes, n_rotations;        /* anchor generator: sizes x rotations base anchors, one range */
  float anchor_range[6];               /* x0, y0, z0, x1, y1, z1 */
  float anchor_sizes[12];              /* n_sizes x (w, l, h) */
  float anchor_rotations[4];
  int32_t nms_pre, max_num, use_rotate_nms;   /* test_cfg */
  float score_thr, nms_thr;
  float dir_offset, dir_limit_offset;         /* Anchor3DHead(dir_offset, dir_limit_offset) */
  int32_t winograd;           /* 1: F(m x m, 3x3) form for the eligible layers (default of the Python host), 0: direct only */
  int32_t winograd_tile;      /* 0: automatic (6 on planes >= 16384 positions, else 4) | 2 | 4 | 6 */
  /* indoor necks (zero for the outdoor ones; the anchor / test_cfg fields above are ignored for them) */
  int32_t fast_n_blocks[3];        /* FastIndoorImVoxelNeck(n_blocks) */
  int32_t unet_channels[4];        /* ImVoxelNeck(channels); [0] = fpn_channels; [3] = 0: three scales (two output levels) */
  int32_t unet_down_layers[4];     /* ImVoxelNeck(n_blocks, "down"), e.g. 1,2,3,4 */
  int32_t unet_up_layers[3];       /* ImVoxelNeck(n_blocks, "up") in decode order (coarse first), e.g. 3,2,1 */
  /* (0.3.0) the rest of the path inside the handle; all zero = the 0.2.0 behaviour */
  int32_t head_type;               /* IVX_HEAD_NONE: the indoor handle ends at the neck levels | IVX_HEAD_SCANNET (n_reg 6: ScanNetImVoxelHead /
                                      HeadV2, aligned 3-D NMS) | IVX_HEAD_SUNRGBD (n_reg 7: SunRgbdImVoxelHead / HeadV2, multi-class BEV NMS);
                                      n_convs = 0 as in every reference config (dense_heads/imvoxel_head_v2.py, imvoxel_head.py) */
  int32_t head_classes;            /* n_classes */
  int32_t head_nms_pre;            /* test_cfg.nms_pre (candidates per level; also max_num of the SUN RGB-D NMS) */
  int32_t head_use_rotate_nms;     /* SUN RGB-D test_cfg.use_rotate_nms */
  float head_score_thr, head_nms_thr;   /* test_cfg.score_thr; iou_thr (ScanNet) / nms_thr (SUN RGB-D) */
  int32_t dcn_stages[4];           /* ResNet stage_with_dcn: DCNv2 (ModulatedDeformConv2dPack, deform_groups 1) as conv2 of every bottleneck of
                                      the stage (configs/imvoxelnet/imvoxelnet_nuscenes.py:13-14); keys backbone.layerS.B.conv2.conv_offset.* */
  int32_t layout_head;             /* 1: LayoutHead(n_channels 2048, linear_size) on C5 (SUN RGB-D Total configs, dense_heads/layout_head.py);
                                      its predicted angles replace the extrinsics of the metas at test time; keys head_2d.{angle,layout}_mlp.* */
  int32_t layout_linear_size;
  int32_t wino_operands;           /* (0.3.1) ivx_conv_desc.wino_operands of the layers that run in the Winograd form: IVX_F32 (0) = fp32 MFMA,
                                      IVX_F16_PAIR = fp16 (hi, lo) operand pairs where the layer allows it (tile 4 / 6, Cin % 32 == 0); with it the
                                      3x3x3 neck layers the Winograd form does not take (stride 2 in x / y; under 2000 positions) run in the
                                      IVX_BF16_PAIR form above (split pass + three-product kernel; Cin % 32 == 0, Cout >= 64, >= 256 positions) */
  int32_t trunk_operands;          /* (0.4.0) IVX_F32 (0): the 2-D trunk (ResNet-50 + FPN) on fp32 MFMA.  IVX_F16_PAIR: its activations chained as
                                      fp16 (hi, lo) pair tensors on the 16-bit matrix cores (ivx_conv_fwd_pio: three fp16 MFMA products per
                                      multiply-add, fp32 accumulate, device-side power-of-two scales, no conversion passes) wherever a tensor's
                                      consumers are all convolutions with Cin % 32 == 0; the stem, DCNv2 columns, the LayoutHead and what
                                      leaves the trunk (FPN level 0, C5) stay fp32 */
  int32_t storage;                 /* (0.4.0) IVX_F32 (0, the reference's precision and the one every parity claim is made for) or IVX_BF16: the
                                      optional reduced-precision mode inside the handle (BASELINE config 5 names it) -- activations and weights
                                      stored as bf16 (v_mfma_f32_32x32x16_bf16, fp32 accumulation and epilogues, the 7x7 stem as a 4x4 convolution
                                      over 2x2 space-to-depth blocks of the image), head outputs and the detection tails fp32.  The image input and
                                      the detection outputs keep their types; the tensors of the sub-path entry points (ivx_backbone_fpn_fwd's
                                      fpn0, ivx_neck3d_*_fwd's volume / levels, ivx_model_forward_levels' levels) are bf16.  Not with DCNv2 stages
                                      or a LayoutHead; the Winograd and pair forms are fp32-storage forms and are not used */
} ivx_model_cfg;

int ivx_create(const ivx_model_cfg *cfg, ivx_model **out);
int ivx_destroy(ivx_model *m);
int ivx_weights_load(ivx_model *m, const char *key, const float *data_host, const int64_t *shape, int32_t ndim);
int ivx_weights_finalize(ivx_model *m, ivx_stream_t stream);
/* Whole path.  input: image batch [B*V,3,H,W] NCHW fp32 (with_trunk) or FPN level-0 maps [B*V,1,H/4,W/4,Cf] channels-last;
 * H, W = padded image size (multiples of 32).  proj [B,V,3,4], new_origin [B,3], crop_hw [B,2] int32 as for
 * ivx_backproject_mean_fwd (device).  Outputs as ivx_anchor_head_get_bboxes; out_valid [B,X,Y,Z] u8 or NULL. */
int64_t ivx_model_workspace_bytes(ivx_model *m, int32_t B, int32_t V, int32_t H, int32_t W);
int ivx_model_forward(ivx_model *m, const float *input, int32_t B, int32_t V, int32_t H, int32_t W, const float *proj,
                      const float *new_origin, const int32_t *crop_hw, void *workspace, int64_t workspace_bytes,
                      float *out_boxes, float *out_scores, int64_t *out_labels, int32_t *out_count, uint8_t *out_valid,
                      ivx_stream_t stream);
/* Sub-paths on the same handle: backbone(img) + neck(x)[0] (detectors/imvoxelnet.py:48,50) and neck_3d(x) (:79).
 * fpn0 [BV,1,H/4,W/4,Cf]; volume [B,X,Y,Z,Cf] -> out [B,X',Y',1,Cout] (ivx_neck3d_out_dims). */
int64_t ivx_backbone_fpn_workspace_bytes(ivx_model *m, int32_t BV, int32_t H, int32_t W);
int ivx_backbone_fpn_fwd(ivx_model *m, const float *img, int32_t BV, int32_t H, int32_t W, float *fpn0, void *workspace,
                         int64_t workspace_bytes, ivx_stream_t stream);
int64_t ivx_neck3d_workspace_bytes(ivx_model *m, int32_t B);
int ivx_neck3d_out_dims(ivx_model *m, int32_t B, int32_t *X, int32_t *Y, int32_t *C);
int ivx_neck3d_kitti_fwd(ivx_model *m, const float *volume, int32_t B, float *out, void *workspace, int64_t workspace_bytes,
                         ivx_stream_t stream);
int ivx_neck3d_nuscenes_fwd(ivx_model *m, const float *volume, int32_t B, float *out, void *workspace,
                            int64_t workspace_bytes, ivx_stream_t stream);
/* Indoor necks: three output levels (two for a 3-scale ImVoxelNeck: the unused entry is ignored / zero), finest first, channels-last [B, X_l, Y_l, Z_l, Cout] (neck_3d(x) of
 * detectors/imvoxelnet.py:79 for FastIndoorImVoxelNeck / ImVoxelNeck).  dims[l] = {X, Y, Z, C} of level l.
 * State-dict keys: neck_3d.down_layer_{i}.{j}.{conv1,norm1,conv2,norm2,downsample.0,downsample.1}.*,
 * neck_3d.up_block_{1,2}.{0,1,3,4}.*, neck_3d.out_block_{i}.{0,1}.* (fast); neck_3d.model.layers_down.*,
 * neck_3d.model.proj.{k}.{conv,norm}.*, neck_3d.model.layers_up_conv.{k}.weight, neck_3d.model.layers_up_res.{k}.{j}.*,
 * neck_3d.conv_blocks.{l}.{0,1}.* (unet). */
int ivx_neck3d_levels(ivx_model *m, int32_t B, int32_t dims[3][4]);
int ivx_neck3d_fast_fwd(ivx_model *m, const float *volume, int32_t B, float *const out_levels[3], void *workspace,
                        int64_t workspace_bytes, ivx_stream_t stream);
int ivx_neck3d_unet_fwd(ivx_model *m, const float *volume, int32_t B, float *const out_levels[3], void *workspace,
                        int64_t workspace_bytes, ivx_stream_t stream);
/* Whole indoor path up to the head: backbone + FPN level 0 + unprojection + neck_3d (extract_feat,
 * detectors/imvoxelnet.py:43-88).  Arguments as ivx_model_forward; workspace from ivx_model_workspace_bytes. */
int ivx_model_forward_levels(ivx_model *m, const float *input, int32_t B, int32_t V, int32_t H, int32_t W, const float *proj,
                             const float *new_origin, const int32_t *crop_hw, void *workspace, int64_t workspace_bytes,
                             float *const out_levels[3], uint8_t *out_valid, ivx_stream_t stream);
/* simple_test as ONE call for every family (detectors/imvoxelnet.py:93-106): image batch [B*V,3,H,W] (with_trunk = 0: the FPN level-0
 * maps [B*V,1,H/4,W/4,Cf] channels-last) -> detections, the per-sample camera
 * set-up of :114-129 / :139 / :67-68 included (computed on the host in the library's fixed fp32 order and uploaded into the
 * workspace).  One ivx_sample_meta per sample = the fields of img_meta the path reads.
 *   anchor families (KITTI / nuScenes, DCNv2 stages included): outputs as ivx_model_forward, max_num rows per sample
 *   indoor families with head_type != 0: out_boxes [B, M, 7] rows of the returned box object's tensor (x, y, z of the bottom face,
 *     dx, dy, dz, yaw), M = ivx_model_max_detections(...): ScanNet the total number of candidates (nothing is cut), SUN RGB-D nms_pre
 *   LayoutHead (layout_head = 1, V = 1): the predicted (pitch, roll) replace metas[b].extrinsics (may be NULL); out_angles host [B,2],
 *     out_layout host [B,7] (centre, exp(size), yaw) or NULL -- the forward synchronises the stream once to read them, as the
 *     reference does.
 * Asynchronous on `stream` otherwise; `metas` and what it points to are consumed before the call returns. */
typedef struct ivx_sample_meta {
  float intrinsic[16];         /* img_meta['lidar2img']['intrinsic'], 4x4 row-major (rows 0..2 x cols 0..2 are used) */
  const float *extrinsics;     /* host [V,4,4] row-major: img_meta['lidar2img']['extrinsic'] */
  float origin[3];             /* img_meta['lidar2img']['origin'] */
  int32_t img_h, img_w;        /* img_meta['img_shape'][:2] (before padding to H x W) */
  int32_t ori_h;               /* img_meta['ori_shape'][0] */
} ivx_sample_meta;
int64_t ivx_model_detect_workspace_bytes(ivx_model *m, int32_t B, int32_t V, int32_t H, int32_t W);
int32_t ivx_model_max_detections(ivx_model *m, int32_t B, int32_t V, int32_t H, int32_t W);
int ivx_model_detect(ivx_model *m, const float *img, int32_t B, int32_t V, int32_t H, int32_t W, const ivx_sample_meta *metas /*host [B]*/,
                     void *workspace, int64_t workspace_bytes, float *out_boxes, float *out_scores, int64_t *out_labels, int32_t *out_count,
                     uint8_t *out_valid, float *out_angles /*host*/, float *out_layout /*host*/, ivx_stream_t stream);
/* BASELINE config 5's named mode inside the handle: "bf16 with fp8 2-D conv MFMA".  On a handle with storage = IVX_BF16 and the trunk
 * (no DCNv2 stages): runs ONE bf16 pass of ResNet-50 + FPN over the given images (img [BV,3,H,W] fp32; synchronises the stream once),
 * records max |output| of conv1 and conv2 of every bottleneck, and from then on stores the INSIDE of every bottleneck as OCP e4m3:
 * conv1 reads the bf16 residual stream and writes e4m3 (per-tensor scale amax * margin / 448), conv2 and conv3 run on
 * v_mfma_f32_32x32x16_fp8_fp8 (e4m3 activations, e4m3 filters with one scale per output channel), conv3 adds the bf16 shortcut and writes
 * bf16 -- the residual stream itself is never re-quantised (ImVoxelNet.calibrate_fp8(residual='bf16') of the Python host; both hosts
 * run the same kernels).  Plans are rebuilt on the next forward.  workspace as for ivx_backbone_fpn_fwd.  ivx_amax_bf16: the max |x|
 * reduction it uses (out: device float, zeroed by the caller). */
int ivx_model_calibrate_fp8(ivx_model *m, const float *img, int32_t BV, int32_t H, int32_t W, float margin, void *workspace,
                            int64_t workspace_bytes, ivx_stream_t stream);
/* The same with the variant chosen: ResNet stages below first_stage (0 .. 4) keep plain bf16 bottlenecks; conv2_bf16 != 0: conv1 / conv2 stay bf16
 * convolutions, conv2 writes e4m3 and only conv3 (e4m3 input and filters) runs on the fp8 matrix cores.  (2, 1): FPN level 0 within ~2.2 % rms of the fp32
 * oracle (BASELINE config 5 "bf16 with fp8 2D-conv MFMA" at a usable accuracy); (0, 0) = ivx_model_calibrate_fp8 (3.6 %). */
int ivx_model_calibrate_fp8_ex(ivx_model *m, const float *img, int32_t BV, int32_t H, int32_t W, float margin, int32_t first_stage, int32_t conv2_bf16,
                               void *workspace, int64_t workspace_bytes, ivx_stream_t stream);
int ivx_amax_bf16(const void *x, int64_t n, float *out, ivx_stream_t stream);

/* Host-only LayoutHead arithmetic in a fixed fp32 order (both hosts of the library use it): angle = limit_period(raw) and layout =
 * (centre, exp(size), yaw) (layout_head.py:52-74); the camera extrinsic of detectors/imvoxelnet.py:164-187 from (pitch, roll). */
int ivx_layout_head_decode(const float *angle_raw /*[2]*/, const float *layout_raw /*[7]*/, float *angle /*[2]*/, float *layout /*[7] or NULL*/);
int ivx_layout_extrinsics(const float *angles /*[2]*/, float *extrinsic4x4);

/* Optional stage timing (measurement only): while enabled, every launch group of the forward calls is bracketed by a pair
 * of HIP events on the caller's stream; the Winograd layers run as their three stages so each is timed.  stage: 0 direct
 * conv, 1 Winograd input transform, 2 grouped GEMM, 3 output transform, 4 unprojection, 5 anchor tail.  flops = FLOPs the
 * launch executes, bytes = algorithmic bytes of a transform / unprojection launch.  Read after synchronising the stream. */
typedef struct ivx_trace_rec {
  int32_t step, stage, is3d;
  float ms;        /* duration of the launch group */
  float start_ms;  /* its start, relative to the first record since tracing was enabled */
  double flops, bytes;
  char name[48];
} ivx_trace_rec;
int ivx_model_trace(ivx_model *m, int32_t level);   /* 0 off | 2 every launch group | 1 coarse: the 3-D neck stages, the
                                                       unprojection and the tail individually, the 2-D trunk as one span (stage 6) |
                                                       3 only the grouped Winograd-domain GEMM launches (stage 2): nine event pairs per KITTI step */
int32_t ivx_model_trace_count(ivx_model *m);
int ivx_model_trace_read(ivx_model *m, int32_t i, ivx_trace_rec *rec);

/* Host-only helpers (no device work, usable without a GPU): the built-in anchor grid for an (H, W) map
 * (Anchor3DRangeGenerator, anchor_3d_generator.py:82-209), and the per-sample camera set-up of
 * detectors/imvoxelnet.py:114-129 / :139 in a fixed fp32 operation order: proj[v] = (K[:3,:3] with rows 0,1 / ratio) @ E_v[:3]
 * as an FMA chain over k; new_origin = origin - n_voxels / 2 * voxel_size. */
int ivx_model_anchors(ivx_model *m, int32_t H, int32_t W, float *anchors_host, int64_t capacity);
int ivx_compute_projection(const float *intrinsic4x4, const float *extrinsics /* [V,4,4] */, int32_t V, double ratio,
                           float *proj /* [V,3,4] */);
int ivx_voxel_new_origin(const float *origin, const int32_t *n_voxels, const float *voxel_size, float *new_origin);
/* Host-only: eval BatchNorm + conv bias as the conv epilogue's affine (all host pointers, n channels; bias may be NULL):
 * scale = gamma / sqrt(var + eps), shift = beta + (bias - mean) * scale, IEEE fp32 in this order. */
int ivx_fold_batchnorm(const float *gamma, const float *beta, const float *mean, const float *var, const float *bias, float eps,
                       int32_t n, float *scale, float *shift);


/* ---------------------------------------------------------------------------------------
 * The export names of SURVEY.md section 8(b)'s minimum set that this header spells differently -- the same entry points under the survey's
 * names (csrc/api_common.cpp forwards; INTEGRATION.md B has the table):
 *   ivx_anchor_head_decode  = ivx_anchor_head_get_bboxes       (Anchor3DHead.get_bboxes_single, dense_heads/anchor3d_head.py)
 *   ivx_fcos3d_head_decode  = ivx_fcos_head_level_candidates   (ImVoxelHead / V2 per-level decode, dense_heads/imvoxel_head*.py)
 *   ivx_nms_rotated_bev     = ivx_nms_bev with rotated = 1     (nms_gpu, ops/iou3d/src/iou3d.cpp:95-147)
 *   ivx_nms_aligned3d       = ivx_aligned_3d_nms               (core/post_processing/box3d_nms.py:91-138)                              */
int ivx_anchor_head_decode(const ivx_anchor_head_desc *d, const float *head_out, const float *anchors, void *workspace, int64_t workspace_bytes,
                           float *out_boxes, float *out_scores, int64_t *out_labels, int32_t *out_count, int64_t *cand_idx, float *cand_boxes,
                           float *cand_scores, ivx_stream_t stream);
int ivx_fcos3d_head_decode(const float *head_out, const uint8_t *valid0, const float *level_vs, const float *level_new_origin, float scale, int32_t B,
                           int32_t nx, int32_t ny, int32_t nz, int32_t CH, int32_t n_classes, int32_t n_reg, int32_t level, int32_t X, int32_t Y,
                           int32_t Z, int32_t nms_pre, void *workspace, int64_t workspace_bytes, float *cand_boxes, float *cand_scores,
                           int32_t *cand_count, ivx_stream_t stream);
int ivx_nms_rotated_bev(const float *boxes_sorted, int32_t n, float thresh, void *workspace, int64_t workspace_bytes, int64_t *keep, int32_t *num_out,
                        ivx_stream_t stream);
int ivx_nms_aligned3d(const float *boxes, const float *scores, const int64_t *classes, int32_t n, float thresh, int64_t *pick, int32_t *num_out,
                      ivx_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* IMVOXEL_H_ */
