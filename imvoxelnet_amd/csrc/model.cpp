// Model-level C-ABI (include/imvoxel.h, section "model handle"): the anchor-head ImVoxelNet forward path as ONE native
// object -- layer sequence, weight packing, Winograd / tile selection, workspace planning and execution live here, so a
// host without Python runs  image -> ResNet-50 -> FPN level 0 -> unprojection -> Kitti/NuScenes neck -> Anchor3DHead ->
// decode + NMS  through ivx_create / ivx_weights_load / ivx_model_forward (tests/c/e2e_small.c is such a host).
//
// Reference structure restated (not its code): mmdet3d/models/detectors/imvoxelnet.py:45-106 (the sequence),
// necks/imvoxelnet.py:94-154,191-230 (the two stack necks), dense_heads/anchor3d_head.py:122-153 (three 1x1 convs, run
// as one fused conv), mmdet 2.10 ResNet(depth=50, style='pytorch') / FPN (level 0 only; parity unpinned).
// Host-only C++ (no kernels): every device operation is one of this library's own C-ABI entry points.
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdarg.h>
#include <string.h>

#include <algorithm>
#include <cstdlib>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "../../include/imvoxel.h"

void ivx_set_error(const char *fmt, ...);

#define M_REQUIRE(cond, ...)        \
  do {                              \
    if (!(cond)) {                  \
      ivx_set_error(__VA_ARGS__);   \
      return IVX_ERR_INVALID_ARG;   \
    }                               \
  } while (0)
#define M_HIP(call, what)                                                        \
  do {                                                                           \
    hipError_t e_ = (call);                                                      \
    if (e_ != hipSuccess) {                                                      \
      ivx_set_error("%s: %s", what, hipGetErrorString(e_));                      \
      return IVX_ERR_HIP;                                                        \
    }                                                                            \
  } while (0)
#define M_TRY(call)                 \
  do {                              \
    int rc_ = (call);               \
    if (rc_ != IVX_OK) return rc_;  \
  } while (0)

namespace {

inline int64_t align256(int64_t v) { return (v + 255) / 256 * 256; }

// IEEE binary16 <-> binary32 on the host, round to nearest even (bit-level: the CPU restatement of the ABI is built with a g++ that has
// no _Float16).  Same bits as a device-side (_Float16) conversion and as torch's .to(float16).
inline uint16_t f32_to_f16_bits(float f) {
  uint32_t x;
  memcpy(&x, &f, 4);
  const uint32_t sign = (x >> 16) & 0x8000u, ax = x & 0x7fffffffu;
  if (ax >= 0x7f800000u) return (uint16_t)(sign | (ax > 0x7f800000u ? 0x7e00u : 0x7c00u));
  if (ax < 0x38800000u) {                       // below 2^-14: a subnormal half = the value in units of 2^-24
    float a;
    memcpy(&a, &ax, 4);
    const float r = a * 16777216.0f;            // exact
    return (uint16_t)(sign | (uint32_t)lrintf(r));   // default rounding mode: to nearest even; 1024 = the smallest normal half
  }
  const uint32_t mant = ax & 0x7fffffu, exp = (ax >> 23) - 127 + 15;
  uint32_t h = (exp << 10) | (mant >> 13);
  const uint32_t rem = mant & 0x1fffu;
  if (rem > 0x1000u || (rem == 0x1000u && (h & 1u))) ++h;   // a carry out of the mantissa bumps the exponent; 65520 and above -> inf
  return (uint16_t)(sign | h);
}
inline uint16_t f32_to_bf16_bits(float f) {        // round to nearest even (torch's .to(bfloat16)); NaN stays NaN
  uint32_t x;
  memcpy(&x, &f, 4);
  if ((x & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((x >> 16) | 0x40u);
  x += 0x7fffu + ((x >> 16) & 1u);
  return (uint16_t)(x >> 16);
}
inline float f16_bits_to_f32(uint16_t h) {
  const uint32_t sign = (uint32_t)(h & 0x8000u) << 16, e = (h >> 10) & 31u, m = h & 0x3ffu;
  float f;
  if (e == 0) {
    f = (float)m * 5.9604644775390625e-08f;     // m * 2^-24
    uint32_t x;
    memcpy(&x, &f, 4);
    x |= sign;
    memcpy(&f, &x, 4);
    return f;
  }
  const uint32_t x = sign | (e == 31 ? 0x7f800000u | (m << 13) : ((e - 15 + 127) << 23) | (m << 13));
  memcpy(&f, &x, 4);
  return f;
}

struct HostTensor {
  std::vector<int64_t> shape;
  std::vector<float> data;
};

// One Conv{2,3}d [+bias] [+eval BN] [+ReLU] of the reference, in its deploy form (imvoxelnet_amd/conv.py FusedConv).
struct ConvLayer {
  std::string name;
  int dims = 3;                       // 2: [Cout,Cin,kh,kw] weights, 3: [Cout,Cin,kd,kh,kw]
  int cin = 0, cout = 0;
  int k[3] = {1, 1, 1}, s[3] = {1, 1, 1}, p[3] = {0, 0, 0};
  bool relu = false;
  bool conv_t = false;                // nn.ConvTranspose3d(k2, s2) weights [Cin,Cout,2,2,2] run as a 1x1x1 GEMM with 8*Cout columns (out_mode 1)
  bool dcn_cols = false;              // DCNv2 main conv: weights [Cout,C,3,3] run as a 1x1 conv over the 9*C columns of ivx_dcn_im2col_fwd
  int cout_zero = 0;                  // output channels appended with zero filters / zero bias (conv_offset: 27 -> 28, so that Cout % 4 == 0
                                      // and the layer can read pair tensors; the column kernel takes the map's channel count as its row stride)
  bool linear = false;                // nn.Linear weights [Cout,Cin] run as a 1x1 conv on a [B,1,1,1,Cin] tensor (LayoutHead MLPs)
  std::vector<std::string> w_keys;    // > 1: filter banks concatenated along Cout (the fused head conv)
  std::vector<std::string> b_keys;    // parallel to w_keys; "" = no bias
  std::string bn;                     // BatchNorm prefix or ""
  // deploy form
  int cin_pad = 0, layout = 0;
  bool wino_cand = false, wino2d = false, identity_epilogue = true;
  float *w = nullptr, *w0 = nullptr, *scale = nullptr, *shift = nullptr;
  std::map<int, float *> u;           // tile * 8 + operand type -> transformed filters
  // fp16-pair form (cfg.trunk_operands = IVX_F16_PAIR; ivx_conv_fwd_pio): pair filters, scale / s_w, and the terms of the output bound
  bool stem_s2d = false;              // bf16 storage: the 7x7 stride-2 stem as a 4x4 stride-1 conv over 2x2 space-to-depth blocks (weights re-indexed at pack time)
  bool out_f32 = false;               // bf16 storage: this layer writes fp32 (the head convs: the tails decode fp32 scores / deltas)
  // e4m3 interior of the bottlenecks (ivx_model_calibrate_fp8 on a bf16 handle): 1 conv1 (bf16 in, e4m3 out), 2 conv2 (e4m3 in / filters /
  // out), 3 conv3 (e4m3 in / filters, bf16 out + bf16 shortcut); 0: not part of it
  int fp8_role = 0;
  int stage = -1;                     // ResNet stage (0 .. 3) of a bottleneck conv, else -1
  int fp8_eff = 0;                    // the role in effect after ivx_model_calibrate_fp8[_ex] (variant / first stage applied to fp8_role)
  std::vector<float> w_tap, scale_h, shift_h;   // roles 2 / 3: the fp32 filters (tap-major) kept for the quantisation; the epilogue vectors of every role
  float *wq = nullptr, *scale_q = nullptr, *shift_q = nullptr;
  int layout_q = 0;
  double out_scale_q = 1.0;
  bool pair_ok = false;
  float *wpair = nullptr, *scale_p = nullptr;
  // split-operand (IVX_BF16_PAIR) form of the 3x3x3 neck layers the Winograd form does not take (strided; fewer than 2000 positions): filters as
  // conv.py pack_pair_weights(layout 1); the layer's input goes through ivx_bf16_pair_split into the step's workspace (conv.py FusedConv._pair)
  bool split_cand = false;
  float *wsplit = nullptr;
  float wbound = 0.f, sbound = 0.f;
  // the first block of stage 1 in one launch (ivx_bottleneck_proj_fwd_pio): on conv3, the layer index of the block's shortcut conv and the joint filter
  // bank of the two (ivx_bottleneck_proj_pack: BN scales folded into the filters, scale_proj = 1 / s_w, shift_proj = shift3 + shiftd) with its bound terms
  int proj_peer = -1;
  bool proj_keep = false;             // pack_layer keeps w_tap / scale_h / shift_h of this layer for the joint bank
  float *wproj = nullptr, *scale_proj = nullptr, *shift_proj = nullptr;
  float proj_wb3 = 0.f, proj_sb = 0.f, proj_wbd = 0.f;
  float *wstem = nullptr;             // the 7x7 stem in the pair chain: fragment-ordered pair filters of the one-launch stem (csrc/stem.hip); scale_p with them
};

enum StepKind { ST_IMG2CL, ST_IMG_S2D, ST_CONV, ST_MAXPOOL, ST_LIFT, ST_TAIL, ST_UPSAMPLE, ST_DCN_COL, ST_AVGPOOL, ST_LAYOUT, ST_FCOS, ST_INDOOR_TAIL };

struct Step {
  StepKind kind;
  int layer = -1;
  int in = -1, res = -1, out = -1, out2 = -1;   // tensor ids (out2: the valid mask of the lift)
  int res_mode = 0;
  int res_after_act = 0;     // the residual is added after the ReLU (skip adds of the U-shaped necks)
  float post_scale = 1.0f;   // Atlas decoder: (x + y) / 2
  int aux = 0;               // ST_DCN_COL: stride of the deformable conv; ST_FCOS: the level
};

struct TInfo {
  int B = 0, D = 0, H = 0, W = 0, C = 0;
  int64_t bytes = 0, off = -1;
  int first = -1, last = -1;
  bool raw = false;          // a byte buffer (candidate lists): `bytes` is set explicitly
  int esz = 4;               // bytes per element: 4, or 2 for a bf16 tensor (cfg.storage = IVX_BF16)
  int fmt = 0;               // 0: fp32; IVX_F16_PAIR: fp16 (hi, lo) pairs with a device-side scale (same bytes)
  int64_t slot = -1;         // arena offset of the tensor's scalar block: IVX_AMAX_SLOTS words of max |tensor|, then its scale; -1: none
  int64_t elems() const { return (int64_t)B * D * H * W * C; }
};

struct PlanStep {
  ivx_conv_desc d;        // CONV: the descriptor handed to the conv entry point (Winograd: the transformed-axes view)
  int tile = 0;           // 0: direct kernel, else F(tile x tile, 3x3)
  int64_t ws = 0;
  // max |tensor| for the fp16-pair operand scale of a Winograd layer whose input is another Winograd layer's output: the producer's
  // output transform leaves one maximum per workgroup (arena offset amax_out, amax_n entries), the consumer's input stage reduces
  // those instead of reading the whole tensor again (amax_in / amax_in_n); -1: the input stage reduces the tensor itself
  int64_t amax_out = -1, amax_in = -1;
  int amax_n = 0, amax_in_n = 0;
  int pio = 0;            // CONV: the fp16-pair form (ivx_conv_fwd_pio); MAXPOOL: pair output (aux = the stem's layer for the bound)
  int64_t split = 0;      // CONV: > 0 = the split-operand form: bytes of the (hi, lo) bf16 copy of the input at the start of the workspace (d.in_dtype = IVX_BF16_PAIR)
  int bound_layer = -1;
  // identity bottleneck in one launch (ivx_bottleneck_fwd_pio, csrc/bottleneck.hip): fuse 1 = conv1's step, which runs the whole block and
  // writes conv3's output tensor (fuse_out); fuse 2 = conv2's / conv3's step, covered by it
  int fuse = 0, fuse_out = -1;
  // the shortcut conv of a block with a downsample runs on the handle's SIDE stream next to conv1 / conv2 (independent: both read the block's
  // input): side = site index + 1 on the shortcut conv's step, join = site index + 1 on the step that reads its output as the residual
  int side = 0, join = 0;
};

struct Plan {
  std::vector<TInfo> t;
  std::vector<PlanStep> ps;
  int64_t arena = 0, ws_off = 0, ws_bytes = 0, tail_ws = 0, total = 0;
  ivx_anchor_head_desc tail;
  // camera block at the start of the arena (ivx_model_detect: the host-built per-batch set-up is uploaded there)
  int64_t cam_bytes = 0, cam_proj = 0, cam_origin = 0, cam_crop = 0, cam_lvl_vs[3] = {0, 0, 0}, cam_lvl_no[3] = {0, 0, 0};
  int n_views = 1;
  ivx_indoor_tail_desc itail;          // indoor families with a head
  int max_det = 0;                     // rows per sample of the detection outputs
  int64_t scal_off = 0, scal_bytes = 0;   // scalar blocks of the pair-chained tensors (zeroed at the start of every forward)
  int64_t ws2_off = 0;                    // second split-K workspace (launches on the side stream), 0: none
  int n_sides = 0;
};

}  // namespace

struct ivx_model {
  ivx_model_cfg cfg;
  std::vector<ConvLayer> layers;
  std::vector<Step> steps;
  int n_tensors = 0;
  // step ranges [begin, end) and boundary tensors
  int trunk0 = 0, trunk1 = 0, lift_step = -1, neck0 = 0, neck1 = 0, head_step = -1, tail_step = -1;
  int t_img = -1, t_fpn0 = -1, t_volume = -1, t_valid = -1, t_neck = -1, t_head = -1;
  std::vector<int> t_levels;           // indoor necks: the output levels, finest first
  // anchor-free head (cfg.head_type): per level the fused head conv output and the candidate buffers; the LayoutHead's tensors
  int head0 = -1;                      // first step after the neck (head convs, candidates, cross-level tail)
  std::vector<int> t_headout, t_cb, t_cs, t_cc;
  float head_scales[3] = {1.f, 1.f, 1.f};   // bbox_head.scales.{l}.scale (mmcv Scale)
  int t_c5 = -1, t_angle = -1, t_layout = -1, layout_step = -1;
  std::vector<std::vector<char>> cam_ring;   // host staging of the camera block (ring: a forward may still be reading the previous one)
  size_t cam_next = 0;
  std::map<std::string, HostTensor> weights;
  bool finalized = false;
  std::vector<float> anchors_host;     // [H*W*A, 7] for the (H, W) below; regenerated when the grid changes
  int anchors_h = 0, anchors_w = 0;
  float *anchors_dev = nullptr;
  size_t anchors_cap = 0;              // floats anchors_dev holds
  std::vector<float> anchors_given;    // supplied through ivx_weights_load("anchors", ...)
  std::map<std::string, std::unique_ptr<Plan>> plans;
  std::vector<void *> owned;           // device allocations (weights, filters, anchors)
  std::vector<float> pack_a, pack_b;   // host staging of ivx_weights_finalize (released there)
  bool fp8_on = false;                 // the bottleneck interiors are e4m3 (ivx_model_calibrate_fp8)
  float *calib_dev = nullptr;          // during the calibration pass: one max |output| per layer (device)
  // optional stage timing (ivx_model_trace): one record per launch group, events recorded on the caller's stream
  bool trace_on = false;
  bool trace_skip = false;             // the open trace_begin recorded nothing (level 3, a stage other than the GEMM)
  int trace_level = 2;                 // 2: every launch group; 1: the 3-D neck stages, the unprojection and the tail individually,
                                       //    the 2-D trunk (image layout change .. FPN level 0) as ONE span (stage 6)
  struct TraceRec { int step, stage, is3d; double flops, bytes; hipEvent_t e0, e1; std::string name; };
  std::vector<TraceRec> trace;
  std::vector<hipEvent_t> event_pool;
  size_t events_used = 0;
  hipStream_t side = nullptr;          // side stream (created on first use) + fork / join events per site
  hipEvent_t ev_fork = nullptr;
  std::vector<hipEvent_t> ev_join;
};

namespace {

// ---------------------------------------------------------------------------------------------- graph construction
int new_tensor(ivx_model *m) { return m->n_tensors++; }

int add_conv(ivx_model *m, ConvLayer L, int in, int res = -1, int res_mode = 0) {
  m->layers.push_back(std::move(L));
  Step s;
  s.kind = ST_CONV;
  s.layer = (int)m->layers.size() - 1;
  s.in = in;
  s.res = res;
  s.res_mode = res >= 0 ? (res_mode ? res_mode : 1) : 0;
  s.out = new_tensor(m);
  m->steps.push_back(s);
  return s.out;
}

ConvLayer conv2d(const std::string &name, int cin, int cout, int k, int stride, int pad, bool relu, const std::string &w,
                 const std::string &b, const std::string &bn) {
  ConvLayer L;
  L.name = name; L.dims = 2; L.cin = cin; L.cout = cout;
  L.k[0] = 1; L.k[1] = k; L.k[2] = k;
  L.s[0] = 1; L.s[1] = stride; L.s[2] = stride;
  L.p[0] = 0; L.p[1] = pad; L.p[2] = pad;
  L.relu = relu; L.w_keys = {w}; L.b_keys = {b}; L.bn = bn;
  return L;
}

ConvLayer conv3d(const std::string &name, int cin, int cout, const int s[3], const int p[3], bool relu, const std::string &w,
                 const std::string &b, const std::string &bn) {
  ConvLayer L;
  L.name = name; L.dims = 3; L.cin = cin; L.cout = cout;
  for (int a = 0; a < 3; ++a) { L.k[a] = 3; L.s[a] = s[a]; L.p[a] = p[a]; }
  L.relu = relu; L.w_keys = {w}; L.b_keys = {b}; L.bn = bn;
  return L;
}

// ResNet-50 (style 'pytorch': stride on the 3x3) + FPN level 0 (lateral 1x1 -> nearest x2 top-down add -> 3x3).
void build_trunk(ivx_model *m) {
  m->trunk0 = (int)m->steps.size();
  m->t_img = new_tensor(m);
  const bool bf16 = m->cfg.storage == IVX_BF16;
  Step a; a.kind = bf16 ? ST_IMG_S2D : ST_IMG2CL; a.in = m->t_img; a.out = new_tensor(m);
  m->steps.push_back(a);
  ConvLayer stem = bf16 ? conv2d("backbone.conv1", 16, 64, 4, 1, 1, true, "backbone.conv1.weight", "", "backbone.bn1")
                        : conv2d("backbone.conv1", 3, 64, 7, 2, 3, true, "backbone.conv1.weight", "", "backbone.bn1");
  stem.stem_s2d = bf16;
  int x = add_conv(m, stem, a.out);
  Step mp; mp.kind = ST_MAXPOOL; mp.in = x; mp.out = new_tensor(m);
  m->steps.push_back(mp);
  x = mp.out;
  const int blocks[4] = {3, 4, 6, 3};
  int cin = 64, feats[4];
  for (int i = 0; i < 4; ++i) {
    const int planes = 64 << i;
    for (int j = 0; j < blocks[i]; ++j) {
      const std::string pre = "backbone.layer" + std::to_string(i + 1) + "." + std::to_string(j) + ".";
      const int stride = (j == 0 && i > 0) ? 2 : 1;
      int idt = x;
      int ds_layer = -1;
      if (j == 0) {
        idt = add_conv(m, conv2d(pre + "downsample", cin, planes * 4, 1, stride, 0, false, pre + "downsample.0.weight", "", pre + "downsample.1"), x);
        ds_layer = (int)m->layers.size() - 1;
      }
      // candidate of the one-launch projection block (make_plan decides per shape): stride 1, Cin = planes = 64, no DCN, fp32 storage + pair chain
      const bool proj_cand = j == 0 && stride == 1 && cin == 64 && planes == 64 && !bf16 && !m->cfg.dcn_stages[i] && m->cfg.trunk_operands == IVX_F16_PAIR;
      if (proj_cand) m->layers[ds_layer].proj_keep = true;
      ConvLayer c1 = conv2d(pre + "conv1", cin, planes, 1, 1, 0, true, pre + "conv1.weight", "", pre + "bn1");
      c1.fp8_role = (bf16 && !m->cfg.dcn_stages[i]) ? 1 : 0;
      c1.stage = i;
      int y = add_conv(m, c1, x);
      if (m->cfg.dcn_stages[i]) {
        // ModulatedDeformConv2dPack (mmcv; configs/imvoxelnet/imvoxelnet_nuscenes.py:13-14): conv_offset (3x3, bias) -> 27 raw channels,
        // ivx_dcn_im2col_fwd builds the modulated, bilinearly sampled columns, the main conv is a 1x1 over K = 9 * C
        ConvLayer co = conv2d(pre + "conv2.conv_offset", planes, 28, 3, stride, 1, false, pre + "conv2.conv_offset.weight",
                              pre + "conv2.conv_offset.bias", "");
        co.cout_zero = 1;                                   // 27 raw channels + one zero channel
        const int off = add_conv(m, co, y);
        Step dc; dc.kind = ST_DCN_COL; dc.in = y; dc.res = off; dc.out = new_tensor(m); dc.aux = stride;
        m->steps.push_back(dc);
        ConvLayer c2 = conv2d(pre + "conv2", 9 * planes, planes, 1, 1, 0, true, pre + "conv2.weight", "", pre + "bn2");
        c2.dcn_cols = true;
        y = add_conv(m, c2, dc.out);
      } else {
        ConvLayer c2 = conv2d(pre + "conv2", planes, planes, 3, stride, 1, true, pre + "conv2.weight", "", pre + "bn2");
        c2.fp8_role = bf16 ? 2 : 0;
        c2.stage = i;
        y = add_conv(m, c2, y);
      }
      ConvLayer c3 = conv2d(pre + "conv3", planes, planes * 4, 1, 1, 0, true, pre + "conv3.weight", "", pre + "bn3");
      c3.fp8_role = (bf16 && !m->cfg.dcn_stages[i]) ? 3 : 0;
      c3.stage = i;
      if (proj_cand) { c3.proj_keep = true; c3.proj_peer = ds_layer; }
      x = add_conv(m, c3, y, idt, 1);
      cin = planes * 4;
    }
    feats[i] = x;
  }
  if (m->cfg.layout_head) {
    // LayoutHead (dense_heads/layout_head.py:8-50): x.mean(dim=(2,3)) of C5, two MLPs Linear-ReLU-(Dropout)-Linear-ReLU-(Dropout)-Linear
    m->t_c5 = feats[3];
    Step ap; ap.kind = ST_AVGPOOL; ap.in = feats[3]; ap.out = new_tensor(m);
    m->steps.push_back(ap);
    const int ls = m->cfg.layout_linear_size;
    auto mlp = [&](const std::string &name, int n_out) {
      int t = ap.out, ci = 2048;
      const int idx[3] = {0, 3, 6}, co[3] = {ls, ls, n_out};
      for (int q = 0; q < 3; ++q) {
        const std::string pre = "head_2d." + name + "." + std::to_string(idx[q]) + ".";
        ConvLayer L = conv2d(pre, ci, co[q], 1, 1, 0, q < 2, pre + "weight", pre + "bias", "");
        L.linear = true;
        t = add_conv(m, L, t);
        ci = co[q];
      }
      return t;
    };
    m->t_angle = mlp("angle_mlp", 2);
    m->t_layout = mlp("layout_mlp", 7);
    Step ly; ly.kind = ST_LAYOUT; ly.in = m->t_angle; ly.res = m->t_layout;
    m->layout_step = (int)m->steps.size();
    m->steps.push_back(ly);
  }
  const int cf = m->cfg.fpn_channels, cins[4] = {256, 512, 1024, 2048};
  int lat = -1;
  for (int i = 3; i >= 0; --i) {
    const std::string pre = "neck.lateral_convs." + std::to_string(i) + ".conv.";
    lat = add_conv(m, conv2d("neck.lateral" + std::to_string(i), cins[i], cf, 1, 1, 0, false, pre + "weight", pre + "bias", ""), feats[i],
                   i == 3 ? -1 : lat, 2);
  }
  m->t_fpn0 = add_conv(m, conv2d("neck.fpn_conv0", cf, cf, 3, 1, 1, false, "neck.fpn_convs.0.conv.weight", "neck.fpn_convs.0.conv.bias", ""), lat);
  m->trunk1 = (int)m->steps.size();
}

// KittiImVoxelNeck / NuScenesImVoxelNeck: block, conv, block, conv, block, conv (necks/imvoxelnet.py:99-113, 131-145).
void build_neck(ivx_model *m, int t_in) {
  m->neck0 = (int)m->steps.size();
  const int c = m->cfg.fpn_channels;
  const int one[3] = {1, 1, 1};
  int strides[3][3], pads[3][3];
  if (m->cfg.neck_type == IVX_NECK_KITTI) {
    const int s_[3][3] = {{1, 1, 2}, {1, 1, 2}, {1, 1, 1}}, p_[3][3] = {{1, 1, 1}, {1, 1, 1}, {0, 0, 0}};
    memcpy(strides, s_, sizeof(s_)); memcpy(pads, p_, sizeof(p_));
  } else {
    const int s_[3][3] = {{2, 2, 2}, {1, 1, 2}, {1, 1, 1}}, p_[3][3] = {{1, 1, 1}, {1, 1, 1}, {1, 1, 0}};
    memcpy(strides, s_, sizeof(s_)); memcpy(pads, p_, sizeof(p_));
  }
  const int chans[4] = {c, c * 2, c * 4, m->cfg.neck_out_channels};
  int x = t_in;
  for (int g = 0; g < 3; ++g) {
    const std::string b = "neck_3d.model." + std::to_string(2 * g) + ".", d = "neck_3d.model." + std::to_string(2 * g + 1) + ".";
    const int ch = chans[g];
    int y = add_conv(m, conv3d(b + "conv1", ch, ch, one, one, true, b + "conv1.weight", "", b + "bn1"), x);
    x = add_conv(m, conv3d(b + "conv2", ch, ch, one, one, true, b + "conv2.weight", "", b + "bn2"), y, x, 1);
    x = add_conv(m, conv3d(d + "0", ch, chans[g + 1], strides[g], pads[g], true, d + "0.weight", d + "0.bias", d + "1"), x);
  }
  m->t_neck = x;
  m->neck1 = (int)m->steps.size();
}


ConvLayer conv3d_k(const std::string &name, int cin, int cout, int k, int stride, int pad, bool relu, const std::string &w, const std::string &b,
                   const std::string &bn) {
  const int s3[3] = {stride, stride, stride}, p3[3] = {pad, pad, pad};
  ConvLayer L = conv3d(name, cin, cout, s3, p3, relu, w, b, bn);
  for (int a = 0; a < 3; ++a) L.k[a] = k;
  return L;
}

// BasicBlock3d (necks/imvoxelnet.py:191-230): conv3-BN-ReLU-conv3-BN-(+x)-ReLU
int add_block3d(ivx_model *m, const std::string &pre, int ch, int x, const char *n1 = "bn1", const char *n2 = "bn2") {
  int y = add_conv(m, conv3d_k(pre + "conv1", ch, ch, 3, 1, 1, true, pre + "conv1.weight", "", pre + n1), x);
  return add_conv(m, conv3d_k(pre + "conv2", ch, ch, 3, 1, 1, true, pre + "conv2.weight", "", pre + n2), y, x, 1);
}

int add_upsample(ivx_model *m, int x) {
  Step u; u.kind = ST_UPSAMPLE; u.in = x; u.out = new_tensor(m);
  m->steps.push_back(u);
  return u.out;
}

// FastIndoorImVoxelNeck (necks/imvoxelnet.py:8-67): BasicBlock3dV2 stacks (stride 2 from level 1 on, 1x1x1-BN shortcut when
// strided), ConvTranspose(k2,s2)-BN-ReLU-conv3-BN-ReLU up-blocks with skip ADD after the activation, conv3-BN-ReLU out-blocks.
void build_neck_fast(ivx_model *m, int t_in) {
  m->neck0 = (int)m->steps.size();
  int c = m->cfg.fpn_channels, x = t_in;
  std::vector<int> down, chans;
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < m->cfg.fast_n_blocks[i]; ++j) {
      const std::string pre = "neck_3d.down_layer_" + std::to_string(i) + "." + std::to_string(j) + ".";
      const int stride = (j == 0 && i > 0) ? 2 : 1, co = stride == 2 ? 2 * c : c;
      int idt = x;
      if (stride != 1) idt = add_conv(m, conv3d_k(pre + "downsample", c, co, 1, 2, 0, false, pre + "downsample.0.weight", "", pre + "downsample.1"), x);
      int y = add_conv(m, conv3d_k(pre + "conv1", c, co, 3, stride, 1, true, pre + "conv1.weight", "", pre + "norm1"), x);
      x = add_conv(m, conv3d_k(pre + "conv2", co, co, 3, 1, 1, true, pre + "conv2.weight", "", pre + "norm2"), y, idt, 1);
      c = co;
    }
    down.push_back(x);
    chans.push_back(c);
  }
  std::vector<int> outs(3, -1);
  for (int i = 2; i >= 0; --i) {
    if (i < 2) {
      const std::string pre = "neck_3d.up_block_" + std::to_string(i + 1) + ".";
      ConvLayer t = conv3d_k(pre + "0", chans[i + 1], 8 * chans[i], 1, 1, 0, true, pre + "0.weight", "", pre + "1");
      t.conv_t = true;
      x = add_conv(m, t, x);
      x = add_conv(m, conv3d_k(pre + "3", chans[i], chans[i], 3, 1, 1, true, pre + "3.weight", "", pre + "4"), x, down[i], 1);
      m->steps.back().res_after_act = 1;            // relu(bn(conv(.))) + skip   (:30-31)
    }
    const std::string po = "neck_3d.out_block_" + std::to_string(i) + ".";
    outs[i] = add_conv(m, conv3d_k(po + "0", chans[i], m->cfg.neck_out_channels, 3, 1, 1, true, po + "0.weight", "", po + "1"), x);
  }
  m->t_levels = outs;
  m->neck1 = (int)m->steps.size();
}

// ImVoxelNeck = Atlas EncoderDecoder + per-level conv3(bias)-BN-ReLU (necks/imvoxelnet.py:70-91, 297-372; cond_proj False)
void build_neck_unet(ivx_model *m, int t_in) {
  m->neck0 = (int)m->steps.size();
  const int *ch = m->cfg.unet_channels, *dl = m->cfg.unet_down_layers, *ul = m->cfg.unet_up_layers;
  const int n = ch[3] > 0 ? 4 : 3;                   // encoder scales (the reference configs use 4; 3 = a two-level decoder)
  int x = t_in;
  std::vector<int> xs;
  for (int i = 0; i < n; ++i) {
    const std::string pre = "neck_3d.model.layers_down." + std::to_string(i) + ".";
    int k0 = 0;
    if (i > 0) {
      x = add_conv(m, conv3d_k(pre + "0", ch[i - 1], ch[i], 3, 2, 1, true, pre + "0.weight", "", pre + "1"), x);
      k0 = 4;                                        // Sequential: conv, norm, dropout, relu precede the blocks
    }
    for (int j = 0; j < dl[i]; ++j) x = add_block3d(m, pre + std::to_string(k0 + j) + ".", ch[i], x);
    xs.push_back(x);
  }
  std::vector<int> out;                              // coarse -> fine
  for (int i = 0; i < n - 1; ++i) {
    const int c_in = ch[n - 1 - i], c_out = ch[n - 2 - i];
    const std::string pu = "neck_3d.model.layers_up_conv." + std::to_string(i) + ".", pp = "neck_3d.model.proj." + std::to_string(i) + ".";
    x = add_upsample(m, x);
    x = add_conv(m, conv3d_k(pu, c_in, c_out, 1, 1, 0, false, pu + "weight", "", ""), x);
    x = add_conv(m, conv3d_k(pp + "conv", c_out, c_out, 1, 1, 0, true, pp + "conv.weight", "", pp + "norm"), xs[n - 2 - i], x, 1);
    m->steps.back().res_after_act = 1;               // (x + relu(bn(conv(skip)))) / 2   (:366-367)
    m->steps.back().post_scale = 0.5f;
    for (int j = 0; j < ul[i]; ++j) x = add_block3d(m, "neck_3d.model.layers_up_res." + std::to_string(i) + "." + std::to_string(j) + ".", c_out, x);
    out.push_back(x);
  }
  std::vector<int> lv(n - 1, -1);
  for (int l = 0; l < n - 1; ++l) {                  // level l (finest first) = out[n - 2 - l]
    const std::string pb = "neck_3d.conv_blocks." + std::to_string(l) + ".";
    lv[l] = add_conv(m, conv3d_k(pb + "0", ch[l], m->cfg.neck_out_channels, 3, 1, 1, true, pb + "0.weight", pb + "0.bias", pb + "1"), out[n - 2 - l]);
  }
  m->t_levels = lv;
  m->neck1 = (int)m->steps.size();
}

void build_graph(ivx_model *m) {
  int t_feat;
  if (m->cfg.with_trunk) {
    build_trunk(m);
    t_feat = m->t_fpn0;
  } else {
    m->t_fpn0 = t_feat = new_tensor(m);
  }
  Step lift; lift.kind = ST_LIFT; lift.in = t_feat; lift.out = m->t_volume = new_tensor(m); lift.out2 = m->t_valid = new_tensor(m);
  m->lift_step = (int)m->steps.size();
  m->steps.push_back(lift);
  if (m->cfg.neck_type == IVX_NECK_FAST || m->cfg.neck_type == IVX_NECK_UNET) {   // indoor families: the handle ends at the neck levels
    if (m->cfg.neck_type == IVX_NECK_FAST) build_neck_fast(m, m->t_volume);
    else build_neck_unet(m, m->t_volume);
    m->head0 = (int)m->steps.size();
    if (m->cfg.head_type != IVX_HEAD_NONE) {
      // ImVoxelHeadV2 / ImVoxelHead with n_convs = 0 (dense_heads/imvoxel_head_v2.py:64-80, 305-313): centerness | reg | cls 3x3x3 convs
      // as ONE conv shared by the levels, then per level the candidate kernel and ONE cross-level tail
      const int R = m->cfg.head_type == IVX_HEAD_SCANNET ? 6 : 7, nc = m->cfg.head_classes, oc = m->cfg.neck_out_channels;
      ConvLayer h = conv3d_k("bbox_head", oc, 1 + R + nc, 3, 1, 1, false, "", "", "");
      h.out_f32 = true;
      h.w_keys = {"bbox_head.centerness_conv.weight", "bbox_head.reg_conv.weight", "bbox_head.cls_conv.weight"};
      h.b_keys = {"", "", "bbox_head.cls_conv.bias"};
      m->layers.push_back(h);
      const int hl = (int)m->layers.size() - 1;
      for (size_t l = 0; l < m->t_levels.size(); ++l) {
        Step c; c.kind = ST_CONV; c.layer = hl; c.in = m->t_levels[l]; c.out = new_tensor(m);
        m->steps.push_back(c);
        m->t_headout.push_back(c.out);
      }
      for (size_t l = 0; l < m->t_levels.size(); ++l) {
        Step f; f.kind = ST_FCOS; f.in = m->t_headout[l]; f.res = m->t_valid; f.aux = (int)l;
        f.out = new_tensor(m); f.out2 = new_tensor(m);
        m->t_cb.push_back(f.out); m->t_cs.push_back(f.out2); m->t_cc.push_back(new_tensor(m));
        m->steps.push_back(f);
      }
      Step t; t.kind = ST_INDOOR_TAIL; t.in = m->t_cb[0];
      m->tail_step = (int)m->steps.size();
      m->steps.push_back(t);
    }
    return;
  }
  build_neck(m, m->t_volume);
  // Anchor3DHead: conv_cls | conv_reg | conv_dir_cls as one 1x1 conv (anchor3d_head.py:122-130,138-153)
  const int A = m->cfg.n_sizes * m->cfg.n_rotations, oc = m->cfg.neck_out_channels;
  ConvLayer h = conv2d("bbox_head", oc, A * (m->cfg.num_classes + 7 + 2), 1, 1, 0, false, "", "", "");
  h.out_f32 = true;
  h.w_keys = {"bbox_head.conv_cls.weight", "bbox_head.conv_reg.weight", "bbox_head.conv_dir_cls.weight"};
  h.b_keys = {"bbox_head.conv_cls.bias", "bbox_head.conv_reg.bias", "bbox_head.conv_dir_cls.bias"};
  m->head_step = (int)m->steps.size();
  m->t_head = add_conv(m, h, m->t_neck);
  Step tail; tail.kind = ST_TAIL; tail.in = m->t_head;
  m->tail_step = (int)m->steps.size();
  m->steps.push_back(tail);
}

// ---------------------------------------------------------------------------------------------- weights
int dev_upload(ivx_model *m, const std::vector<float> &h, float **out, hipStream_t st) {
  void *d = nullptr;
  M_HIP(hipMalloc(&d, std::max<size_t>(h.size(), 1) * sizeof(float)), "hipMalloc (weights)");
  m->owned.push_back(d);
  M_HIP(hipMemcpyAsync(d, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice, st), "hipMemcpyAsync (weights)");
  *out = (float *)d;
  return IVX_OK;
}

// ... from a staging buffer that is about to be reused: the copy has completed when this returns
int dev_upload_sync(ivx_model *m, const float *h, size_t n, float **out, hipStream_t st) {
  void *d = nullptr;
  M_HIP(hipMalloc(&d, std::max<size_t>(n, 1) * sizeof(float)), "hipMalloc (weights)");
  m->owned.push_back(d);
  M_HIP(hipMemcpyAsync(d, h, n * sizeof(float), hipMemcpyHostToDevice, st), "hipMemcpyAsync (weights)");
  M_HIP(hipStreamSynchronize(st), "hipStreamSynchronize (weights)");
  *out = (float *)d;
  return IVX_OK;
}

// The same for a buffer that is REPLACED (a second ivx_model_calibrate_fp8): the previous device copy is released first -- after a
// stream synchronisation, a forward in flight may still read it -- instead of piling up in m->owned until ivx_destroy.
int dev_reupload_sync(ivx_model *m, const float *h, size_t n, float **out, hipStream_t st) {
  if (*out) {
    M_HIP(hipStreamSynchronize(st), "hipStreamSynchronize (replacing weights)");
    auto it = std::find(m->owned.begin(), m->owned.end(), (void *)*out);
    if (it != m->owned.end()) m->owned.erase(it);
    (void)hipFree(*out);
    *out = nullptr;
  }
  return dev_upload_sync(m, h, n, out, st);
}

const HostTensor *find_w(const ivx_model *m, const std::string &key) {
  auto it = m->weights.find(key);
  return it == m->weights.end() ? nullptr : &it->second;
}

// Deploy form of one layer: imvoxelnet_amd/conv.py FusedConv.__init__ in C++ (same fp32 operations in the same order).
int pack_layer(ivx_model *m, ConvLayer &L, hipStream_t st, std::string *missing) {
  const int kd = L.k[0], kh = L.k[1], kw = L.k[2], taps = kd * kh * kw;
  const bool bf16 = m->cfg.storage == IVX_BF16;      // bf16 storage (conv.py FusedConv with dtype bfloat16): 8-channel chunks, 64-channel K chunks
  L.cin_pad = bf16 ? (L.cin + 7) / 8 * 8 : (L.cin + 3) / 4 * 4;
  L.layout = (L.cin_pad % (bf16 ? 64 : 32) == 0) ? 1 : 0;
  // staging buffers live in the handle and are reused by every layer (fresh 100 MB vectors per layer cost seconds of first-touch
  // page faults in a sandboxed container); only channel padding needs the zero fill
  std::vector<float> &wp = m->pack_a, &wc = m->pack_b;
  const size_t n_w = (size_t)L.cout * taps * L.cin_pad;
  if (wp.size() < n_w) wp.resize(n_w);
  if (L.cin_pad != L.cin) std::fill(wp.begin(), wp.begin() + n_w, 0.f);
  std::vector<float> bias(L.cout, 0.f);
  bool has_bias = false;
  int co0 = 0;
  for (size_t q = 0; q < L.w_keys.size(); ++q) {
    const HostTensor *w = find_w(m, L.w_keys[q]);
    if (!w) { *missing += L.w_keys[q] + " "; return IVX_OK; }
    const size_t nd = w->shape.size();
    if (L.conv_t) {   // [Cin, Cr, 2,2,2] -> columns n = ((a*2+e)*2+f)*Cr + co of a 1x1x1 GEMM (conv.py FusedConvTranspose2x)
      const int cr = L.cout / 8;
      M_REQUIRE(nd == 5 && w->shape[0] == L.cin && w->shape[1] == cr && w->shape[2] == 2 && w->shape[3] == 2 && w->shape[4] == 2,
                "ivx_weights_finalize: %s must be [Cin, Cout, 2, 2, 2]", L.w_keys[q].c_str());
      for (int ci = 0; ci < L.cin; ++ci)
        for (int co = 0; co < cr; ++co)
          for (int tap = 0; tap < 8; ++tap) wp[(size_t)(tap * cr + co) * L.cin_pad + ci] = w->data[((size_t)ci * cr + co) * 8 + tap];
      co0 = L.cout;
      continue;
    }
    if (L.dcn_cols) {   // [Cout, C, 3, 3] -> the 1x1 filter over the (tap, c) columns of ivx_dcn_im2col_fwd: k = tap * C + c
      const int C9 = L.cin / 9;
      M_REQUIRE(nd == 4 && w->shape[0] == L.cout && w->shape[1] == C9 && w->shape[2] == 3 && w->shape[3] == 3,
                "ivx_weights_finalize: %s must be [Cout, C, 3, 3]", L.w_keys[q].c_str());
      for (int co = 0; co < L.cout; ++co)
        for (int c = 0; c < C9; ++c)
          for (int t = 0; t < 9; ++t) wp[(size_t)co * L.cin_pad + (size_t)t * C9 + c] = w->data[((size_t)co * C9 + c) * 9 + t];
      co0 = L.cout;
      continue;
    }
    if (L.stem_s2d) {   // [64, 3, 7, 7] -> the 4x4 conv over 2x2 space-to-depth blocks (backbones.stem_s2d_weights): channel (a*2+e)*3 + c of
                        // tap (th, tw) is w[co][c][2*th + a][2*tw + e] (the eighth row / column and channels 12..15 are zero)
      M_REQUIRE(nd == 4 && w->shape[1] == 3 && w->shape[2] == 7 && w->shape[3] == 7 && (int)w->shape[0] == L.cout,
                "ivx_weights_finalize: bf16 storage is built for the 3-channel 7x7 stem (%s)", L.w_keys[q].c_str());
      std::fill(wp.begin(), wp.begin() + n_w, 0.f);
      for (int co = 0; co < L.cout; ++co)
        for (int c = 0; c < 3; ++c)
          for (int r = 0; r < 7; ++r)
            for (int cc = 0; cc < 7; ++cc) {
              const int th = r >> 1, a = r & 1, tw = cc >> 1, e = cc & 1;
              wp[((size_t)co * 16 + th * 4 + tw) * L.cin_pad + (a * 2 + e) * 3 + c] = w->data[(((size_t)co * 3 + c) * 7 + r) * 7 + cc];
            }
      co0 = L.cout;
      continue;
    }
    M_REQUIRE((L.linear ? nd == 2 : nd == (size_t)(L.dims + 2)) && w->shape[1] == L.cin, "ivx_weights_finalize: %s has the wrong rank / input channels",
              L.w_keys[q].c_str());
    const int co_n = (int)w->shape[0];
    int64_t want = (int64_t)co_n * L.cin * taps;
    M_REQUIRE((int64_t)w->data.size() == want && co0 + co_n <= L.cout, "ivx_weights_finalize: %s has the wrong shape", L.w_keys[q].c_str());
    for (int co = 0; co < co_n; ++co) {      // torch [Cout,Cin,(kd,)kh,kw] -> [Cout,kd,kh,kw,Cin_pad]; contiguous writes (the filter
      const float *src = w->data.data() + (size_t)co * L.cin * taps;                  // block of one output channel stays in L2)
      float *dst = wp.data() + (size_t)(co0 + co) * taps * L.cin_pad;
      for (int t = 0; t < taps; ++t)
        for (int ci = 0; ci < L.cin; ++ci) dst[(size_t)t * L.cin_pad + ci] = src[(size_t)ci * taps + t];
    }
    if (!L.b_keys[q].empty()) {
      const HostTensor *b = find_w(m, L.b_keys[q]);
      if (!b) { *missing += L.b_keys[q] + " "; return IVX_OK; }
      M_REQUIRE((int)b->data.size() == co_n, "ivx_weights_finalize: %s has the wrong size", L.b_keys[q].c_str());
      for (int co = 0; co < co_n; ++co) bias[co0 + co] = b->data[co];
      has_bias = true;
    }
    co0 += co_n;
  }
  if (L.cout_zero > 0 && co0 + L.cout_zero == L.cout) {       // appended zero channels (bias stays 0)
    std::fill(wp.begin() + (size_t)co0 * taps * L.cin_pad, wp.begin() + n_w, 0.f);
    co0 = L.cout;
  }
  M_REQUIRE(co0 == L.cout, "ivx_weights_finalize: layer %s: %d output channels loaded, %d expected", L.name.c_str(), co0, L.cout);
  // Winograd candidate (FusedConv: 3x3 on the transformed axes with stride 1, fp32, unpadded Cin, >= 64 channels 3-D / 128 2-D)
  L.wino2d = kd == 1 && kh == 3 && kw == 3 && L.s[0] == 1 && L.s[1] == 1 && L.s[2] == 1;
  const bool wino3d = kd == 3 && kh == 3 && L.s[0] == 1 && L.s[1] == 1;
  const int min_ch = L.wino2d ? 128 : 64;
  L.wino_cand = !bf16 && (wino3d || L.wino2d) && L.cin_pad == L.cin && L.cout % 4 == 0 && std::max(L.cin, L.cout) >= min_ch && L.cin % 4 == 0;
  if (L.wino_cand) M_TRY(dev_upload_sync(m, wp.data(), n_w, &L.w0, st));   // tap-major [Cout,kd,kh,kw,Cin]; (1,3,3) is the same memory as (3,3,1)
  const int ck = bf16 ? 64 : 32;
  const float *w_src = wp.data();
  if (L.layout == 1) {           // chunk-major K: [Cout, Cin/32, kd,kh,kw, 32] (bf16: 64-channel chunks)
    if (wc.size() < n_w) wc.resize(n_w);
    const int nch = L.cin_pad / ck;
    for (int co = 0; co < L.cout; ++co)
      for (int ch = 0; ch < nch; ++ch)
        for (int t = 0; t < taps; ++t)
          memcpy(&wc[(((size_t)co * nch + ch) * taps + t) * ck], &wp[((size_t)co * taps + t) * L.cin_pad + ch * ck], ck * sizeof(float));
    w_src = wc.data();
  }
  if (bf16) {                    // round to nearest even, two values per uploaded word
    std::vector<uint16_t> wb(n_w + (n_w & 1));
    for (size_t i = 0; i < n_w; ++i) wb[i] = f32_to_bf16_bits(w_src[i]);
    M_TRY(dev_upload_sync(m, reinterpret_cast<const float *>(wb.data()), (n_w + 1) / 2, &L.w, st));
  } else {
    M_TRY(dev_upload_sync(m, w_src, n_w, &L.w, st));
  }
  const int n_aff = L.conv_t ? L.cout / 8 : L.cout;     // the transposed conv's BN has the REAL channel count (epilogue indexes n % Cr)
  std::vector<float> scale(n_aff, 1.f), shift(bias.begin(), bias.begin() + n_aff);
  if (!L.bn.empty()) {
    const HostTensor *g = find_w(m, L.bn + ".weight"), *b = find_w(m, L.bn + ".bias"), *mu = find_w(m, L.bn + ".running_mean"),
                     *var = find_w(m, L.bn + ".running_var");
    if (!g || !b || !mu || !var) { *missing += L.bn + ".{weight,bias,running_mean,running_var} "; return IVX_OK; }
    M_REQUIRE((int)g->data.size() == n_aff && (int)b->data.size() == n_aff && (int)mu->data.size() == n_aff && (int)var->data.size() == n_aff,
              "ivx_weights_finalize: BatchNorm %s has the wrong size", L.bn.c_str());
    std::vector<float> bias_in(shift);
    M_TRY(ivx_fold_batchnorm(g->data.data(), b->data.data(), mu->data.data(), var->data.data(), bias_in.data(), 1e-5f, n_aff, scale.data(),
                             shift.data()));
  }
  if (L.proj_keep) {                 // kept for ivx_bottleneck_proj_pack (ivx_weights_finalize, after every layer is packed)
    L.w_tap.assign(wp.begin(), wp.begin() + n_w);
    L.scale_h = scale;
    L.shift_h = shift;
  }
  if (L.fp8_role) {                  // kept for ivx_model_calibrate_fp8
    if (L.fp8_role >= 2) L.w_tap.assign(wp.begin(), wp.begin() + n_w);
    L.scale_h = scale;
    L.shift_h = shift;
  }
  L.identity_epilogue = !has_bias && L.bn.empty();
  if (!L.identity_epilogue) {
    M_TRY(dev_upload(m, scale, &L.scale, st));
    M_TRY(dev_upload(m, shift, &L.shift, st));
  }
  // Split-operand form (conv.py FusedConv: _split_cand / pack_pair_weights): every fp32 filter as hi = bf16(w), lo = bf16(w - hi), per 16 channels
  // [hi x16 | lo x16], chunk-major [Cout, 2 Cin / 64, taps, 64]
  static const bool split_rule = !(getenv("IVX_CONV_PAIR") && atoi(getenv("IVX_CONV_PAIR")) == 0);      // IVX_CONV_PAIR=0: off in both hosts (A/B)
  L.split_cand = split_rule && !bf16 && m->cfg.wino_operands == IVX_F16_PAIR && L.dims == 3 && kd == 3 && kh == 3 && kw == 3 && !L.conv_t && !L.linear && !L.dcn_cols &&
                 L.cin_pad == L.cin && L.cin % 32 == 0 && L.cout >= 64;
  if (L.split_cand) {
    std::vector<uint16_t> wb(2 * n_w);
    const int nch = L.cin / 32;
    for (int co = 0; co < L.cout; ++co)
      for (int ch = 0; ch < nch; ++ch)
        for (int t = 0; t < taps; ++t) {
          const float *src = &wp[((size_t)co * taps + t) * L.cin + ch * 32];
          uint16_t *dst = &wb[(((size_t)co * nch + ch) * taps + t) * 64];
          for (int g = 0; g < 2; ++g)
            for (int j = 0; j < 16; ++j) {
              const float w = src[g * 16 + j];
              const uint16_t hi = f32_to_bf16_bits(w);
              uint32_t hb = (uint32_t)hi << 16;
              float hf;
              memcpy(&hf, &hb, 4);
              dst[g * 32 + j] = hi;
              dst[g * 32 + 16 + j] = f32_to_bf16_bits(w - hf);
            }
        }
    M_TRY(dev_upload_sync(m, reinterpret_cast<const float *>(wb.data()), n_w, &L.wsplit, st));
  }
  // fp16-pair form of the 2-D trunk (cfg.trunk_operands): pair filters + scale / s_w, and the bound terms (also for the layers that
  // stay fp32: the stem's bound scales the max-pool's pair output)
  const bool trunk_layer = L.name.rfind("backbone.", 0) == 0 || L.name.rfind("neck.", 0) == 0;
  if (m->cfg.trunk_operands == IVX_F16_PAIR && !bf16 && trunk_layer && !L.conv_t && !L.linear) {
    L.pair_ok = L.cin_pad == L.cin && L.cin % 32 == 0 && L.cout % 4 == 0;
    std::vector<uint16_t> packed(L.pair_ok ? 2 * n_w : 0);
    std::vector<float> sp(L.pair_ok ? L.cout : 0);
    M_TRY(ivx_pair_pack_filters(wp.data(), L.cout, taps, L.cin_pad, scale.data(), shift.data(), L.pair_ok ? packed.data() : nullptr,
                                L.pair_ok ? sp.data() : nullptr, &L.wbound, &L.sbound));
    if (L.pair_ok) {
      M_TRY(dev_upload_sync(m, reinterpret_cast<const float *>(packed.data()), n_w, &L.wpair, st));
      M_TRY(dev_upload_sync(m, sp.data(), sp.size(), &L.scale_p, st));
    }
    if (L.name == "backbone.conv1" && L.cin == 3 && L.cout == 64 && kd == 1 && kh == 7 && kw == 7 && L.s[1] == 2 && L.s[2] == 2 && L.p[1] == 3 && L.p[2] == 3 &&
        L.w_keys.size() == 1) {        // the one-launch stem + max-pool (ivx_stem_pool_fwd_pair)
      const HostTensor *w = find_w(m, L.w_keys[0]);
      std::vector<uint16_t> fr((size_t)ivx_stem_pool_filter_bytes() / 2);
      std::vector<float> sps(64);
      M_TRY(ivx_stem_pool_pack_filters(w->data.data(), scale.data(), fr.data(), sps.data()));
      M_TRY(dev_upload_sync(m, reinterpret_cast<const float *>(fr.data()), fr.size() / 2, &L.wstem, st));
      M_TRY(dev_upload_sync(m, sps.data(), sps.size(), &L.scale_p, st));
    }
  }
  return IVX_OK;
}

// Anchor3DRangeGenerator.grid_anchors for one range per size (anchor_3d_generator.py:82-209): centres are
// linspace(min, max, n) in fp32 (first half counted from the start, second half from the end, as torch.linspace),
// flat order y (slow), x, size, rotation (fast).  A host may instead load the reference generator's own output
// through ivx_weights_load("anchors", ...).
void make_anchors(const ivx_model_cfg &c, int H, int W, std::vector<float> *out) {
  const int S = c.n_sizes, R = c.n_rotations;
  out->assign((size_t)H * W * S * R * 7, 0.f);
  auto lin = [](float a, float b, int n, int i) {
    if (n == 1) return a;
    const float step = (b - a) / (float)(n - 1);
    return i < n / 2 ? a + step * (float)i : b - step * (float)(n - 1 - i);
  };
  for (int y = 0; y < H; ++y)
    for (int x = 0; x < W; ++x)
      for (int s = 0; s < S; ++s)
        for (int r = 0; r < R; ++r) {
          float *a = out->data() + ((((size_t)y * W + x) * S + s) * R + r) * 7;
          a[0] = lin(c.anchor_range[0], c.anchor_range[3], W, x);
          a[1] = lin(c.anchor_range[1], c.anchor_range[4], H, y);
          a[2] = c.anchor_range[2];      // one z level: linspace(z, z', 1) = z
          a[3] = c.anchor_sizes[s * 3 + 0]; a[4] = c.anchor_sizes[s * 3 + 1]; a[5] = c.anchor_sizes[s * 3 + 2];
          a[6] = c.anchor_rotations[r];
        }
}

// ---------------------------------------------------------------------------------------------- planning
struct Range { int s0, s1; };

int conv_out(const ConvLayer &L, const TInfo &in, TInfo *o) {
  const int dims[3] = {in.D, in.H, in.W};
  int od[3];
  for (int a = 0; a < 3; ++a) {
    M_REQUIRE(dims[a] + 2 * L.p[a] >= L.k[a], "layer %s: kernel larger than the padded input", L.name.c_str());
    od[a] = (dims[a] + 2 * L.p[a] - L.k[a]) / L.s[a] + 1;
  }
  o->B = in.B; o->D = od[0]; o->H = od[1]; o->W = od[2]; o->C = L.cout;
  if (L.conv_t) { o->D = 2 * in.D; o->H = 2 * in.H; o->W = 2 * in.W; o->C = L.cout / 8; }
  return IVX_OK;
}

constexpr int64_t SPLIT_MIN_POS = 256;      // conv.py FusedConv.SPLIT_MIN_POS
// The Winograd decision of FusedConv.wino_tile (conv.py): returns the tile (0 = direct) and the descriptor to run.
int plan_conv(ivx_model *m, ConvLayer &L, const TInfo &in, const Step &st, const TInfo *res, PlanStep *ps, hipStream_t stream, int out_fmt = 0) {
  ivx_conv_desc d;
  memset(&d, 0, sizeof(d));
  d.B = in.B; d.D = in.D; d.H = in.H; d.W = in.W; d.Cin = in.C; d.Cout = L.cout;
  d.KD = L.k[0]; d.KH = L.k[1]; d.KW = L.k[2];
  d.sd = L.s[0]; d.sh = L.s[1]; d.sw = L.s[2];
  d.pd = L.p[0]; d.ph = L.p[1]; d.pw = L.p[2];
  d.relu = L.relu ? 1 : 0;
  d.res_mode = st.res_mode;
  if (st.res_mode == 2) { d.res_h = res->H; d.res_w = res->W; }
  d.wgt_layout = L.layout; d.post_scale = st.post_scale;
  d.res_after_act = st.res_after_act;
  d.out_mode = L.conv_t ? 1 : 0;
  M_REQUIRE(in.C == L.cin_pad, "layer %s: input has %d channels, expected %d", L.name.c_str(), in.C, L.cin_pad);
  if (m->cfg.storage == IVX_BF16) {   // bf16 storage: the direct kernel on bf16 operands, fp32 accumulate; the head convs write fp32
    d.in_dtype = IVX_BF16;
    d.out_dtype = L.out_f32 ? IVX_F32 : IVX_BF16;
    if (m->fp8_on && L.fp8_eff) {     // e4m3 interior of a bottleneck: tensor scales are folded into scale_q / shift_q
      d.in_dtype = L.fp8_eff == 1 ? IVX_BF16 : IVX_FP8;
      d.out_dtype = L.fp8_eff == 3 ? IVX_BF16 : IVX_FP8;
      if (L.fp8_eff >= 2) d.wgt_layout = L.layout_q;
      d.res_scale = 1.0f;
    }
  }
  if (in.fmt == IVX_F16_PAIR) {      // the fp16-pair form: direct kernel on the 16-bit matrix cores, scales on the device
    M_REQUIRE(L.pair_ok && L.wpair, "internal: layer %s reads a pair tensor but has no pair filters", L.name.c_str());
    d.in_dtype = IVX_F16_PAIR; d.out_dtype = out_fmt; d.wgt_layout = 1;
    ivx_pair_io io;
    memset(&io, 0, sizeof(io));
    float f; uint32_t u;             // workspace query: the pointers are only tested for NULL
    io.in_scale = &f; io.res_scale = &f; io.out_scale = &f; io.amax_in = &u; io.amax_res = &u; io.amax_out = &u;
    io.res_dtype = res ? res->fmt : 0;
    ps->d = d;
    ps->ws = ivx_conv_pio_workspace_bytes(&d, &io);
    M_REQUIRE(ps->ws >= 0, "layer %s: %s", L.name.c_str(), ivx_last_error());
    ps->tile = 0;
    ps->pio = 1;
    return IVX_OK;
  }
  // An fp32-input convolution reads its residual as fp32 (or as the storage type): a pair residual here would be read as garbage without
  // an error (round-4 advisor).  Unreachable with ResNet-50 -- a tensor is a pair only when every INPUT consumer reads pairs, and a
  // residual's producer and consumer sit in the same bottleneck -- but the format rule must not rest on that.
  M_REQUIRE(!res || res->fmt == 0, "internal: layer %s has an fp32 input and a pair residual (the pair chain broke between them)", L.name.c_str());
  int tile = 0;
  ivx_conv_desc dw = d;
  if (L.wino_cand && !L.conv_t && m->cfg.winograd && m->cfg.storage == IVX_F32 && (st.res_mode == 0 || st.res_mode == 1) && (int64_t)in.B * in.D * in.H * in.W >= 2000) {
    if (L.wino2d) {   // [B,1,H,W,C] as [B,H,W,1,C] with a 3x3x1 kernel
      dw.D = in.H; dw.H = in.W; dw.W = 1;
      dw.KD = 3; dw.KH = 3; dw.KW = 1; dw.sd = dw.sh = dw.sw = 1;
      dw.pd = L.p[1]; dw.ph = L.p[2]; dw.pw = 0;
    }
    const int64_t plane = (int64_t)(dw.D + 2 * dw.pd - 2) * (dw.H + 2 * dw.ph - 2);
    tile = m->cfg.winograd_tile ? m->cfg.winograd_tile : (plane >= 16384 ? 6 : 4);
    ivx_conv_desc probe = dw;
    probe.relu = 0; probe.res_mode = 0; probe.wgt_layout = 0;
    if (!ivx_conv_winograd_supported(&probe, tile)) tile = 0;
    // operands of the transformed-domain GEMMs (FusedConv._wino_operands): fp16 pairs where the layer's channel count allows
    if (tile >= 4 && m->cfg.wino_operands == IVX_F16_PAIR && L.cin % (L.layout == 1 ? 32 : 16) == 0) dw.wino_operands = IVX_F16_PAIR;
  }
  if (tile) {
    ps->d = dw;
    ps->ws = ivx_conv_winograd_workspace_bytes(&dw, tile);
    M_REQUIRE(ps->ws >= 0, "layer %s: %s", L.name.c_str(), ivx_last_error());
    const int ukey = tile * 8 + dw.wino_operands;
    if (!L.u.count(ukey)) {   // transformed filters for this tile and operand type, made once (ops.conv_winograd_weights)
      ivx_conv_desc wd;
      memset(&wd, 0, sizeof(wd));
      wd.B = 1; wd.D = 4; wd.H = 4; wd.W = std::max(dw.KW, 1); wd.Cin = L.cin; wd.Cout = L.cout;
      wd.KD = 3; wd.KH = 3; wd.KW = dw.KW; wd.sd = wd.sh = wd.sw = 1; wd.pd = wd.ph = 1; wd.pw = dw.KW / 2;
      wd.wgt_layout = L.layout; wd.post_scale = 1.0f; wd.wino_operands = dw.wino_operands;
      const int64_t n = ivx_conv_winograd_weight_elems(&wd, tile);
      M_REQUIRE(n > 0, "layer %s: %s", L.name.c_str(), ivx_last_error());
      void *u = nullptr;
      M_HIP(hipMalloc(&u, (size_t)n * sizeof(float)), "hipMalloc (Winograd filters)");
      m->owned.push_back(u);
      M_TRY(ivx_conv_winograd_weights(&wd, tile, L.w0, (float *)u, stream));
      // One-time work of planning a new shape.  The *_workspace_bytes / *_dims queries plan on the NULL stream and the forward
      // that follows may run on a non-blocking stream: the transformed filters must be complete before this returns, whatever
      // stream later reads them.
      M_HIP(hipStreamSynchronize(stream), "hipStreamSynchronize (Winograd filters)");
      L.u[ukey] = (float *)u;
    }
  } else {
    // the split-operand form for the 3x3x3 layers the Winograd form did not take (conv.py FusedConv.takes_pair_form, the default rule): the
    // strided convolutions of the necks and the layers of their coarsest levels; fp32 tensors on both sides, 16-bit matrix cores in between
    if (L.split_cand && L.wsplit && m->cfg.storage == IVX_F32 && (int64_t)in.B * in.D * in.H * in.W >= SPLIT_MIN_POS && ivx_conv_pair_supported(&d) == 1) {
      ps->split = align256((int64_t)in.elems() * 4);
      d.in_dtype = IVX_BF16_PAIR;
      d.wgt_layout = 1;
    }
    ps->d = d;
    ps->ws = ivx_conv_workspace_bytes(&d);
    M_REQUIRE(ps->ws >= 0, "layer %s: %s", L.name.c_str(), ivx_last_error());
    ps->ws += ps->split;
  }
  ps->tile = tile;
  return IVX_OK;
}

// Shapes, per-step descriptors and a liveness-based arena for the steps [r.s0, r.s1).  `ext` tensors are caller-owned.
int make_plan(ivx_model *m, Range r, const std::map<int, TInfo> &inputs, int n_views, Plan *pl, hipStream_t stream) {
  pl->t.assign(m->n_tensors, TInfo());
  pl->ps.assign(m->steps.size(), PlanStep());
  for (auto &kv : inputs) { pl->t[kv.first] = kv.second; pl->t[kv.first].first = -2; }
  const ivx_model_cfg &c = m->cfg;
  // fp16-pair chaining of the 2-D trunk (cfg.trunk_operands): a tensor is stored as pairs when its producer can write them (the
  // max-pool after the stem, a convolution that itself reads pairs) and every step that reads it as an INPUT is a trunk convolution
  // with pair filters; residual uses take either format.  Tensors that leave the trunk (FPN level 0, C5 for the LayoutHead, DCNv2
  // columns) therefore stay fp32.  Every tensor of the chain owns a scalar block (amax slots + scale) in the arena.
  const bool pair_mode = c.trunk_operands == IVX_F16_PAIR && c.with_trunk && c.storage == IVX_F32;
  const int esz = c.storage == IVX_BF16 ? 2 : 4;       // bytes per activation element (head outputs: always 4)
  int n_slots = 0;
  auto wants_pair = [&](int t, const TInfo &ti) {
    if (!pair_mode || ti.C % 16 || ti.elems() * 4 >= (1LL << 31)) return false;
    bool any = false;
    for (int j = r.s0; j < r.s1; ++j) {
      const Step &q = m->steps[j];
      if (q.in != t) continue;
      if (j >= m->trunk0 && j < m->trunk1 && q.kind == ST_DCN_COL) {
        // the column kernel reads pairs when it can also WRITE pairs: the contraction conv has pair filters and 9x the map fits 2 GiB
        bool ok = ti.elems() * 9 * 4 < (1LL << 31);
        for (int j2 = j + 1; j2 < r.s1 && ok; ++j2)
          if (m->steps[j2].in == q.out && (m->steps[j2].kind != ST_CONV || !m->layers[m->steps[j2].layer].pair_ok)) ok = false;
        if (!ok) return false;
        any = true;
        continue;
      }
      if (j < m->trunk0 || j >= m->trunk1 || q.kind != ST_CONV || !m->layers[q.layer].pair_ok) return false;
      {   // a wide 3x3 layer on a large map keeps fp32 tensors and runs in its Winograd form with pair operands in the transformed domain
          // (conv.py FusedConv.prefers_winograd: 256 -> 256 at 120x160x50 2.25 vs 3.14 ms for the direct pair form)
        const ConvLayer &Lq = m->layers[q.layer];
        // Only the FPN output conv: the bottleneck's conv2 always takes the direct pair form in BOTH hosts (backbones.py _Bottleneck.forward_cl
        // does not consult prefers_winograd; a rule applied to every wide trunk 3x3 here would let the two copies drift apart at 167+ views).
        if (Lq.name.rfind("neck.fpn_conv", 0) == 0 && Lq.wino2d && Lq.wino_cand && c.winograd && c.wino_operands == IVX_F16_PAIR && Lq.cin >= 256 &&
            Lq.cout >= 256 && Lq.cin % 32 == 0 && (int64_t)ti.B * ti.H * ti.W >= 200000)
          return false;
      }
      any = true;
    }
    return any;
  };
  for (int i = r.s0; i < r.s1; ++i) {
    const Step &s = m->steps[i];
    const TInfo &in = pl->t[s.in];
    M_REQUIRE(in.B > 0, "internal: step %d reads an unplanned tensor", i);
    TInfo o;
    o.fmt = 0; o.slot = -1;
    switch (s.kind) {
      case ST_IMG2CL:
        o = in; o.C = 4; o.fmt = 0; o.slot = -1;
        if (pair_mode) o.slot = n_slots++;                 // max |image| for the bound of the stem's output
        break;
      case ST_IMG_S2D:                                     // bf16 storage: [BV, 3, H, W] fp32 -> [BV, 1, H/2+1, W/2+1, 16] bf16 blocks
        M_REQUIRE(in.H % 2 == 0 && in.W % 2 == 0, "bf16 storage: the image height and width must be even");
        o = in; o.H = in.H / 2 + 1; o.W = in.W / 2 + 1; o.C = 16;
        break;
      case ST_MAXPOOL: {
        o = in; o.H = (in.H + 2 - 3) / 2 + 1; o.W = (in.W + 2 - 3) / 2 + 1; o.fmt = 0; o.slot = -1;
        int prod = -1;                                     // the stem: its bound terms and the image's maximum scale the pooled map
        for (int j = r.s0; j < i; ++j)
          if (m->steps[j].out == s.in && m->steps[j].kind == ST_CONV) prod = j;
        if (prod >= 0 && pl->t[m->steps[prod].in].slot >= 0 && pl->t[m->steps[prod].in].fmt == 0 && wants_pair(s.out, o)) {
          o.fmt = IVX_F16_PAIR; o.slot = n_slots++;
          pl->ps[i].pio = 1; pl->ps[i].bound_layer = m->steps[prod].layer;
        }
        break;
      }
      case ST_UPSAMPLE: o = in; o.D = 2 * in.D; o.H = 2 * in.H; o.W = 2 * in.W; break;
      case ST_CONV: {
        ConvLayer &L = m->layers[s.layer];
        M_TRY(conv_out(L, in, &o));
        int out_fmt = 0;
        if (in.fmt == IVX_F16_PAIR) {                      // reads pairs: its epilogue leaves max |out|, and writes pairs when the consumers take them
          o.slot = n_slots++;
          if (o.C % 16 == 0 && in.slot >= 0 && (s.res < 0 || pl->t[s.res].slot >= 0) && wants_pair(s.out, o)) out_fmt = IVX_F16_PAIR;
          o.fmt = out_fmt;
        }
        M_TRY(plan_conv(m, L, in, s, s.res >= 0 ? &pl->t[s.res] : nullptr, &pl->ps[i], stream, out_fmt));
        pl->ws_bytes = std::max(pl->ws_bytes, pl->ps[i].ws);
        break;
      }
      case ST_LIFT: {
        const int V = n_views;
        M_REQUIRE(in.B % V == 0, "the %d feature maps do not divide into scenes of %d views", in.B, V);
        o.B = in.B / V; o.D = c.n_voxels[0]; o.H = c.n_voxels[1]; o.W = c.n_voxels[2]; o.C = in.C;
        TInfo v = o; v.C = 1;
        v.bytes = align256(v.elems());          // u8 mask
        v.first = i;
        pl->t[s.out2] = v;
        break;
      }
      case ST_TAIL: break;
      case ST_DCN_COL: {
        const TInfo &om = pl->t[s.res];          // the conv_offset output fixes the output grid
        o = om; o.C = 9 * in.C; o.fmt = 0; o.slot = -1;
        if (in.fmt == IVX_F16_PAIR) {            // (wants_pair let the input be a pair tensor only if the columns can be one too)
          M_REQUIRE(in.slot >= 0 && wants_pair(s.out, o), "internal: DCNv2 columns of a pair tensor must be a pair tensor");
          o.fmt = IVX_F16_PAIR; o.slot = n_slots++;
          pl->ps[i].pio = 1;
        }
        break;
      }
      case ST_AVGPOOL: o = in; o.D = 1; o.H = 1; o.W = 1; break;
      case ST_LAYOUT: break;
      case ST_FCOS: {
        const int n = in.D * in.H * in.W, k = (c.head_nms_pre > 0 && c.head_nms_pre < n) ? c.head_nms_pre : n;
        const int R = c.head_type == IVX_HEAD_SCANNET ? 6 : 7;
        const int64_t fw = ivx_fcos_head_workspace_bytes(in.B, n, c.head_nms_pre);
        M_REQUIRE(fw >= 0, "indoor head level %d: more than 65536 candidates per level (nms_pre %d)", s.aux, c.head_nms_pre);
        pl->ws_bytes = std::max(pl->ws_bytes, fw);
        pl->itail.k[s.aux] = k;
        TInfo cb; cb.B = in.B; cb.D = cb.H = 1; cb.W = k; cb.C = R; cb.raw = true; cb.bytes = align256((int64_t)in.B * k * R * 4); cb.first = i;
        TInfo cs = cb; cs.C = c.head_classes; cs.bytes = align256((int64_t)in.B * k * c.head_classes * 4);
        TInfo cc = cb; cc.W = 1; cc.C = 1; cc.bytes = align256((int64_t)in.B * 4);
        pl->t[s.out] = cb; pl->t[s.out2] = cs; pl->t[m->t_cc[s.aux]] = cc;
        break;
      }
      case ST_INDOOR_TAIL: {
        ivx_indoor_tail_desc &d = pl->itail;
        d.B = in.B; d.n_levels = (int)m->t_cb.size(); d.n_classes = c.head_classes; d.n_reg = c.head_type == IVX_HEAD_SCANNET ? 6 : 7;
        d.use_rotate_nms = c.head_use_rotate_nms; d.score_thr = c.head_score_thr; d.nms_thr = c.head_nms_thr;
        int K = 0;
        for (int l = 0; l < d.n_levels; ++l) K += d.k[l];
        // ScanNet: nothing is cut after the NMS (imvoxel_head_v2.py:528-545); SUN RGB-D: max_num = nms_pre (:411-413)
        d.max_num = c.head_type == IVX_HEAD_SCANNET ? K : std::min<int64_t>(c.head_nms_pre > 0 ? c.head_nms_pre : K, (int64_t)K * c.head_classes);
        pl->max_det = d.max_num;
        const int64_t tw = ivx_indoor_tail_workspace_bytes(&d);
        M_REQUIRE(tw >= 0, "indoor tail: %s", ivx_last_error());
        pl->ws_bytes = std::max(pl->ws_bytes, tw);
        break;
      }
    }
    if (s.kind == ST_LAYOUT || s.kind == ST_FCOS || s.kind == ST_INDOOR_TAIL) continue;
    if (s.kind == ST_TAIL) {
      const TInfo &nk = pl->t[m->t_neck];      // [B, X', Y', 1, C]: the reference's H = Y', W = X' (necks/imvoxelnet.py:120)
      ivx_anchor_head_desc &d = pl->tail;
      memset(&d, 0, sizeof(d));
      d.B = nk.B; d.H = nk.H; d.W = nk.D; d.CH = in.C;
      d.num_anchors = c.n_sizes * c.n_rotations; d.num_classes = c.num_classes;
      d.cls_off = 0; d.reg_off = d.num_anchors * c.num_classes; d.dir_off = d.reg_off + d.num_anchors * 7;
      d.nms_pre = c.nms_pre; d.max_num = c.max_num; d.use_rotate_nms = c.use_rotate_nms; d.hw_transposed = 1;
      d.score_thr = c.score_thr; d.nms_thr = c.nms_thr; d.dir_offset = c.dir_offset; d.dir_limit_offset = c.dir_limit_offset;
      pl->tail_ws = ivx_anchor_head_workspace_bytes(&d);
      M_REQUIRE(pl->tail_ws >= 0, "anchor tail: %s", ivx_last_error());
      pl->ws_bytes = std::max(pl->ws_bytes, pl->tail_ws);
      pl->max_det = c.max_num;
      continue;
    }
    if (s.kind != ST_CONV && s.kind != ST_MAXPOOL && s.kind != ST_IMG2CL && s.kind != ST_DCN_COL) { o.fmt = 0; o.slot = -1; }   // (`o = in` above copies the input's)
    o.esz = (s.kind == ST_CONV && m->layers[s.layer].out_f32) ? 4 : esz;
    if (s.kind == ST_CONV && m->fp8_on && (m->layers[s.layer].fp8_eff == 1 || m->layers[s.layer].fp8_eff == 2)) o.esz = 1;
    o.bytes = align256(o.elems() * o.esz);
    o.first = i;
    pl->t[s.out] = o;
  }
  // liveness
  for (int i = r.s0; i < r.s1; ++i) {
    const Step &s = m->steps[i];
    for (int t : {s.in, s.res})
      if (t >= 0) pl->t[t].last = std::max(pl->t[t].last, i);
    if (s.kind == ST_INDOOR_TAIL)            // reads every level's candidate buffers
      for (size_t l = 0; l < m->t_cb.size(); ++l)
        for (int t : {m->t_cb[l], m->t_cs[l], m->t_cc[l]}) pl->t[t].last = std::max(pl->t[t].last, i);
  }
  // Identity bottlenecks of the pair chain that the one-launch kernel takes: conv1 1x1 (4P -> P) -> conv2 3x3 s1 p1 (P -> P) -> conv3 1x1
  // (P -> 4P) + the block's input as the shortcut, every tensor a pair tensor, P in {64, 128} (ResNet-50's stages 1 and 2).  The block's
  // output is written by conv1's step, so it is allocated there; the two P-channel intermediates stay unwritten.
  // (backbones.py _Bottleneck.forward_cl applies the same rule: both hosts launch the same kernels.)
  static const bool fuse_on = !(getenv("IVX_FUSE_BOTTLENECK") && atoi(getenv("IVX_FUSE_BOTTLENECK")) == 0);
  // The head of the chain in one launch (ivx_stem_pool_fwd_pair): image layout change + 7x7 stem + max-pool -> only max |image| is computed
  // at the layout step, the stem's step is skipped, the max-pool's step runs the fused kernel on the caller's NCHW image.
  static const bool fuse_stem = !(getenv("IVX_FUSE_STEM") && atoi(getenv("IVX_FUSE_STEM")) == 0);
  if (pair_mode && fuse_stem)
    for (int i = std::max(r.s0, m->trunk0); i + 2 < std::min(r.s1, m->trunk1); ++i) {
      const Step &s0 = m->steps[i], &s1 = m->steps[i + 1], &s2 = m->steps[i + 2];
      if (s0.kind != ST_IMG2CL || s1.kind != ST_CONV || s2.kind != ST_MAXPOOL || s1.in != s0.out || s2.in != s1.out) continue;
      const ConvLayer &L = m->layers[s1.layer];
      if (!L.wstem || !pl->ps[i + 2].pio || pl->t[s0.out].slot < 0 || pl->t[s1.out].last != i + 2 || pl->t[s0.out].last != i + 1) continue;
      pl->ps[i].fuse = 3; pl->ps[i + 1].fuse = 2; pl->ps[i + 2].fuse = 4;
    }
  // The first block of stage 1 (steps: shortcut conv, conv1, conv2, conv3 + shortcut) in one launch (ivx_bottleneck_proj_fwd_pio; backbones.py
  // _Bottleneck.fuses_proj applies the same rule): the shortcut conv's step and conv2 / conv3 are skipped, conv1's step (fuse 5) runs the block; the
  // shortcut tensor and the two intermediates stay unwritten.
  if (pair_mode && fuse_on)
    for (int i = std::max(r.s0, m->trunk0); i + 3 < std::min(r.s1, m->trunk1); ++i) {
      const Step &sd = m->steps[i], &s1 = m->steps[i + 1], &s2 = m->steps[i + 2], &s3 = m->steps[i + 3];
      if (sd.kind != ST_CONV || s1.kind != ST_CONV || s2.kind != ST_CONV || s3.kind != ST_CONV) continue;
      const ConvLayer &Ld = m->layers[sd.layer], &L1 = m->layers[s1.layer], &L2 = m->layers[s2.layer], &L3 = m->layers[s3.layer];
      if (!L3.wproj || L3.proj_peer != sd.layer) continue;
      const TInfo &x = pl->t[sd.in], &yd = pl->t[sd.out], &y1 = pl->t[s1.out], &y2 = pl->t[s2.out], &o = pl->t[s3.out];
      auto is1x1 = [](const ConvLayer &L) { return L.k[0] == 1 && L.k[1] == 1 && L.k[2] == 1 && L.s[1] == 1 && L.s[2] == 1 && L.p[1] == 0 && L.p[2] == 0; };
      const int P = L1.cout;
      if (!(pl->ps[i + 1].pio && pl->ps[i + 2].pio && pl->ps[i + 3].pio) || sd.res >= 0 || s1.res >= 0 || s2.res >= 0 || s1.in != sd.in || s2.in != s1.out ||
          s3.in != s2.out || s3.res != sd.out || s3.res_mode != 1 || s3.res_after_act || s3.post_scale != 1.0f)
        continue;
      if (!is1x1(Ld) || !is1x1(L1) || !is1x1(L3) || !(L2.k[0] == 1 && L2.k[1] == 3 && L2.k[2] == 3 && L2.s[1] == 1 && L2.s[2] == 1 && L2.p[1] == 1 && L2.p[2] == 1) ||
          L2.dcn_cols || Ld.relu || !L1.relu || !L2.relu || !L3.relu || L1.cin != Ld.cin || L2.cin != P || L2.cout != P || L3.cin != P || L3.cout != 4 * P ||
          Ld.cout != 4 * P)
        continue;
      if (x.fmt != IVX_F16_PAIR || y1.fmt != IVX_F16_PAIR || y2.fmt != IVX_F16_PAIR || o.fmt != IVX_F16_PAIR || x.D != 1 || x.slot < 0 || o.slot < 0 ||
          yd.last != i + 3 || y1.last != i + 2 || y2.last != i + 3)
        continue;
      ivx_bottleneck_desc bd = {x.B, x.H, x.W, P};
      if (!ivx_bottleneck_proj_supported(&bd, L1.cin)) continue;
      pl->ps[i].fuse = 2;
      pl->ps[i + 1].fuse = 5; pl->ps[i + 1].fuse_out = s3.out;
      pl->ps[i + 2].fuse = pl->ps[i + 3].fuse = 2;
      pl->t[s3.out].first = i + 1;
      i += 3;
    }
  if (pair_mode && fuse_on)
    for (int i = std::max(r.s0, m->trunk0); i + 2 < std::min(r.s1, m->trunk1); ++i) {
      const Step &s1 = m->steps[i], &s2 = m->steps[i + 1], &s3 = m->steps[i + 2];
      if (pl->ps[i].fuse || pl->ps[i + 1].fuse || pl->ps[i + 2].fuse) continue;
      if (s1.kind != ST_CONV || s2.kind != ST_CONV || s3.kind != ST_CONV) continue;
      const ConvLayer &L1 = m->layers[s1.layer], &L2 = m->layers[s2.layer], &L3 = m->layers[s3.layer];
      const TInfo &x = pl->t[s1.in], &y1 = pl->t[s1.out], &y2 = pl->t[s2.out], &o = pl->t[s3.out];
      auto is1x1 = [](const ConvLayer &L) { return L.k[0] == 1 && L.k[1] == 1 && L.k[2] == 1 && L.s[1] == 1 && L.s[2] == 1 && L.p[1] == 0 && L.p[2] == 0; };
      const int P = L1.cout;
      if (!(pl->ps[i].pio && pl->ps[i + 1].pio && pl->ps[i + 2].pio) || s1.res >= 0 || s2.res >= 0 || s2.in != s1.out || s3.in != s2.out ||
          s3.res != s1.in || s3.res_mode != 1 || s3.res_after_act || s3.post_scale != 1.0f)
        continue;
      if (!is1x1(L1) || !is1x1(L3) || !(L2.k[0] == 1 && L2.k[1] == 3 && L2.k[2] == 3 && L2.s[1] == 1 && L2.s[2] == 1 && L2.p[1] == 1 && L2.p[2] == 1) ||
          L2.dcn_cols || !L1.relu || !L2.relu || !L3.relu || L1.cin != 4 * P || L2.cin != P || L2.cout != P || L3.cin != P || L3.cout != 4 * P)
        continue;
      if (x.fmt != IVX_F16_PAIR || y1.fmt != IVX_F16_PAIR || y2.fmt != IVX_F16_PAIR || o.fmt != IVX_F16_PAIR || x.D != 1 || x.slot < 0 || o.slot < 0 ||
          y1.last != i + 1 || y2.last != i + 2)
        continue;
      ivx_bottleneck_desc bd = {x.B, x.H, x.W, P};
      if (!ivx_bottleneck_supported(&bd)) continue;
      pl->ps[i].fuse = 1; pl->ps[i].fuse_out = s3.out;
      pl->ps[i + 1].fuse = pl->ps[i + 2].fuse = 2;
      pl->t[s3.out].first = i;
      i += 2;
    }
  // Shortcut convs on the side stream: step i = the downsample conv, steps i + 1 .. j - 1 read the same block input / each other, step j takes
  // the shortcut as its residual.  The block input stays allocated until the join (the side launch may still be reading it).
  static const bool side_on = !(getenv("IVX_SIDE_STREAM") && atoi(getenv("IVX_SIDE_STREAM")) == 0);
  if (side_on && c.with_trunk)
    for (int i = std::max(r.s0, m->trunk0); i + 2 < std::min(r.s1, m->trunk1); ++i) {
      const Step &s0 = m->steps[i];
      if (s0.kind != ST_CONV || pl->ps[i].tile != 0 || pl->ps[i].fuse != 0 || s0.res >= 0) continue;
      const std::string &nm = m->layers[s0.layer].name;
      if (nm.size() < 10 || nm.compare(nm.size() - 10, 10, "downsample") != 0) continue;
      int j = -1;
      for (int k = i + 1; k < std::min(r.s1, m->trunk1) && k <= i + 6; ++k)
        if (m->steps[k].res == s0.out) { j = k; break; }
      if (j < 0 || j < i + 2) continue;
      bool ok = true;
      for (int k = i + 1; k < j && ok; ++k) ok = m->steps[k].in != s0.out && m->steps[k].res != s0.out && pl->ps[k].side == 0 && pl->ps[k].join == 0;
      if (!ok) continue;
      pl->ps[i].side = pl->ps[j].join = ++pl->n_sides;
      pl->t[s0.in].last = std::max(pl->t[s0.in].last, j);
    }
  std::vector<int> keep = {m->t_fpn0, m->t_volume, m->t_valid, m->t_neck, m->t_head, m->t_angle, m->t_layout};
  keep.insert(keep.end(), m->t_levels.begin(), m->t_levels.end());
  for (int t : keep)
    if (t >= 0 && pl->t[t].first >= r.s0) pl->t[t].last = r.s1;   // boundary tensors may be read back by the caller
  // first-fit arena with coalescing free list
  struct Blk { int64_t off, size; };
  std::vector<Blk> free_list;
  // camera block (ivx_model_detect): proj [B,V,3,4] | new_origin [B,3] | crop [B,2] i32 | per head level: voxel size [B,3], origin [B,3]
  int64_t top = 0;
  {
    const TInfo &vol = pl->t[m->t_volume];
    const int Bs = vol.B > 0 ? vol.B : 1;
    auto take = [&](int64_t bytes) { const int64_t at = top; top += align256(bytes); return at; };
    pl->n_views = n_views;
    pl->cam_proj = take((int64_t)Bs * n_views * 12 * 4);
    pl->cam_origin = take((int64_t)Bs * 3 * 4);
    pl->cam_crop = take((int64_t)Bs * 2 * 4);
    for (int l = 0; l < 3; ++l) { pl->cam_lvl_vs[l] = take((int64_t)Bs * 3 * 4); pl->cam_lvl_no[l] = take((int64_t)Bs * 3 * 4); }
    pl->cam_bytes = top;
  }
  auto release = [&](int64_t off, int64_t size) {
    free_list.push_back({off, size});
    std::sort(free_list.begin(), free_list.end(), [](const Blk &a, const Blk &b) { return a.off < b.off; });
    std::vector<Blk> merged;
    for (const Blk &b : free_list) {
      if (!merged.empty() && merged.back().off + merged.back().size == b.off) merged.back().size += b.size;
      else merged.push_back(b);
    }
    free_list.swap(merged);
  };
  for (int i = r.s0; i < r.s1; ++i) {
    const Step &s = m->steps[i];
    const int extra = s.kind == ST_FCOS ? m->t_cc[s.aux] : pl->ps[i].fuse_out;
    for (int t : {s.out, s.out2, extra}) {
      if (t < 0 || pl->t[t].first != i) continue;
      TInfo &ti = pl->t[t];
      bool placed = false;
      for (size_t f = 0; f < free_list.size(); ++f)
        if (free_list[f].size >= ti.bytes) {
          ti.off = free_list[f].off;
          free_list[f].off += ti.bytes; free_list[f].size -= ti.bytes;
          if (free_list[f].size == 0) free_list.erase(free_list.begin() + f);
          placed = true;
          break;
        }
      if (!placed) { ti.off = top; top += ti.bytes; }
    }
    for (int t = 0; t < m->n_tensors; ++t) {   // tensors whose last reader is this step die now
      TInfo &ti = pl->t[t];
      if (ti.off >= 0 && ti.first >= r.s0 && ti.last == i && ti.first != -2) release(ti.off, ti.bytes);
    }
    // an output nobody reads (should not happen) stays allocated
  }
  // per-workgroup maxima between chained Winograd layers (see PlanStep)
  for (int i = r.s0; i < r.s1; ++i) {
    const Step &s = m->steps[i];
    PlanStep &pc = pl->ps[i];
    if (s.kind != ST_CONV || pc.tile == 0 || pc.d.wino_operands != IVX_F16_PAIR) continue;
    for (int j = i - 1; j >= r.s0; --j) {
      const Step &q = m->steps[j];
      if (q.out != s.in) continue;
      PlanStep &pp = pl->ps[j];
      if (q.kind == ST_LIFT) {          // the single-view unprojection leaves per-workgroup maxima too
        const TInfo &vol = pl->t[q.out];
        const int32_t nb = ivx_backproject_amax_blocks(vol.B, n_views, vol.D, vol.H, vol.W);
        if (nb > 0) {
          if (pp.amax_out < 0) { pp.amax_out = top; pp.amax_n = nb; top += align256((int64_t)nb * 4); }
          pc.amax_in = pp.amax_out;
          pc.amax_in_n = pp.amax_n;
        }
      }
      if (q.kind == ST_CONV && pp.tile > 0) {
        if (pp.amax_out < 0) {
          const int32_t nb = ivx_conv_winograd_output_blocks(&pp.d, pp.tile);
          M_REQUIRE(nb > 0, "layer %s: %s", m->layers[q.layer].name.c_str(), ivx_last_error());
          pp.amax_out = top;
          pp.amax_n = nb;
          top += align256((int64_t)nb * 4);
        }
        pc.amax_in = pp.amax_out;
        pc.amax_in_n = pp.amax_n;
      }
      break;
    }
  }
  if (n_slots > 0) {        // scalar blocks of the pair chain: IVX_AMAX_SLOTS words + the scale, 512 bytes apart
    top = align256(top);
    pl->scal_off = top;
    pl->scal_bytes = (int64_t)n_slots * 512;
    top += pl->scal_bytes;
  }
  pl->arena = align256(top);
  pl->ws_off = pl->arena;
  pl->ws_bytes = align256(pl->ws_bytes);
  pl->total = pl->arena + pl->ws_bytes;
  if (pl->n_sides > 0) { pl->ws2_off = pl->total; pl->total += pl->ws_bytes; }      // the side stream's own split-K workspace
  return IVX_OK;
}

std::string plan_key(const char *what, int B, int V, int H, int W) {
  char buf[96];
  snprintf(buf, sizeof(buf), "%s:%d:%d:%d:%d", what, B, V, H, W);
  return buf;
}

TInfo tinfo(int B, int D, int H, int W, int C, int esz = 4) {
  TInfo t; t.B = B; t.D = D; t.H = H; t.W = W; t.C = C; t.esz = esz; t.bytes = align256(t.elems() * esz);
  return t;
}

int get_plan(ivx_model *m, const char *what, Range r, int B, int V, int H, int W, const std::map<int, TInfo> &inputs, Plan **out,
             hipStream_t stream) {
  M_REQUIRE(m->finalized, "%s: call ivx_weights_finalize first", what);
  const std::string key = plan_key(what, B, V, H, W);
  auto it = m->plans.find(key);
  if (it == m->plans.end()) {
    std::unique_ptr<Plan> pl(new Plan());
    M_TRY(make_plan(m, r, inputs, V, pl.get(), stream));
    it = m->plans.emplace(key, std::move(pl)).first;
  }
  *out = it->second.get();
  return IVX_OK;
}

// ---------------------------------------------------------------------------------------------- execution
struct Bind {               // caller-owned buffers by tensor id
  std::map<int, void *> ext;
  const float *proj = nullptr, *new_origin = nullptr;
  const int32_t *crop = nullptr;
  int V = 1;
  float *boxes = nullptr, *scores = nullptr;
  int64_t *labels = nullptr;
  int32_t *count = nullptr;
  // ivx_model_detect: per-level geometry of the anchor-free head (device), the host metas (LayoutHead: the projection is built
  // from the predicted angles in the middle of the forward) and the host outputs of the LayoutHead
  const float *lvl_vs[3] = {nullptr, nullptr, nullptr}, *lvl_no[3] = {nullptr, nullptr, nullptr};
  const ivx_sample_meta *metas = nullptr;
  float *proj_dev = nullptr;           // writable alias of proj inside the camera block
  float *angles_host = nullptr, *layout_host = nullptr;
};

int trace_begin(ivx_model *m, int step, int stage, int is3d, double flops, double bytes, const std::string &name, hipStream_t st) {
  if (!m->trace_on) return IVX_OK;
  m->trace_skip = m->trace_level == 3 && stage != 2;      // level 3: only the grouped Winograd-domain GEMM launches carry events
  if (m->trace_skip) return IVX_OK;
  while (m->event_pool.size() < m->events_used + 2) {
    hipEvent_t e;
    M_HIP(hipEventCreate(&e), "hipEventCreate");
    m->event_pool.push_back(e);
  }
  ivx_model::TraceRec r{step, stage, is3d, flops, bytes, m->event_pool[m->events_used], m->event_pool[m->events_used + 1], name};
  m->events_used += 2;
  M_HIP(hipEventRecord(r.e0, st), "hipEventRecord");
  m->trace.push_back(r);
  return IVX_OK;
}

int trace_end(ivx_model *m, hipStream_t st) {
  if (!m->trace_on || m->trace_skip) return IVX_OK;
  M_HIP(hipEventRecord(m->trace.back().e1, st), "hipEventRecord");
  return IVX_OK;
}

// The anchor grid of the tail for this (H, W): generated (or taken from ivx_weights_load("anchors")) and uploaded on first use of
// a grid; the previous grid's device buffer is reused when it is large enough, freed otherwise.  Not on the steady-state path.
int ensure_anchors(ivx_model *m, const ivx_anchor_head_desc &d, hipStream_t st, const char *who) {
  if (m->anchors_h == d.H && m->anchors_w == d.W && m->anchors_dev) return IVX_OK;
  const size_t n = (size_t)d.H * d.W * d.num_anchors * 7;
  if (!m->anchors_given.empty()) {
    M_REQUIRE(m->anchors_given.size() == n, "%s: the loaded anchors have %zu values, the %d x %d grid needs %zu", who, m->anchors_given.size(),
              d.H, d.W, n);
    m->anchors_host = m->anchors_given;
  } else {
    make_anchors(m->cfg, d.H, d.W, &m->anchors_host);
  }
  M_HIP(hipStreamSynchronize(st), "hipStreamSynchronize (anchors)");     // a forward in flight may still read the old grid
  if (m->anchors_dev && m->anchors_cap < n) {
    auto it = std::find(m->owned.begin(), m->owned.end(), (void *)m->anchors_dev);
    if (it != m->owned.end()) m->owned.erase(it);
    (void)hipFree(m->anchors_dev);
    m->anchors_dev = nullptr;
  }
  if (!m->anchors_dev) {
    void *p = nullptr;
    M_HIP(hipMalloc(&p, std::max<size_t>(n, 1) * sizeof(float)), "hipMalloc (anchors)");
    m->owned.push_back(p);
    m->anchors_dev = (float *)p;
    m->anchors_cap = n;
  }
  M_HIP(hipMemcpyAsync(m->anchors_dev, m->anchors_host.data(), n * sizeof(float), hipMemcpyHostToDevice, st), "hipMemcpyAsync (anchors)");
  M_HIP(hipStreamSynchronize(st), "hipStreamSynchronize (anchors)");     // anchors_host may be rewritten by the next grid
  m->anchors_h = d.H; m->anchors_w = d.W;
  return IVX_OK;
}

int run_steps(ivx_model *m, const Plan &pl, Range r, const Bind &bd, void *workspace, int64_t workspace_bytes, hipStream_t st,
              const char *who) {
  M_REQUIRE(workspace && ((uintptr_t)workspace & 255) == 0, "%s: workspace must be non-null and 256-byte aligned", who);
  if (workspace_bytes < pl.total) {
    ivx_set_error("%s: workspace too small (%lld < %lld); size it with the matching *_workspace_bytes call", who, (long long)workspace_bytes,
                  (long long)pl.total);
    return IVX_ERR_WORKSPACE;
  }
  char *base = (char *)workspace;
  void *ws = base + pl.ws_off;
  auto ptr = [&](int t) -> void * {
    auto it = bd.ext.find(t);
    if (it != bd.ext.end()) return it->second;
    return base + pl.t[t].off;
  };
  // scalar block of a pair-chained tensor: IVX_AMAX_SLOTS words of max |tensor|, then its scale
  auto slotp = [&](int t) -> uint32_t * { return (t >= 0 && pl.t[t].slot >= 0) ? (uint32_t *)(base + pl.scal_off + pl.t[t].slot * 512) : nullptr; };
  auto scalep = [&](int t) -> float * { uint32_t *q = slotp(t); return q ? (float *)(q + IVX_AMAX_SLOTS) : nullptr; };
  if (pl.scal_bytes > 0 && r.s0 <= m->trunk0 && r.s1 > m->trunk0)
    M_HIP(hipMemsetAsync(base + pl.scal_off, 0, (size_t)pl.scal_bytes, st), "hipMemsetAsync (amax slots)");
  const bool span2d = m->trace_on && m->trace_level == 1 && m->cfg.with_trunk;     // coarse tracing: the trunk is one span
  std::vector<char> forked((size_t)std::max(pl.n_sides, 1), 0);      // sites whose shortcut conv went to the side stream in this run
  double span_flops = 0.0, span_bytes = 0.0;     // products issued / bytes every launch of the span must move (inputs + outputs + filters, as executed)
  for (int i = r.s0; i < r.s1; ++i) {
    const Step &s = m->steps[i];
    const TInfo &in = pl.t[s.in];
    const bool in_span = span2d && i >= m->trunk0 && i < m->trunk1;
    if (in_span && i == m->trunk0) M_TRY(trace_begin(m, i, 6, 0, 0.0, 0.0, "2-D trunk (ResNet-50 + FPN level 0)", st));
    const bool was_on = m->trace_on;
    if (in_span) {
      m->trace_on = false;                       // no per-launch events inside the span
      {   // algorithmic bytes of the launch this step issues (a step covered by a one-launch kernel issues none)
        const PlanStep &ps = pl.ps[i];
        auto tb = [&](int t) { return t >= 0 ? (double)pl.t[t].elems() * pl.t[t].esz : 0.0; };
        if (s.kind == ST_CONV && ps.fuse != 2) {
          const ConvLayer &L = m->layers[s.layer];
          if (ps.fuse == 1) {
            const ConvLayer &L2 = m->layers[m->steps[i + 1].layer], &L3 = m->layers[m->steps[i + 2].layer];
            span_bytes += tb(s.in) + tb(ps.fuse_out) + 4.0 * ((double)L.cout * L.cin + 9.0 * L2.cout * L2.cin + (double)L3.cout * L3.cin);
          } else if (ps.fuse == 5) {
            const ConvLayer &L2 = m->layers[m->steps[i + 1].layer], &L3 = m->layers[m->steps[i + 2].layer];
            span_bytes += tb(s.in) + tb(ps.fuse_out) + 4.0 * ((double)L.cout * L.cin + 9.0 * L2.cout * L2.cin + (double)L3.cout * (L3.cin + L.cin));
          } else {
            span_bytes += tb(s.in) + tb(s.out) + tb(s.res) + (double)pl.t[s.in].esz * L.cout * L.cin_pad * L.k[0] * L.k[1] * L.k[2];
          }
        } else if (s.kind == ST_MAXPOOL && ps.fuse == 4) {
          span_bytes += tb(s.out) + tb(m->t_img);      // the one-launch head reads the image, writes the pooled map
        } else if (s.kind == ST_IMG2CL && ps.fuse == 3) {
          span_bytes += tb(s.in);                      // the amax pass
        } else if (s.kind != ST_CONV && s.kind != ST_LAYOUT) {
          span_bytes += tb(s.in) + tb(s.out) + tb(s.res);
        }
      }
      if (s.kind == ST_CONV) {
        const PlanStep &ps = pl.ps[i];
        const TInfo &o = pl.t[s.out];
        const ConvLayer &L = m->layers[s.layer];
        if (ps.tile) {
          int32_t a_, b_, zo;
          M_TRY(ivx_conv_out_dims(&ps.d, &a_, &b_, &zo));
          const int n = ps.tile + 2;
          const double tiles = (double)o.B * ((ps.d.D + 2 * ps.d.pd - 2 + ps.tile - 1) / ps.tile) * ((ps.d.H + 2 * ps.d.ph - 2 + ps.tile - 1) / ps.tile);
          span_flops += (ps.d.wino_operands ? 3.0 : 1.0) * ivx_conv_winograd_issued_fraction(&ps.d) * 2.0 * n * n * tiles * zo * ps.d.Cout * ps.d.KW * ps.d.Cin;   // as the per-launch records count
        } else {
          span_flops += ((ps.pio || (ps.fuse == 2 && L.wstem)) ? 3.0 : 1.0) * 2.0 * o.elems() * L.cin * L.k[0] * L.k[1] * L.k[2];      // pair form (the one-launch stem too): three products per multiply-add
        }
      }
    }
    struct Restore { ivx_model *m; bool on; ~Restore() { m->trace_on = on; } } restore{m, was_on};   // also on an error return
    switch (s.kind) {
      case ST_IMG2CL:
        if (pl.ps[i].fuse == 3) {              // one-launch stem: only the image's maximum is needed here
          M_TRY(ivx_amax_f32((const float *)ptr(s.in), in.elems(), slotp(s.out), st));
          break;
        }
        if (slotp(s.out))
          M_TRY(ivx_nchw_to_nhwc_amax((const float *)ptr(s.in), in.B, 3, (int64_t)in.H * in.W, 4, (float *)ptr(s.out), slotp(s.out), st));
        else
          M_TRY(ivx_nchw_to_nhwc((const float *)ptr(s.in), in.B, 3, (int64_t)in.H * in.W, 4, (float *)ptr(s.out), st));
        break;
      case ST_IMG_S2D:
        M_TRY(ivx_image_s2d_bf16((const float *)ptr(s.in), in.B, in.H, in.W, ptr(s.out), st));
        break;
      case ST_MAXPOOL:
        if (m->cfg.storage == IVX_BF16) {
          M_TRY(ivx_maxpool2d_fwd_bf16(ptr(s.in), in.B, in.H, in.W, in.C, 3, 2, 1, ptr(s.out), st));
          break;
        }
        if (pl.ps[i].pio) {        // fp32 stem output -> pair tensor, scaled by the bound of the stem's output from the image's maximum
          const ConvLayer &Ls = m->layers[pl.ps[i].bound_layer];
          int t_img4 = -1;
          for (int j = r.s0; j < i; ++j)
            if (m->steps[j].out == s.in) t_img4 = m->steps[j].in;
          M_REQUIRE(slotp(t_img4), "internal: the pair max-pool needs the image's amax slots");
          if (pl.ps[i].fuse == 4) {            // layout change + stem + max-pool in one launch, from the caller's NCHW image
            int t_img = -1;
            for (int j = r.s0; j < i; ++j)
              if (m->steps[j].out == t_img4) t_img = m->steps[j].in;
            M_REQUIRE(t_img >= 0, "internal: the one-launch stem needs the image");
            const TInfo &im = pl.t[t_img];
            const double px = (double)in.elems() / in.C;
            M_TRY(trace_begin(m, i, 0, 0, 3.0 * 2.0 * px * 64 * 147, 0.0, Ls.name + " + max-pool (one launch)", st));
            M_TRY(ivx_stem_pool_fwd_pair((const float *)ptr(t_img), im.B, im.H, im.W, Ls.wstem, Ls.scale_p, Ls.shift, Ls.wbound, Ls.sbound, slotp(t_img4),
                                         ptr(s.out), scalep(s.out), slotp(s.out), st));
            M_TRY(trace_end(m, st));
            break;
          }
          M_TRY(ivx_maxpool2d_fwd_pair((const float *)ptr(s.in), in.B, in.H, in.W, in.C, 3, 2, 1, ptr(s.out), slotp(t_img4), Ls.wbound, Ls.sbound,
                                       scalep(s.out), slotp(s.out), st));
        } else
          M_TRY(ivx_maxpool2d_fwd((const float *)ptr(s.in), in.B, in.H, in.W, in.C, 3, 2, 1, (float *)ptr(s.out), st));
        break;
      case ST_UPSAMPLE:
        if (m->cfg.storage == IVX_BF16)
          M_TRY(ivx_upsample_trilinear2x_fwd_bf16(ptr(s.in), in.B, in.D, in.H, in.W, in.C, ptr(s.out), st));
        else
          M_TRY(ivx_upsample_trilinear2x_fwd((const float *)ptr(s.in), in.B, in.D, in.H, in.W, in.C, (float *)ptr(s.out), st));
        break;
      case ST_CONV: {
        const ConvLayer &L = m->layers[s.layer];
        const PlanStep &ps = pl.ps[i];
        if (ps.fuse == 2) break;               // conv2 / conv3 of a bottleneck that conv1's step ran
        if (ps.fuse == 5) {                    // conv1's step of the one-launch projection block (the shortcut conv's step i - 1 was skipped)
          const ConvLayer &L2 = m->layers[m->steps[i + 1].layer], &L3 = m->layers[m->steps[i + 2].layer];
          const int t_out = ps.fuse_out;
          ivx_bottleneck_desc bd = {in.B, in.H, in.W, L.cout};
          ivx_bottleneck_io io;
          memset(&io, 0, sizeof(io));
          io.in_scale = scalep(s.in); io.amax_in = slotp(s.in); io.out_scale = scalep(t_out); io.amax_out = slotp(t_out);
          io.wbound[0] = L.wbound; io.sbound[0] = L.sbound; io.wbound[1] = L2.wbound; io.sbound[1] = L2.sbound;
          io.wbound[2] = L3.proj_wb3; io.sbound[2] = L3.proj_sb;
          const double px = (double)in.elems() / in.C;
          M_TRY(trace_begin(m, i, 0, 0, 3.0 * 2.0 * px * ((double)L.cin * L.cout + 13.0 * L.cout * L.cout + 4.0 * L.cout * L.cin),
                            4.0 * in.elems() + 4.0 * px * 4.0 * L.cout, L.name + " ..conv3+ds (one launch)", st));
          M_TRY(ivx_bottleneck_proj_fwd_pio(&bd, L.cin, &io, L3.proj_wbd, ptr(s.in), L.wpair, L.scale_p, L.shift, L2.wpair, L2.scale_p, L2.shift, L3.wproj,
                                            L3.scale_proj, L3.shift_proj, ptr(t_out), st));
          M_TRY(trace_end(m, st));
          break;
        }
        if (ps.fuse == 1) {
          const ConvLayer &L2 = m->layers[m->steps[i + 1].layer], &L3 = m->layers[m->steps[i + 2].layer];
          const int t_out = ps.fuse_out;
          ivx_bottleneck_desc bd = {in.B, in.H, in.W, L.cout};
          ivx_bottleneck_io io;
          memset(&io, 0, sizeof(io));
          io.in_scale = scalep(s.in); io.amax_in = slotp(s.in); io.out_scale = scalep(t_out); io.amax_out = slotp(t_out);
          io.wbound[0] = L.wbound; io.sbound[0] = L.sbound; io.wbound[1] = L2.wbound; io.sbound[1] = L2.sbound;
          io.wbound[2] = L3.wbound; io.sbound[2] = L3.sbound;
          const double px = (double)in.elems() / in.C;
          M_TRY(trace_begin(m, i, 0, 0, 3.0 * 2.0 * px * (17.0 * L.cout * L.cout), 2.0 * 4.0 * in.elems(), L.name + " .. conv3 (one launch)", st));
          M_TRY(ivx_bottleneck_fwd_pio(&bd, &io, ptr(s.in), L.wpair, L.scale_p, L.shift, L2.wpair, L2.scale_p, L2.shift, L3.wpair, L3.scale_p, L3.shift,
                                       ptr(t_out), st));
          M_TRY(trace_end(m, st));
          break;
        }
        const void *res = s.res >= 0 ? ptr(s.res) : nullptr;
        const TInfo &o = pl.t[s.out];
        const int is3d = in.D > 1 && in.W > 1;      // a 3-D neck layer (the head conv sees [B,X',Y',1,C])
        // the shortcut conv of a block runs on the side stream next to conv1 / conv2 (with per-launch tracing everything stays on `st`)
        hipStream_t cst = st;
        void *cws = ws;
        if (ps.side > 0 && !(m->trace_on && m->trace_level != 3) && pl.ws2_off > 0) {
          if (!m->side) {
            M_HIP(hipStreamCreateWithFlags(&m->side, hipStreamNonBlocking), "hipStreamCreateWithFlags");
            M_HIP(hipEventCreateWithFlags(&m->ev_fork, hipEventDisableTiming), "hipEventCreateWithFlags");
          }
          while ((int)m->ev_join.size() < pl.n_sides) {
            hipEvent_t e;
            M_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming), "hipEventCreateWithFlags");
            m->ev_join.push_back(e);
          }
          M_HIP(hipEventRecord(m->ev_fork, st), "hipEventRecord");
          M_HIP(hipStreamWaitEvent(m->side, m->ev_fork, 0), "hipStreamWaitEvent");
          cst = m->side;
          cws = base + pl.ws2_off;
          forked[ps.side - 1] = 1;
        }
        if (ps.join > 0 && forked[ps.join - 1]) {
          M_HIP(hipStreamWaitEvent(st, m->ev_join[ps.join - 1], 0), "hipStreamWaitEvent");
          forked[ps.join - 1] = 0;
        }
        if (ps.tile && m->trace_on) {   // the three stages as separate launches so each gets its own pair of events
          const int n = ps.tile + 2, tiles = o.B * ((ps.d.D + 2 * ps.d.pd - 2 + ps.tile - 1) / ps.tile) * ((ps.d.H + 2 * ps.d.ph - 2 + ps.tile - 1) / ps.tile);
          int32_t zo_d, zo_h, zo;
          M_TRY(ivx_conv_out_dims(&ps.d, &zo_d, &zo_h, &zo));
          const double vb = 4.0 * n * n * tiles * ps.d.W * ps.d.Cin, mb = 4.0 * n * n * tiles * zo * ps.d.Cout;
          M_TRY(trace_begin(m, i, 1, is3d, 0.0, 4.0 * in.elems() + vb, L.name, st));
          M_TRY(ivx_conv_winograd_input_amax(&ps.d, ps.tile, ptr(s.in), ws, pl.ws_bytes, ps.amax_in >= 0 ? (const float *)(base + ps.amax_in) : nullptr,
                                             ps.amax_in_n, st));
          M_TRY(trace_end(m, st));
          // flops: the matrix-core products the stage issues (pair operands: hi*hi + hi*lo + lo*hi per multiply-add)
          M_TRY(trace_begin(m, i, 2, is3d, (ps.d.wino_operands ? 3.0 : 1.0) * ivx_conv_winograd_issued_fraction(&ps.d) * 2.0 * n * n * tiles * zo * ps.d.Cout * ps.d.KW * ps.d.Cin,
                            vb + mb, L.name, st));      // (the z-blocked tile skips the taps outside a 3-slice column)
          M_TRY(ivx_conv_winograd_gemm(&ps.d, ps.tile, L.u.at(ps.tile * 8 + ps.d.wino_operands), ws, pl.ws_bytes, st));
          M_TRY(trace_end(m, st));
          M_TRY(trace_begin(m, i, 3, is3d, 0.0, mb + 4.0 * o.elems() * (res ? 2 : 1), L.name, st));
          M_TRY(ivx_conv_winograd_output_amax(&ps.d, ps.tile, L.scale, L.shift, res, ptr(s.out), ws, pl.ws_bytes,
                                              ps.amax_out >= 0 ? (float *)(base + ps.amax_out) : nullptr, st));
          M_TRY(trace_end(m, st));
          break;
        }
        M_TRY(trace_begin(m, i, 0, is3d, ((ps.pio || ps.split) ? 3.0 : 1.0) * 2.0 * o.elems() * L.cin * L.k[0] * L.k[1] * L.k[2], 0.0, L.name, st));
        if (ps.pio) {
          ivx_pair_io io;
          memset(&io, 0, sizeof(io));
          io.in_scale = scalep(s.in);
          io.res_dtype = s.res >= 0 ? pl.t[s.res].fmt : 0;
          io.res_scale = io.res_dtype == IVX_F16_PAIR ? scalep(s.res) : nullptr;
          io.out_scale = scalep(s.out);
          io.amax_in = slotp(s.in); io.amax_res = slotp(s.res); io.amax_out = slotp(s.out);
          io.wbound = L.wbound; io.sbound = L.sbound;
          M_TRY(ivx_conv_fwd_pio(&ps.d, &io, ptr(s.in), L.wpair, L.scale_p, L.shift, res, ptr(s.out), cws, pl.ws_bytes, cst));
        } else if (ps.tile) {
          M_TRY(ivx_conv_winograd_input_amax(&ps.d, ps.tile, ptr(s.in), ws, pl.ws_bytes, ps.amax_in >= 0 ? (const float *)(base + ps.amax_in) : nullptr,
                                             ps.amax_in_n, st));
          M_TRY(ivx_conv_winograd_gemm(&ps.d, ps.tile, L.u.at(ps.tile * 8 + ps.d.wino_operands), ws, pl.ws_bytes, st));
          M_TRY(ivx_conv_winograd_output_amax(&ps.d, ps.tile, L.scale, L.shift, res, ptr(s.out), ws, pl.ws_bytes,
                                              ps.amax_out >= 0 ? (float *)(base + ps.amax_out) : nullptr, st));
        } else if (m->fp8_on && L.fp8_eff) {
          M_TRY(ivx_conv_fwd_ws(&ps.d, ptr(s.in), L.fp8_eff >= 2 ? L.wq : L.w, L.scale_q, L.shift_q, res, ptr(s.out), cws, pl.ws_bytes, cst));
        } else if (ps.split) {     // (hi, lo) bf16 copy of the input at the start of the workspace, then the three-product kernel
          M_TRY(ivx_bf16_pair_split((const float *)ptr(s.in), in.elems(), cws, cst));
          M_TRY(ivx_conv_fwd_ws(&ps.d, cws, L.wsplit, L.scale, L.shift, res, ptr(s.out), (char *)cws + ps.split, pl.ws_bytes - ps.split, cst));
        } else {
          M_TRY(ivx_conv_fwd_ws(&ps.d, ptr(s.in), L.w, L.scale, L.shift, res, ptr(s.out), cws, pl.ws_bytes, cst));
          if (m->calib_dev && (L.fp8_eff == 1 || L.fp8_eff == 2))       // calibration pass: max |output| of the tensors that will be e4m3
            M_TRY(ivx_amax_bf16(ptr(s.out), o.elems(), m->calib_dev + s.layer, st));
        }
        M_TRY(trace_end(m, st));
        if (cst != st) M_HIP(hipEventRecord(m->ev_join[ps.side - 1], cst), "hipEventRecord");
        break;
      }
      case ST_LIFT: {
        M_REQUIRE(bd.proj && bd.new_origin && bd.crop, "%s: proj / new_origin / crop_hw are required", who);
        const TInfo &o = pl.t[s.out];
        M_TRY(trace_begin(m, i, 4, 1, 0.0, (double)in.esz * in.elems() + (double)o.esz * o.elems() + (double)o.elems() / o.C, "unprojection", st));
        if (m->cfg.storage == IVX_BF16) {
          // one view: the lift is a gather-copy (no arithmetic on the features: the reference divides by a count of 1), so a bf16 map
          // with C channels goes through the fp32 kernel as C / 2 32-bit words; several views: sum and division in fp32 (bf16 kernel).
          // (V == 1 takes backproject_single_view_kernel, which only loads and stores the words -- no add, no divide, so no denormal
          // flush, no -0 -> +0 and no Inf/NaN arithmetic on the bit patterns of bf16 pairs: a bit copy whatever the float mode.)
          if (bd.V == 1 && in.C % 8 == 0)
            M_TRY(ivx_backproject_mean_fwd((const float *)ptr(s.in), o.B, 1, in.H, in.W, in.C / 2, bd.proj, bd.new_origin, bd.crop, m->cfg.voxel_size, o.D,
                                           o.H, o.W, (float *)ptr(s.out), (uint8_t *)ptr(s.out2), st));
          else
            M_TRY(ivx_backproject_mean_fwd_bf16(ptr(s.in), o.B, bd.V, in.H, in.W, in.C, bd.proj, bd.new_origin, bd.crop, m->cfg.voxel_size, o.D, o.H, o.W,
                                                ptr(s.out), (uint8_t *)ptr(s.out2), st));
          M_TRY(trace_end(m, st));
          break;
        }
        M_TRY(ivx_backproject_mean_fwd_amax((const float *)ptr(s.in), o.B, bd.V, in.H, in.W, in.C, bd.proj, bd.new_origin, bd.crop, m->cfg.voxel_size,
                                            o.D, o.H, o.W, (float *)ptr(s.out), (uint8_t *)ptr(s.out2),
                                            pl.ps[i].amax_out >= 0 ? (float *)(base + pl.ps[i].amax_out) : nullptr, st));
        M_TRY(trace_end(m, st));
        break;
      }
      case ST_DCN_COL: {
        const TInfo &om = pl.t[s.res];
        M_TRY(trace_begin(m, i, 0, 0, 0.0, 4.0 * pl.t[s.out].elems(), "dcn columns", st));
        if (pl.ps[i].pio)            // inside the pair chain: pair map in, pair columns out (same scale)
          M_TRY(ivx_dcn_im2col_fwd_pair(ptr(s.in), scalep(s.in), (const float *)ptr(s.res), in.B, in.H, in.W, in.C, 3, 3, s.aux, 1, 1, om.C,
                                        ptr(s.out), scalep(s.out), slotp(s.out), st));
        else
          M_TRY(ivx_dcn_im2col_fwd((const float *)ptr(s.in), (const float *)ptr(s.res), in.B, in.H, in.W, in.C, 3, 3, s.aux, 1, 1, om.C,
                                   (float *)ptr(s.out), st));
        M_TRY(trace_end(m, st));
        break;
      }
      case ST_AVGPOOL:
        M_TRY(ivx_global_avgpool_fwd((const float *)ptr(s.in), in.B, (int64_t)in.D * in.H * in.W, in.C, (float *)ptr(s.out), st));
        break;
      case ST_LAYOUT: {
        // LayoutHead._forward_single (layout_head.py:52-74) + the projection from the PREDICTED angles (detectors/imvoxelnet.py:59-61,
        // 121-124, 164-187).  The one place where the forward returns to the host: 9 numbers per sample, as in the reference (its
        // extrinsics are built from `angles` with CPU ops).
        if (!bd.metas && r.s1 == m->trunk1) break;      // ivx_backbone_fpn_fwd: the FPN map does not depend on the LayoutHead
        M_REQUIRE(bd.metas && bd.proj_dev, "%s: a handle with a LayoutHead runs through ivx_model_detect (the projection is built from its angles)", who);
        const int B = in.B;
        std::vector<float> a((size_t)B * 2), l((size_t)B * 7), pj((size_t)B * 12);
        M_HIP(hipMemcpyAsync(a.data(), ptr(s.in), a.size() * 4, hipMemcpyDeviceToHost, st), "hipMemcpyAsync (angles)");
        M_HIP(hipMemcpyAsync(l.data(), ptr(s.res), l.size() * 4, hipMemcpyDeviceToHost, st), "hipMemcpyAsync (layout)");
        M_HIP(hipStreamSynchronize(st), "hipStreamSynchronize (LayoutHead)");
        for (int b = 0; b < B; ++b) {
          float ang[2], E[16];
          M_TRY(ivx_layout_head_decode(&a[(size_t)b * 2], &l[(size_t)b * 7], ang, bd.layout_host ? bd.layout_host + (size_t)b * 7 : nullptr));
          if (bd.angles_host) { bd.angles_host[(size_t)b * 2] = ang[0]; bd.angles_host[(size_t)b * 2 + 1] = ang[1]; }
          M_TRY(ivx_layout_extrinsics(ang, E));
          const ivx_sample_meta &mt = bd.metas[b];
          M_TRY(ivx_compute_projection(mt.intrinsic, E, 1, (double)mt.ori_h / ((double)mt.img_h / 4.0), &pj[(size_t)b * 12]));
        }
        M_HIP(hipMemcpyAsync(bd.proj_dev, pj.data(), pj.size() * 4, hipMemcpyHostToDevice, st), "hipMemcpyAsync (projection)");
        M_HIP(hipStreamSynchronize(st), "hipStreamSynchronize (projection)");      // pj dies at the end of this block
        break;
      }
      case ST_FCOS: {
        const ivx_model_cfg &c = m->cfg;
        const int l = s.aux, R = c.head_type == IVX_HEAD_SCANNET ? 6 : 7;
        const TInfo &vol = pl.t[m->t_volume];
        M_REQUIRE(bd.lvl_vs[l] && bd.lvl_no[l], "%s: the head levels need their geometry (ivx_model_detect)", who);
        M_TRY(trace_begin(m, i, 5, 0, 0.0, 0.0, "indoor head candidates", st));
        M_TRY(ivx_fcos_head_level_candidates((const float *)ptr(s.in), (const uint8_t *)ptr(s.res), bd.lvl_vs[l], bd.lvl_no[l], m->head_scales[l], in.B,
                                             in.D, in.H, in.W, in.C, c.head_classes, R, l, vol.D, vol.H, vol.W, c.head_nms_pre, ws, pl.ws_bytes,
                                             (float *)ptr(s.out), (float *)ptr(s.out2), (int32_t *)ptr(m->t_cc[l]), st));
        M_TRY(trace_end(m, st));
        break;
      }
      case ST_INDOOR_TAIL: {
        M_REQUIRE(bd.boxes && bd.scores && bd.labels && bd.count, "%s: output buffers are required", who);
        const float *cb[4] = {nullptr, nullptr, nullptr, nullptr}, *cs[4] = {nullptr, nullptr, nullptr, nullptr};
        for (size_t l = 0; l < m->t_cb.size(); ++l) { cb[l] = (const float *)ptr(m->t_cb[l]); cs[l] = (const float *)ptr(m->t_cs[l]); }
        M_TRY(trace_begin(m, i, 5, 0, 0.0, 0.0, "indoor tail", st));
        M_TRY(ivx_indoor_tail_get_bboxes(&pl.itail, cb, cs, ws, pl.ws_bytes, bd.boxes, bd.scores, bd.labels, bd.count, st));
        M_TRY(trace_end(m, st));
        break;
      }
      case ST_TAIL: {
        const ivx_anchor_head_desc &d = pl.tail;
        M_TRY(ensure_anchors(m, d, st, who));          // no-op on the steady-state path
        M_REQUIRE(bd.boxes && bd.scores && bd.labels && bd.count, "%s: output buffers are required", who);
        M_TRY(trace_begin(m, i, 5, 0, 0.0, 0.0, "anchor tail", st));
        M_TRY(ivx_anchor_head_get_bboxes(&d, (const float *)ptr(s.in), m->anchors_dev, ws, pl.ws_bytes, bd.boxes, bd.scores, bd.labels,
                                         bd.count, nullptr, nullptr, nullptr, st));
        M_TRY(trace_end(m, st));
        break;
      }
    }
    m->trace_on = was_on;
    if (in_span && i == m->trunk1 - 1) {         // close the trunk span: its record is the last one opened at level 1
      for (auto it = m->trace.rbegin(); it != m->trace.rend(); ++it)
        if (it->stage == 6) {
          it->flops = span_flops;
          it->bytes = span_bytes;
          M_HIP(hipEventRecord(it->e1, st), "hipEventRecord");
          break;
        }
    }
  }
  return IVX_OK;
}

}  // namespace

// ================================================================================================ C-ABI
extern "C" int ivx_create(const ivx_model_cfg *cfg, ivx_model **out) {
  M_REQUIRE(cfg && out, "ivx_create: null argument");
  M_REQUIRE(cfg->neck_type >= IVX_NECK_KITTI && cfg->neck_type <= IVX_NECK_UNET, "ivx_create: neck_type must be IVX_NECK_KITTI | NUSCENES | FAST | UNET");
  const bool indoor = cfg->neck_type == IVX_NECK_FAST || cfg->neck_type == IVX_NECK_UNET;
  if (cfg->neck_type == IVX_NECK_FAST)
    M_REQUIRE(cfg->fast_n_blocks[0] >= 1 && cfg->fast_n_blocks[1] >= 1 && cfg->fast_n_blocks[2] >= 1, "ivx_create: fast_n_blocks must be >= 1 per level");
  if (cfg->neck_type == IVX_NECK_UNET) {
    M_REQUIRE(cfg->unet_channels[0] == cfg->fpn_channels, "ivx_create: unet_channels[0] must equal fpn_channels");
    const int nu = cfg->unet_channels[3] > 0 ? 4 : 3;     // 4 scales (the reference configs) or 3 (unet_channels[3] = 0)
    for (int i = 0; i < nu; ++i) M_REQUIRE(cfg->unet_channels[i] > 0 && cfg->unet_channels[i] % 4 == 0 && cfg->unet_down_layers[i] >= 0, "ivx_create: bad unet_channels / unet_down_layers");
    for (int i = 0; i < nu - 1; ++i) M_REQUIRE(cfg->unet_up_layers[i] >= 0, "ivx_create: bad unet_up_layers");
  }
  M_REQUIRE(cfg->fpn_channels > 0 && cfg->fpn_channels % 4 == 0 && cfg->neck_out_channels > 0 && cfg->neck_out_channels % 4 == 0,
            "ivx_create: channel counts must be positive multiples of 4");
  M_REQUIRE(cfg->n_voxels[0] > 0 && cfg->n_voxels[1] > 0 && cfg->n_voxels[2] > 0, "ivx_create: bad n_voxels");
  M_REQUIRE(indoor || (cfg->num_classes >= 1 && cfg->n_sizes >= 1 && cfg->n_sizes <= 4 && cfg->n_rotations >= 1 && cfg->n_rotations <= 4),
            "ivx_create: 1..4 anchor sizes / rotations, >= 1 class");
  M_REQUIRE(indoor || (cfg->nms_pre > 0 && cfg->max_num > 0), "ivx_create: nms_pre and max_num must be positive");
  M_REQUIRE(cfg->wino_operands == IVX_F32 || cfg->wino_operands == IVX_F16_PAIR, "ivx_create: wino_operands IVX_F32 | IVX_F16_PAIR");
  M_REQUIRE(cfg->trunk_operands == IVX_F32 || cfg->trunk_operands == IVX_F16_PAIR, "ivx_create: trunk_operands IVX_F32 | IVX_F16_PAIR");
  M_REQUIRE(cfg->storage == IVX_F32 || cfg->storage == IVX_BF16, "ivx_create: storage IVX_F32 | IVX_BF16");
  if (cfg->storage == IVX_BF16) {
    M_REQUIRE(!cfg->layout_head && !cfg->dcn_stages[0] && !cfg->dcn_stages[1] && !cfg->dcn_stages[2] && !cfg->dcn_stages[3],
              "ivx_create: bf16 storage is not built for the DCNv2 stages / the LayoutHead");
    M_REQUIRE(cfg->fpn_channels % 8 == 0 && cfg->neck_out_channels % 8 == 0, "ivx_create: bf16 storage needs channel counts that are multiples of 8");
  }
  M_REQUIRE(cfg->winograd_tile == 0 || cfg->winograd_tile == 2 || cfg->winograd_tile == 4 || cfg->winograd_tile == 6, "ivx_create: winograd_tile 0 | 2 | 4 | 6");
  M_REQUIRE(cfg->head_type >= IVX_HEAD_NONE && cfg->head_type <= IVX_HEAD_SUNRGBD, "ivx_create: head_type 0 (none) | IVX_HEAD_SCANNET | IVX_HEAD_SUNRGBD");
  if (cfg->head_type != IVX_HEAD_NONE) {
    M_REQUIRE(indoor, "ivx_create: the anchor-free heads sit on the FAST / UNET necks (the stack necks end in the Anchor3DHead)");
    M_REQUIRE(cfg->head_classes >= 1 && (cfg->head_type == IVX_HEAD_SCANNET || cfg->head_classes <= 64), "ivx_create: bad head_classes");
    M_REQUIRE(cfg->head_nms_pre > 0 && cfg->head_nms_pre <= 65536, "ivx_create: head_nms_pre must be in 1..65536");
  }
  M_REQUIRE(!cfg->layout_head || (cfg->with_trunk && cfg->layout_linear_size > 0), "ivx_create: the LayoutHead needs the trunk and a positive layout_linear_size");
  for (int i = 0; i < 4; ++i) M_REQUIRE(cfg->dcn_stages[i] == 0 || cfg->with_trunk, "ivx_create: dcn_stages without the trunk");
  ivx_model *m = new ivx_model();
  m->cfg = *cfg;
  build_graph(m);
  *out = m;
  return IVX_OK;
}

extern "C" int ivx_destroy(ivx_model *m) {
  if (!m) return IVX_OK;
  for (void *p : m->owned) (void)hipFree(p);
  for (hipEvent_t e : m->event_pool) (void)hipEventDestroy(e);
  for (hipEvent_t e : m->ev_join) (void)hipEventDestroy(e);
  if (m->ev_fork) (void)hipEventDestroy(m->ev_fork);
  if (m->side) (void)hipStreamDestroy(m->side);
  delete m;
  return IVX_OK;
}

extern "C" int ivx_weights_load(ivx_model *m, const char *key, const float *data, const int64_t *shape, int32_t ndim) {
  M_REQUIRE(m && key && data && (shape || ndim == 0) && ndim >= 0 && ndim <= 6, "ivx_weights_load: bad argument");
  M_REQUIRE(!m->finalized || !strcmp(key, "anchors"), "ivx_weights_load: the model is finalized; create a new handle to change weights");
  HostTensor t;
  int64_t n = 1;
  for (int i = 0; i < ndim; ++i) {
    M_REQUIRE(shape[i] > 0, "ivx_weights_load: %s: non-positive dimension", key);
    t.shape.push_back(shape[i]);
    n *= shape[i];
  }
  t.data.assign(data, data + n);
  if (!strcmp(key, "anchors")) {
    M_REQUIRE(ndim == 2 && shape[1] == 7, "ivx_weights_load: anchors must be [n, 7]");
    m->anchors_given = t.data;
    m->anchors_h = m->anchors_w = 0;      // re-upload on the next forward
    return IVX_OK;
  }
  std::string k = key;
  if (k.rfind("module.", 0) == 0) k = k.substr(7);     // DataParallel prefix (mmcv load_checkpoint strips it)
  m->weights[k] = std::move(t);
  return IVX_OK;
}

extern "C" int ivx_weights_finalize(ivx_model *m, ivx_stream_t stream) {
  M_REQUIRE(m, "ivx_weights_finalize: null handle");
  M_REQUIRE(!m->finalized, "ivx_weights_finalize: already finalized");
  std::string missing;
  for (const ConvLayer &L : m->layers) {          // completeness first: no device work for an incomplete state dict
    for (size_t q = 0; q < L.w_keys.size(); ++q) {
      if (!find_w(m, L.w_keys[q])) missing += L.w_keys[q] + " ";
      if (!L.b_keys[q].empty() && !find_w(m, L.b_keys[q])) missing += L.b_keys[q] + " ";
    }
    if (!L.bn.empty())
      for (const char *sfx : {".weight", ".bias", ".running_mean", ".running_var"})
        if (!find_w(m, L.bn + sfx)) missing += L.bn + sfx + " ";
  }
  if (missing.empty())
    for (ConvLayer &L : m->layers) M_TRY(pack_layer(m, L, (hipStream_t)stream, &missing));
  if (!missing.empty()) {
    ivx_set_error("ivx_weights_finalize: missing state-dict keys: %.400s", missing.c_str());
    return IVX_ERR_INVALID_ARG;
  }
  for (ConvLayer &L3 : m->layers) {               // joint filter bank of conv3 + shortcut conv of the one-launch projection block
    if (L3.proj_peer < 0) continue;
    ConvLayer &Ld = m->layers[L3.proj_peer];
    if (L3.pair_ok && Ld.pair_ok && L3.cout == Ld.cout && L3.cout == 4 * L3.cin && !L3.w_tap.empty() && !Ld.w_tap.empty()) {
      const int P = L3.cin, Cin = Ld.cin;
      std::vector<uint16_t> packed((size_t)2 * 4 * P * (Cin + P));
      std::vector<float> sc(4 * P), sf(4 * P);
      M_TRY(ivx_bottleneck_proj_pack(L3.w_tap.data(), L3.scale_h.data(), L3.shift_h.data(), Ld.w_tap.data(), Ld.scale_h.data(), Ld.shift_h.data(), P, Cin,
                                     packed.data(), sc.data(), sf.data(), &L3.proj_wb3, &L3.proj_sb, &L3.proj_wbd));
      M_TRY(dev_upload_sync(m, reinterpret_cast<const float *>(packed.data()), packed.size() / 2, &L3.wproj, (hipStream_t)stream));
      M_TRY(dev_upload_sync(m, sc.data(), sc.size(), &L3.scale_proj, (hipStream_t)stream));
      M_TRY(dev_upload_sync(m, sf.data(), sf.size(), &L3.shift_proj, (hipStream_t)stream));
    }
    for (ConvLayer *L : {&L3, &Ld}) { std::vector<float>().swap(L->w_tap); std::vector<float>().swap(L->scale_h); std::vector<float>().swap(L->shift_h); }
  }
  for (size_t l = 0; l < m->t_headout.size() && l < 3; ++l) {     // mmcv Scale of the anchor-free heads (imvoxel_head_v2.py:78-80): exp(scale * reg)
    const HostTensor *sc = find_w(m, "bbox_head.scales." + std::to_string(l) + ".scale");
    if (!sc || sc->data.size() != 1) {
      ivx_set_error("ivx_weights_finalize: missing state-dict keys: bbox_head.scales.%d.scale", (int)l);
      return IVX_ERR_INVALID_ARG;
    }
    m->head_scales[l] = sc->data[0];
  }
  M_HIP(hipStreamSynchronize((hipStream_t)stream), "hipStreamSynchronize");   // host staging vectors die here
  m->weights.clear();
  std::vector<float>().swap(m->pack_a);
  std::vector<float>().swap(m->pack_b);
  m->finalized = true;
  return IVX_OK;
}

namespace {
int check_img(const ivx_model *m, int B, int V, int H, int W, const char *who) {
  M_REQUIRE(m, "%s: null handle", who);
  M_REQUIRE(B > 0 && V > 0 && H > 0 && W > 0, "%s: non-positive dims", who);
  M_REQUIRE(H % 32 == 0 && W % 32 == 0, "%s: the padded image must be a multiple of 32 (Pad(size_divisor=32)); got %d x %d", who, H, W);
  return IVX_OK;
}
}  // namespace

// ---- whole path
// whole = false: up to the anchor tail (stack necks) or the neck levels (indoor necks) -- ivx_model_forward / _forward_levels;
// whole = true: every step, i.e. including the anchor-free head and its tail -- ivx_model_detect
static int plan_forward(ivx_model *m, int B, int V, int H, int W, Plan **pl, Range *r, hipStream_t st, bool whole = false) {
  std::map<int, TInfo> in;
  const int end = (!whole && m->head0 >= 0) ? m->head0 : (int)m->steps.size();
  if (m->cfg.with_trunk) {
    in[m->t_img] = tinfo(B * V, 1, H, W, 3);
    *r = {m->trunk0, end};
  } else {
    in[m->t_fpn0] = tinfo(B * V, 1, H / 4, W / 4, m->cfg.fpn_channels, m->cfg.storage == IVX_BF16 ? 2 : 4);
    *r = {m->lift_step, end};
  }
  return get_plan(m, whole ? "detect" : "forward", *r, B, V, H, W, in, pl, st);
}

extern "C" int64_t ivx_model_workspace_bytes(ivx_model *m, int32_t B, int32_t V, int32_t H, int32_t W) {
  if (check_img(m, B, V, H, W, "ivx_model_workspace_bytes") != IVX_OK) return -1;
  Plan *pl; Range r;
  if (plan_forward(m, B, V, H, W, &pl, &r, nullptr) != IVX_OK) return -1;
  return pl->total;
}

extern "C" int ivx_model_forward(ivx_model *m, const float *input, int32_t B, int32_t V, int32_t H, int32_t W, const float *proj,
                                 const float *new_origin, const int32_t *crop_hw, void *workspace, int64_t workspace_bytes,
                                 float *out_boxes, float *out_scores, int64_t *out_labels, int32_t *out_count, uint8_t *out_valid,
                                 ivx_stream_t stream) {
  M_TRY(check_img(m, B, V, H, W, "ivx_model_forward"));
  M_REQUIRE(input, "ivx_model_forward: null input");
  M_REQUIRE(m->t_head >= 0, "ivx_model_forward: the handle holds an indoor neck (no anchor head); use ivx_model_forward_levels or ivx_model_detect");
  M_REQUIRE(!m->cfg.layout_head, "ivx_model_forward: a handle with a LayoutHead runs through ivx_model_detect");
  Plan *pl; Range r;
  M_TRY(plan_forward(m, B, V, H, W, &pl, &r, (hipStream_t)stream));
  Bind bd;
  bd.ext[m->cfg.with_trunk ? m->t_img : m->t_fpn0] = (void *)input;
  if (out_valid) bd.ext[m->t_valid] = out_valid;
  bd.proj = proj; bd.new_origin = new_origin; bd.crop = crop_hw; bd.V = V;
  bd.boxes = out_boxes; bd.scores = out_scores; bd.labels = out_labels; bd.count = out_count;
  return run_steps(m, *pl, r, bd, workspace, workspace_bytes, (hipStream_t)stream, "ivx_model_forward");
}

// ---- 2-D trunk alone
static int plan_trunk(ivx_model *m, int BV, int H, int W, Plan **pl, hipStream_t st) {
  M_REQUIRE(m->cfg.with_trunk, "ivx_backbone_fpn_fwd: the handle was created without the 2-D trunk (with_trunk = 0)");
  std::map<int, TInfo> in;
  in[m->t_img] = tinfo(BV, 1, H, W, 3);
  return get_plan(m, "trunk", {m->trunk0, m->trunk1}, BV, 1, H, W, in, pl, st);
}

extern "C" int64_t ivx_backbone_fpn_workspace_bytes(ivx_model *m, int32_t BV, int32_t H, int32_t W) {
  if (check_img(m, BV, 1, H, W, "ivx_backbone_fpn_workspace_bytes") != IVX_OK) return -1;
  Plan *pl;
  if (plan_trunk(m, BV, H, W, &pl, nullptr) != IVX_OK) return -1;
  return pl->total;
}

extern "C" int ivx_backbone_fpn_fwd(ivx_model *m, const float *img, int32_t BV, int32_t H, int32_t W, float *fpn0, void *workspace,
                                    int64_t workspace_bytes, ivx_stream_t stream) {
  M_TRY(check_img(m, BV, 1, H, W, "ivx_backbone_fpn_fwd"));
  M_REQUIRE(img && fpn0, "ivx_backbone_fpn_fwd: null argument");
  Plan *pl;
  M_TRY(plan_trunk(m, BV, H, W, &pl, (hipStream_t)stream));
  Bind bd;
  bd.ext[m->t_img] = (void *)img;
  bd.ext[m->t_fpn0] = fpn0;
  return run_steps(m, *pl, {m->trunk0, m->trunk1}, bd, workspace, workspace_bytes, (hipStream_t)stream, "ivx_backbone_fpn_fwd");
}

// ---- e4m3 interior of the bottlenecks ("bf16 with fp8 2-D conv MFMA", BASELINE config 5) on a bf16 handle
namespace {
// fp32 -> OCP e4m3 (bias 7, 3 mantissa bits, max 448, no infinities), round to nearest even; |x| <= 448 by construction here
inline uint8_t f32_to_e4m3_bits(float f) {
  uint32_t x;
  memcpy(&x, &f, 4);
  const uint8_t sign = (uint8_t)((x >> 24) & 0x80u);
  float a = fabsf(f);
  if (!(a == a)) return (uint8_t)(sign | 0x7fu);
  if (a > 448.f) a = 448.f;
  if (a < 0.015625f) return (uint8_t)(sign | (uint8_t)lrintf(a * 512.0f));       // below 2^-6: units of 2^-9 (8 = the smallest normal)
  int e;
  const float mant = frexpf(a, &e) * 2.0f;                                         // a = mant * 2^(e-1), mant in [1, 2)
  int q = (int)lrintf((mant - 1.0f) * 8.0f), ex = e - 1;
  if (q == 8) { q = 0; ++ex; }
  return (uint8_t)(sign | (uint8_t)(((ex + 7) << 3) | q));
}
}  // namespace

extern "C" int ivx_model_calibrate_fp8(ivx_model *m, const float *img, int32_t BV, int32_t H, int32_t W, float margin, void *workspace,
                                       int64_t workspace_bytes, ivx_stream_t stream) {
  return ivx_model_calibrate_fp8_ex(m, img, BV, H, W, margin, 0, 0, workspace, workspace_bytes, stream);
}

// first_stage: ResNet stages below it keep plain bf16 bottlenecks; conv2_bf16: conv1 and conv2 stay bf16 convolutions, conv2 WRITES e4m3 and only
// conv3 runs on the fp8 matrix cores (e4m3 input + e4m3 filters).  (2, 1) is the variant the noise budget prices at ~2.1 % of FPN level 0
// (tools/fp8_noise_budget.py, profiles/r06_config5.md); (0, 0) the round-3 mode (every interior tensor and both filter banks e4m3: 3.6 %).
extern "C" int ivx_model_calibrate_fp8_ex(ivx_model *m, const float *img, int32_t BV, int32_t H, int32_t W, float margin, int32_t first_stage,
                                          int32_t conv2_bf16, void *workspace, int64_t workspace_bytes, ivx_stream_t stream) {
  M_TRY(check_img(m, BV, 1, H, W, "ivx_model_calibrate_fp8"));
  M_REQUIRE(img && margin > 0.f && first_stage >= 0 && first_stage <= 4, "ivx_model_calibrate_fp8: bad argument");
  M_REQUIRE(m->cfg.storage == IVX_BF16 && m->cfg.with_trunk, "ivx_model_calibrate_fp8: needs a handle with storage = IVX_BF16 and the 2-D trunk");
  hipStream_t st = (hipStream_t)stream;
  m->fp8_on = false;                         // a second calibration starts from the bf16 trunk again
  for (ConvLayer &L : m->layers) {
    L.fp8_eff = 0;
    if (!L.fp8_role || L.stage < first_stage) continue;
    L.fp8_eff = !conv2_bf16 ? L.fp8_role : (L.fp8_role == 1 ? 0 : (L.fp8_role == 2 ? 1 : 3));
  }
  m->plans.clear();
  Plan *pl;
  M_TRY(plan_trunk(m, BV, H, W, &pl, st));
  const size_t nl = m->layers.size();
  void *cd = nullptr;
  M_HIP(hipMalloc(&cd, nl * sizeof(float)), "hipMalloc (calibration maxima)");
  M_HIP(hipMemsetAsync(cd, 0, nl * sizeof(float), st), "hipMemsetAsync");
  Bind bd;
  bd.ext[m->t_img] = (void *)img;
  m->calib_dev = (float *)cd;
  const int rc = run_steps(m, *pl, {m->trunk0, m->trunk1}, bd, workspace, workspace_bytes, st, "ivx_model_calibrate_fp8");
  m->calib_dev = nullptr;
  std::vector<float> amax(nl, 0.f);
  hipError_t e1 = hipMemcpyAsync(amax.data(), cd, nl * sizeof(float), hipMemcpyDeviceToHost, st), e2 = hipStreamSynchronize(st);
  (void)hipFree(cd);
  if (rc != IVX_OK) return rc;
  M_HIP(e1, "hipMemcpyAsync (calibration maxima)");
  M_HIP(e2, "hipStreamSynchronize");
  // scales and e4m3 filters (conv.py FusedConv with dtype / out_dtype FP8, in the same fp32 operation order)
  double s_prev = 1.0;                       // scale of the e4m3 tensor the next layer of the block reads
  for (size_t li = 0; li < nl; ++li) {
    ConvLayer &L = m->layers[li];
    if (!L.fp8_eff) continue;
    const double s_in = L.fp8_eff == 1 ? 1.0 : s_prev;
    const double s_out = L.fp8_eff == 3 ? 1.0 : std::max((double)amax[li], 1e-12) * (double)margin / 448.0;
    L.out_scale_q = s_out;
    const int taps = L.k[0] * L.k[1] * L.k[2];
    std::vector<float> ws_(L.cout, 1.0f);
    if (L.fp8_eff >= 2) {
      M_REQUIRE(L.cin_pad % 16 == 0 && !L.w_tap.empty(), "ivx_model_calibrate_fp8: layer %s: e4m3 filters need Cin %% 16 == 0", L.name.c_str());
      const size_t per = (size_t)taps * L.cin_pad, n_w = per * L.cout;
      L.layout_q = L.cin_pad % 128 == 0 ? 1 : 0;
      std::vector<uint8_t> q(n_w + 3, 0);
      const int nch = L.cin_pad / 128;
      for (int co = 0; co < L.cout; ++co) {
        const float *src = L.w_tap.data() + (size_t)co * per;
        float mx = 0.f;
        for (size_t k = 0; k < per; ++k) mx = std::max(mx, fabsf(src[k]));
        ws_[co] = std::max(mx, 1e-12f) / 448.0f;                                       // one scale per output channel: max |w| -> 448
        for (int t = 0; t < taps; ++t)
          for (int c = 0; c < L.cin_pad; ++c) {
            const size_t dst = L.layout_q ? (((size_t)co * nch + c / 128) * taps + t) * 128 + (c % 128) : ((size_t)co * taps + t) * L.cin_pad + c;
            q[dst] = f32_to_e4m3_bits(src[(size_t)t * L.cin_pad + c] / ws_[co]);
          }
      }
      M_TRY(dev_reupload_sync(m, reinterpret_cast<const float *>(q.data()), (n_w + 3) / 4, &L.wq, st));
    }
    const float r = (float)(s_in / s_out);
    const float so = (float)s_out;
    std::vector<float> sc(L.cout), sf(L.cout);
    for (int co = 0; co < L.cout; ++co) {
      sc[co] = (L.scale_h[co] * ws_[co]) * r;                                          // bn_scale * s_w[co] * (s_in / s_out)
      sf[co] = L.shift_h[co] / so;                                                     // bn_shift / s_out
    }
    M_TRY(dev_reupload_sync(m, sc.data(), sc.size(), &L.scale_q, st));
    M_TRY(dev_reupload_sync(m, sf.data(), sf.size(), &L.shift_q, st));
    s_prev = s_out;
  }
  m->fp8_on = true;
  m->plans.clear();                           // element sizes and dtypes changed
  return IVX_OK;
}

// ---- 3-D neck alone
static int plan_neck(ivx_model *m, int B, Plan **pl, hipStream_t st) {
  std::map<int, TInfo> in;
  in[m->t_volume] = tinfo(B, m->cfg.n_voxels[0], m->cfg.n_voxels[1], m->cfg.n_voxels[2], m->cfg.fpn_channels, m->cfg.storage == IVX_BF16 ? 2 : 4);
  return get_plan(m, "neck", {m->neck0, m->neck1}, B, 1, 0, 0, in, pl, st);
}

extern "C" int64_t ivx_neck3d_workspace_bytes(ivx_model *m, int32_t B) {
  if (!m || B <= 0) { ivx_set_error("ivx_neck3d_workspace_bytes: bad argument"); return -1; }
  Plan *pl;
  if (plan_neck(m, B, &pl, nullptr) != IVX_OK) return -1;
  return pl->total;
}

extern "C" int ivx_neck3d_out_dims(ivx_model *m, int32_t B, int32_t *X, int32_t *Y, int32_t *C) {
  M_REQUIRE(m && X && Y && C && B > 0, "ivx_neck3d_out_dims: bad argument");
  Plan *pl;
  M_REQUIRE(m->t_neck >= 0, "ivx_neck3d_out_dims: the handle holds an indoor neck; use ivx_neck3d_levels");
  M_TRY(plan_neck(m, B, &pl, nullptr));
  const TInfo &o = pl->t[m->t_neck];
  M_REQUIRE(o.W == 1, "the z axis must collapse to 1 (got %d); necks/imvoxelnet.py:119,150", o.W);
  *X = o.D; *Y = o.H; *C = o.C;
  return IVX_OK;
}

static int neck_fwd(ivx_model *m, int want_type, const float *volume, int32_t B, float *out, void *workspace, int64_t workspace_bytes,
                    ivx_stream_t stream, const char *who) {
  M_REQUIRE(m && volume && out && B > 0, "%s: bad argument", who);
  M_REQUIRE(m->cfg.neck_type == want_type, "%s: the handle holds the other stack neck", who);
  Plan *pl;
  M_TRY(plan_neck(m, B, &pl, (hipStream_t)stream));
  M_REQUIRE(pl->t[m->t_neck].W == 1, "%s: the z axis must collapse to 1 (got %d); necks/imvoxelnet.py:119,150", who, pl->t[m->t_neck].W);
  Bind bd;
  bd.ext[m->t_volume] = (void *)volume;
  bd.ext[m->t_neck] = out;
  return run_steps(m, *pl, {m->neck0, m->neck1}, bd, workspace, workspace_bytes, (hipStream_t)stream, who);
}

extern "C" int ivx_neck3d_kitti_fwd(ivx_model *m, const float *volume, int32_t B, float *out, void *workspace, int64_t workspace_bytes,
                                    ivx_stream_t stream) {
  return neck_fwd(m, IVX_NECK_KITTI, volume, B, out, workspace, workspace_bytes, stream, "ivx_neck3d_kitti_fwd");
}

extern "C" int ivx_neck3d_nuscenes_fwd(ivx_model *m, const float *volume, int32_t B, float *out, void *workspace, int64_t workspace_bytes,
                                       ivx_stream_t stream) {
  return neck_fwd(m, IVX_NECK_NUSCENES, volume, B, out, workspace, workspace_bytes, stream, "ivx_neck3d_nuscenes_fwd");
}

// Host-only: the anchor grid this handle uses for an (H, W) map (n = H*W*A rows of 7); for hosts that want to inspect it.
// ---- indoor necks: three levels
extern "C" int ivx_neck3d_levels(ivx_model *m, int32_t B, int32_t dims[3][4]) {
  M_REQUIRE(m && dims && B > 0, "ivx_neck3d_levels: bad argument");
  M_REQUIRE(!m->t_levels.empty(), "ivx_neck3d_levels: the handle holds a stack neck; use ivx_neck3d_out_dims");
  Plan *pl;
  M_TRY(plan_neck(m, B, &pl, nullptr));
  for (int l = 0; l < 3; ++l) {
    dims[l][0] = dims[l][1] = dims[l][2] = dims[l][3] = 0;      // a 3-scale U-Net has two levels: the third entry stays 0
    if (l >= (int)m->t_levels.size()) continue;
    const TInfo &o = pl->t[m->t_levels[l]];
    dims[l][0] = o.D; dims[l][1] = o.H; dims[l][2] = o.W; dims[l][3] = o.C;
  }
  return IVX_OK;
}

static int neck_levels_fwd(ivx_model *m, int want_type, const float *volume, int32_t B, float *const out_levels[3], void *workspace,
                           int64_t workspace_bytes, ivx_stream_t stream, const char *who) {
  M_REQUIRE(m && volume && out_levels && B > 0, "%s: bad argument", who);
  M_REQUIRE(m->cfg.neck_type == want_type, "%s: the handle holds another neck", who);
  for (size_t l = 0; l < m->t_levels.size(); ++l) M_REQUIRE(out_levels[l], "%s: null output level %d", who, (int)l);
  Plan *pl;
  M_TRY(plan_neck(m, B, &pl, (hipStream_t)stream));
  Bind bd;
  bd.ext[m->t_volume] = (void *)volume;
  for (size_t l = 0; l < m->t_levels.size(); ++l) bd.ext[m->t_levels[l]] = out_levels[l];
  return run_steps(m, *pl, {m->neck0, m->neck1}, bd, workspace, workspace_bytes, (hipStream_t)stream, who);
}

extern "C" int ivx_neck3d_fast_fwd(ivx_model *m, const float *volume, int32_t B, float *const out_levels[3], void *workspace,
                                   int64_t workspace_bytes, ivx_stream_t stream) {
  return neck_levels_fwd(m, IVX_NECK_FAST, volume, B, out_levels, workspace, workspace_bytes, stream, "ivx_neck3d_fast_fwd");
}

extern "C" int ivx_neck3d_unet_fwd(ivx_model *m, const float *volume, int32_t B, float *const out_levels[3], void *workspace,
                                   int64_t workspace_bytes, ivx_stream_t stream) {
  return neck_levels_fwd(m, IVX_NECK_UNET, volume, B, out_levels, workspace, workspace_bytes, stream, "ivx_neck3d_unet_fwd");
}

extern "C" int ivx_model_forward_levels(ivx_model *m, const float *input, int32_t B, int32_t V, int32_t H, int32_t W, const float *proj,
                                        const float *new_origin, const int32_t *crop_hw, void *workspace, int64_t workspace_bytes,
                                        float *const out_levels[3], uint8_t *out_valid, ivx_stream_t stream) {
  M_TRY(check_img(m, B, V, H, W, "ivx_model_forward_levels"));
  M_REQUIRE(input && out_levels, "ivx_model_forward_levels: null argument");
  M_REQUIRE(!m->t_levels.empty(), "ivx_model_forward_levels: the handle holds a stack neck + anchor head; use ivx_model_forward");
  M_REQUIRE(!m->cfg.layout_head, "ivx_model_forward_levels: a handle with a LayoutHead runs through ivx_model_detect");
  for (size_t l = 0; l < m->t_levels.size(); ++l) M_REQUIRE(out_levels[l], "ivx_model_forward_levels: null output level %d", (int)l);
  Plan *pl; Range r;
  M_TRY(plan_forward(m, B, V, H, W, &pl, &r, (hipStream_t)stream));
  Bind bd;
  bd.ext[m->cfg.with_trunk ? m->t_img : m->t_fpn0] = (void *)input;
  if (out_valid) bd.ext[m->t_valid] = out_valid;
  for (size_t l = 0; l < m->t_levels.size(); ++l) bd.ext[m->t_levels[l]] = out_levels[l];
  bd.proj = proj; bd.new_origin = new_origin; bd.crop = crop_hw; bd.V = V;
  return run_steps(m, *pl, r, bd, workspace, workspace_bytes, (hipStream_t)stream, "ivx_model_forward_levels");
}

// ---- the whole of simple_test in one call, every family: the per-batch camera set-up is computed here (host), uploaded into the
// camera block of the workspace, and the handle runs trunk -> unprojection -> neck -> head -> tail.
static int plan_detect(ivx_model *m, int B, int V, int H, int W, Plan **pl, Range *r, hipStream_t st) {
  M_REQUIRE(m->t_head >= 0 || m->cfg.head_type != IVX_HEAD_NONE,
            "ivx_model_detect: the handle was created without a head (head_type 0): it ends at the neck levels, use ivx_model_forward_levels");
  M_REQUIRE(!m->cfg.layout_head || V == 1, "ivx_model_detect: the LayoutHead predicts ONE camera per sample (V must be 1, got %d)", V);
  return plan_forward(m, B, V, H, W, pl, r, st, true);
}

extern "C" int64_t ivx_model_detect_workspace_bytes(ivx_model *m, int32_t B, int32_t V, int32_t H, int32_t W) {
  if (check_img(m, B, V, H, W, "ivx_model_detect_workspace_bytes") != IVX_OK) return -1;
  Plan *pl; Range r;
  if (plan_detect(m, B, V, H, W, &pl, &r, nullptr) != IVX_OK) return -1;
  return pl->total;
}

extern "C" int32_t ivx_model_max_detections(ivx_model *m, int32_t B, int32_t V, int32_t H, int32_t W) {
  if (check_img(m, B, V, H, W, "ivx_model_max_detections") != IVX_OK) return -1;
  Plan *pl; Range r;
  if (plan_detect(m, B, V, H, W, &pl, &r, nullptr) != IVX_OK) return -1;
  return pl->max_det;
}

extern "C" int ivx_model_detect(ivx_model *m, const float *img, int32_t B, int32_t V, int32_t H, int32_t W, const ivx_sample_meta *metas,
                                void *workspace, int64_t workspace_bytes, float *out_boxes, float *out_scores, int64_t *out_labels,
                                int32_t *out_count, uint8_t *out_valid, float *out_angles, float *out_layout, ivx_stream_t stream) {
  M_TRY(check_img(m, B, V, H, W, "ivx_model_detect"));
  M_REQUIRE(img && metas, "ivx_model_detect: null argument");
  hipStream_t st = (hipStream_t)stream;
  Plan *pl; Range r;
  M_TRY(plan_detect(m, B, V, H, W, &pl, &r, st));
  M_REQUIRE(workspace && ((uintptr_t)workspace & 255) == 0, "ivx_model_detect: workspace must be non-null and 256-byte aligned");
  if (workspace_bytes < pl->total) {
    ivx_set_error("ivx_model_detect: workspace too small (%lld < %lld); size it with ivx_model_detect_workspace_bytes", (long long)workspace_bytes,
                  (long long)pl->total);
    return IVX_ERR_WORKSPACE;
  }
  // host camera set-up (detectors/imvoxelnet.py:114-129, :139, :67-68; imvoxel_head_v2.py:206-214 for the head levels), in the library's
  // fixed fp32 operation order
  if (m->cam_ring.empty()) m->cam_ring.resize(8);
  std::vector<char> &hs = m->cam_ring[m->cam_next++ % m->cam_ring.size()];
  hs.assign((size_t)pl->cam_bytes, 0);
  const ivx_model_cfg &c = m->cfg;
  for (int b = 0; b < B; ++b) {
    const ivx_sample_meta &mt = metas[b];
    M_REQUIRE(mt.img_h > 0 && mt.img_w > 0 && mt.ori_h > 0 && mt.img_h <= H && mt.img_w <= W, "ivx_model_detect: sample %d: bad img_shape / ori_shape", b);
    if (!c.layout_head) {
      M_REQUIRE(mt.extrinsics, "ivx_model_detect: sample %d: null extrinsics", b);
      M_TRY(ivx_compute_projection(mt.intrinsic, mt.extrinsics, V, (double)mt.ori_h / ((double)mt.img_h / 4.0),
                                   (float *)(hs.data() + pl->cam_proj) + (size_t)b * V * 12));
    }
    M_TRY(ivx_voxel_new_origin(mt.origin, c.n_voxels, c.voxel_size, (float *)(hs.data() + pl->cam_origin) + (size_t)b * 3));
    int32_t *crop = (int32_t *)(hs.data() + pl->cam_crop) + (size_t)b * 2;
    crop[0] = mt.img_h / 4; crop[1] = mt.img_w / 4;
    for (size_t l = 0; l < m->t_headout.size(); ++l) {
      const TInfo &ho = pl->t[m->t_headout[l]];
      const int32_t nl[3] = {ho.D, ho.H, ho.W};
      float *vs = (float *)(hs.data() + pl->cam_lvl_vs[l]) + (size_t)b * 3;
      for (int a = 0; a < 3; ++a) vs[a] = c.voxel_size[a] * (float)(1 << l);       // voxel_size * 2^level (exact)
      M_TRY(ivx_voxel_new_origin(mt.origin, nl, vs, (float *)(hs.data() + pl->cam_lvl_no[l]) + (size_t)b * 3));
    }
  }
  char *base = (char *)workspace;
  M_HIP(hipMemcpyAsync(base, hs.data(), (size_t)pl->cam_bytes, hipMemcpyHostToDevice, st), "hipMemcpyAsync (camera block)");
  Bind bd;
  bd.ext[m->cfg.with_trunk ? m->t_img : m->t_fpn0] = (void *)img;     // (with_trunk = 0: the FPN level-0 maps, channels-last)
  if (out_valid) bd.ext[m->t_valid] = out_valid;
  bd.proj = (const float *)(base + pl->cam_proj); bd.proj_dev = (float *)(base + pl->cam_proj);
  bd.new_origin = (const float *)(base + pl->cam_origin); bd.crop = (const int32_t *)(base + pl->cam_crop); bd.V = V;
  for (int l = 0; l < 3; ++l) { bd.lvl_vs[l] = (const float *)(base + pl->cam_lvl_vs[l]); bd.lvl_no[l] = (const float *)(base + pl->cam_lvl_no[l]); }
  bd.metas = metas; bd.angles_host = out_angles; bd.layout_host = out_layout;
  bd.boxes = out_boxes; bd.scores = out_scores; bd.labels = out_labels; bd.count = out_count;
  return run_steps(m, *pl, r, bd, workspace, workspace_bytes, st, "ivx_model_detect");
}

// ---- LayoutHead host arithmetic (no device work), fixed fp32 operation order shared by every host of the library.
// angle = limit_period(angle) (layout_head.py:53, structures/utils.py:5-18: val - floor(val / pi + 0.5) * pi); layout = (centre, exp(size), yaw) (:70-74)
extern "C" int ivx_layout_head_decode(const float *angle_raw, const float *layout_raw, float *angle, float *layout) {
  M_REQUIRE(angle_raw && layout_raw && angle, "ivx_layout_head_decode: null argument");
  const float PI = 3.14159265358979323846f;
  for (int q = 0; q < 2; ++q) {
    const float t = floorf(angle_raw[q] / PI + 0.5f);
    angle[q] = angle_raw[q] - t * PI;
  }
  if (layout) {
    for (int q = 0; q < 3; ++q) layout[q] = layout_raw[q];
    for (int q = 3; q < 6; ++q) layout[q] = expf(layout_raw[q]);
    layout[6] = layout_raw[6];
  }
  return IVX_OK;
}

// get_extrinsics (detectors/imvoxelnet.py:164-187): camera extrinsic [4,4] from the predicted (pitch, roll), yaw = 0; the rotation in
// Total3DUnderstanding axes, moved to the depth-box axes (column order 2, 0, 1; third row negated).
extern "C" int ivx_layout_extrinsics(const float *angles, float *extrinsic4x4) {
  M_REQUIRE(angles && extrinsic4x4, "ivx_layout_extrinsics: null argument");
  const float cp = cosf(angles[0]), sp = sinf(angles[0]), cr = cosf(angles[1]), sr = sinf(angles[1]);
  float r[3][3];                                  // yaw = 0: cos = 1, sin = 0 (the products with them are exact)
  r[0][0] = cp;       r[0][1] = -(cr * sp);  r[0][2] = sp * sr;
  r[1][0] = sp;       r[1][1] = cp * cr;     r[1][2] = -(cp * sr);
  r[2][0] = 0.0f;     r[2][1] = sr;          r[2][2] = cr;
  float q[3][3];                                  // t @ r.T with t = [[0,0,1],[0,-1,0],[-1,0,0]]
  for (int j = 0; j < 3; ++j) { q[0][j] = r[j][2]; q[1][j] = -r[j][1]; q[2][j] = -r[j][0]; }
  const int perm[3] = {2, 0, 1};
  for (int i = 0; i < 16; ++i) extrinsic4x4[i] = 0.0f;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) extrinsic4x4[i * 4 + j] = (i == 2 ? -1.0f : 1.0f) * q[i][perm[j]];
  extrinsic4x4[15] = 1.0f;
  return IVX_OK;
}

extern "C" int ivx_model_anchors(ivx_model *m, int32_t H, int32_t W, float *anchors_host, int64_t capacity) {
  M_REQUIRE(m && anchors_host && H > 0 && W > 0, "ivx_model_anchors: bad argument");
  std::vector<float> a;
  make_anchors(m->cfg, H, W, &a);
  M_REQUIRE(capacity >= (int64_t)a.size(), "ivx_model_anchors: capacity %lld < %zu values", (long long)capacity, a.size());
  memcpy(anchors_host, a.data(), a.size() * sizeof(float));
  return IVX_OK;
}

// ---- host-side camera set-up (no device work): detectors/imvoxelnet.py:114-129 and :139 in fixed fp32 operation order, so
// the result does not depend on which BLAS kernel the host's matmul picks (the reference's `intrinsic @ extrinsic[:3]` is an
// FMA chain over k on the CPUs it was pinned on; tests/golden/backproject_cases.npz holds its outputs).
extern "C" int ivx_compute_projection(const float *intrinsic4x4, const float *extrinsics, int32_t V, double ratio, float *proj) {
  M_REQUIRE(intrinsic4x4 && extrinsics && proj && V > 0, "ivx_compute_projection: bad argument");
  float K[9];
  const float r = (float)ratio;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      float v = intrinsic4x4[i * 4 + j];
      if (i < 2) v = v / r;                       // intrinsic[:2] /= ratio
      K[i * 3 + j] = v;
    }
  for (int v = 0; v < V; ++v) {
    const float *E = extrinsics + (size_t)v * 16;  // rows 0..2 of the 4x4 matrix
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 4; ++j) {
        float acc = K[i * 3 + 0] * E[0 * 4 + j];
        acc = fmaf(K[i * 3 + 1], E[1 * 4 + j], acc);
        acc = fmaf(K[i * 3 + 2], E[2 * 4 + j], acc);
        proj[(size_t)v * 12 + i * 4 + j] = acc;
      }
  }
  return IVX_OK;
}

extern "C" int ivx_voxel_new_origin(const float *origin, const int32_t *n_voxels, const float *voxel_size, float *new_origin) {
  M_REQUIRE(origin && n_voxels && voxel_size && new_origin, "ivx_voxel_new_origin: null argument");
  for (int a = 0; a < 3; ++a) {                    // origin - n_voxels / 2. * voxel_size   (get_points, :139)
    const float half = (float)n_voxels[a] / 2.0f;
    const float ext = half * voxel_size[a];
    new_origin[a] = origin[a] - ext;
  }
  return IVX_OK;
}

// ---- optional stage timing: events around every launch group of the forward calls made while enabled.  Stages: 0 direct
// conv (or an un-split Winograd layer), 1 Winograd input transform, 2 grouped GEMM, 3 output transform, 4 unprojection,
// 5 anchor tail.  Read the records after synchronising the stream; enabling clears them.
extern "C" int ivx_model_trace(ivx_model *m, int32_t enable) {
  M_REQUIRE(m, "ivx_model_trace: null handle");
  m->trace.clear();
  m->events_used = 0;
  m->trace_on = enable != 0;
  if (enable >= 1 && enable <= 3) m->trace_level = enable;
  return IVX_OK;
}

extern "C" int32_t ivx_model_trace_count(ivx_model *m) { return m ? (int32_t)m->trace.size() : -1; }

extern "C" int ivx_model_trace_read(ivx_model *m, int32_t i, ivx_trace_rec *rec) {
  M_REQUIRE(m && rec && i >= 0 && i < (int32_t)m->trace.size(), "ivx_model_trace_read: bad index");
  const ivx_model::TraceRec &r = m->trace[i];
  float ms = 0.f, start = 0.f;
  M_HIP(hipEventElapsedTime(&ms, r.e0, r.e1), "hipEventElapsedTime (synchronise the stream first)");
  M_HIP(hipEventElapsedTime(&start, m->trace[0].e0, r.e0), "hipEventElapsedTime");
  rec->step = r.step; rec->stage = r.stage; rec->is3d = r.is3d; rec->ms = ms; rec->start_ms = start; rec->flops = r.flops; rec->bytes = r.bytes;
  snprintf(rec->name, sizeof(rec->name), "%s", r.name.c_str());
  return IVX_OK;
}

// Host-only: filters of the fp16-pair form (include/imvoxel.h).  Both hosts call this, so the kernels get the same bits from either.
extern "C" int ivx_pair_pack_filters(const float *w, int32_t Cout, int32_t taps, int32_t Cin, const float *scale, const float *shift, void *packed,
                                     float *scale_out, float *wbound, float *sbound) {
  M_REQUIRE(w && Cout > 0 && taps > 0 && Cin > 0 && wbound && sbound, "ivx_pair_pack_filters: bad argument");
  M_REQUIRE(!packed || (Cin % 32 == 0 && scale_out), "ivx_pair_pack_filters: pair filters need Cin %% 32 == 0 and scale_out");
  const size_t per = (size_t)taps * Cin;
  float amax = 0.f;
  double wb = 0.0, sb = 0.0;
  for (int co = 0; co < Cout; ++co) {
    double l1 = 0.0;
    const float *r = w + (size_t)co * per;
    for (size_t k = 0; k < per; ++k) {
      const float a = fabsf(r[k]);
      amax = a > amax ? a : amax;
      l1 += (double)a;
    }
    wb = std::max(wb, fabs((double)(scale ? scale[co] : 1.0f)) * l1);
    sb = std::max(sb, fabs((double)(shift ? shift[co] : 0.0f)));
  }
  *wbound = nextafterf((float)wb, INFINITY);      // never below the exact value
  *sbound = nextafterf((float)sb, INFINITY);
  if (!packed) return IVX_OK;
  float sw = 1.0f;
  if (amax > 0.f && amax < 3.0e38f) {             // as ivx_pow2_scale on the device: max |w| * s_w in [2^14, 2^15)
    int e;
    (void)frexpf(amax, &e);
    int k = 15 - e;
    k = k < -120 ? -120 : (k > 120 ? 120 : k);
    sw = ldexpf(1.0f, k);
  }
  const float inv = 1.0f / sw;
  for (int co = 0; co < Cout; ++co) scale_out[co] = (scale ? scale[co] : 1.0f) * inv;
  uint16_t *o = (uint16_t *)packed;
  const int nch = Cin / 32;
  for (int co = 0; co < Cout; ++co)
    for (int ch = 0; ch < nch; ++ch)
      for (int t = 0; t < taps; ++t) {
        uint16_t *dst = o + (((size_t)co * nch + ch) * taps + t) * 64;
        const float *src = w + ((size_t)co * taps + t) * Cin + (size_t)ch * 32;
        for (int g = 0; g < 2; ++g)
          for (int e = 0; e < 16; ++e) {
            const float y = src[g * 16 + e] * sw;
            const uint16_t h = f32_to_f16_bits(y);
            dst[g * 32 + e] = h;
            dst[g * 32 + 16 + e] = f32_to_f16_bits(y - f16_bits_to_f32(h));
          }
      }
  return IVX_OK;
}

// Host-only: the joint filter bank of conv3 and the shortcut conv of a stage's first block (include/imvoxel.h, csrc/bottleneck.hip projection form)
extern "C" int ivx_bottleneck_proj_pack(const float *w3, const float *scale3, const float *shift3, const float *wd, const float *scaled, const float *shiftd,
                                        int32_t P, int32_t Cin, void *packed, float *scale_out, float *shift_out, float *wbound3, float *sbound,
                                        float *wboundd) {
  M_REQUIRE(w3 && scale3 && shift3 && wd && scaled && shiftd && packed && scale_out && shift_out && wbound3 && sbound && wboundd,
            "ivx_bottleneck_proj_pack: null argument");
  M_REQUIRE(P > 0 && Cin > 0 && P % 32 == 0 && Cin % 32 == 0, "ivx_bottleneck_proj_pack: P and Cin must be multiples of 32");
  const int C = 4 * P, K = Cin + P;
  std::vector<float> bank((size_t)C * K);
  double b3 = 0.0, bd = 0.0;
  for (int n = 0; n < C; ++n) {
    double l3 = 0.0, ld = 0.0;
    for (int k = 0; k < Cin; ++k) {
      const float v = (float)((double)scaled[n] * (double)wd[(size_t)n * Cin + k]);
      bank[(size_t)n * K + k] = v;
      ld += fabs((double)v);
    }
    for (int k = 0; k < P; ++k) {
      const float v = (float)((double)scale3[n] * (double)w3[(size_t)n * P + k]);
      bank[(size_t)n * K + Cin + k] = v;
      l3 += fabs((double)v);
    }
    b3 = std::max(b3, l3);
    bd = std::max(bd, ld);
    shift_out[n] = shift3[n] + shiftd[n];
  }
  *wbound3 = nextafterf((float)b3, INFINITY);
  *wboundd = nextafterf((float)bd, INFINITY);
  float wb_all;
  return ivx_pair_pack_filters(bank.data(), C, 1, K, nullptr, shift_out, packed, scale_out, &wb_all, sbound);
}

// Host-only: the pair filters of the one-launch stem (csrc/stem.hip) in the order its wave reads them: [column tile 2][step 11][hi, lo][lane 64]
// [8 halves]; lane = h * 32 + n_l holds output channel nt * 32 + n_l, k = (c, ky) pair 2 * step + h (the 22nd: zeros), 8 filter columns
// (the eighth: zero).  Same scale rule as ivx_pair_pack_filters: s_w puts max |w| into [2^14, 2^15); scale_out = scale / s_w.
extern "C" int64_t ivx_stem_pool_filter_bytes(void) { return 2 * 11 * 2 * 64 * 8 * 2; }
extern "C" int ivx_stem_pool_pack_filters(const float *w, const float *scale, void *packed, float *scale_out) {
  M_REQUIRE(w && packed && scale_out, "ivx_stem_pool_pack_filters: null argument");
  float amax = 0.f;
  for (int i = 0; i < 64 * 3 * 49; ++i) amax = std::max(amax, fabsf(w[i]));
  float sw = 1.0f;
  if (amax > 0.f && amax < 3.0e38f) {
    int e;
    (void)frexpf(amax, &e);
    int k = 15 - e;
    k = k < -120 ? -120 : (k > 120 ? 120 : k);
    sw = ldexpf(1.0f, k);
  }
  for (int co = 0; co < 64; ++co) scale_out[co] = (scale ? scale[co] : 1.0f) / sw;
  uint16_t *o = (uint16_t *)packed;
  for (int nt = 0; nt < 2; ++nt)
    for (int kk = 0; kk < 11; ++kk)
      for (int lane = 0; lane < 64; ++lane)
        for (int e = 0; e < 8; ++e) {
          const int n = nt * 32 + (lane & 31), pq = 2 * kk + (lane >> 5);
          float y = 0.f;
          if (pq < 21 && e < 7) y = w[((n * 3 + pq / 7) * 7 + pq % 7) * 7 + e] * sw;
          const uint16_t h = f32_to_f16_bits(y);
          o[((((size_t)nt * 11 + kk) * 2 + 0) * 64 + lane) * 8 + e] = h;
          o[((((size_t)nt * 11 + kk) * 2 + 1) * 64 + lane) * 8 + e] = f32_to_f16_bits(y - f16_bits_to_f32(h));
        }
  return IVX_OK;
}

// Host-only: eval-mode BatchNorm (and the conv bias) as the epilogue's per-channel affine, in IEEE fp32 with a fixed
// operation order:  scale = gamma / sqrt(var + eps);  shift = beta + (bias - mean) * scale.   bias may be NULL (zeros).
// Both hosts of the conv kernels (the native model handle and the Python FusedConv) call this, so they feed the kernels
// the same bits whatever vector math library the host's tensor package uses.
extern "C" int ivx_fold_batchnorm(const float *gamma, const float *beta, const float *mean, const float *var, const float *bias, float eps,
                                  int32_t n, float *scale, float *shift) {
  M_REQUIRE(gamma && beta && mean && var && scale && shift && n > 0, "ivx_fold_batchnorm: bad argument");
  for (int c = 0; c < n; ++c) {
    const float s = gamma[c] / sqrtf(var[c] + eps);
    const float t = ((bias ? bias[c] : 0.0f) - mean[c]) * s;
    scale[c] = s;
    shift[c] = beta[c] + t;
  }
  return IVX_OK;
}
