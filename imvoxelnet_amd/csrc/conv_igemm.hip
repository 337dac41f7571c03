// Channels-last implicit-GEMM convolution on the CDNA4 matrix cores: exact fp32 (default) or bf16 storage.
//
//   GEMM view:  M = B*Do*Ho*Wo output positions, N = Cout, K = KD*KH*KW*Cin; im2col is done on the fly.
//   MFMA:       v_mfma_f32_32x32x2_f32 (f32 in, f32 accumulate == an fmaf chain, 64 cyc/SIMD), or
//               v_mfma_f32_32x32x16_bf16 for the optional bf16 storage mode (fp32 accumulate).
//   Workgroup:  256 threads = 4 wave64, block tile BM x BN = (WR*TM*32) x (WC*TN*32),
//               each wave owns TM x TN MFMA tiles of 32x32 (16 accumulator VGPRs each).
//   Kernels:    conv_igemm_v4_kernel (production: buffer loads with hardware zero-fill, LDS-DMA staging, XOR
//               swizzled LDS rows, chunk-major K order, split-K / grid-tail plan) -- see its header below;
//               conv_igemm_f32_kernel (generic: global loads into registers one slab ahead, ds_write_b128 into
//               padded 144-byte LDS rows, 64-bit addressing) for tensors >= 2 GiB or kernel extents > 8;
//               conv_naive_f32_kernel (validation only).
//   Fragments:  lane l reads 16 bytes of row (l & 31); half-wave h = l>>5 takes the k-chunk 2*kk + h, so one
//               ds_read_b128 per operand feeds 4 fp32 MFMAs (or 1 bf16 MFMA).  The k permutation is the same for A
//               and B, so the sum over k is unchanged.
//   Epilogue:   y = acc*scale[n] + shift[n] (+ residual, same-shape or nearest-upsampled) (ReLU) ..., stored with
//               32 consecutive lanes on 32 consecutive channels (128-byte rows).
//   Grid:       M tiles with an XCD-aware remap (block b runs on XCD b % 8; every XCD gets a contiguous range of
//               M tiles so halo re-reads hit its own L2).
//
// Reference call sites replaced: see include/imvoxel.h (ivx_conv_fwd).
#include "ivx_common.h"
#include <stdlib.h>

// This file is compiled six times (imvoxelnet_amd/_build.py), each translation unit instantiating one family of the LDS-DMA kernel, so that
// the families build in parallel:  IVX_CONV_TU 0 = the host side, the generic / naive / reduce kernels;  1 = fp32;  2 = bf16 and e4m3;
// 3 = bf16 (hi, lo) pair operands;  4 = fp16 pair operands;  5 = the z-halo kernel (conv_wino_halo_kernel).
#ifndef IVX_CONV_TU
#define IVX_CONV_TU 0
#endif
struct ConvParams;
struct ConvPlan;
int ivx_conv_launch_f32(ConvParams &p, const ConvPlan &pl, hipStream_t st);
int ivx_conv_launch_lowp(ConvParams &p, const ConvPlan &pl, hipStream_t st);
int ivx_conv_launch_pair_bf16(ConvParams &p, const ConvPlan &pl, hipStream_t st);
int ivx_conv_launch_pair_f16(ConvParams &p, const ConvPlan &pl, hipStream_t st);
int ivx_conv_launch_halo(ConvParams &p, int cfg, hipStream_t st);    // TU 5: the z-halo kernel of the Winograd-domain GEMMs (pair operands)
int ivx_conv_launch_fold4(ConvParams &p, const IvxWinoFold &f, int n2, hipStream_t st);    // TU 5: GEMM + output transform of F(4x4,3x3) fused
int ivx_conv_fold4_blocks(long long M, int Cout);

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

struct ConvParams {
  const float *in, *wgt, *scale, *shift, *res;
  float *out;
  int B, D, H, W, Cin;
  int Cout, KD, KH, KW;
  int sd, sh, sw, pd, ph, pw;
  int Do, Ho, Wo;
  int M, K;
  int relu, res_mode, rH, rW;
  int kmode;  // 0: k = tap*Cin + ci   1: k = (ci/32)*taps*32 + tap*32 + ci%32
  int out_mode;       // 1: ConvTranspose(k2,s2) scatter: column n = tap*Cr + co goes to output voxel 2*(d,h,w) + tap
  int Cr;             // real output channels (Cout / 8 when out_mode == 1, else Cout)
  int res_after_act;  // add the residual after the activation (skip connections of the U-shaped necks)
  float post_scale;   // final multiplier (Atlas neck (x + y) / 2); 1 = none
  int ksplit;         // > 1: split-K, grid.y = slice; raw partial sums go to `partial` [ksplit][8*q_count*BM][Cout]
  float *partial;
  // M-tile range of this launch (LDS-DMA kernel): XCD x owns tiles [x*q_total, (x+1)*q_total); this launch covers the
  // q_count tiles starting at q_begin inside every XCD's range.  Whole problem: q_begin 0, q_count q_total.
  int q_total, q_begin, q_count, bm;
  int in_bf16;        // in / wgt are bf16 (LDS-DMA kernel only); accumulation is always fp32
  int out_bf16;       // out / res are bf16 (converted round-to-nearest-even in the epilogue)
  // grouped launch (LDS-DMA kernel, grid.z = group; the Winograd-domain GEMMs): group g reads in + g*g_in, wgt + g*g_w
  // and writes out + g*g_out (element strides; all 0 for an ordinary convolution)
  long long g_in, g_w, g_out;
  int groups;
  int narrow_epilogue;  // A/B knob: 1 = the one-channel-per-lane epilogue everywhere (ivx_conv_set_epilogue_mode)
  int in_fp8;         // in / wgt are OCP e4m3 bytes (v_mfma_f32_32x32x16_fp8_fp8); dequantisation is folded into scale[]
  int out_fp8;        // out / res are e4m3 bytes (saturating round-to-nearest-even at the store)
  float res_scale;    // multiplier of the residual (fp8 storage: res_scale_of_tensor / out_scale); 1 otherwise
  int in_pair;        // 1 / 2: in / wgt hold every fp32 value as a (hi, lo) bf16 / fp16 pair, 16-channel groups [hi16 | lo16] (IVX_*_PAIR); Cin and K
                      // count bf16 elements (2 per real channel); the K loop issues hi*hi + hi*lo + lo*hi per group
  // fp16-pair ACTIVATIONS with device-side power-of-two scales (ivx_pair_io, include/imvoxel.h): the 2-D trunk chained on the 16-bit
  // matrix cores without split passes -- the producing epilogue writes the operand its consumer reads.
  int pio;                       // any of the fields below is in use (in_pair == 2 kernels, the split-K reduction, the validation kernel)
  const float *in_scale_p;       // device: the scale the pair input was written with (NULL: 1); the accumulator is divided by it
  int out_pair, res_pair;        // out / res are IVX_F16_PAIR tensors [.., 2*Cout] (independently of each other)
  const float *res_scale_p;      // device: scale of the pair residual
  float *out_scale_p;            // device: receives the scale chosen for the output
  const unsigned *amax_in, *amax_res;   // device, IVX_AMAX_SLOTS words: bits of max |in| / max |res| (true values); out_pair only
  unsigned *amax_out;            // device, IVX_AMAX_SLOTS words the epilogue accumulates max |out| into (atomic max), or NULL
  float wbound, sbound;          // |out| <= amax_in * wbound + sbound (+ amax_res): max_co |scale[co]| * sum_k |w[co][k]| and max_co |shift[co]|
  // one device word copied by workgroup 0 of the launch (grouped Winograd-domain GEMMs: the filter scale travels to the workspace header
  // the output transform reads -- was a 4-byte hipMemcpyAsync, i.e. one more launch per neck layer); NULL = none
  const unsigned *cp_src;
  unsigned *cp_dst;
  // Phase stagger of the first round of workgroups (pair-IO launches with a residual epilogue, run_conv): workgroups whose XCD-local index q has bit
  // stagger_shift set and q < stagger_first wait stagger_ticks (100 MHz) before their first request, so that the K loops of one half of the resident
  // workgroups overlap the epilogues of the other half for the rest of the launch.  0 ticks = off.
  int stagger_ticks, stagger_shift, stagger_first;
#ifdef IVX_CONV_TIMELINE
  unsigned long long *tl;        // debug build only (tools/conv_timeline.py): 8 words per workgroup of conv_igemm_v4_kernel's pair-IO path --
                                 // s_memrealtime (100 MHz) at entry, after the prologue barrier, after the K loop, at the end; HW_ID; XCC_ID
#endif
};

// What a wave needs of the pair-IO state: multipliers of the accumulator / the residual (exact: powers of two) and the output scale.
struct PairIO { float inv_in, inv_res, s_out; bool sat; float post; };      // post = ConvParams::post_scale (carried here: see conv_igemm_v4_kernel)
// Every wave computes the same values from the same device words (no communication): the output scale comes from a BOUND of the
// output -- |out| <= max|in| * wbound + sbound + max|res| with the MEASURED maxima of the operands, which their producers left in the
// amax slots -- so nothing has to pass over the output before it is written as (hi, lo) halves.  A bound that is loose by a factor L
// costs nothing up to L = 2^18: a value's error is max(2^-22 |x|, 2^-25 / s), i.e. relative to the tensor's maximum max(2^-22, 2^-40 L).
__device__ __forceinline__ PairIO conv_pair_io(const ConvParams &p) {
  PairIO io = {1.0f, 1.0f, 1.0f, false, p.post_scale};
  if (p.in_scale_p) io.inv_in = 1.0f / *p.in_scale_p;
  if (p.res_pair && p.res_scale_p) io.inv_res = 1.0f / *p.res_scale_p;
  if (p.out_pair) {
    const int lane = threadIdx.x & 63;
    float a = p.amax_in ? __uint_as_float(p.amax_in[lane]) : 0.f;
    float r = (p.amax_res && p.res_mode) ? __uint_as_float(p.amax_res[lane]) : 0.f;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      a = fmaxf(a, __shfl_xor(a, o));
      r = fmaxf(r, __shfl_xor(r, o));
    }
    const float bound = (a * p.wbound + p.sbound + r) * fabsf(p.post_scale) * 1.001f;
    io.s_out = ivx_pow2_scale(bound);
    // a non-finite bound (an Inf / NaN among the operands) says nothing about the finite outputs: fixed scale 2^-8 and a saturating
    // split, so that the damage stays where fp32 arithmetic keeps it instead of the whole tensor overflowing to Inf
    io.sat = !(bound < 3.0e38f);
    if (io.sat) io.s_out = 0.00390625f;
  }
  return io;
}
// element offset (in halves) of channel n of row `row` (C channels) inside an IVX_F16_PAIR tensor: hi; lo is 16 further
__device__ __forceinline__ size_t pair_off(size_t row, int n, int C) { return row * (size_t)(2 * C) + (size_t)((n >> 4) * 32 + (n & 15)); }

struct fp8_t { unsigned char v; };       // storage element of the fp8 instantiation (size 1)
typedef long i64_t;

__device__ __forceinline__ float fp8_to_f32(unsigned char b) { return __builtin_amdgcn_cvt_f32_fp8((int)b, 0); }
// saturating e4m3 conversion of two floats -> low / high byte pair (NaN stays NaN)
__device__ __forceinline__ unsigned int f32x2_to_fp8(float a, float b) {
  a = __builtin_fminf(__builtin_fmaxf(a, -448.f), 448.f);
  b = __builtin_fminf(__builtin_fmaxf(b, -448.f), 448.f);
  return (unsigned int)__builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false) & 0xffffu;
}

__device__ __forceinline__ float conv_ld_res(const ConvParams &p, size_t i) {
  if (p.res_pair) {      // (generic one-element paths only: split-K reduction, validation kernel); the caller multiplies by 1 / scale
    const _Float16 *r = reinterpret_cast<const _Float16 *>(p.res) + pair_off(i / p.Cout, (int)(i % p.Cout), p.Cout);
    return (float)r[0] + (float)r[16];
  }
  if (p.out_fp8) return fp8_to_f32(reinterpret_cast<const unsigned char *>(p.res)[i]);
  return p.out_bf16 ? (float)reinterpret_cast<const __bf16 *>(p.res)[i] : p.res[i];
}
__device__ __forceinline__ void conv_st_out(const ConvParams &p, size_t i, float v) {
  if (p.out_pair) {      // v is already multiplied by the output scale
    _Float16 *o = reinterpret_cast<_Float16 *>(p.out) + pair_off(i / p.Cout, (int)(i % p.Cout), p.Cout);
    const _Float16 h = (_Float16)v;
    o[0] = h;
    o[16] = (_Float16)(v - (float)h);
    return;
  }
  if (p.out_fp8)
    reinterpret_cast<unsigned char *>(p.out)[i] = (unsigned char)(f32x2_to_fp8(v, 0.f) & 0xffu);
  else if (p.out_bf16)
    reinterpret_cast<__bf16 *>(p.out)[i] = (__bf16)v;
  else
    p.out[i] = v;
}

#define IVX_BK 32
#define IVX_LDK 36

// Row base (in floats) of the nearest-upsampled residual for output row m (res_mode 2; FPN top-down,
// F.interpolate(mode='nearest', size=...): src = min(floor(dst * in/out), in-1), identity / >>1 when exact).
__device__ __noinline__ size_t res2_row_base(int m, int Ho, int Wo, int rH, int rW, int Cout) {
  const int ow = m % Wo;
  const int t = m / Wo;
  const int oh = t % Ho;
  const int b = t / Ho;  // Do == 1 for this mode
  int sh_ = (Ho == rH) ? oh : (Ho == 2 * rH ? (oh >> 1) : (int)floorf(oh * ((float)rH / Ho)));
  int sw_ = (Wo == rW) ? ow : (Wo == 2 * rW ? (ow >> 1) : (int)floorf(ow * ((float)rW / Wo)));
  sh_ = sh_ < rH - 1 ? sh_ : rH - 1;
  sw_ = sw_ < rW - 1 ? sw_ : rW - 1;
  return (((size_t)b * rH + sh_) * rW + sw_) * Cout;
}

// y = act(acc*scale + shift [+ res]) [+ res] [* post_scale] for one output element; `oidx` is its flat offset.
__device__ __forceinline__ float conv_finish(const ConvParams &p, float acc, float sc, float sf, size_t ridx, float rscale) {
  float v = acc * sc + sf;
  // one fused multiply-add, spelled out so that every epilogue variant rounds the same way; res_scale is 1 outside the fp8 mode,
  // where fma(r, 1, v) == v + r exactly
  if (p.res_mode && !p.res_after_act) v = __builtin_fmaf(conv_ld_res(p, ridx), rscale, v);
  if (p.relu) v = v > 0.f ? v : 0.f;
  if (p.res_mode && p.res_after_act) v = __builtin_fmaf(conv_ld_res(p, ridx), rscale, v);
  return v * p.post_scale;
}
__device__ __forceinline__ float conv_finish(const ConvParams &p, float acc, float sc, float sf, size_t ridx) {
  return conv_finish(p, acc, sc, sf, ridx, p.res_scale);
}

// Flat output offset of (row m, column n) for out_mode 1 (ConvTranspose3d kernel 2, stride 2): row m is the INPUT
// voxel (b,d,h,w) (the GEMM is a 1x1x1 conv), column n = ((a*2+e)*2+f)*Cr + co.
__device__ __noinline__ size_t up2_row_base(int m, int D, int H, int W, int Cr) {
  const int w = m % W;
  int t = m / W;
  const int h = t % H;
  t /= H;
  const int d = t % D;
  const int b = t / D;
  return ((((size_t)b * 2 * D + 2 * d) * 2 * H + 2 * h) * 2 * W + 2 * w) * Cr;
}

template <int TM, int TN>
__device__ __forceinline__ void conv_epilogue(const ConvParams &p, f32x16 (&acc)[TM][TN], int m0, int n0, int wr, int wc,
                                              int lane, size_t obase = 0) {
  const int col_l = lane & 31, hh = lane >> 5;
  float sc[TN], sf[TN];
  int nn[TN];
  size_t coff[TN];  // out_mode 1: per-column offset inside the 2x2x2 output cell
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    nn[j] = n0 + (wc * TN + j) * 32 + col_l;
    const bool nok = nn[j] < p.Cout;
    int ch = nn[j];
    coff[j] = 0;
    if (p.out_mode == 1 && nok) {
      const int tap = nn[j] / p.Cr;
      ch = nn[j] - tap * p.Cr;
      const int a = tap >> 2, e = (tap >> 1) & 1, f = tap & 1;
      coff[j] = (((size_t)a * 2 * p.H + e) * 2 * p.W + f) * p.Cr + ch;
    }
    sc[j] = (nok && p.scale) ? p.scale[ch] : 1.0f;
    sf[j] = (nok && p.shift) ? p.shift[ch] : 0.0f;
  }
#pragma unroll
  for (int i = 0; i < TM; ++i) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = m0 + (wr * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
      if (m >= p.M) continue;
      if (p.out_mode == 1) {
        const size_t ob = up2_row_base(m, p.D, p.H, p.W, p.Cr);
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          if (nn[j] >= p.Cout) continue;
          conv_st_out(p, ob + coff[j], conv_finish(p, acc[i][j][r], sc[j], sf[j], ob + coff[j]));
        }
        continue;
      }
      size_t rbase = (size_t)m * p.Cout;
      if (p.res_mode == 2) rbase = res2_row_base(m, p.Ho, p.Wo, p.rH, p.rW, p.Cout);
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        if (nn[j] >= p.Cout) continue;
        conv_st_out(p, obase + (size_t)m * p.Cout + nn[j], conv_finish(p, acc[i][j][r], sc[j], sf[j], rbase + nn[j]));
      }
    }
  }
}

// Wide-store epilogue of the LDS-DMA kernel (fp32 output, plain layout, Cout % 4 == 0; residual: none, same shape, or the
// nearest-upsampled coarser FPN level).
// The MFMA accumulator layout gives a lane ONE channel of 16 rows, so the direct epilogue above issues 16 dword stores per
// 32x32 tile and lane (64 per wave for a 64x64 quadrant): the store issue, not the bytes, sets its duration, and while a
// wave sits in it its SIMD runs with one MFMA wave fewer.  With short K (the Winograd-domain GEMMs: K = 192 .. 768, the
// 1x1 layers of the 2-D trunk) that is the largest loss of the kernel.  Here each 32x32 tile is transposed through the
// wave's own slice of the (now idle) staging LDS -- 16 ds_write_b32, 4 ds_read_b128 -- so that a lane owns 4 consecutive
// channels: the tile leaves as 4 dwordx4 stores per lane (8 rows x 128 bytes per wave instruction, whole cache lines), the
// residual arrives as 4 dwordx4 loads, scale / shift as one float4 each.  No barrier is needed: after the K loop's last
// barrier every wave holds its remaining fragments in registers, and a wave touches only its own 4 KB slice.
template <int TM, int TN>
__device__ __forceinline__ void conv_epilogue_wide(const ConvParams &p, f32x16 (&acc)[TM][TN], int m0, int n0, int wr, int wc,
                                                   int lane, float *stage, size_t obase) {
  const int col_l = lane & 31, hh = lane >> 5;
  const int rrow = lane >> 3, c4 = (lane & 7) * 4;     // read side: row rrow + 8q, channels c4 .. c4+3 of the tile
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int nb = n0 + (wc * TN + j) * 32 + c4;
    const bool nok = nb < p.Cout;                        // Cout % 4 == 0: a 4-channel chunk is inside or outside as a whole
    f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sf = {0.f, 0.f, 0.f, 0.f};
    if (nok && p.scale) sc = *reinterpret_cast<const f32x4 *>(p.scale + nb);
    if (nok && p.shift) sf = *reinterpret_cast<const f32x4 *>(p.shift + nb);
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) stage[((r & 3) + 8 * (r >> 2) + 4 * hh) * 32 + col_l] = acc[i][j][r];
      const int mb = m0 + (wr * TM + i) * 32 + rrow;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        f32x4 v = *reinterpret_cast<const f32x4 *>(stage + (rrow + 8 * q) * 32 + c4);
        const int m = mb + 8 * q;
        if (m < p.M && nok) {
          const size_t o = (size_t)m * p.Cout + nb;
          v = v * sc + sf;
          f32x4 rr = {0.f, 0.f, 0.f, 0.f};
          if (p.res_mode == 2) rr = *reinterpret_cast<const f32x4 *>(p.res + res2_row_base(m, p.Ho, p.Wo, p.rH, p.rW, p.Cout) + nb);
          else if (p.res_mode) rr = *reinterpret_cast<const f32x4 *>(p.res + o);
          if (p.res_mode && !p.res_after_act) v += rr;
          if (p.relu) {
            v[0] = v[0] > 0.f ? v[0] : 0.f; v[1] = v[1] > 0.f ? v[1] : 0.f;
            v[2] = v[2] > 0.f ? v[2] : 0.f; v[3] = v[3] > 0.f ? v[3] : 0.f;
          }
          if (p.res_mode && p.res_after_act) v += rr;
          v *= p.post_scale;
          *reinterpret_cast<f32x4 *>(p.out + obase + o) = v;
        }
      }
    }
  }
}

// Pair-IO variant of the wide epilogue (in_pair == 2 kernels; ConvParams::pio): the accumulator is divided by the input's scale, the
// residual is fp32 or an fp16 (hi, lo) pair tensor with its own scale, the output is fp32 or a pair tensor written with the scale of
// conv_pair_io -- a lane's 4 channels are 8 bytes of hi halves and, 32 bytes further, 8 bytes of lo halves -- and the maximum of the
// stored values (true values, before the output scale) goes to the amax slots for the next layer's bound.
// epilogue of 4 consecutive channels nb .. nb + 3 of output row m (pair IO): v = the accumulators; returns max |stored value|
typedef unsigned u32x4r __attribute__((ext_vector_type(4)));
typedef unsigned u32x2r __attribute__((ext_vector_type(2)));
// the residual of 4 consecutive channels nb .. nb + 3 of row `row` as it lies in memory: 4 floats, or (pair tensor) 8 bytes of hi halves
// and, 32 bytes further, 8 bytes of lo halves
__device__ __forceinline__ u32x4r conv_pio_res_load(const ConvParams &p, const size_t row, const int nb) {
  if (p.res_pair) {
    const _Float16 *rp = reinterpret_cast<const _Float16 *>(p.res) + pair_off(row, nb, p.Cout);
    const u32x2r h = *reinterpret_cast<const u32x2r *>(rp), l = *reinterpret_cast<const u32x2r *>(rp + 16);
    return u32x4r{h.x, h.y, l.x, l.y};
  }
  return *reinterpret_cast<const u32x4r *>(p.res + row * (size_t)p.Cout + nb);
}
__device__ __forceinline__ f32x4 conv_pio_res_decode(const ConvParams &p, const PairIO io, const u32x4r raw) {
  typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
  f32x4 rr;
  if (p.res_pair) {
    const f16x4 rh = __builtin_bit_cast(f16x4, u32x2r{raw.x, raw.y}), rl = __builtin_bit_cast(f16x4, u32x2r{raw.z, raw.w});
#pragma unroll
    for (int e = 0; e < 4; ++e) rr[e] = ((float)rh[e] + (float)rl[e]) * io.inv_res;
  } else {
    rr = __builtin_bit_cast(f32x4, raw);
  }
  return rr;
}
// v = the accumulators, rr = the decoded residual (zeros without one); stores the 4 channels, returns max |stored value|
__device__ __forceinline__ float conv_pio_finish4_rr(const ConvParams &p, const PairIO io, f32x4 v, const int m, const int nb, const f32x4 sc, const f32x4 sf,
                                                     const f32x4 rr) {
  typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
  v = v * sc + sf;
  if (p.res_mode && !p.res_after_act) v += rr;
  if (p.relu) {
    v[0] = v[0] > 0.f ? v[0] : 0.f; v[1] = v[1] > 0.f ? v[1] : 0.f;
    v[2] = v[2] > 0.f ? v[2] : 0.f; v[3] = v[3] > 0.f ? v[3] : 0.f;
  }
  if (p.res_mode && p.res_after_act) v += rr;
  v *= io.post;
  if (p.out_pair) {
    f16x4 h, l;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float y = v[e] * io.s_out;
      if (io.sat) y = (y > 65504.f && y < __builtin_inff()) ? 65504.f : ((y < -65504.f && y > -__builtin_inff()) ? -65504.f : y);
      h[e] = (_Float16)y;
      l[e] = (_Float16)(y - (float)h[e]);
    }
    _Float16 *op = reinterpret_cast<_Float16 *>(p.out) + pair_off((size_t)m, nb, p.Cout);
    *reinterpret_cast<f16x4 *>(op) = h;
    *reinterpret_cast<f16x4 *>(op + 16) = l;
  } else {
    *reinterpret_cast<f32x4 *>(p.out + (size_t)m * p.Cout + nb) = v;
  }
  return fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3])));
}
__device__ __forceinline__ float conv_pio_finish4(const ConvParams &p, const PairIO io, f32x4 v, const int m, const int nb, const f32x4 sc, const f32x4 sf) {
  f32x4 rr = {0.f, 0.f, 0.f, 0.f};
  if (p.res_mode) {
    const size_t row = p.res_mode == 2 ? res2_row_base(m, p.Ho, p.Wo, p.rH, p.rW, 1) : (size_t)m;
    rr = conv_pio_res_decode(p, io, conv_pio_res_load(p, row, nb));
  }
  return conv_pio_finish4_rr(p, io, v, m, nb, sc, sf, rr);
}

// Measured and removed (round 4, profiles/r04_pio_epilogue_prefetch_ab.md): requesting the residual of block b + 2 before block b is
// transposed and stored (two blocks = 32 registers in flight) so that a wave waits for one memory round trip instead of TM * TN --
// the 128 x 128 tile went from 5 to 26 spilled registers and every residual layer got SLOWER (64 -> 256 at 50 views 0.67 -> 0.77 ms,
// 128 -> 512 0.41 -> 0.45, 256 -> 1024 0.27 -> 0.29): the serial load -> store chain of a wave is hidden by the other fifteen waves of
// the CU; what bounds these layers is not the residual's latency.  (A one-block-ahead form with the request issued before the block is
// consumed: 55 spilled registers, 0.68 -> 1.11 ms.)
// RPF (round 5): the residual of the whole tile was requested BEFORE the K loop (conv_pio_res_prefetch: raw bits, 4 dwords per lane, block
// and pass -- 64 registers for the 128 x 128 tile) and is decoded here; taken when p.res_mode == 1, everything else goes the path above.
template <int TM, int TN>
__device__ __forceinline__ void conv_pio_res_prefetch(const ConvParams &p, u32x4r (&rraw)[TM][TN][4], int m0, int n0, int wr, int wc, int lane) {
  const int rrow = lane >> 3, c4 = (lane & 7) * 4;
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int nb = n0 + (wc * TN + j) * 32 + c4;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int m = m0 + (wr * TM + i) * 32 + rrow + 8 * q;
        rraw[i][j][q] = u32x4r{0u, 0u, 0u, 0u};
        if (m < p.M && nb < p.Cout) rraw[i][j][q] = conv_pio_res_load(p, (size_t)m, nb);
      }
  }
}
// The maximum is committed ONCE PER WORKGROUP (round 5): every wave leaves its maximum in the first word of its own staging slice, one
// barrier, wave 0 reduces and issues the atomic.  The 64 slots of a tensor lie in two cache lines, so atomics on different slots still
// serialise at one memory channel; with one atomic per WAVE the 960 .. 3840 waves of a trunk launch whose slots had just been zeroed
// queued there and the last workgroups of the launch waited 10 - 20 us in their epilogue (workgroup timelines inside the model,
// profiles/r05_trunk_wg_timeline.md; isolated launches do not show it: their slots already hold the maximum and the read skips the atomic).
template <int TM, int TN, int RPF = 0>
__device__ __forceinline__ void conv_epilogue_wide_pio(const ConvParams &p, const PairIO io, f32x16 (&acc)[TM][TN], int m0, int n0, int wr, int wc,
                                                       int lane, float *stage, int wid, int nw, const u32x4r (*rraw)[TN][4] = nullptr) {
  const int col_l = lane & 31, hh = lane >> 5;
  const int rrow = lane >> 3, c4 = (lane & 7) * 4;
  float omax = 0.f;
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int nb = n0 + (wc * TN + j) * 32 + c4;
    const bool nok = nb < p.Cout;
    f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sf = {0.f, 0.f, 0.f, 0.f};
    if (nok && p.scale) sc = *reinterpret_cast<const f32x4 *>(p.scale + nb);
    if (nok && p.shift) sf = *reinterpret_cast<const f32x4 *>(p.shift + nb);
    sc *= io.inv_in;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) stage[((r & 3) + 8 * (r >> 2) + 4 * hh) * 32 + col_l] = acc[i][j][r];
      const int mb = m0 + (wr * TM + i) * 32 + rrow;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 v = *reinterpret_cast<const f32x4 *>(stage + (rrow + 8 * q) * 32 + c4);
        const int m = mb + 8 * q;
        if (m < p.M && nok) {
          if (RPF && p.res_mode == 1) omax = fmaxf(omax, conv_pio_finish4_rr(p, io, v, m, nb, sc, sf, conv_pio_res_decode(p, io, rraw[i][j][q])));
          else omax = fmaxf(omax, conv_pio_finish4(p, io, v, m, nb, sc, sf));
        }
      }
    }
  }
  if (p.amax_out) {        // (uniform over the workgroup)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) omax = fmaxf(omax, __shfl_xor(omax, o));
    if (lane == 0) stage[0] = omax;
    __syncthreads();
    if (wid == 0) ivx_amax_commit(p.amax_out, lane < nw ? stage[lane * 1024] : 0.f, (int)blockIdx.x);
  }
}

// bf16-output variant of the wide epilogue (Cout % 8 == 0).  The one-channel-per-lane epilogue stores 2 bytes per lane --
// sub-dword writes, which the memory side handles as read-modify-write: the Cout-expanding 1x1 layers of the bf16 trunk ran
// at ~1.1 TB/s of algorithmic traffic.  Here NJ (2 when TN is even, else 1) 32x32 tiles are transposed through the wave's LDS
// slice so that a lane owns 8 consecutive channels of a row: the tile pair leaves as 16-byte stores (8 bf16), 8 lanes to a
// 128-byte row segment; the residual arrives as 16-byte loads.  Rounding is the same round-to-nearest-even conversion.
template <int TM, int TN, int NJ>
__device__ __forceinline__ void conv_epilogue_wide_bf16(const ConvParams &p, f32x16 (&acc)[TM][TN], int m0, int n0, int wr, int wc,
                                                        int lane, float *stage, size_t obase) {
  static_assert(TN % NJ == 0, "tile pairs");
  constexpr int LPR = 4 * NJ;        // lanes per row: 8 channels each over NJ*32 channels
  constexpr int RPP = 64 / LPR;      // rows per pass
  const int col_l = lane & 31, hh = lane >> 5;
  const int rrow = lane / LPR, c8 = (lane % LPR) * 8;
  const float *rd = stage + (c8 >> 5) * 1024 + (c8 & 31);
  __bf16 *outp = reinterpret_cast<__bf16 *>(p.out);
  const __bf16 *resp = reinterpret_cast<const __bf16 *>(p.res);
#pragma unroll
  for (int j = 0; j < TN; j += NJ) {
    const int nb = n0 + (wc * TN + j) * 32 + c8;
    const bool nok = nb < p.Cout;                        // Cout % 8 == 0: an 8-channel chunk is inside or outside as a whole
    f32x4 sc0 = {1.f, 1.f, 1.f, 1.f}, sc1 = sc0, sf0 = {0.f, 0.f, 0.f, 0.f}, sf1 = sf0;
    if (nok && p.scale) { sc0 = *reinterpret_cast<const f32x4 *>(p.scale + nb); sc1 = *reinterpret_cast<const f32x4 *>(p.scale + nb + 4); }
    if (nok && p.shift) { sf0 = *reinterpret_cast<const f32x4 *>(p.shift + nb); sf1 = *reinterpret_cast<const f32x4 *>(p.shift + nb + 4); }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int jj = 0; jj < NJ; ++jj)
#pragma unroll
        for (int r = 0; r < 16; ++r) stage[jj * 1024 + ((r & 3) + 8 * (r >> 2) + 4 * hh) * 32 + col_l] = acc[i][j + jj][r];
      const int mb = m0 + (wr * TM + i) * 32 + rrow;
#pragma unroll
      for (int q = 0; q < 32 / RPP; ++q) {
        f32x4 v0 = *reinterpret_cast<const f32x4 *>(rd + (rrow + RPP * q) * 32);
        f32x4 v1 = *reinterpret_cast<const f32x4 *>(rd + (rrow + RPP * q) * 32 + 4);
        const int m = mb + RPP * q;
        if (m < p.M && nok) {
          const size_t o = (size_t)m * p.Cout + nb;
          v0 = v0 * sc0 + sf0;
          v1 = v1 * sc1 + sf1;
          f32x4 r0 = {0.f, 0.f, 0.f, 0.f}, r1 = r0;
          if (p.res_mode) {
            const size_t ro = p.res_mode == 2 ? res2_row_base(m, p.Ho, p.Wo, p.rH, p.rW, p.Cout) + nb : o;
            const bf16x8 rb = *reinterpret_cast<const bf16x8 *>(resp + ro);
#pragma unroll
            for (int e = 0; e < 4; ++e) { r0[e] = (float)rb[e] * p.res_scale; r1[e] = (float)rb[e + 4] * p.res_scale; }
          }
          if (p.res_mode && !p.res_after_act) { v0 += r0; v1 += r1; }
          if (p.relu) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { v0[e] = v0[e] > 0.f ? v0[e] : 0.f; v1[e] = v1[e] > 0.f ? v1[e] : 0.f; }
          }
          if (p.res_mode && p.res_after_act) { v0 += r0; v1 += r1; }
          v0 *= p.post_scale;
          v1 *= p.post_scale;
          bf16x8 ob;
#pragma unroll
          for (int e = 0; e < 4; ++e) { ob[e] = (__bf16)v0[e]; ob[e + 4] = (__bf16)v1[e]; }
          *reinterpret_cast<bf16x8 *>(outp + obase + o) = ob;
        }
      }
    }
  }
}

// e4m3-output variant (Cout % 16 == 0): a lane owns 16 consecutive channels of a row -- one 16-byte store of 16 fp8 values,
// 4 lanes to a 64-byte row segment of a tile pair; the residual (e4m3, its own scale) arrives as one 16-byte load.
template <int TM, int TN, int NJ>
__device__ __forceinline__ void conv_epilogue_wide_fp8(const ConvParams &p, f32x16 (&acc)[TM][TN], int m0, int n0, int wr, int wc,
                                                       int lane, float *stage, size_t obase) {
  static_assert(TN % NJ == 0, "tile pairs");
  constexpr int LPR = 2 * NJ;        // lanes per row: 16 channels each over NJ*32 channels
  constexpr int RPP = 64 / LPR;      // rows per pass
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  const int col_l = lane & 31, hh = lane >> 5;
  const int rrow = lane / LPR, c16 = (lane % LPR) * 16;
  const float *rd = stage + (c16 >> 5) * 1024 + (c16 & 31);
  unsigned char *outp = reinterpret_cast<unsigned char *>(p.out);
  const unsigned char *resp = reinterpret_cast<const unsigned char *>(p.res);
#pragma unroll
  for (int j = 0; j < TN; j += NJ) {
    const int nb = n0 + (wc * TN + j) * 32 + c16;
    const bool nok = nb < p.Cout;                        // Cout % 16 == 0
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int jj = 0; jj < NJ; ++jj)
#pragma unroll
        for (int r = 0; r < 16; ++r) stage[jj * 1024 + ((r & 3) + 8 * (r >> 2) + 4 * hh) * 32 + col_l] = acc[i][j + jj][r];
      const int mb = m0 + (wr * TM + i) * 32 + rrow;
#pragma unroll
      for (int q = 0; q < 32 / RPP; ++q) {
        const int m = mb + RPP * q;
        if (!(m < p.M && nok)) continue;
        const size_t o = (size_t)m * p.Cout + nb;
        u32x4 rb = {0u, 0u, 0u, 0u};
        if (p.res_mode) rb = *reinterpret_cast<const u32x4 *>(resp + (p.res_mode == 2 ? res2_row_base(m, p.Ho, p.Wo, p.rH, p.rW, p.Cout) + nb : o));
        u32x4 ob;
        // four channels at a time (scale / shift come from the cache again for every row: keeps the live registers low -- the
        // first version held 16 scales + 16 shifts + 16 values + 16 residuals beside the 64 accumulators and spilled 75 VGPRs)
#pragma unroll
        for (int w = 0; w < 4; ++w) {
          f32x4 v = *reinterpret_cast<const f32x4 *>(rd + (rrow + RPP * q) * 32 + 4 * w);
          f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sf = {0.f, 0.f, 0.f, 0.f};
          if (p.scale) sc = *reinterpret_cast<const f32x4 *>(p.scale + nb + 4 * w);
          if (p.shift) sf = *reinterpret_cast<const f32x4 *>(p.shift + nb + 4 * w);
          v = v * sc + sf;
          f32x4 r = {0.f, 0.f, 0.f, 0.f};
          if (p.res_mode) {
            const f32x2 lo = __builtin_amdgcn_cvt_pk_f32_fp8((int)rb[w], false), hi = __builtin_amdgcn_cvt_pk_f32_fp8((int)rb[w], true);
            r[0] = lo[0]; r[1] = lo[1]; r[2] = hi[0]; r[3] = hi[1];
          }
          if (p.res_mode && !p.res_after_act) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = __builtin_fmaf(r[e], p.res_scale, v[e]);     // as conv_finish
          }
          if (p.relu) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : 0.f;
          }
          if (p.res_mode && p.res_after_act) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = __builtin_fmaf(r[e], p.res_scale, v[e]);
          }
          v *= p.post_scale;
          ob[w] = f32x2_to_fp8(v[0], v[1]) | (f32x2_to_fp8(v[2], v[3]) << 16);
        }
        *reinterpret_cast<u32x4 *>(outp + obase + o) = ob;
      }
    }
  }
}

template <int TM, int TN, int WR, int WC>
__global__ __launch_bounds__(256) void conv_igemm_f32_kernel(const ConvParams p) {
  constexpr int BM = WR * TM * 32, BN = WC * TN * 32;
  constexpr int AR = BM / 32, BR = BN / 32;
  static_assert(WR * WC == 4, "4 waves per workgroup");
  __shared__ __attribute__((aligned(16))) float smem[2 * (BM + BN) * IVX_LDK];
  float *As = smem;
  float *Bs = smem + 2 * BM * IVX_LDK;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = tid >> 6;
  const int wr = wid / WC, wc = wid % WC;

  // XCD-aware bijective remap of the M-tile index.
  int bid = blockIdx.x;
  {
    const int nb = gridDim.x;
    const int q = nb >> 3, r = nb & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int m0 = bid * BM;
  const int n0 = blockIdx.y * BN;

  // ---- loader state -------------------------------------------------------------------------
  const int cc = tid & 7;   // 16-byte chunk column inside the slab
  const int lr = tid >> 3;  // 0..31
  int a_bd[AR], a_id[AR], a_ih[AR], a_iw[AR];
#pragma unroll
  for (int j = 0; j < AR; ++j) {
    const int m = m0 + lr + 32 * j;
    if (m < p.M) {
      const int ow = m % p.Wo;
      int t = m / p.Wo;
      const int oh = t % p.Ho;
      t /= p.Ho;
      const int od = t % p.Do;
      const int b = t / p.Do;
      a_bd[j] = b * p.D;
      a_id[j] = od * p.sd - p.pd;
      a_ih[j] = oh * p.sh - p.ph;
      a_iw[j] = ow * p.sw - p.pw;
    } else {
      a_bd[j] = 0;
      a_id[j] = -(1 << 28);  // fails the bounds test for every tap
      a_ih[j] = 0;
      a_iw[j] = 0;
    }
  }
  // per-thread k cursor: k4 = slab*32 + cc*4 = tap*Cin + kc
  int k4 = cc * 4;
  int kc, ka, ke, kf;
  if (p.kmode == 1) {  // chunk-major: slab s = (channel chunk s / taps, tap s % taps)
    kc = cc * 4;
    ka = ke = kf = 0;
  } else {
    const int tap = k4 / p.Cin;
    kc = k4 - tap * p.Cin;
    kf = tap % p.KW;
    const int t2 = tap / p.KW;
    ke = t2 % p.KH;
    ka = t2 / p.KH;
  }
  const int S = (p.K + IVX_BK - 1) / IVX_BK;

  f32x4 ra[AR], rb[BR];
  auto load_slab = [&]() {
    const bool kok = k4 < p.K;
#pragma unroll
    for (int j = 0; j < AR; ++j) {
      const int id = a_id[j] + ka, ih = a_ih[j] + ke, iw = a_iw[j] + kf;
      const bool ok = kok && (unsigned)id < (unsigned)p.D && (unsigned)ih < (unsigned)p.H &&
                      (unsigned)iw < (unsigned)p.W;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (ok) {
        const size_t off = (((size_t)(a_bd[j] + id) * p.H + ih) * p.W + iw) * (size_t)p.Cin + kc;
        v = *reinterpret_cast<const f32x4 *>(p.in + off);
      }
      ra[j] = v;
    }
#pragma unroll
    for (int j = 0; j < BR; ++j) {
      const int n = n0 + lr + 32 * j;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (kok && n < p.Cout) v = *reinterpret_cast<const f32x4 *>(p.wgt + (size_t)n * p.K + k4);
      rb[j] = v;
    }
  };
  auto advance_k = [&]() {
    k4 += IVX_BK;
    if (p.kmode == 1) {  // next tap of the same 32-channel chunk; after the last tap move to the next chunk
      if (++kf == p.KW) {
        kf = 0;
        if (++ke == p.KH) {
          ke = 0;
          if (++ka == p.KD) {
            ka = 0;
            kc += IVX_BK;
          }
        }
      }
      return;
    }
    kc += IVX_BK;
    while (kc >= p.Cin) {
      kc -= p.Cin;
      if (++kf == p.KW) {
        kf = 0;
        if (++ke == p.KH) {
          ke = 0;
          ++ka;
        }
      }
    }
  };
  auto store_slab = [&](int buf) {
    float *Ab = As + buf * BM * IVX_LDK + lr * IVX_LDK + cc * 4;
    float *Bb = Bs + buf * BN * IVX_LDK + lr * IVX_LDK + cc * 4;
#pragma unroll
    for (int j = 0; j < AR; ++j) *reinterpret_cast<f32x4 *>(Ab + 32 * j * IVX_LDK) = ra[j];
#pragma unroll
    for (int j = 0; j < BR; ++j) *reinterpret_cast<f32x4 *>(Bb + 32 * j * IVX_LDK) = rb[j];
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  load_slab();
  store_slab(0);
  __syncthreads();

  const int frag_off = (lane & 31) * IVX_LDK + 4 * (lane >> 5);
  for (int s = 0; s < S; ++s) {
    const int cur = s & 1;
    const bool more = (s + 1) < S;
    if (more) {
      advance_k();
      load_slab();
    }
    const float *Ac = As + cur * BM * IVX_LDK + wr * TM * 32 * IVX_LDK + frag_off;
    const float *Bc = Bs + cur * BN * IVX_LDK + wc * TN * 32 * IVX_LDK + frag_off;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      f32x4 a[TM], b[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) a[i] = *reinterpret_cast<const f32x4 *>(Ac + i * 32 * IVX_LDK + kk * 8);
#pragma unroll
      for (int j = 0; j < TN; ++j) b[j] = *reinterpret_cast<const f32x4 *>(Bc + j * 32 * IVX_LDK + kk * 8);
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][q], b[j][q], acc[i][j], 0, 0, 0);
    }
    if (more) store_slab(cur ^ 1);
    __syncthreads();
  }

  conv_epilogue<TM, TN>(p, acc, m0, n0, wr, wc, lane);
}

// ------------------------------------------------------------------------------------------------
// v4, the production kernel: same tiling as the generic kernel above, cheaper slab loop.
//   * operands are fetched with raw buffer loads: an out-of-image tap, a row past M or a k past K gets the
//     out-of-range offset 0x80000000 and the hardware returns zeros -- no exec-mask branches, no 64-bit address
//     arithmetic in the loop (needs the tensor < 2 GiB); per-row tap validity is a precomputed bit mask (kernel
//     extents <= 8), so the per-slab work per load is one AND/compare, one add, one select;
//   * LDS-DMA staging (buffer_load_dwordx4 ... offen lds): the operands go global -> LDS without passing through
//     VGPRs, so the slab loop has no ds_write pass and no vmcnt -> ds_write dependency;
//   * MFMA fragments are double-buffered in registers: the ds_read_b128 of k-group kk+1 is issued before the MFMAs
//     of group kk.  The DMA writes
// wave-uniform base + lane*16 B, so LDS rows are unpadded 128-byte rows and bank conflicts are avoided by an XOR
// swizzle applied on the SOURCE side (lane with slot c of row r fetches k-chunk c ^ ((r >> SW_SH) & SW_MSK)) and
// again on the fragment read.  BK (K-slab depth) is 32 or 16; 16 halves the LDS so three workgroups fit on a CU.
// s_waitcnt vmcnt(0) (expcnt / lgkmcnt untouched): the LDS-DMA loads are VMEM operations; the compiler does not order
// them against the ds_reads of a LATER loop iteration, so the wait before the publishing barrier is explicit.
__device__ __forceinline__ void lds_dma_wait_all() { __builtin_amdgcn_s_waitcnt(0x0f70); }

// One matrix instruction on a (hi | lo) operand pair: PAIR 1 = bf16 halves, 2 = fp16 halves (same shape and rate)
template <int PAIR>
__device__ __forceinline__ f32x16 pair_mfma(const f32x4 a, const f32x4 b, const f32x16 c) {
  if constexpr (PAIR == 2) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// s_waitcnt vmcnt(n) with n = PER * newer, newer a runtime value in 0 .. 7 (the tail of a deep ring: fewer slabs are outstanding than in the
// steady state).  gfx9 encoding: vmcnt[3:0] in bits 3:0, vmcnt[5:4] in bits 15:14; expcnt / lgkmcnt fields all ones = untouched.  A count
// above 63 is clamped (waiting for fewer outstanding requests than needed is always correct).
template <int N>
__device__ __forceinline__ void lds_dma_wait_imm() {
  constexpr int n = N > 63 ? 63 : N;
  __builtin_amdgcn_s_waitcnt(0x0f70 | (n & 15) | ((n >> 4) << 14));
}
template <int PER>
__device__ __forceinline__ void lds_dma_wait_slabs(const int newer) {
  switch (newer) {
    case 1: lds_dma_wait_imm<PER>(); break;
    case 2: lds_dma_wait_imm<2 * PER>(); break;
    case 3: lds_dma_wait_imm<3 * PER>(); break;
    case 4: lds_dma_wait_imm<4 * PER>(); break;
    case 5: lds_dma_wait_imm<5 * PER>(); break;
    case 6: lds_dma_wait_imm<6 * PER>(); break;
    case 7: lds_dma_wait_imm<7 * PER>(); break;
    default: lds_dma_wait_imm<0>(); break;      // 0 (and anything else: waiting for everything is always correct)
  }
}

// NB (round 5): depth of the LDS ring.  2 = the form described above (prefetch distance two slabs).  With NB buffers the DMA of slab s + NB is
// issued when slab s retires, so NB slabs are in flight per workgroup: the layers of the 2-D trunk whose launch has fewer tiles than the chip
// has workgroup slots (every /8 .. /32 layer at KITTI's batch of 4: 240 - 480 tiles of 0.7 - 1.5 us per slab with 6 - 12 MFMAs of work in it)
// are bound by that latency, and resident workgroups cannot hide it when there are not enough tiles to be resident.  The products are
// accumulated in the same order: results are bit-identical to NB = 2.
template <typename T, int TM, int TN, int WR, int WC, int BK, int WPE, int UNI, int PAIR = 0, int NB = 2, int RPF = 0>
__global__ __launch_bounds__(64 * WR * WC, WPE) void conv_igemm_v4_kernel(const ConvParams p, const unsigned in_bytes,
                                                                  const unsigned w_bytes) {
  constexpr int BM = WR * TM * 32, BN = WC * TN * 32;
  constexpr int EL = sizeof(T);        // 4 (fp32) or 2 (bf16)
  constexpr int EPC = 16 / EL;         // elements per 16-byte chunk
  constexpr int CK = 128 / EL;         // channels per chunk of the chunk-major K order (128 bytes)
  constexpr int NCH = BK / EPC;        // 16-byte chunks per LDS row (8 or 4)
  constexpr int NT = 64 * WR * WC;     // 4 waves (256 threads) or 8 waves (512: the B slab is shared by twice the MFMAs)
  constexpr int RP = NT / NCH;         // rows covered by one pass of the workgroup's threads
  constexpr int AR = BM / RP, BR = BN / RP;
  constexpr int SW_SH = NCH == 8 ? 1 : 2, SW_MSK = NCH - 1;   // slot c of row r holds k-chunk c ^ ((r >> SW_SH) & SW_MSK)
  static_assert(NCH == 8 || NCH == 4, "BK: LDS rows are 128 or 64 bytes");
  static_assert(BM % RP == 0 && BN % RP == 0, "tile rows");
  static_assert(WR * WC == 4 || WR * WC == 8 || WR * WC == 16, "4, 8 or 16 waves per workgroup");
  // LDS-DMA staging: rows are BK elements (128 or 64 B), unpadded
  static_assert(NB >= 2 && NB <= 8, "ring depth");
  static_assert(NB == 2 || (NB - 1) * (AR + BR) <= 63, "deep ring: vmcnt immediates of the tail waits (6 bits)");
  __shared__ __attribute__((aligned(16))) T smem[NB * (BM + BN) * BK];
  T *As = smem;
  T *Bs = smem + NB * BM * BK;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = tid >> 6;
  const int wr = wid / WC, wc = wid % WC;

  // 1-D grid of 8 * ceil(Mt/8) * Nt workgroups.  Workgroup b runs on XCD b % 8 (observed dispatch rule; used for
  // speed only): XCD x owns the contiguous M-tiles [x*q, (x+1)*q) so halo re-reads of neighbouring x-slabs hit its own
  // L2, and inside an XCD the Nt workgroups that share one A-tile are consecutive, so they run together and the
  // A-tile is fetched into that L2 once instead of once per N-tile.  Ids past the last M-tile exit (< 8*Nt of them).
  int mt, nt, ct;   // ct: compact tile id inside this launch (split-K partial rows)
  {
    const int Nt = (p.Cout + BN - 1) / BN;
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const int lt = idx / Nt;
    nt = idx - lt * Nt;
    mt = xcd * p.q_total + p.q_begin + lt;
    ct = xcd * p.q_count + lt;
  }
  if (mt * BM >= p.M) return;
  const int m0 = mt * BM;
  const int n0 = nt * BN;
  if (p.cp_dst && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0) *p.cp_dst = *p.cp_src;
#ifdef IVX_CONV_TIMELINE
  unsigned long long tl0 = __builtin_amdgcn_s_memrealtime(), tl1 = 0, tl2 = 0;
#endif
  if constexpr (PAIR == 2) {
    if (p.stagger_ticks > 0) {      // (uniform per workgroup)
      const unsigned q = blockIdx.x >> 3;
      if (q < (unsigned)p.stagger_first && ((q >> p.stagger_shift) & 1u)) {
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)p.stagger_ticks) __builtin_amdgcn_s_sleep(32);
      }
    }
  }

  const size_t gz = blockIdx.z;   // group of a grouped launch (strides are 0 otherwise)
  const __amdgpu_buffer_rsrc_t rs_in =
      __builtin_amdgcn_make_buffer_rsrc((void *)((const char *)p.in + gz * (size_t)p.g_in * EL), 0, in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_w =
      __builtin_amdgcn_make_buffer_rsrc((void *)((const char *)p.wgt + gz * (size_t)p.g_w * EL), 0, w_bytes, 0x00020000);

  const int lr = tid / NCH;
  const int cc = (tid & (NCH - 1)) ^ ((lr >> SW_SH) & SW_MSK);   // the k-chunk this lane fetches (its LDS slot is tid % NCH)
  const int wid_u = __builtin_amdgcn_readfirstlane(wid);
  int a_off[AR];
  unsigned a_msk[AR];
#pragma unroll
  for (int j = 0; j < AR; ++j) {
    const int m = m0 + lr + RP * j;
    a_off[j] = 0;
    a_msk[j] = 0;
    if (m < p.M) {
      const int ow = m % p.Wo;
      int t = m / p.Wo;
      const int oh = t % p.Ho;
      t /= p.Ho;
      const int od = t % p.Do;
      const int b = t / p.Do;
      const int id0 = od * p.sd - p.pd, ih0 = oh * p.sh - p.ph, iw0 = ow * p.sw - p.pw;
      a_off[j] = (((b * p.D + id0) * p.H + ih0) * p.W + iw0) * p.Cin;
      unsigned md = 0, mh = 0, mw = 0;
      for (int a = 0; a < p.KD; ++a) md |= ((unsigned)(id0 + a) < (unsigned)p.D) ? (1u << a) : 0u;
      for (int e = 0; e < p.KH; ++e) mh |= ((unsigned)(ih0 + e) < (unsigned)p.H) ? (1u << e) : 0u;
      for (int f = 0; f < p.KW; ++f) mw |= ((unsigned)(iw0 + f) < (unsigned)p.W) ? (1u << f) : 0u;
      a_msk[j] = md | (mh << 8) | (mw << 16);
    }
  }
  int b_off[BR];
#pragma unroll
  for (int j = 0; j < BR; ++j) {
    const int n = n0 + lr + RP * j;
    b_off[j] = n < p.Cout ? n * p.K : -1;
  }
  // slab range of this workgroup (split-K: grid.y slices the K loop)
  const int S_all = (p.K + BK - 1) / BK;
  int s_begin = 0, s_end = S_all;
  if (p.ksplit > 1) {
    const int per = (S_all + p.ksplit - 1) / p.ksplit;
    s_begin = blockIdx.y * per;
    s_end = s_begin + per < S_all ? s_begin + per : S_all;
  }
  // K-loop state.  UNI (the production case: chunk-major weights, or tap-major with Cin % BK == 0): every lane of a
  // slab works on the same filter tap, so tap indices, the tap's input offset and its validity mask are wave-uniform
  // (scalar registers / SALU) and the per-slab vector work is one and/compare/add/select per A row and one add per B
  // row.  Otherwise (tiny Cin: the 3-channel stem, odd test shapes) the tap differs per 16-byte chunk and the state is
  // per lane.
  const int S = s_end - s_begin;
  const unsigned OOB = 0x80000000u;
  typedef __attribute__((address_space(3))) void *lds_ptr_t;
  int k4 = s_begin * BK + cc * EPC;      // per lane (generic path)
  int kc, ka, ke, kf;                    // generic: per lane; UNI: uniform (kc without the lane's chunk offset)
  int khalf = 0;   // 64-byte rows, chunk-major: which half of the 128-byte channel chunk this slab covers
  if (p.kmode == 1) {  // chunk-major: slab s = (CK-channel chunk, tap, half)
    constexpr int HPS = CK / BK;                       // slabs per (chunk, tap)
    const int ntap = p.KD * p.KH * p.KW;
    const int chunk = s_begin / (ntap * HPS);
    const int rem = s_begin - chunk * ntap * HPS;
    const int tap = rem / HPS;
    khalf = rem - tap * HPS;
    kc = chunk * CK + khalf * (CK / 2) + (UNI ? 0 : cc * EPC);
    kf = tap % p.KW;
    const int t2 = tap / p.KW;
    ke = t2 % p.KH;
    ka = t2 / p.KH;
  } else {
    const int kfirst = UNI ? s_begin * BK : k4;
    const int tap = kfirst / p.Cin;
    kc = kfirst - tap * p.Cin;
    kf = tap % p.KW;
    const int t2 = tap / p.KW;
    ke = t2 % p.KH;
    ka = t2 / p.KH;
  }
  int a_base[AR];          // UNI: a_off + the lane's chunk offset (elements)
  unsigned b_base[BR];     // UNI: byte offset of (row n, the lane's chunk) or the out-of-range marker
  unsigned kb = (unsigned)(s_begin * BK) * EL;   // UNI: byte offset of the slab inside a weight row
  if constexpr (UNI) {
#pragma unroll
    for (int j = 0; j < AR; ++j) a_base[j] = a_off[j] + cc * EPC;
#pragma unroll
    for (int j = 0; j < BR; ++j) b_base[j] = b_off[j] >= 0 ? (unsigned)(b_off[j] + cc * EPC) * EL : OOB;
  }

  auto load_slab = [&](int buf) {
    T *Ab = As + buf * BM * BK + wid_u * (64 / NCH) * BK;   // wave-uniform base; the DMA adds lane*16 B
    T *Bb = Bs + buf * BN * BK + wid_u * (64 / NCH) * BK;
    if constexpr (UNI) {
      const int delta = ((ka * p.H + ke) * p.W + kf) * p.Cin + kc;                       // scalar
      const unsigned tap = (1u << ka) | (1u << (8 + ke)) | (1u << (16 + kf));           // scalar
#pragma unroll
      for (int j = 0; j < AR; ++j) {
        const unsigned vo = ((a_msk[j] & tap) == tap) ? (unsigned)(a_base[j] + delta) * EL : OOB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in, (lds_ptr_t)(Ab + RP * j * BK), 16, vo, 0, 0, 0);
      }
#pragma unroll
      for (int j = 0; j < BR; ++j) {
        const unsigned vo = b_base[j] + kb;   // an out-of-range marker stays out of range: kb < 2^31
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_ptr_t)(Bb + RP * j * BK), 16, vo, 0, 0, 0);
      }
    } else {
      // branch-free: a k past K turns the tap mask into all-ones, which no row mask can satisfy
      const unsigned kbad = (k4 < p.K) ? 0u : 0xffffffffu;
      const int delta = ((ka * p.H + ke) * p.W + kf) * p.Cin + kc;
      const unsigned tap = ((1u << ka) | (1u << (8 + ke)) | (1u << (16 + kf))) | kbad;
#pragma unroll
      for (int j = 0; j < AR; ++j) {
        const unsigned good = ((a_msk[j] & tap) == tap) ? 0xffffffffu : 0u;
        const unsigned vo = (((unsigned)(a_off[j] + delta) * EL) & good) | (OOB & ~good);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in, (lds_ptr_t)(Ab + RP * j * BK), 16, vo, 0, 0, 0);
      }
#pragma unroll
      for (int j = 0; j < BR; ++j) {
        const unsigned good = (b_off[j] >= 0 ? 0xffffffffu : 0u) & ~kbad;
        const unsigned vo = (((unsigned)(b_off[j] + k4) * EL) & good) | (OOB & ~good);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_ptr_t)(Bb + RP * j * BK), 16, vo, 0, 0, 0);
      }
    }
  };
  auto advance_k = [&]() {
    k4 += BK;
    kb += BK * EL;
    if (p.kmode == 1) {  // next tap of the same channel chunk; after the last tap move to the next chunk
      if (BK < CK) {
        khalf ^= 1;
        kc += khalf ? CK / 2 : -(CK / 2);
        if (khalf) return;
      }
      if (++kf == p.KW) {
        kf = 0;
        if (++ke == p.KH) {
          ke = 0;
          if (++ka == p.KD) {
            ka = 0;
            kc += CK;
          }
        }
      }
      return;
    }
    kc += BK;
    while (kc >= p.Cin) {
      kc -= p.Cin;
      if (++kf == p.KW) {
        kf = 0;
        if (++ke == p.KH) {
          ke = 0;
          ++ka;
        }
      }
    }
  };
  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // RPF: the tile's residual is requested here, in front of the K loop, and consumed by the epilogue (pair IO, res_mode 1): the Cout-expanding
  // 1x1 layers of the bottlenecks have 4 .. 32 slabs of K and twice the output's bytes to move; with the residual requested block by block
  // in the epilogue the launch ran as K loop + (load -> store) chains, 90 us for 283 MB at 64 -> 256 / 96 x 320 x 4 against 46 without one.
  u32x4r rraw[RPF ? TM : 1][RPF ? TN : 1][4];
  if constexpr (RPF) {
    static_assert(PAIR == 2, "residual prefetch: the pair-IO epilogue");
    if (p.pio && p.res_mode == 1 && p.ksplit <= 1) conv_pio_res_prefetch<TM, TN>(p, rraw, m0, n0, wr, wc, lane);
  }
  // Software pipeline over two LDS buffers with a prefetch distance of two slabs.  The fragments of a slab's LAST k-step
  // are read into registers before the barrier, so after the barrier the buffer of the CURRENT slab is already free:
  // the DMA of slab s+2 is issued into it right there and has the last MFMA group of slab s plus all but the last
  // group of slab s+1 to land (a full slab of MFMA time), instead of being issued only at the top of the next slab.
  load_slab(0);
#pragma unroll
  for (int j = 1; j < NB; ++j)
    if (S > j) {
      advance_k();
      load_slab(j);
    }
  if constexpr (NB == 2) {
    lds_dma_wait_all();
  } else {
    const int newer = (S - 1 < NB - 1) ? S - 1 : NB - 1;      // slabs issued after slab 0
    lds_dma_wait_slabs<AR + BR>(newer);
  }
  // (NB > 2: a RAW barrier -- the workgroup fence of __syncthreads() waits for vmcnt(0), i.e. for every request of the ring, which makes
  // any ring deeper than two a two-deep one: seen in the ISA, round 5.  The clobbers keep LDS accesses on their side of it.)
  if constexpr (NB == 2) {
    __syncthreads();   // slabs 0 and 1 landed in every wave's view
  } else {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  }

  const int frow = (lane & 31) * BK, fsw = ((lane & 31) >> SW_SH) & SW_MSK, fh = lane >> 5;
#ifdef IVX_CONV_TIMELINE
  tl1 = __builtin_amdgcn_s_memrealtime();
#endif
  int cur = 0;
  for (int s = 0; s < S; ++s) {
    const T *Ac = As + cur * BM * BK + wr * TM * 32 * BK + frow;
    const T *Bc = Bs + cur * BN * BK + wc * TN * 32 * BK + frow;
    // one 16-byte read per operand tile and k-step: half-wave h takes chunk 2*kk + h (4 fp32 k -> 4 MFMAs 32x32x2,
    // or 8 bf16 k -> 1 MFMA 32x32x16); A and B use the same k permutation, so the sum over k is unchanged
    f32x4 fa[2][TM], fb[2][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) fa[0][i] = *reinterpret_cast<const f32x4 *>(Ac + i * 32 * BK + ((fh ^ fsw) * EPC));
#pragma unroll
    for (int j = 0; j < TN; ++j) fb[0][j] = *reinterpret_cast<const f32x4 *>(Bc + j * 32 * BK + ((fh ^ fsw) * EPC));
#pragma unroll
    for (int kk = 0; kk < NCH / 2; ++kk) {
      const int cb = kk & 1, nb = cb ^ 1;
      // PAIR: k-group 2t holds the hi halves of 16 channels, 2t+1 their lo halves; the odd step needs both register buffers
      // (hi in nb, lo in cb), so the read of the next group is issued after its MFMAs instead of before
      const bool defer = PAIR && (kk & 1);
      if (kk < NCH / 2 - 1) {
        if (!defer) {
#pragma unroll
          for (int i = 0; i < TM; ++i) fa[nb][i] = *reinterpret_cast<const f32x4 *>(Ac + i * 32 * BK + (((2 * (kk + 1) + fh) ^ fsw) * EPC));
#pragma unroll
          for (int j = 0; j < TN; ++j) fb[nb][j] = *reinterpret_cast<const f32x4 *>(Bc + j * 32 * BK + (((2 * (kk + 1) + fh) ^ fsw) * EPC));
        }
      } else {
        // every wave holds its last fragments of buffer `cur`; slab s+1 (other buffer) must have landed: each wave
        // waits for its own DMA (issued one barrier ago), the barrier then publishes all of them
        if constexpr (NB == 2) {
          lds_dma_wait_all();
        } else {           // slab s + 1 has landed when only the slabs issued after it are outstanding: min(NB - 2, S - s - 2) of them
          int newer = S - s - 2;
          newer = newer < 0 ? 0 : (newer > NB - 2 ? NB - 2 : newer);
          lds_dma_wait_slabs<AR + BR>(newer);
        }
        if constexpr (NB == 2) {
          __syncthreads();
        } else {           // this wave's fragment reads of slab s are complete (they sit in registers: the compiler waited for them)
          __builtin_amdgcn_s_waitcnt(0xc07f);      // lgkmcnt(0)
          asm volatile("" ::: "memory");
          __builtin_amdgcn_s_barrier();
          asm volatile("" ::: "memory");
        }
        if (s + NB < S) {
          advance_k();
          load_slab(cur);
        }
      }
      if constexpr (EL == 4) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cb][i][q], fb[cb][j][q], acc[i][j], 0, 0, 0);
      } else if constexpr (EL == 2 && PAIR) {
        static_assert(NCH % 4 == 0, "pair operands: a slab holds whole [hi16 | lo16] groups");
        if ((kk & 1) == 0) {   // hi * hi
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
              acc[i][j] = pair_mfma<PAIR>(fa[cb][i], fb[cb][j], acc[i][j]);
        } else {               // hi * lo + lo * hi (hi: buffer nb, lo: buffer cb)
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
              acc[i][j] = pair_mfma<PAIR>(fa[nb][i], fb[cb][j], acc[i][j]);
              acc[i][j] = pair_mfma<PAIR>(fa[cb][i], fb[nb][j], acc[i][j]);
            }
          if (kk < NCH / 2 - 1) {
#pragma unroll
            for (int i = 0; i < TM; ++i) fa[nb][i] = *reinterpret_cast<const f32x4 *>(Ac + i * 32 * BK + (((2 * (kk + 1) + fh) ^ fsw) * EPC));
#pragma unroll
            for (int j = 0; j < TN; ++j) fb[nb][j] = *reinterpret_cast<const f32x4 *>(Bc + j * 32 * BK + (((2 * (kk + 1) + fh) ^ fsw) * EPC));
          }
        }
      } else if constexpr (EL == 2) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[cb][i]), __builtin_bit_cast(bf16x8, fb[cb][j]),
                                                               acc[i][j], 0, 0, 0);
      } else {   // e4m3: the lane's 16-byte chunk holds 16 k -> two MFMAs 32x32x16 (8 bytes per operand each)
        typedef i64_t i64x2 __attribute__((ext_vector_type(2)));
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(__builtin_bit_cast(i64x2, fa[cb][i])[q], __builtin_bit_cast(i64x2, fb[cb][j])[q],
                                                                    acc[i][j], 0, 0, 0);
      }
    }
    cur = cur + 1 == NB ? 0 : cur + 1;
  }
#ifdef IVX_CONV_TIMELINE
  tl2 = __builtin_amdgcn_s_memrealtime();
#endif
  if (p.ksplit > 1) {
    // raw partial sums; ivx split-K reduce kernel applies the epilogue
    float *part = p.partial + ((size_t)blockIdx.y * 8 * p.q_count + ct) * BM * p.Cout;   // rows of this tile, slice y
    const int col_l = lane & 31, hh = lane >> 5;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + (wr * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
        if (m >= p.M) continue;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const int n = n0 + (wc * TN + j) * 32 + col_l;
          if (n < p.Cout) part[(size_t)(m - m0) * p.Cout + n] = acc[i][j][r];
        }
      }
    return;
  }
  if constexpr (PAIR == 2) {
    if (p.pio) {        // fill_params guarantees out_mode 0, Cout % 16 == 0, no grouped launch
      static_assert(sizeof(smem) >= (size_t)NT / 64 * 4096, "4 KB of staging LDS per wave for the transposed epilogue");
      PairIO io = conv_pair_io(p);
      // the three multipliers are the same in every lane: say so (SGPRs).  As VGPR values they were spilled at four workgroups per CU and
      // RELOADED FROM SCRATCH in every pass of the epilogue -- two more memory round trips per pass, each behind an `s_waitcnt vmcnt(0)` that
      // also waits for the pass's own stores (ISA, round 5)
      io.inv_in = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, io.inv_in)));
      io.inv_res = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, io.inv_res)));
      io.s_out = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, io.s_out)));
      io.post = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, p.post_scale)));      // (the compiler kept
      // {post_scale, ksplit, partial} of the by-value ConvParams in a 16-byte private array and re-read post_scale from scratch in every pass)
      if (p.out_scale_p && blockIdx.x == 0 && tid == 0) *p.out_scale_p = io.s_out;     // (workgroup 0 owns M-tile q_begin of XCD 0: never out of range)
      if constexpr (RPF) conv_epilogue_wide_pio<TM, TN, 1>(p, io, acc, m0, n0, wr, wc, lane, reinterpret_cast<float *>(smem) + wid_u * 1024, wid_u, NT / 64, rraw);
      else conv_epilogue_wide_pio<TM, TN>(p, io, acc, m0, n0, wr, wc, lane, reinterpret_cast<float *>(smem) + wid_u * 1024, wid_u, NT / 64);
#ifdef IVX_CONV_TIMELINE
      if (p.tl && tid == 0) {
        unsigned long long *t = p.tl + (size_t)blockIdx.x * 8;
        t[0] = tl0; t[1] = tl1; t[2] = tl2; t[3] = __builtin_amdgcn_s_memrealtime();
        t[4] = __builtin_amdgcn_s_getreg(63492); t[5] = __builtin_amdgcn_s_getreg(63508);
      }
#endif
      return;
    }
  }
  if (p.out_mode == 0 && !p.out_bf16 && !p.out_fp8 && (p.Cout & 3) == 0 && !p.narrow_epilogue) {
    static_assert(sizeof(smem) >= (size_t)NT / 64 * 4096, "4 KB of staging LDS per wave for the transposed epilogue");
    conv_epilogue_wide<TM, TN>(p, acc, m0, n0, wr, wc, lane, reinterpret_cast<float *>(smem) + wid_u * 1024, gz * (size_t)p.g_out);
    return;
  }
  if constexpr (EL < 4)    // e4m3 outputs come from bf16 (the stem) or e4m3 inputs only: keeps the fp32 instantiations as they were
  if (p.out_mode == 0 && p.out_fp8 && (p.Cout & 15) == 0 && !p.narrow_epilogue) {
    constexpr int NJ = (TN % 2 == 0 && sizeof(smem) >= (size_t)NT / 64 * 8192) ? 2 : 1;
    static_assert(sizeof(smem) >= (size_t)NT / 64 * 4096 * NJ, "staging LDS for the transposed fp8 epilogue");
    conv_epilogue_wide_fp8<TM, TN, NJ>(p, acc, m0, n0, wr, wc, lane, reinterpret_cast<float *>(smem) + wid_u * 1024 * NJ, gz * (size_t)p.g_out);
    return;
  }
  if (p.out_mode == 0 && p.out_bf16 && (p.Cout & 7) == 0 && !p.narrow_epilogue) {
    constexpr int NJ = (TN % 2 == 0 && sizeof(smem) >= (size_t)NT / 64 * 8192) ? 2 : 1;
    static_assert(sizeof(smem) >= (size_t)NT / 64 * 4096 * NJ, "staging LDS for the transposed bf16 epilogue");
    conv_epilogue_wide_bf16<TM, TN, NJ>(p, acc, m0, n0, wr, wc, lane, reinterpret_cast<float *>(smem) + wid_u * 1024 * NJ, gz * (size_t)p.g_out);
    return;
  }
  conv_epilogue<TM, TN>(p, acc, m0, n0, wr, wc, lane, gz * (size_t)p.g_out);
}

// Epilogue + store of one output element (row m, column n) from its finished accumulator: shared by the validation
// kernel and the split-K reduction.
#if IVX_CONV_TU == 0
// (pair IO: the scales of conv_pair_io; returns |stored value| before the output scale, for the amax slots)
__device__ __forceinline__ float conv_store_one(const ConvParams &p, int m, int n, float acc, const PairIO io = {1.0f, 1.0f, 1.0f, false, 1.0f}) {
  if (p.pio) {
    const size_t idx = (size_t)m * p.Cout + n;
    size_t ridx = idx;
    if (p.res_mode == 2) ridx = res2_row_base(m, p.Ho, p.Wo, p.rH, p.rW, p.Cout) + n;
    const float v = conv_finish(p, acc, (p.scale ? p.scale[n] : 1.0f) * io.inv_in, p.shift ? p.shift[n] : 0.0f, ridx, p.res_pair ? io.inv_res : 1.0f);
    float y = p.out_pair ? v * io.s_out : v;
    if (p.out_pair && io.sat) y = (y > 65504.f && y < __builtin_inff()) ? 65504.f : ((y < -65504.f && y > -__builtin_inff()) ? -65504.f : y);
    conv_st_out(p, idx, y);
    return fabsf(v);
  }
  if (p.out_mode == 1) {
    const int tap = n / p.Cr, ch = n - tap * p.Cr;
    const int a2 = tap >> 2, e2 = (tap >> 1) & 1, f2 = tap & 1;
    const size_t o = up2_row_base(m, p.D, p.H, p.W, p.Cr) + (((size_t)a2 * 2 * p.H + e2) * 2 * p.W + f2) * p.Cr + ch;
    conv_st_out(p, o, conv_finish(p, acc, p.scale ? p.scale[ch] : 1.0f, p.shift ? p.shift[ch] : 0.0f, o));
    return 0.f;
  }
  const size_t idx = (size_t)m * p.Cout + n;
  size_t ridx = idx;
  if (p.res_mode == 2) ridx = res2_row_base(m, p.Ho, p.Wo, p.rH, p.rW, p.Cout) + n;
  conv_st_out(p, idx, conv_finish(p, acc, p.scale ? p.scale[n] : 1.0f, p.shift ? p.shift[n] : 0.0f, ridx));
  return 0.f;
}

// Split-K reduction: out = epilogue(sum over slices in slice order) -- deterministic.  Partial rows are compact:
// row cr = (xcd*q_count + local_tile)*bm + row_in_tile  <->  m = (xcd*q_total + q_begin + local_tile)*bm + row_in_tile.
__global__ __launch_bounds__(256) void conv_splitk_reduce_kernel(const ConvParams p) {
  const size_t rows = (size_t)8 * p.q_count * p.bm;
  const size_t total = rows * p.Cout;
  PairIO io = {1.0f, 1.0f, 1.0f, false, p.post_scale};
  if (p.pio) {
    io = conv_pair_io(p);
    if (p.out_scale_p && blockIdx.x == 0 && threadIdx.x == 0) *p.out_scale_p = io.s_out;
  }
  float omax = 0.f;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const int n = (int)(idx % p.Cout);
    const size_t cr = idx / p.Cout;
    const int ctile = (int)(cr / p.bm), rit = (int)(cr - (size_t)ctile * p.bm);
    const int xcd = ctile / p.q_count, lt = ctile - xcd * p.q_count;
    const long long m = (long long)(xcd * p.q_total + p.q_begin + lt) * p.bm + rit;
    if (m >= p.M) continue;
    float acc = 0.f;
    for (int z = 0; z < p.ksplit; ++z) acc += p.partial[(size_t)z * total + idx];
    omax = fmaxf(omax, conv_store_one(p, (int)m, n, acc, io));
  }
  if (p.pio && p.amax_out) {        // (uniform)
    __shared__ float red[4];
    ivx_amax_commit_wg(p.amax_out, omax, red, (int)blockIdx.x);
  }
}

// The same reduction for pair IO (the chained fp16-pair trunk), four channels per thread: 16-byte partial loads, the epilogue of
// conv_pio_finish4 (8-byte hi / lo stores).  32-bit indexing: split K is only planned for small outputs (the launcher checks).
__global__ __launch_bounds__(256) void conv_splitk_reduce_pio_kernel(const ConvParams p) {
  const PairIO io = conv_pair_io(p);
  if (p.out_scale_p && blockIdx.x == 0 && threadIdx.x == 0) *p.out_scale_p = io.s_out;
  const unsigned C4 = (unsigned)p.Cout >> 2, rows = 8u * (unsigned)p.q_count * (unsigned)p.bm, total4 = rows * C4;
  const size_t total = (size_t)rows * p.Cout;
  float omax = 0.f;
  for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += gridDim.x * blockDim.x) {
    const unsigned cr = i / C4, nb = (i - cr * C4) * 4;
    const unsigned ctile = cr / (unsigned)p.bm, rit = cr - ctile * (unsigned)p.bm;
    const unsigned xcd = ctile / (unsigned)p.q_count, lt = ctile - xcd * (unsigned)p.q_count;
    const long long m = (long long)(xcd * p.q_total + p.q_begin + lt) * p.bm + rit;
    if (m >= p.M) continue;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int z = 0; z < p.ksplit; ++z) acc += *reinterpret_cast<const f32x4 *>(p.partial + (size_t)z * total + (size_t)cr * p.Cout + nb);
    f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sf = {0.f, 0.f, 0.f, 0.f};
    if (p.scale) sc = *reinterpret_cast<const f32x4 *>(p.scale + nb);
    if (p.shift) sf = *reinterpret_cast<const f32x4 *>(p.shift + nb);
    sc *= io.inv_in;
    omax = fmaxf(omax, conv_pio_finish4(p, io, acc, (int)m, (int)nb, sc, sf));
  }
  if (p.amax_out) {        // (uniform)
    __shared__ float red[4];
    ivx_amax_commit_wg(p.amax_out, omax, red, (int)blockIdx.x);
  }
}

// Validation kernel: one thread per output element, sequential fmaf over (kd,kh,kw,ci).
__global__ __launch_bounds__(256) void conv_naive_f32_kernel(const ConvParams p) {
  const size_t total = (size_t)p.M * p.Cout;
  PairIO io = {1.0f, 1.0f, 1.0f, false, p.post_scale};
  if (p.pio) {
    io = conv_pair_io(p);
    if (p.out_scale_p && blockIdx.x == 0 && threadIdx.x == 0) *p.out_scale_p = io.s_out;
  }
  float omax = 0.f;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const int n = (int)(idx % p.Cout);
    const int m = (int)(idx / p.Cout);
    const int ow = m % p.Wo;
    int t = m / p.Wo;
    const int oh = t % p.Ho;
    t /= p.Ho;
    const int od = t % p.Do;
    const int b = t / p.Do;
    float acc = 0.f;
    for (int a = 0; a < p.KD; ++a) {
      const int id = od * p.sd - p.pd + a;
      if ((unsigned)id >= (unsigned)p.D) continue;
      for (int e = 0; e < p.KH; ++e) {
        const int ih = oh * p.sh - p.ph + e;
        if ((unsigned)ih >= (unsigned)p.H) continue;
        for (int f = 0; f < p.KW; ++f) {
          const int iw = ow * p.sw - p.pw + f;
          if ((unsigned)iw >= (unsigned)p.W) continue;
          const size_t xo = ((((size_t)b * p.D + id) * p.H + ih) * p.W + iw) * p.Cin;
          const int tap = (a * p.KH + e) * p.KW + f;
          const int ntap = p.KD * p.KH * p.KW;
          const size_t wo = (size_t)n * p.K;
          const int ck = p.in_fp8 ? 128 : (p.in_bf16 ? 64 : 32);   // channels per 128-byte chunk of the chunk-major order
          if (p.in_pair) {   // the value a (hi, lo) pair stands for is hi + lo (exact in fp32): plain fp32 products of those
            for (int c = 0; c < p.Cin / 2; ++c) {
              const int eh = (c >> 4) * 32 + (c & 15), el = eh + 16;
              const size_t wh = p.kmode == 1 ? (size_t)((eh / ck) * ntap + tap) * ck + (eh % ck) : (size_t)tap * p.Cin + eh;
              if (p.in_pair == 2) {
                const _Float16 *xp = reinterpret_cast<const _Float16 *>(p.in) + xo, *wp = reinterpret_cast<const _Float16 *>(p.wgt) + wo;
                acc = fmaf((float)xp[eh] + (float)xp[el], (float)wp[wh] + (float)wp[wh + 16], acc);
              } else {
                const __bf16 *xp = reinterpret_cast<const __bf16 *>(p.in) + xo, *wp = reinterpret_cast<const __bf16 *>(p.wgt) + wo;
                acc = fmaf((float)xp[eh] + (float)xp[el], (float)wp[wh] + (float)wp[wh + 16], acc);
              }
            }
            continue;
          }
          for (int c = 0; c < p.Cin; ++c) {
            const size_t wi = wo + (p.kmode == 1 ? (size_t)((c / ck) * ntap + tap) * ck + (c % ck) : (size_t)tap * p.Cin + c);
            const float xv = p.in_fp8 ? fp8_to_f32(reinterpret_cast<const unsigned char *>(p.in)[xo + c])
                                      : (p.in_bf16 ? (float)reinterpret_cast<const __bf16 *>(p.in)[xo + c] : p.in[xo + c]);
            const float wv = p.in_fp8 ? fp8_to_f32(reinterpret_cast<const unsigned char *>(p.wgt)[wi])
                                      : (p.in_bf16 ? (float)reinterpret_cast<const __bf16 *>(p.wgt)[wi] : p.wgt[wi]);
            acc = fmaf(xv, wv, acc);
          }
        }
      }
    }
    omax = fmaxf(omax, conv_store_one(p, m, n, acc, io));
  }
  if (p.pio && p.amax_out) {        // (uniform)
    __shared__ float red[4];
    ivx_amax_commit_wg(p.amax_out, omax, red, (int)blockIdx.x);
  }
}

#ifdef IVX_CONV_TIMELINE
static unsigned long long *g_timeline = nullptr;
extern "C" int ivx_conv_set_timeline(void *buf) { g_timeline = (unsigned long long *)buf; return 0; }
#endif
static thread_local int g_plan_mode = 0;
// Tuning knob (A/B; per calling thread): 1 = the round-1 tile rule of plan_conv, 0 (default) = the scored choice.
extern "C" int ivx_conv_set_plan_mode(int mode) {
  g_plan_mode = mode;
  return IVX_OK;
}

static thread_local int g_narrow_epilogue = 0;
// Tuning knob (A/B experiments; per calling thread): 1 = one-channel-per-lane epilogue stores in the LDS-DMA kernel,
// 0 (default) = the LDS-transposed wide-store epilogue where it applies.
extern "C" int ivx_conv_set_epilogue_mode(int narrow) {
  g_narrow_epilogue = narrow ? 1 : 0;
  return IVX_OK;
}

static int fill_params(const ivx_conv_desc *d, const void *in, const void *wgt, const float *scale,
                       const float *shift, const void *res, void *out, ConvParams *p, const ivx_pair_io *io = nullptr) {
  IVX_REQUIRE(d && in && wgt && out, "ivx_conv_fwd: null argument");
  IVX_REQUIRE(d->B > 0 && d->D > 0 && d->H > 0 && d->W > 0 && d->Cin > 0 && d->Cout > 0, "ivx_conv_fwd: non-positive dims");
  IVX_REQUIRE(d->Cin % 4 == 0, "ivx_conv_fwd: Cin (%d) must be a multiple of 4 (pad the input channels)", d->Cin);
  IVX_REQUIRE(d->KD > 0 && d->KH > 0 && d->KW > 0 && d->sd > 0 && d->sh > 0 && d->sw > 0, "ivx_conv_fwd: bad kernel/stride");
  IVX_REQUIRE(d->pd >= 0 && d->ph >= 0 && d->pw >= 0, "ivx_conv_fwd: negative padding");
  int32_t Do, Ho, Wo;
  if (ivx_conv_out_dims(d, &Do, &Ho, &Wo) != IVX_OK) return IVX_ERR_INVALID_ARG;
  const int64_t M = (int64_t)d->B * Do * Ho * Wo;
  const bool pair = d->in_dtype == IVX_BF16_PAIR || d->in_dtype == IVX_F16_PAIR;
  const int cin_el = pair ? 2 * d->Cin : d->Cin;     // stored elements per voxel (a pair tensor holds two bf16 per channel)
  const int64_t K = (int64_t)d->KD * d->KH * d->KW * cin_el;
  IVX_REQUIRE(M < (1LL << 31) - 512 && K < (1LL << 30), "ivx_conv_fwd: problem too large for 32-bit row index");
  IVX_REQUIRE((int64_t)d->B * d->D * d->H * d->W < (1LL << 31), "ivx_conv_fwd: input voxel count exceeds 2^31");
  IVX_REQUIRE(d->res_mode >= 0 && d->res_mode <= 2, "ivx_conv_fwd: bad res_mode");
  IVX_REQUIRE(d->out_mode == 0 || (d->out_mode == 1 && d->KD == 1 && d->KH == 1 && d->KW == 1 && d->sd == 1 && d->sh == 1 && d->sw == 1 &&
                                     d->pd == 0 && d->ph == 0 && d->pw == 0 && d->Cout % 8 == 0 && d->res_mode != 2),
              "ivx_conv_fwd: out_mode 1 (ConvTranspose k2 s2) needs a 1x1x1 stride-1 GEMM with Cout = 8 * real channels");
  IVX_REQUIRE(d->in_dtype >= IVX_F32 && d->in_dtype <= IVX_F16_PAIR && d->out_dtype >= IVX_F32 && (d->out_dtype <= IVX_FP8 || (io && d->out_dtype == IVX_F16_PAIR)),
              "ivx_conv_fwd: dtypes are IVX_F32 (0), IVX_BF16 (1), IVX_FP8 (2) or, for the input, IVX_BF16_PAIR (3) / IVX_F16_PAIR (4); an IVX_F16_PAIR "
              "output needs ivx_conv_fwd_pio");
  if (io) {
    IVX_REQUIRE(d->in_dtype == IVX_F16_PAIR && d->out_mode == 0 && d->Cout % 4 == 0 && (d->out_dtype == IVX_F32 || d->out_dtype == IVX_F16_PAIR),
                "ivx_conv_fwd_pio: needs an IVX_F16_PAIR input, out_mode 0, Cout %% 4 == 0 and an IVX_F32 or IVX_F16_PAIR output");
    IVX_REQUIRE(io->res_dtype == IVX_F32 || io->res_dtype == IVX_F16_PAIR, "ivx_conv_fwd_pio: res_dtype is IVX_F32 or IVX_F16_PAIR");
    IVX_REQUIRE(d->out_dtype != IVX_F16_PAIR || (d->Cout % 16 == 0 && io->out_scale && io->amax_in && (d->res_mode == 0 || io->amax_res)),
                "ivx_conv_fwd_pio: a pair output needs Cout %% 16 == 0, out_scale, amax_in (and amax_res with a residual)");
    IVX_REQUIRE(!(d->res_mode && io->res_dtype == IVX_F16_PAIR) || (d->Cout % 16 == 0 && io->res_scale), "ivx_conv_fwd_pio: a pair residual needs Cout %% 16 == 0 and res_scale");
  }
  IVX_REQUIRE(!pair || (d->Cin % 16 == 0 && d->out_mode == 0), "ivx_conv_fwd: bf16-pair input needs Cin %% 16 == 0 and out_mode 0");
  IVX_REQUIRE(d->in_dtype != IVX_FP8 || d->Cin % 16 == 0, "ivx_conv_fwd: fp8 input needs Cin %% 16 == 0");
  IVX_REQUIRE((d->in_dtype != IVX_FP8 && d->out_dtype != IVX_FP8) || d->out_mode == 0, "ivx_conv_fwd: fp8 is built for out_mode 0 only");
  const int ck = d->in_dtype == IVX_FP8 ? 128 : (d->in_dtype == IVX_BF16 ? 64 : 32);   // pair: 32 real channels = 64 stored elements
  IVX_REQUIRE(d->wgt_layout == 0 || (d->wgt_layout == 1 && d->Cin % ck == 0),
              "ivx_conv_fwd: wgt_layout 1 needs Cin %% %d == 0 (128-byte channel chunks)", ck);
  IVX_REQUIRE(d->in_dtype == IVX_F32 || d->Cin % 8 == 0, "ivx_conv_fwd: bf16 input needs Cin %% 8 == 0");
  IVX_REQUIRE(d->res_mode == 0 || res, "ivx_conv_fwd: res_mode set but res is NULL");
  if (d->res_mode == 2) {
    IVX_REQUIRE(Do == 1 && d->res_h > 0 && d->res_w > 0, "ivx_conv_fwd: res_mode 2 needs a 2-D output and res dims");
  }
  p->in = (const float *)in; p->wgt = (const float *)wgt; p->scale = scale; p->shift = shift;
  p->res = d->res_mode ? (const float *)res : nullptr; p->out = (float *)out;
  p->in_bf16 = d->in_dtype == IVX_BF16 || pair; p->out_bf16 = d->out_dtype == IVX_BF16;
  p->in_pair = d->in_dtype == IVX_BF16_PAIR ? 1 : (d->in_dtype == IVX_F16_PAIR ? 2 : 0);
  p->in_fp8 = d->in_dtype == IVX_FP8; p->out_fp8 = d->out_dtype == IVX_FP8;
  p->res_scale = d->res_scale == 0.f ? 1.0f : d->res_scale;
  p->B = d->B; p->D = d->D; p->H = d->H; p->W = d->W; p->Cin = cin_el;
  p->Cout = d->Cout; p->KD = d->KD; p->KH = d->KH; p->KW = d->KW;
  p->sd = d->sd; p->sh = d->sh; p->sw = d->sw; p->pd = d->pd; p->ph = d->ph; p->pw = d->pw;
  p->Do = Do; p->Ho = Ho; p->Wo = Wo; p->M = (int)M; p->K = (int)K;
  p->relu = d->relu; p->res_mode = d->res_mode; p->rH = d->res_h; p->rW = d->res_w; p->kmode = d->wgt_layout;
  p->out_mode = d->out_mode; p->Cr = d->out_mode == 1 ? d->Cout / 8 : d->Cout; p->res_after_act = d->res_after_act;
  p->post_scale = d->post_scale == 0.f ? 1.0f : d->post_scale;
  p->ksplit = 1; p->partial = nullptr; p->q_total = 0; p->q_begin = 0; p->q_count = 0; p->bm = 0;
  p->stagger_ticks = 0; p->stagger_shift = 0; p->stagger_first = 0;
  p->groups = 1; p->g_in = p->g_w = p->g_out = 0;
  p->narrow_epilogue = g_narrow_epilogue;
#ifdef IVX_CONV_TIMELINE
  p->tl = g_timeline;
#endif
  p->cp_src = nullptr; p->cp_dst = nullptr;
  p->pio = 0; p->in_scale_p = nullptr; p->out_pair = 0; p->res_pair = 0; p->res_scale_p = nullptr; p->out_scale_p = nullptr;
  p->amax_in = p->amax_res = nullptr; p->amax_out = nullptr; p->wbound = 0.f; p->sbound = 0.f;
  if (io) {
    p->pio = 1;
    p->in_scale_p = io->in_scale;
    p->out_pair = d->out_dtype == IVX_F16_PAIR; p->out_bf16 = 0;
    p->res_pair = d->res_mode && io->res_dtype == IVX_F16_PAIR;
    p->res_scale_p = io->res_scale; p->out_scale_p = p->out_pair ? io->out_scale : nullptr;
    p->amax_in = io->amax_in; p->amax_res = d->res_mode ? io->amax_res : nullptr; p->amax_out = io->amax_out;
    p->wbound = io->wbound; p->sbound = io->sbound;
  }
  return IVX_OK;
}

extern "C" int ivx_conv_out_dims(const ivx_conv_desc *d, int32_t *Do, int32_t *Ho, int32_t *Wo) {
  IVX_REQUIRE(d && Do && Ho && Wo, "ivx_conv_out_dims: null argument");
  const int od = (d->D + 2 * d->pd - d->KD) / d->sd + 1;
  const int oh = (d->H + 2 * d->ph - d->KH) / d->sh + 1;
  const int ow = (d->W + 2 * d->pw - d->KW) / d->sw + 1;
  IVX_REQUIRE(d->D + 2 * d->pd >= d->KD && d->H + 2 * d->ph >= d->KH && d->W + 2 * d->pw >= d->KW && od > 0 && oh > 0 && ow > 0,
              "ivx_conv_out_dims: kernel larger than padded input");
  *Do = od; *Ho = oh; *Wo = ow;
  return IVX_OK;
}

// Generic kernel (64-bit addressing, any kernel extent; weight layout 0 only): the fallback for tensors >= 2 GiB.
template <int TM, int TN, int WR, int WC>
static void launch_cfg(const ConvParams &p, hipStream_t st) {
  constexpr int BM = WR * TM * 32, BN = WC * TN * 32;
  dim3 grid((p.M + BM - 1) / BM, (p.Cout + BN - 1) / BN);
  hipLaunchKernelGGL((conv_igemm_f32_kernel<TM, TN, WR, WC>), grid, dim3(256), 0, st, p);
}

#endif   // IVX_CONV_TU == 0

template <typename T, int TM, int TN, int WR, int WC, int BK, int WPE = 1, int PAIR = 0, int NB = 2, int RPF = 0>
static void launch_v4(ConvParams &p, hipStream_t st) {
  constexpr int BM = WR * TM * 32, BN = WC * TN * 32;
  const int64_t in_bytes = (int64_t)p.B * p.D * p.H * p.W * p.Cin * sizeof(T);
  const int64_t w_bytes = (int64_t)p.Cout * p.K * sizeof(T);
  const long long Mt = (p.M + BM - 1) / BM, Nt = (p.Cout + BN - 1) / BN;
  p.bm = BM;
  if (p.q_total == 0) {  // whole problem in one launch
    p.q_total = (int)((Mt + 7) / 8);
    p.q_begin = 0;
    p.q_count = p.q_total;
  }
  const long long g1 = 8LL * p.q_count * Nt;
  const dim3 grid((unsigned)g1, p.ksplit > 1 ? p.ksplit : 1, p.groups > 1 ? p.groups : 1);
  // every slab lies inside one filter tap -> uniform K-loop state
  const bool uni = p.kmode == 1 || p.Cin % BK == 0;
  if constexpr (NB > 2) {      // deep ring: the uniform K-loop state only (every pair layer of the trunk: chunk-major filters)
    if (!uni) {
      launch_v4<T, TM, TN, WR, WC, BK, WPE, PAIR, 2>(p, st);
      return;
    }
    auto kern = conv_igemm_v4_kernel<T, TM, TN, WR, WC, BK, WPE, 1, PAIR, NB, RPF>;
    hipLaunchKernelGGL(kern, grid, dim3(64 * WR * WC), 0, st, p, (unsigned)in_bytes, (unsigned)w_bytes);
  } else {
    auto kern = uni ? conv_igemm_v4_kernel<T, TM, TN, WR, WC, BK, WPE, 1, PAIR, 2, RPF> : conv_igemm_v4_kernel<T, TM, TN, WR, WC, BK, WPE, 0, PAIR>;
    hipLaunchKernelGGL(kern, grid, dim3(64 * WR * WC), 0, st, p, (unsigned)in_bytes, (unsigned)w_bytes);
  }
}



struct ConvPlan {
  int cfg;       // tile / kernel selector (see launch_one)
  int ksplit;    // > 1: split K over the whole problem (small outputs)
  int tail_ks;   // > 1: the last partial round of M-tiles runs as a second launch with K split tail_ks ways
  int q_total;   // M-tiles per XCD (LDS-DMA kernels)
  int qa;        // M-tiles per XCD covered by the first launch when tail_ks > 1
  int bm;        // tile rows of the chosen config
  int64_t ws_bytes;
};


#if IVX_CONV_TU == 0
static thread_local int g_halo_mode = -1;   // A/B knob (ivx_conv_set_halo_mode): -1 default rule | 0 never | 1 .. 14 / 30 .. 34 force that z-halo config of the stride-1 layers, 21 .. 24 / 41 .. 45 of the z-stride-2 layers
extern "C" int ivx_conv_set_halo_mode(int mode) {
  g_halo_mode = mode;
  return IVX_OK;
}
static thread_local int g_tile_override = 0;
// Tuning knob (A/B experiments, tools/conv_bench.py; per calling thread): 0 = automatic choice, else force a tile
// config of launch_one (1..7 generic kernel, 41..53 LDS-DMA fp32, 61..73 LDS-DMA bf16).
extern "C" int ivx_conv_set_tile_override(int cfg) {
  g_tile_override = cfg;
  return IVX_OK;
}

struct TileInfo { int bm, bn, bk, wg_per_cu; };
static bool tile_info(int cfg, TileInfo *t) {
  switch (cfg) {
    case 41: *t = {128, 128, 32, 2}; return true;
    case 43: *t = {128, 64, 32, 3}; return true;
    case 44: *t = {128, 32, 32, 4}; return true;
    case 46: *t = {64, 64, 32, 5}; return true;
    case 51: *t = {128, 128, 16, 3}; return true;
    case 53: *t = {128, 64, 16, 5}; return true;
    case 52: *t = {256, 64, 16, 3}; return true;
    case 54: *t = {128, 128, 16, 4}; return true;
    case 55: *t = {128, 128, 16, 5}; return true;
    case 56: *t = {128, 64, 16, 6}; return true;
    case 57: *t = {256, 64, 16, 4}; return true;
    case 58: *t = {256, 128, 16, 2}; return true;
    case 59: *t = {128, 256, 16, 2}; return true;
    case 47: *t = {64, 64, 16, 6}; return true;
    case 48: *t = {64, 128, 16, 5}; return true;
    case 49: *t = {128, 64, 16, 5}; return true;
    case 74: *t = {128, 128, 32, 4}; return true;
    case 91: *t = {128, 128, 64, 4}; return true;
    case 92: *t = {128, 64, 64, 5}; return true;
    case 93: *t = {64, 64, 64, 6}; return true;
    case 94: *t = {128, 128, 128, 2}; return true;
    case 81: *t = {256, 128, 32, 2}; return true;
    case 82: *t = {256, 256, 32, 1}; return true;
    case 83: *t = {256, 128, 64, 1}; return true;
    // bf16 (cfg + 20: same tile, same LDS bytes, BK counts bf16 elements)
    case 61: *t = {128, 128, 64, 2}; return true;
    case 63: *t = {128, 64, 64, 3}; return true;
    case 64: *t = {128, 32, 64, 4}; return true;
    case 66: *t = {64, 64, 64, 5}; return true;
    case 67: *t = {64, 64, 32, 6}; return true;
    case 71: *t = {128, 128, 32, 3}; return true;
    case 73: *t = {128, 64, 32, 5}; return true;
    case 75: *t = {128, 64, 32, 5}; return true;
    case 76: *t = {256, 64, 32, 4}; return true;
    case 85: *t = {128, 256, 32, 2}; return true;
    case 72: *t = {256, 64, 32, 3}; return true;
    // deep-ring forms of 66 / 74 (pair operands): NB = 4 -> 64 KB of LDS, two workgroups per CU; NB = 3 -> 48 KB, three
    case 166: *t = {64, 64, 64, 2}; return true;
    case 174: *t = {128, 128, 32, 2}; return true;
    case 474: *t = {128, 128, 32, 3}; return true;
    case 475: *t = {128, 128, 32, 2}; return true;
    case 574: *t = {128, 128, 32, 2}; return true;
    case 177: *t = {128, 128, 32, 2}; return true;
    case 179: *t = {128, 128, 64, 1}; return true;
    case 77: *t = {128, 128, 32, 2}; return true;
    case 183: *t = {256, 256, 32, 1}; return true;
    default: return false;
  }
}

static bool dma_applicable(const ConvParams &p) {
  const int el = p.in_fp8 ? 1 : (p.in_bf16 ? 2 : 4);
  const int64_t in_b = (int64_t)p.B * p.D * p.H * p.W * p.Cin * el, w_b = (int64_t)p.Cout * p.K * el;
  return in_b < (1LL << 31) && w_b < (1LL << 31) && p.KD <= 8 && p.KH <= 8 && p.KW <= 8;
}

// Kernel / tile / split-K choice (measured per layer on MI355X with tools/conv_bench.py).
//  * Big problems want the 128-row tiles with the best MFMA : staging ratio; BK 16 keeps LDS at 32 KB and the
//    128 x 128 kernel is compiled for <= 128 registers (launch bound 4 waves/SIMD), so four workgroups share a CU
//    (+2.5 % per layer over three; measured).  When 128 x 128 tiles would not fill the workgroup slots several
//    times over -- every ResNet/FPN layer at KITTI resolution, the indoor necks -- 64 x 64 tiles win by occupancy.
//  * When even 64 x 64 tiles leave most of the 256 CUs idle and K is long (ResNet stage 4, FPN laterals on C5, the
//    coarse levels of the indoor necks), K is split across grid.y and a second pass sums the slices.
//  * Grid tail: with uniform tiles the last partial round costs a whole workgroup time on a few CUs while the rest
//    idle (measured: 10 044 tiles on 768 slots = 13.08 rounds runs 4 % slower per tile than exactly 13 rounds).
//    When the remainder is at most half a round, the full rounds run as one launch and the remainder as a second
//    launch with K split so that it fills every slot once (needs the caller's workspace: ivx_conv_fwd_ws).
//  * The LDS-DMA kernel (cfg 4x/5x fp32, 6x/7x bf16) is used whenever its preconditions hold, else the same tile on
//    the generic kernel (cfg 1..7).
static int conv_skinny_rule() {
  static const int v = getenv("IVX_CONV_SKINNY") ? atoi(getenv("IVX_CONV_SKINNY")) : 1;
  return v;
}
static ConvPlan plan_conv(const ConvParams &p, bool allow_ws) {
  ConvPlan pl = {g_tile_override, 1, 1, 0, 0, 0, 0};
  const bool dma_ok = dma_applicable(p);
  bool small = false;
  if (pl.cfg == 0 && dma_ok && !p.in_bf16 && p.Cout > 32 && g_plan_mode == 0) {
    // fp32 LDS-DMA kernel: score the three tile shapes by (tile efficiency) x (useful fraction of the padded tile grid) x (fill
    // of the last round of workgroup slots) and take the best.  The round-1 rule (128 x 128 from 2500 tiles on, else 64 x 64)
    // left the mid-sized layers of the 2-D trunk -- 900 .. 2500 big tiles, i.e. 0.9 .. 2.4 rounds of 1024 slots -- on 64 x 64
    // tiles at ~0.8 of the big tile's rate, or on a half-empty second round; 128 x 64 at five per CU sits in between.
    // Efficiencies are the measured per-tile rates on long layers relative to 128 x 128 (141 / 131 / 113 TFLOP/s).
    // Short K (the 1x1 layers up to 512 input channels): one or two K slabs per tile, the tile's time is its epilogue and the
    // latency of its loads, and six 128 x 64 workgroups per CU (56) overlap those better than four 128 x 128: interleaved A/B
    // at 50 views (tools/conv_ab.py, profiles/r02_conv_ab_f32.log) 64->256 0.547 vs 0.640 ms, 128->512 0.378 vs 0.417,
    // 256->1024 0.293 vs 0.310, 512->2048 0.269 vs 0.290; from K = 1024 on the 128 x 128 tile is ahead again.
    const bool short_k = p.K <= 512;
    const int c64 = p.K <= 640 ? 47 : 46;
    const int cand[3] = {54, short_k ? 56 : 49, c64};
    const double eff[3] = {short_k ? 0.93 : 1.00, short_k ? 1.00 : 0.93, short_k ? 0.85 : 0.80};
    double best = -1.0;
    for (int i = 0; i < 3; ++i) {
      TileInfo ti;
      tile_info(cand[i], &ti);
      const double tiles = (double)((p.M + ti.bm - 1) / ti.bm) * ((p.Cout + ti.bn - 1) / ti.bn);
      const double util = (double)p.M * p.Cout / (tiles * ti.bm * ti.bn);
      const double rounds = tiles / (256.0 * ti.wg_per_cu);
      const double fill = rounds >= 1.0 ? rounds / (double)(long long)(rounds + 0.999999) : rounds;
      const double score = eff[i] * util * fill;
      if (score > best) { best = score; pl.cfg = cand[i]; }
    }
    small = pl.cfg == c64;
    // (Round 6, measured and not adopted -- profiles/r06_skinny_convs.md: 128 x 128 / 256 x 128 tiles + split K for the layers of a few hundred rows
    // under a K loop of 13824 .. 27648 (512 -> 512 at 10 x 10 x 4): 0.084 -> 0.140 / 0.243 ms per launch.  These launches are bound by the number of
    // slabs a workgroup runs one after the other, not by L2 -> LDS bytes: 64 x 64 tiles with K split 27 ways stay.)
  }
  if (pl.cfg == 0) {
    const long long nblk = (long long)((p.M + 127) / 128) * ((p.Cout + 127) / 128);
    if (p.Cout <= 32) {
      pl.cfg = dma_ok ? 44 : 4;
      // the head convs of the indoor configs (Cout = 25) on the coarse levels: 4 .. 25 row tiles under a K loop of 54 .. 108 slabs ran as that
      // many workgroups, one slab after the other (89 us for 69 MFLOP at 10 x 10 x 4); K is split as for every other small layer
      // (IVX_CONV_SKINNY=0: the round-5 rule, for A/B)
      small = dma_ok && !p.in_bf16 && !p.in_fp8 && conv_skinny_rule() != 0;
    } else if (nblk >= 2500) {
      pl.cfg = p.Cout > 64 ? (dma_ok ? 54 : 1) : (dma_ok ? (p.K <= 640 ? 49 : 56) : 3);   // 49: the stem
    } else {
      // short-K layers (1x1 convolutions, 3x3 on 64 channels, the stem) are staging/latency-bound: the 64-byte-row
      // variant at six workgroups per CU keeps more loads in flight; long-K layers prefer the 128-byte rows
      pl.cfg = dma_ok ? (p.K <= 640 ? 47 : 46) : 6;
      small = true;
    }
  }
  if (p.in_fp8) {
    // e4m3 storage (2-D trunk of the bf16 + fp8 mode): HBM / latency-bound like the bf16 trunk, same rule: 128 x 128 at four per
    // CU, 128 x 64 for Cout <= 64, 64 x 64 (+ split K) when the tiles would not fill the workgroup slots
    if (g_tile_override >= 91 && g_tile_override <= 94) {
      pl.cfg = g_tile_override;
    } else {
      const long long t128 = (long long)((p.M + 127) / 128) * ((p.Cout + 127) / 128);
      if (p.Cout <= 64) pl.cfg = ((long long)((p.M + 127) / 128) * 2 > 256 * 5) ? 92 : 93;
      else pl.cfg = t128 * 2 > 1024 ? 91 : 93;
      small = pl.cfg == 93;
    }
  }
  if (p.in_bf16 && g_tile_override == 0 && dma_ok && p.Cout > 32 && g_plan_mode == 0) {
    // bf16: interleaved A/B over the layers of the 2-D trunk at 50 views (tools/conv_ab.py, profiles/r02_conv_ab_bf16.log).
    // At 8x the MFMA rate every 1x1 layer and most 3x3 layers are bound by HBM / staging latency, and what matters is how
    // many workgroups a CU holds to overlap one tile's epilogue with another's loads: 128 x 128 with 64-byte LDS rows at four
    // per CU (74) is the best or within 5 % of it on all of them (the 1 / 2 per CU of 82 / 81 / 61 lose 20-40 % on short K);
    // Cout <= 64 takes 128 x 64 (63).  The 8- / 16-wave tiles keep the long-K layers with many tiles (the 3-D necks), and
    // layers that would leave most workgroup slots empty keep the 64 x 64 + split-K path below.
    const long long t128 = (long long)((p.M + 127) / 128) * ((p.Cout + 127) / 128);
    if (p.Cout <= 64) {
      if ((long long)((p.M + 127) / 128) * 2 > 256 * 3) pl.cfg = 63;
    } else if (t128 * 2 > 1024 && !(t128 >= 2500 && p.K >= 1024)) {
      pl.cfg = 74;
    }
    if (pl.cfg == 63 || pl.cfg == 74) small = false;
  }
  if (p.in_bf16 && pl.cfg >= 41 && pl.cfg <= 57) {
    // same tile, bf16 instantiation -- except for the big layers: at 8x the MFMA rate the kernel is bound by the
    // L2 -> LDS staging traffic (ablation: +33 % without the loads), so the 256 x 128 / 256 x 256 workgroups of 8 / 16
    // waves, which move 25 % / 50 % fewer bytes per flop, win there (measured, tools/conv_bench.py --dtype bf16)
    if (g_tile_override == 0 && pl.cfg == 54) pl.cfg = p.Cout > 128 ? 82 : 81;   // 16 / 8 waves: less L2->LDS traffic per flop
    else if (pl.cfg == 54 || pl.cfg == 55) pl.cfg = 74;
    else if (pl.cfg == 48) pl.cfg = 66;
    else if (pl.cfg == 49 || pl.cfg == 56) pl.cfg = 73;
    else if (pl.cfg == 57) pl.cfg = 72;
    else if (pl.cfg <= 53) pl.cfg += 20;   // 41..47, 51..53 -> 61..67, 71..73
  }
  if (p.in_pair && g_tile_override == 0) {   // pair operands are instantiated for a subset of the bf16 tiles
    if (pl.cfg == 64) pl.cfg = 67;
    else if (pl.cfg == 71) pl.cfg = 74;
    else if (pl.cfg == 72) pl.cfg = 73;
  }
  static const int pio_rule = getenv("IVX_PIO_RULE") ? atoi(getenv("IVX_PIO_RULE")) : 1;        // A/B knobs of the round-4 measurements
  static const int pio_splitk = getenv("IVX_PIO_SPLITK") ? atoi(getenv("IVX_PIO_SPLITK")) : 1;
  if (p.in_pair == 2 && p.pio && g_tile_override == 0 && dma_ok && pio_rule) {
    // Chained fp16-pair activations (the 2-D trunk): interleaved A/B of every pair tile on the trunk's layer shapes at KITTI batch 4 and at
    // 50 views (tools/pio_ab.py, profiles/r04_pio_ab_*.md).  128 x 128 with 64-byte LDS rows at four per CU (74) is the best or within
    // 5-10 % of it wherever it fills the chip, also for Cout = 64 at KITTI size (half of its B tile is zero fill, and it still beats
    // 128 x 64 / 256 x 64: 43 vs 50 / 47 us); the 50-view maps of 64 output channels take 256 x 64 (76); layers with few tiles take
    // 64 x 64 with 128-byte rows (66: 63 vs 90 us on 256 -> 256 3x3 at 24 x 80 x 4, 49 vs 58 on 1024 -> 256) + split K.
    const long long t128 = (long long)((p.M + 127) / 128) * ((p.Cout + 127) / 128);
    if (p.Cout <= 64) pl.cfg = p.M >= 400000 ? 76 : (p.M >= 60000 ? 74 : 66);
    else pl.cfg = t128 >= 200 ? 74 : 66;
    small = pl.cfg == 66 && pio_splitk;
    // Round 5: a launch with fewer tiles than the chip has workgroup slots (every /8 .. /32 layer of the trunk at KITTI's batch of 4: 240 - 480
    // tiles) cannot hide its slab latency behind other resident workgroups; it takes the deep-ring form of its tile (166 / 174: FOUR slabs
    // in flight per workgroup, raw barriers; bit-identical results) unless the split-K rule below takes it.  Measured (profiles/
    // r05_trunk_deep_ring.md): trunk span 3.85 -> 3.39 ms at KITTI; tiles <= 512 is the best threshold (1024 / 2048 / all: 3.61 -- above one
    // tile per slot the 64 KB workgroups lose more occupancy than the depth returns).  The first form of the deep ring kept __syncthreads(),
    // whose fence waits for vmcnt(0): no effect in the model and +25 us per launch in isolation.  IVX_PIO_DEEP=0 turns the rule off (A/B).
    static const int pio_deep = getenv("IVX_PIO_DEEP") ? atoi(getenv("IVX_PIO_DEEP")) : 4;
    if (pio_deep == 4) {
      TileInfo td;
      tile_info(pl.cfg, &td);
      const long long tl = (long long)((p.M + td.bm - 1) / td.bm) * ((p.Cout + td.bn - 1) / td.bn);
      const int Sd = (p.K + td.bk - 1) / td.bk;
      static const long long split_tiles = getenv("IVX_PIO_SPLIT_TILES") ? atoll(getenv("IVX_PIO_SPLIT_TILES")) : 320;
      const bool will_split = small && allow_ws && Sd >= 16 && tl <= split_tiles && (tl <= 100 || Sd >= 48);      // (the rule of the split-K block below)
      static const long long deep_tiles = getenv("IVX_PIO_DEEP_TILES") ? atoll(getenv("IVX_PIO_DEEP_TILES")) : 512;
      static const int deep_slabs = getenv("IVX_PIO_DEEP_SLABS") ? atoi(getenv("IVX_PIO_DEEP_SLABS")) : 6;
      if (!will_split && p.kmode == 1 && Sd >= deep_slabs && tl <= deep_tiles && (pl.cfg == 66 || pl.cfg == 74)) {      // (only these two have a deep-ring form)
        pl.cfg += 100;       // 66 -> 166, 74 -> 174
        // ... and the 128 x 128 tile runs on MORE WAVES where the launch leaves SIMDs idle: 16 waves (179: wave tile 32 x 32, 128-byte rows, 128 KB, one
        // workgroup per CU) up to one tile per CU, 8 waves (177: wave tile 64 x 32) up to two.  A slab's LDS-DMA requests, fragment reads, barriers
        // and the epilogue are issued by 4x / 2x the waves (workgroup timelines, profiles/r05_trunk_wg_timeline.md: 512 -> 128 at 48 x 160 x 4
        // 27.1 -> 23.1 -> 20.2 us, 128 -> 128 3x3 46.8 -> 41.1 -> 34.9, 512 -> 2048 + residual at 12 x 40 33.5 -> 26.9 -> 21.9).  Same products in
        // the same order per output: bit-identical.
        static const long long w16_tiles = getenv("IVX_PIO_W16_TILES") ? atoll(getenv("IVX_PIO_W16_TILES")) : 256;
        static const long long w8_tiles = getenv("IVX_PIO_W8_TILES") ? atoll(getenv("IVX_PIO_W8_TILES")) : 512;
        if (pl.cfg == 174) pl.cfg = tl <= w16_tiles ? 179 : (tl <= w8_tiles ? 177 : 174);
        small = false;
      }
    }
    // Round 6: the 256 x 256 tile on 16 waves (183: three-buffer ring, one workgroup per CU; built for the neck's last layer) for the wide layers of the
    // multi-view maps -- half the L2 -> LDS bytes per product of the 128 x 128 tile.  It wins where its tiles fill the 256 CUs in whole rounds: interleaved
    // A/B over the trunk's layers at 50 / 20 ScanNet views, nuScenes and KITTI (tools/pio_ab.py --cfgs 0,74,81,82,183; profiles/r06_tile183_trunk.md):
    // 235 - 1876 tiles at 50 views -5 .. -19 % per launch (1024 -> 256 0.141 -> 0.114 ms, 256 -> 256 3x3 0.239 -> 0.202, 256 -> 1024 + residual 0.225 -> 0.215),
    // 272 tiles (1.06 rounds) +4 .. +10 %, 94 - 136 tiles +30 %; the nearest-upsampled FPN laterals lose 4 %.  Rule: last-round fill of the 256 x 256 grid
    // >= 0.7.  Same products in the same order per output: bit-identical.  IVX_PIO_T183=0 turns it off (A/B).
    static const int pio_t183 = getenv("IVX_PIO_T183") ? atoi(getenv("IVX_PIO_T183")) : 1;
    if (pio_t183 && pl.cfg == 74 && p.in_pair == 2 && p.kmode == 1 && p.res_mode != 2 && p.Cout >= 256 && p.Cout % 256 == 0) {
      const long long t256 = (long long)((p.M + 255) / 256) * (p.Cout / 256);
      const long long rounds = (t256 + 255) / 256;
      if (t256 * 10 >= rounds * 256 * 7) {
        pl.cfg = 183;
        small = false;
      }
    }
  }
  TileInfo t;
  if (!tile_info(pl.cfg, &t) || !dma_ok) return pl;
  const long long Mt = (p.M + t.bm - 1) / t.bm, Nt = (p.Cout + t.bn - 1) / t.bn;
  pl.q_total = (int)((Mt + 7) / 8);
  pl.bm = t.bm;
  if (!allow_ws || g_tile_override != 0) return pl;
  const int S = (p.K + t.bk - 1) / t.bk;
  if (small) {
    const long long tiles = Mt * Nt;
    const long long slots = 256LL * t.wg_per_cu;           // resident workgroups on the chip
    static const long long pio_split_tiles = getenv("IVX_PIO_SPLIT_TILES") ? atoll(getenv("IVX_PIO_SPLIT_TILES")) : 320;
    // Pair IO: at 16-bit MFMA rates a tile is short and a split costs a second launch plus the partial sums' round trip, so it pays only
    // below about one tile per CU, and for the mid-sized layers only when the K loop is long.  Measured through bench.py's trunk span
    // (profiles/r04_pio_split_rule.md): KITTI 4.24 ms with the fp32 rule (2 * tiles <= slots), 4.01 without any split, 3.88 with
    // tiles <= 320; single-view SUN RGB-D 2.42 without, 2.06 with tiles <= 320, 1.86 with tiles <= 100 (its 150 / 160-tile layers have
    // K loops of 16 .. 36 slabs and lose from the split; KITTI's 240-tile layers have 64 .. 144 and gain).
    const bool pio_split = tiles <= pio_split_tiles && (tiles <= 100 || S >= 48);
    if ((p.pio ? pio_split : 2 * tiles <= slots) && S >= 16) {     // under half a round: split K until the slots are filled once
      long long ks = (slots + tiles - 1) / tiles;
      if (ks > S / 8) ks = S / 8;
      if (ks > 32) ks = 32;
      if (ks >= 2) {
        pl.ksplit = (int)ks;
        pl.ws_bytes = ivx_align_up((int64_t)ks * 8 * pl.q_total * t.bm * p.Cout * 4, 256);
      }
    }
    return pl;
  }
  // tail plan
  // (Round 6: not for pair IO.  At 16-bit MFMA rates a tile is short: the second launch + the partial sums' round trip + the reduction cost more than
  // the idle slots of the last round -- the same layers run 5 - 12 % faster as ONE launch (tools/pio_ab.py, cfg 0 vs the same tile forced:
  // nuScenes 512 -> 1024 s2 0.177 -> 0.156 ms, 256 -> 512 s2 0.215 -> 0.195, 256 -> 128 0.257 -> 0.235; profiles/r06_tile183_trunk.md (c)).  IVX_PIO_TAIL=1 restores it.)
  static const int pio_tail = getenv("IVX_PIO_TAIL") ? atoi(getenv("IVX_PIO_TAIL")) : 0;
  if (p.pio && !pio_tail) return pl;
  const long long spx = 32LL * t.wg_per_cu;              // workgroup slots per XCD
  const long long bpx = (long long)pl.q_total * Nt;      // workgroups per XCD
  const long long fr = bpx / spx, rem = bpx - fr * spx;
  if (fr >= 2 && rem > 0 && 2 * rem <= spx && (fr * spx) % Nt == 0) {
    long long ks = spx / rem;
    if (ks > S / 8) ks = S / 8;
    if (ks > 16) ks = 16;
    if (ks >= 2) {
      pl.tail_ks = (int)ks;
      pl.qa = (int)(fr * spx / Nt);
      pl.ws_bytes = ivx_align_up((int64_t)ks * 8 * (pl.q_total - pl.qa) * t.bm * p.Cout * 4, 256);
    }
  }
  return pl;
}

#endif   // IVX_CONV_TU == 0

// pair operands (IVX_BF16_PAIR / IVX_F16_PAIR): the bf16 tiles with the three-product K loop
#if IVX_CONV_TU == 3 || IVX_CONV_TU == 4
template <int PAIR>
static int launch_pair(ConvParams &p, const ConvPlan &pl, hipStream_t st) {
  switch (pl.cfg) {
    case 63: launch_v4<__bf16, 2, 1, 2, 2, 64, 1, PAIR>(p, st); break;
    case 66: launch_v4<__bf16, 1, 1, 2, 2, 64, 1, PAIR>(p, st); break;
    case 67: launch_v4<__bf16, 1, 1, 2, 2, 32, 6, PAIR>(p, st); break;
    case 73: launch_v4<__bf16, 2, 1, 2, 2, 32, 1, PAIR>(p, st); break;
    case 75: launch_v4<__bf16, 2, 1, 2, 2, 32, 5, PAIR>(p, st); break;   // 73 at five workgroups per CU
    case 74: launch_v4<__bf16, 2, 2, 2, 2, 32, 4, PAIR>(p, st); break;
    case 61: launch_v4<__bf16, 2, 2, 2, 2, 64, 1, PAIR>(p, st); break;
    case 81: launch_v4<__bf16, 2, 2, 4, 2, 32, 4, PAIR>(p, st); break;
    case 82: launch_v4<__bf16, 2, 2, 4, 4, 32, 4, PAIR>(p, st); break;
    case 83: launch_v4<__bf16, 2, 2, 4, 2, 64, 2, PAIR>(p, st); break;
    // larger wave tiles: fewer LDS fragment reads per product (the pair loop reads 4 fragments for 3 products; at 16x the fp32 MFMA rate
    // the LDS port, shared by the DMA writes and the fragment reads, is as busy as the matrix pipe)
    case 76: launch_v4<__bf16, 2, 2, 4, 1, 32, 4, PAIR>(p, st); break;   // 256 x 64, wave tile 64 x 64
    // deep LDS rings (NB = 4: four slabs in flight per workgroup, raw barriers): 64 KB, two per CU -- the rule's choice for launches of at
    // most 512 tiles (plan_conv; profiles/r05_trunk_deep_ring.md)
    case 166: launch_v4<__bf16, 1, 1, 2, 2, 64, 2, PAIR, 4>(p, st); break;   // 66 (64 x 64, 128-byte rows)
    case 174: launch_v4<__bf16, 2, 2, 2, 2, 32, 2, PAIR, 4>(p, st); break;   // 74 (128 x 128, 64-byte rows)
    // residual prefetch (RPF; fp16 pairs only): 74 at three / two workgroups per CU, 174
    case 474: if constexpr (PAIR == 2) { launch_v4<__bf16, 2, 2, 2, 2, 32, 3, PAIR, 2, 1>(p, st); break; } else return IVX_ERR_INVALID_ARG;
    case 475: if constexpr (PAIR == 2) { launch_v4<__bf16, 2, 2, 2, 2, 32, 2, PAIR, 2, 1>(p, st); break; } else return IVX_ERR_INVALID_ARG;
    case 574: if constexpr (PAIR == 2) { launch_v4<__bf16, 2, 2, 2, 2, 32, 2, PAIR, 4, 1>(p, st); break; } else return IVX_ERR_INVALID_ARG;
    // the 128 x 128 tile of 174 on 8 / 16 waves (wave tile 64 x 32 / 32 x 32): more waves issue the slab's LDS-DMA requests and barriers in parallel
    case 177: launch_v4<__bf16, 2, 1, 2, 4, 32, 2, PAIR, 4>(p, st); break;
    case 179: launch_v4<__bf16, 1, 1, 4, 4, 64, 1, PAIR, 4>(p, st); break;
    case 77: launch_v4<__bf16, 2, 1, 2, 4, 32, 2, PAIR>(p, st); break;
    case 183: launch_v4<__bf16, 2, 2, 4, 4, 32, 1, PAIR, 3>(p, st); break;   // 82 (256 x 256, 16 waves) with a three-buffer ring: 96 KB
    // (round 5, measured and removed -- profiles/r05_trunk_deep_ring.md (e): rings of six / eight buffers (96 / 128 KB): trunk 3.40 -> 3.9 / 4.05 ms,
    // and 3.43 / 3.47 when only launches of at most one tile per CU take them; a slab-level software pipeline of the four-buffer loop (next slab's
    // fragments read and the next DMA issued in front of the slab's last eight MFMAs): -0.09 us per slab in a long K loop, but 3.42 -> 3.45 ms)
    // (measured and removed, profiles/r05_trunk_deep_ring.md: 74 with three buffers for launches of 513 .. 4096 tiles: trunk 3.44 -> 3.57 ms; on the
    // last neck layer's grouped GEMM 81 / 76 / 74 with three or four buffers: 0.66 / 0.67 / 0.74 / 0.62 ms against 0.57 of tile 82)
    case 85: launch_v4<__bf16, 2, 4, 2, 2, 32, 2, PAIR>(p, st); break;   // 128 x 256, 4 waves, wave tile 64 x 128: no gain over 81 / 82, so the
                                                                         // fragment reads are not what limits the loop (TM = 4 variants: the
                                                                         // compiler keeps the accumulators in scratch, 10x slower; removed)
    default:
      ivx_set_error("ivx_conv_fwd: tile %d has no pair-operand instantiation (61, 63, 66, 67, 73 .. 77, 81 .. 83, 85, 166, 174, 177, 179, 183; fp16 pairs also 474, 475, 574)", pl.cfg);
      return IVX_ERR_INVALID_ARG;
  }
  return IVX_OK;
}

#if IVX_CONV_TU == 3
int ivx_conv_launch_pair_bf16(ConvParams &p, const ConvPlan &pl, hipStream_t st) { return launch_pair<1>(p, pl, st); }
#else
int ivx_conv_launch_pair_f16(ConvParams &p, const ConvPlan &pl, hipStream_t st) { return launch_pair<2>(p, pl, st); }
#endif
#endif

#if IVX_CONV_TU == 5
// Winograd-domain GEMM of a stride-1, pad-1, 1x1x3 convolution along z on pair operands, with the THREE z-taps served from ONE staged
// tile.  The rows of a transformed plane are (tile column, z), z fastest, so the A rows of tap kz are the output rows shifted by
// kz - 1: the generic kernel stages them three times (one K slab per tap: 3 x the L2 -> LDS traffic, 3 x the barriers), and at the 16-bit
// MFMA rate that staging is what the pair GEMMs wait for (5.8 TB/s of LDS-DMA for 3.8 TB/s of HBM traffic, the per-slab
// MFMA work of 0.18 us cannot hide one load).  Here an iteration covers one 16-channel pair group: BM + 2 rows of A (a one-row halo on
// either side) and the three taps' B rows are staged once, and 9 * TM * TN MFMAs (3 taps x hi*hi + hi*lo + lo*hi) run on them; fragments
// of tap kz are read kz rows further down, rows whose z + kz - 1 falls outside the column are zeroed in registers.
// Same LDS row format as conv_igemm_v4_kernel (64-byte rows of 32 stored elements, XOR swizzle by (row >> 2) & 3), same epilogue.
// (T = __bf16: the 2-byte storage element, as in conv_igemm_v4_kernel.)  NBUF staging buffers form a ring with a prefetch distance of
// NBUF - 1 groups: at 16-bit MFMA rates one group of a 128 x 128 tile is 0.5 us of matrix work against 2-3 us of load latency under
// load, so the tiles are large (bytes staged per MFMA fall with the tile edge) and the ring is as deep as the 160 KB of LDS allow.
// TAIL 1: a tile stores BM rows and stages BM + 16 (the last 16 by wave 0).  TAIL 0: a tile stages exactly BM rows and stores the BM - 2
// in the middle, consecutive tiles overlapping by two rows (1.6 % more tiles at BM = 128): 40 KB of LDS for the 128 x 64 tile, i.e. a
// fourth workgroup per CU -- and resident workgroups are what hides the load latency here.
// SW 2 (TAIL 0 only): the z-stride-2 layers.  With an even Z the input row of output row r' and tap kz is 2 r' - 1 + kz whatever the
// column, so a tile stages 2 BM consecutive input rows, stores BM - 1 output rows and reads the fragments of tap kz at rows 2 o + kz;
// only tap 0 can fall outside the column (z' = 0).  One staged row serves 1.5 taps instead of 3.
// ZR 1: a lane whose z - 1 / z + 1 neighbour lies outside the column reads the fragments of that tap from an LDS row that is ALWAYS zero
// instead of zeroing the eight dwords it read (32 v_cndmask per group in the 126 x 64 tile, where the loop has 18 MFMAs): the address is
// chosen once per tile.  The zero row costs no LDS: it is a staged row that no stored output reads -- row BM + 2 of the tail pass, the
// last of the 2 BM rows of the stride-2 form, and for the overlapping stride-1 tiles the last staged row, given up as a source row (a
// tile then stores BM - 3 rows instead of BM - 2) -- whose DMA lanes get an out-of-range offset, i.e. zeros, in every group.
// DI 1 (SW 2 only): the staged rows are de-interleaved -- LDS rows 0 .. BM - 1 hold the even staged rows, BM .. 2 BM - 1 the odd ones -- so
// that the fragment rows of a tap are CONSECUTIVE LDS rows as in the stride-1 form (rows two apart start on two of the four 16-bank groups
// only: 27 % of the LDS cycles of the interleaved form were bank conflicts, profiles/r04_bench_pmc.md).  Only the global row a DMA lane
// fetches changes; tap kz of output row o reads LDS row o (kz 0), BM + o (kz 1), o + 1 (kz 2).
template <typename T, int TM, int TN, int WR, int WC, int WPE, int PAIR, int NBUF, int TAIL = 1, int SW = 1, int ZR = 0, int DI = 0>
__global__ __launch_bounds__(64 * WR * WC, WPE) void conv_wino_halo_kernel(const ConvParams p, const unsigned in_bytes, const unsigned w_bytes) {
  constexpr int BM = WR * TM * 32, BN = WC * TN * 32, NT = 64 * WR * WC;
  constexpr int BK = 32, EPC = 8, NCH = 4, RP = NT / NCH;
  static_assert((SW * BM) % RP == 0 && (3 * BN) % RP == 0, "A rows and the three taps of B rows in whole passes (+ one 16-row tail pass of A)");
  static_assert(SW == 1 || TAIL == 0, "the stride-2 form uses overlapping tiles");
  static_assert(!DI || SW == 2, "de-interleaved staging is the stride-2 form's");
  constexpr int APASS = SW * BM / RP, AROWS = SW * BM + 16 * TAIL;        // staged input rows: plane rows SW * m0 - 1 ...
  constexpr int BMO = TAIL ? BM : (SW == 1 ? BM - 2 - ZR : BM - 1);        // rows a tile stores
  constexpr int ZROW = TAIL ? BM + 2 : SW * BM - 1;                        // ZR: the LDS row of A that stays zero
  constexpr int BPASS = 3 * BN / RP;
  constexpr int BUF = (AROWS + 3 * BN) * BK;                               // elements per buffer
  constexpr int D = NBUF - 1;                                              // prefetch distance in groups
  constexpr int LPG = APASS + BPASS;                                       // DMA instructions per group and wave (wave 0: TAIL more)
  static_assert((D - 1) * (LPG + TAIL) <= 15, "vmcnt immediate");
  __shared__ __attribute__((aligned(16))) T smem[NBUF * BUF];
  static_assert(sizeof(smem) >= (size_t)NT / 64 * 4096, "4 KB of staging LDS per wave for the transposed epilogue");
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wr = wid / WC, wc = wid % WC;
  int mt, nt;
  {
    const int Nt = (p.Cout + BN - 1) / BN;
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const int lt = idx / Nt;
    nt = idx - lt * Nt;
    mt = xcd * p.q_total + lt;
  }
  if (mt * BMO >= p.M) return;
  const int m0 = mt * BMO, n0 = nt * BN;
  const size_t gz = blockIdx.z;
  if (p.cp_dst && blockIdx.x == 0 && blockIdx.z == 0 && tid == 0) *p.cp_dst = *p.cp_src;
#ifdef IVX_CONV_TIMELINE
  const unsigned long long tl0 = __builtin_amdgcn_s_memrealtime();
  unsigned long long tl1 = 0, tl2 = 0;
#endif
  const __amdgpu_buffer_rsrc_t rs_in =
      __builtin_amdgcn_make_buffer_rsrc((void *)((const char *)p.in + gz * (size_t)p.g_in * sizeof(T)), 0, in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_w =
      __builtin_amdgcn_make_buffer_rsrc((void *)((const char *)p.wgt + gz * (size_t)p.g_w * sizeof(T)), 0, w_bytes, 0x00020000);
  const int lr = tid / NCH;
  const int cc = (tid & (NCH - 1)) ^ ((lr >> 2) & 3);            // the k-chunk this lane fetches (its LDS slot is tid % NCH)
  const int wid_u = __builtin_amdgcn_readfirstlane(wid);
  const unsigned OOB = 0x80000000u;
  typedef __attribute__((address_space(3))) void *lds_ptr_t;
  // A: LDS row i holds plane row m0 - 1 + i; the last 16 rows are staged by wave 0 alone (only two of them are used)
  unsigned a_base[APASS + 1];
  a_base[APASS] = OOB;
#pragma unroll
  for (int j = 0; j < APASS + TAIL; ++j) {
    const int li = j < APASS ? lr + RP * j : BM + (lane >> 2);
    const int row = SW * m0 - 1 + (DI ? (li < BM ? 2 * li : 2 * (li - BM) + 1) : li);
    const bool zr = ZR && !TAIL && li == ZROW;
    a_base[j] = (row >= 0 && row < SW * p.M && (j < APASS || (lane >> 2) < 2) && !zr) ? ((unsigned)row * (unsigned)p.Cin + cc * EPC) * (unsigned)sizeof(T) : OOB;
  }
  // B: LDS row t * BN + n holds filter row n0 + n of tap t; chunk-major K: (32 real channels = 64 stored) x tap
  unsigned b_base[BPASS];
#pragma unroll
  for (int j = 0; j < BPASS; ++j) {
    const int r = lr + RP * j, tap = r / BN, n = n0 + (r - tap * BN);
    b_base[j] = n < p.Cout ? ((unsigned)n * (unsigned)p.K + tap * 64 + cc * EPC) * (unsigned)sizeof(T) : OOB;
  }
  const int G = p.Cin / BK;                                    // 16-channel pair groups
  auto load_group = [&](int g, int buf) {
    T *Ab = smem + buf * BUF + wid_u * (64 / NCH) * BK;     // wave-uniform base; the DMA adds lane * 16 B
    T *Bb = Ab + AROWS * BK;
    const unsigned ka = (unsigned)g * BK * (unsigned)sizeof(T);                                             // byte offset of the group inside an A row
    const unsigned kb = ((unsigned)(g >> 1) * 3u * 64u + (unsigned)(g & 1) * 32u) * (unsigned)sizeof(T);    // ... inside a filter row (tap 0)
    // (the offset goes through a named variable: with the conditional written inside the builtin's argument list the host pass of
    // this clang drops the whole kernel without a diagnostic)
#pragma unroll
    for (int j = 0; j < APASS; ++j) {
      const unsigned vo = a_base[j] == OOB ? OOB : a_base[j] + ka;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in, (lds_ptr_t)(Ab + RP * j * BK), 16, vo, 0, 0, 0);
    }
    if (TAIL && wid_u == 0) {
      const unsigned vo = a_base[APASS] == OOB ? OOB : a_base[APASS] + ka;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in, (lds_ptr_t)(smem + buf * BUF + BM * BK), 16, vo, 0, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < BPASS; ++j) {
      const unsigned vo = b_base[j] == OOB ? OOB : b_base[j] + kb;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_ptr_t)(Bb + RP * j * BK), 16, vo, 0, 0, 0);
    }
  };
  // validity of the z - 1 / z + 1 neighbours of this lane's rows (Wo = W = Z rows per tile column)
  const int rl = lane & 31, fh = lane >> 5;
  bool ok0[TM], ok2[TM];
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int z = (m0 + (wr * TM + i) * 32 + rl) % p.Wo;      // output z (Wo = W for stride 1)
    ok0[i] = z >= 1;
    ok2[i] = SW == 2 || z + 1 < p.Wo;                        // stride 2: input z = 2 z' + 1 <= Z - 1 always
  }
  // ZR: element offset of this lane's hi fragment of taps 0 and 2 inside the A area of a buffer (lo: the same ^ 2 chunks)
  int aoff[2][TM];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int ar = DI ? rl + t : SW * rl + 2 * t;
      const bool ok = t == 0 ? ok0[i] : ok2[i];
      aoff[t][i] = ok ? ((DI ? 1 : SW) * (wr * TM + i) * 32 + ar) * BK + ((fh ^ ((ar >> 2) & 3)) * EPC) : ZROW * BK + fh * EPC;
    }
  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

#pragma unroll
  for (int g = 0; g < D; ++g)
    if (g < G) load_group(g, g);
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  int cur = 0, nxt = D % NBUF;          // buffer of group g / of group g + D
  for (int g = 0; g < G; ++g) {
    // group g has landed when at most the D - 1 younger groups are outstanding (this wave's own loads; the barrier publishes all waves')
    if (D > 1 && g + D - 1 < G) {
      if (TAIL && wid_u == 0) __builtin_amdgcn_s_waitcnt(0x0f70 | ((D - 1) * (LPG + 1)));
      else __builtin_amdgcn_s_waitcnt(0x0f70 | ((D - 1) * LPG));
    } else {
      lds_dma_wait_all();
    }
#ifdef IVX_CONV_TIMELINE
    if (g == 0) tl1 = __builtin_amdgcn_s_memrealtime();      // the first group has landed
#endif
    if constexpr (NBUF == 2) {
      __syncthreads();                  // ... and every wave is done reading the buffer of group g - 1, which group g + D now overwrites
    } else {
      // a ring deeper than two needs a RAW barrier: the fence of __syncthreads() waits for vmcnt(0), i.e. for the whole ring (round 5; the
      // three-buffer rings of rounds 3 - 4 were two-deep rings with a third buffer).  This wave's fragment reads of group g - 1 are complete
      // (its MFMAs consumed them); the clobbers keep LDS accesses on their side of the barrier.
      __builtin_amdgcn_s_waitcnt(0xc07f);
      asm volatile("" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
    }
    if (g + D < G) load_group(g + D, nxt);
    const T *Ac = smem + cur * BUF + (DI ? 1 : SW) * wr * TM * 32 * BK;
    const T *Bc = smem + cur * BUF + AROWS * BK + wc * TN * 32 * BK;
#pragma unroll
    for (int kz = 0; kz < 3; ++kz) {
      f32x4 ah[TM], al[TM], bh[TN], bl[TN];
      // A rows of this tap: LDS row SW * o + kz (LDS row 0 is plane row SW * m0 - 1); DI: o + (kz >> 1) in the even (kz 0, 2) / odd (kz 1) half
      const int ar = DI ? rl + (kz >> 1) + (kz & 1) * BM : SW * rl + kz, asw = (ar >> 2) & 3;
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        if (ZR && kz != 1) {
          const T *A0 = smem + cur * BUF;
          ah[i] = *reinterpret_cast<const f32x4 *>(A0 + aoff[kz >> 1][i]);
          al[i] = *reinterpret_cast<const f32x4 *>(A0 + (aoff[kz >> 1][i] ^ (2 * EPC)));
        } else {
          const T *rowp = Ac + ((DI ? 1 : SW) * i * 32 + ar) * BK;
          ah[i] = *reinterpret_cast<const f32x4 *>(rowp + ((fh ^ asw) * EPC));
          al[i] = *reinterpret_cast<const f32x4 *>(rowp + (((2 + fh) ^ asw) * EPC));
        }
      }
      const int bsw = (rl >> 2) & 3;
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const T *rowp = Bc + (kz * BN + j * 32 + rl) * BK;
        bh[j] = *reinterpret_cast<const f32x4 *>(rowp + ((fh ^ bsw) * EPC));
        bl[j] = *reinterpret_cast<const f32x4 *>(rowp + (((2 + fh) ^ bsw) * EPC));
      }
      if (!ZR && kz != 1) {
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          const bool ok = kz == 0 ? ok0[i] : ok2[i];
          ah[i] = ok ? ah[i] : zero4;
          al[i] = ok ? al[i] : zero4;
        }
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          acc[i][j] = pair_mfma<PAIR>(ah[i], bh[j], acc[i][j]);
          acc[i][j] = pair_mfma<PAIR>(ah[i], bl[j], acc[i][j]);
          acc[i][j] = pair_mfma<PAIR>(al[i], bh[j], acc[i][j]);
        }
    }
    cur = cur + 1 == NBUF ? 0 : cur + 1;
    nxt = nxt + 1 == NBUF ? 0 : nxt + 1;
  }
#ifdef IVX_CONV_TIMELINE
  tl2 = __builtin_amdgcn_s_memrealtime();
#endif
  __syncthreads();                      // the staging area of the epilogue overlaps the ring
  if constexpr (TAIL) {
    conv_epilogue_wide<TM, TN>(p, acc, m0, n0, wr, wc, lane, reinterpret_cast<float *>(smem) + wid_u * 1024, gz * (size_t)p.g_out);
  } else {
    // the same LDS-transposed 16-byte stores (conv_epilogue_wide), plain values, rows limited to the BM - 2 this tile owns
    float *stage = reinterpret_cast<float *>(smem) + wid_u * 1024;
    float *outp = p.out + gz * (size_t)p.g_out;
    const int mlim = (m0 + BMO < p.M) ? m0 + BMO : p.M;
    const int col_l = lane & 31, hh = lane >> 5, rrow = lane >> 3, c4 = (lane & 7) * 4;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int nb = n0 + (wc * TN + j) * 32 + c4;
      const bool nok = nb < p.Cout;
#pragma unroll
      for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) stage[((r & 3) + 8 * (r >> 2) + 4 * hh) * 32 + col_l] = acc[i][j][r];
        const int mb = m0 + (wr * TM + i) * 32 + rrow;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 v = *reinterpret_cast<const f32x4 *>(stage + (rrow + 8 * q) * 32 + c4);
          const int m = mb + 8 * q;
          if (m < mlim && nok) *reinterpret_cast<f32x4 *>(outp + (size_t)m * p.Cout + nb) = v;
        }
      }
    }
  }
#ifdef IVX_CONV_TIMELINE
  if (p.tl && tid == 0) {
    unsigned long long *t = p.tl + ((size_t)blockIdx.z * gridDim.x + blockIdx.x) * 8;
    t[0] = tl0; t[1] = tl1; t[2] = tl2; t[3] = __builtin_amdgcn_s_memrealtime();
    t[4] = __builtin_amdgcn_s_getreg(63492); t[5] = __builtin_amdgcn_s_getreg(63508);
  }
#endif
}

template <int TM, int TN, int WR, int WC, int WPE, int PAIR, int NBUF, int TAIL = 1, int SW = 1, int ZR = 0, int DI = 0>
static void launch_halo(ConvParams &p, hipStream_t st) {
  constexpr int BM = (WR * TM * 32) - (TAIL ? 0 : (SW == 1 ? 2 + ZR : 1)), BN = WC * TN * 32;     // rows a tile stores
  const int64_t in_bytes = (int64_t)p.B * p.D * p.H * p.W * p.Cin * 2, w_bytes = (int64_t)p.Cout * p.K * 2;
  const long long Mt = (p.M + BM - 1) / BM, Nt = (p.Cout + BN - 1) / BN;
  p.bm = BM;
  p.q_total = (int)((Mt + 7) / 8); p.q_begin = 0; p.q_count = p.q_total;
  const dim3 grid((unsigned)(8LL * p.q_total * Nt), 1, p.groups > 1 ? p.groups : 1);
  auto kern = conv_wino_halo_kernel<__bf16, TM, TN, WR, WC, WPE, PAIR, NBUF, TAIL, SW, ZR, DI>;
  hipLaunchKernelGGL(kern, grid, dim3(64 * WR * WC), 0, st, p, (unsigned)in_bytes, (unsigned)w_bytes);
}

// z-BLOCKED form of the same GEMM for shallow columns (Z = 3 or 6: the 256- and 128-channel stride-1 layers of the KITTI neck), where the
// halo kernel multiplies zeros: with z the fastest row index a 32-row MFMA block mixes all z, so the taps that fall outside the column
// (kz = 0 at z = 0, kz = 2 at z = Z - 1: 2 of the 3 Z tap-slices) are issued and masked -- 22 % of the matrix work at Z = 3, 11 % at
// Z = 6, on launches whose clock the matrix pipe already pulls down to 1.7 GHz (PMC).  Here a tile is CT = 32 WCOL whole
// columns and the DMA lanes place plane row m0 + c Z + z at LDS row z CT + c (the lane -> global row map of an LDS-DMA is free): every
// 32-row block of the staged tile then holds ONE z, a wave owns three consecutive output z-blocks of 32 columns and 32 channels, and a
// tap whose source block lies outside the column is a wave-uniform skip -- no masks, no zero products, no halo rows (a tile stores every
// row it stages).  V and M keep their layout: only this kernel's addressing differs.  Per output element the products are accumulated
// in the halo kernel's order (group by group, tap 0, 1, 2), so the results are bit-identical.
// Z = output slices per column (3 or 6); SW = z stride (1, or 2 with Z_in = 2 Z: the source block of output block zo and tap kz is
// SW zo + kz - 1, and only zo = 0, kz = 0 falls outside).
template <typename T, int Z, int WCOL, int WN, int WPE, int PAIR, int SW = 1>
__global__ __launch_bounds__(64 * WCOL * (Z / 3) * WN, WPE) void conv_wino_zblk_kernel(const ConvParams p, const unsigned in_bytes, const unsigned w_bytes) {
  static_assert(Z == 3 || Z == 6, "three output z-blocks per wave");
  constexpr int ZI = SW * Z;                     // input slices per column
  constexpr int ZH = Z / 3, NW = WCOL * ZH * WN, NT = 64 * NW;
  constexpr int CT = WCOL * 32, AR = ZI * CT, BN = WN * 32;
  constexpr int NS = 2 * SW + 3;                 // source blocks of a wave's three output blocks
  constexpr int BK = 32, EPC = 8, NCH = 4, RP = NT / NCH;
  constexpr int APASS = (AR + RP - 1) / RP, BPASS = (3 * BN + RP - 1) / RP;
  static_assert(AR % 16 == 0 && (3 * BN) % 16 == 0, "a wave's 16 rows of a pass are inside or outside as a whole");
  constexpr int BUF = (AR + 3 * BN) * BK;
  __shared__ __attribute__((aligned(16))) T smem[2 * BUF];
  static_assert(sizeof(smem) >= (size_t)NW * 4096, "4 KB of staging LDS per wave for the transposed epilogue");
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wid_u = __builtin_amdgcn_readfirstlane(wid);
  const int cg = wid_u / (ZH * WN), zh = (wid_u / WN) % ZH, nb = wid_u % WN;
  int mt, nt;
  {
    const int Nt = (p.Cout + BN - 1) / BN;
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const int lt = idx / Nt;
    nt = idx - lt * Nt;
    mt = xcd * p.q_total + lt;
  }
  const int c0 = mt * CT, n0 = nt * BN;          // first column of the tile
  const int m0 = c0 * Z;                         // its first OUTPUT row (columns x Z)
  if (m0 >= p.M) return;
#ifdef IVX_CONV_TIMELINE
  const unsigned long long tl0 = __builtin_amdgcn_s_memrealtime();
  unsigned long long tl1 = 0, tl2 = 0;
#endif
  const size_t gz = blockIdx.z;
  if (p.cp_dst && blockIdx.x == 0 && blockIdx.z == 0 && tid == 0) *p.cp_dst = *p.cp_src;
  const __amdgpu_buffer_rsrc_t rs_in =
      __builtin_amdgcn_make_buffer_rsrc((void *)((const char *)p.in + gz * (size_t)p.g_in * sizeof(T)), 0, in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_w =
      __builtin_amdgcn_make_buffer_rsrc((void *)((const char *)p.wgt + gz * (size_t)p.g_w * sizeof(T)), 0, w_bytes, 0x00020000);
  const int lr = tid / NCH;
  const int cc = (tid & (NCH - 1)) ^ ((lr >> 2) & 3);            // the k-chunk this lane fetches (its LDS slot is tid % NCH); RP % 16 == 0
  const unsigned OOB = 0x80000000u;
  typedef __attribute__((address_space(3))) void *lds_ptr_t;
  unsigned a_base[APASS], b_base[BPASS];
#pragma unroll
  for (int j = 0; j < APASS; ++j) {
    const int li = lr + RP * j;                                  // LDS row: z * CT + c
    const int z = li / CT, c = li - z * CT;
    const int row = (c0 + c) * ZI + z;                           // input plane row of (column, z)
    a_base[j] = (li < AR && row < SW * p.M) ? ((unsigned)row * (unsigned)p.Cin + cc * EPC) * (unsigned)sizeof(T) : OOB;
  }
#pragma unroll
  for (int j = 0; j < BPASS; ++j) {
    const int r = lr + RP * j, tap = r / BN, n = n0 + (r - tap * BN);
    b_base[j] = (r < 3 * BN && n < p.Cout) ? ((unsigned)n * (unsigned)p.K + tap * 64 + cc * EPC) * (unsigned)sizeof(T) : OOB;
  }
  const int G = p.Cin / BK;
  auto load_group = [&](int g, int buf) {
    T *Ab = smem + buf * BUF + wid_u * (64 / NCH) * BK;     // wave-uniform base; the DMA adds lane * 16 B
    T *Bb = Ab + AR * BK;
    const unsigned ka = (unsigned)g * BK * (unsigned)sizeof(T);
    const unsigned kb = ((unsigned)(g >> 1) * 3u * 64u + (unsigned)(g & 1) * 32u) * (unsigned)sizeof(T);
#pragma unroll
    for (int j = 0; j < APASS; ++j) {
      if (RP * j + wid_u * 16 < AR) {                       // (wave-uniform: the last pass may be a partial one)
        const unsigned vo = a_base[j] == OOB ? OOB : a_base[j] + ka;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in, (lds_ptr_t)(Ab + RP * j * BK), 16, vo, 0, 0, 0);
      }
    }
#pragma unroll
    for (int j = 0; j < BPASS; ++j) {
      if (RP * j + wid_u * 16 < 3 * BN) {
        const unsigned vo = b_base[j] == OOB ? OOB : b_base[j] + kb;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_ptr_t)(Bb + RP * j * BK), 16, vo, 0, 0, 0);
      }
    }
  };
  const int rl = lane & 31, fh = lane >> 5, sw = (rl >> 2) & 3;
  const int z0 = zh * 3;
  f32x16 acc[3];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  load_group(0, 0);
  int cur = 0;
  for (int g = 0; g < G; ++g) {
    lds_dma_wait_all();                 // this wave's loads of group g have landed; the barrier publishes all waves'
#ifdef IVX_CONV_TIMELINE
    if (g == 0) tl1 = __builtin_amdgcn_s_memrealtime();
#endif
    __syncthreads();                    // ... and every wave is done reading the other buffer (group g - 1)
    if (g + 1 < G) load_group(g + 1, cur ^ 1);
    const T *A0 = smem + cur * BUF + (cg * 32 + rl) * BK;
    const T *B0 = smem + cur * BUF + AR * BK + (nb * 32 + rl) * BK;
    f32x4 ah[NS], al[NS];               // source blocks SW z0 - 1 .. SW (z0 + 2) + 1 of this wave's columns
#pragma unroll
    for (int s5 = 0; s5 < NS; ++s5) {
      const int zs = SW * z0 + s5 - 1;
      if (zs >= 0 && zs < ZI) {
        const T *rowp = A0 + zs * CT * BK;
        ah[s5] = *reinterpret_cast<const f32x4 *>(rowp + ((fh ^ sw) * EPC));
        al[s5] = *reinterpret_cast<const f32x4 *>(rowp + (((2 + fh) ^ sw) * EPC));
      }
    }
#pragma unroll
    for (int kz = 0; kz < 3; ++kz) {
      const T *rowp = B0 + kz * BN * BK;
      const f32x4 bh = *reinterpret_cast<const f32x4 *>(rowp + ((fh ^ sw) * EPC));
      const f32x4 bl = *reinterpret_cast<const f32x4 *>(rowp + (((2 + fh) ^ sw) * EPC));
#pragma unroll
      for (int zo = 0; zo < 3; ++zo) {
        const int zs = SW * (z0 + zo) + kz - 1;
        if (zs >= 0 && zs < ZI) {       // (wave-uniform) a tap outside the column is skipped, not multiplied by zeros
          acc[zo] = pair_mfma<PAIR>(ah[SW * zo + kz], bh, acc[zo]);
          acc[zo] = pair_mfma<PAIR>(ah[SW * zo + kz], bl, acc[zo]);
          acc[zo] = pair_mfma<PAIR>(al[SW * zo + kz], bh, acc[zo]);
        }
      }
    }
    cur ^= 1;
  }
#ifdef IVX_CONV_TIMELINE
  tl2 = __builtin_amdgcn_s_memrealtime();
#endif
  __syncthreads();                      // the staging area of the epilogue overlaps the ring
  float *stage = reinterpret_cast<float *>(smem) + wid_u * 1024;
  float *outp = p.out + gz * (size_t)p.g_out;
  const int col_l = lane & 31, hh = lane >> 5, rrow = lane >> 3, c4 = (lane & 7) * 4;
  const int nbc = n0 + nb * 32 + c4;
  const bool nok = nbc < p.Cout;
#pragma unroll
  for (int zo = 0; zo < 3; ++zo) {
#pragma unroll
    for (int r = 0; r < 16; ++r) stage[((r & 3) + 8 * (r >> 2) + 4 * hh) * 32 + col_l] = acc[zo][r];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 v = *reinterpret_cast<const f32x4 *>(stage + (rrow + 8 * q) * 32 + c4);
      const int m = m0 + (cg * 32 + rrow + 8 * q) * Z + z0 + zo;      // plane row of (column, z)
      if (m < p.M && nok) *reinterpret_cast<f32x4 *>(outp + (size_t)m * p.Cout + nbc) = v;
    }
  }
#ifdef IVX_CONV_TIMELINE
  if (p.tl && tid == 0) {
    unsigned long long *t = p.tl + ((size_t)blockIdx.z * gridDim.x + blockIdx.x) * 8;
    t[0] = tl0; t[1] = tl1; t[2] = tl2; t[3] = __builtin_amdgcn_s_memrealtime();
    t[4] = __builtin_amdgcn_s_getreg(63492); t[5] = __builtin_amdgcn_s_getreg(63508);
  }
#endif
}

template <int Z, int WCOL, int WN, int WPE, int SW = 1>
static void launch_zblk(ConvParams &p, hipStream_t st) {
  constexpr int AR = Z * WCOL * 32, BN = WN * 32;          // output rows of a tile
  const int64_t in_bytes = (int64_t)p.B * p.D * p.H * p.W * p.Cin * 2, w_bytes = (int64_t)p.Cout * p.K * 2;
  const long long Mt = (p.M + AR - 1) / AR, Nt = (p.Cout + BN - 1) / BN;
  p.bm = AR;
  p.q_total = (int)((Mt + 7) / 8); p.q_begin = 0; p.q_count = p.q_total;
  const dim3 grid((unsigned)(8LL * p.q_total * Nt), 1, p.groups > 1 ? p.groups : 1);
  auto kern = conv_wino_zblk_kernel<__bf16, Z, WCOL, WN, WPE, 2, SW>;
  hipLaunchKernelGGL(kern, grid, dim3(64 * WCOL * (Z / 3) * WN), 0, st, p, (unsigned)in_bytes, (unsigned)w_bytes);
}

static int zblk_wrong_z(int want, int got) {
  ivx_set_error("ivx_conv_launch_halo: this z-blocked config is built for columns of %d slices, the layer has %d", want, got);
  return IVX_ERR_INVALID_ARG;
}

int ivx_conv_launch_halo(ConvParams &p, int cfg, hipStream_t st) {
  if (p.in_pair != 2) {
    ivx_set_error("ivx_conv_launch_halo: fp16 pair operands only");
    return IVX_ERR_INVALID_ARG;
  }
  switch (cfg) {
    case 1: launch_halo<2, 2, 4, 1, 2, 2, 2>(p, st); break;   // 256 x 64, 4 waves, 2 buffers: 58 KB, two workgroups per CU
    case 2: launch_halo<2, 2, 2, 2, 2, 2, 2>(p, st); break;   // 128 x 128, 4 waves, 2 buffers: 66 KB
    case 3: launch_halo<2, 2, 4, 2, 1, 2, 2>(p, st); break;   // 256 x 128, 8 waves, 2 buffers: 82 KB
    case 4: launch_halo<2, 1, 2, 2, 3, 2, 2>(p, st); break;   // 128 x 64, 4 waves, 2 buffers: 42 KB, three per CU
    case 5: launch_halo<2, 2, 4, 4, 1, 2, 2>(p, st); break;   // 256 x 256, 16 waves, 2 buffers: 130 KB
    case 6: launch_halo<2, 2, 4, 2, 1, 2, 3>(p, st); break;   // 256 x 128, 8 waves, 3 buffers: 123 KB
    case 7: launch_halo<2, 1, 2, 2, 2, 2, 3>(p, st); break;   // 128 x 64, 4 waves, 3 buffers: 63 KB, two per CU
    case 8: launch_halo<2, 2, 4, 1, 1, 2, 3>(p, st); break;   // 256 x 64, 4 waves, 3 buffers: 87 KB
    case 9: launch_halo<2, 2, 2, 2, 1, 2, 3>(p, st); break;   // 128 x 128, 4 waves, 3 buffers: 99 KB
    case 10: launch_halo<2, 1, 2, 2, 4, 2, 2, 0>(p, st); break;  // 126 (of 128) x 64, overlapping tiles, 40 KB: four per CU
    case 11: launch_halo<2, 2, 4, 1, 2, 2, 2, 0>(p, st); break;  // 254 x 64, overlapping tiles, 56 KB: two per CU
    case 12: launch_halo<2, 2, 4, 4, 1, 2, 2, 0>(p, st); break;  // 254 x 256, 16 waves, overlapping tiles, 128 KB
    case 13: launch_halo<2, 2, 4, 2, 1, 2, 2, 0>(p, st); break;  // 254 x 128, 8 waves, overlapping tiles, 80 KB: two per CU
    case 14: launch_halo<2, 2, 2, 2, 2, 2, 2, 0>(p, st); break;  // 126 x 128, 4 waves, overlapping tiles, 64 KB: two per CU
    // z stride 2 (p.W == 2 * p.Wo): 2 BM staged rows
    case 21: launch_halo<2, 1, 2, 2, 2, 2, 2, 0, 2>(p, st); break;  // 127 x 64, 4 waves: 56 KB, two per CU
    case 22: launch_halo<2, 2, 2, 2, 2, 2, 2, 0, 2>(p, st); break;  // 127 x 128, 4 waves: 80 KB, two per CU
    case 23: launch_halo<2, 2, 4, 2, 1, 2, 2, 0, 2>(p, st); break;  // 255 x 128, 8 waves: 112 KB, one per CU
    case 24: launch_halo<1, 2, 4, 1, 2, 2, 2, 0, 2>(p, st); break;  // 127 x 64, 4 waves in M (wave tile 32 x 64): 56 KB
    // ZR: masked taps read a zero row (10 / 13 / 14 / 11 / 22 / 21 with the fragment address selected instead of the fragment zeroed)
    case 30: launch_halo<2, 1, 2, 2, 4, 2, 2, 0, 1, 1>(p, st); break;  // 125 x 64
    // (round 5, with the raw barrier that makes a third buffer a real prefetch distance of two groups -- measured and removed, profiles/
    // r05_fused_gemm_output.md: 125 x 64 with three / four buffers 0.535 / 0.600 ms against 0.493 (64 channels); 253 x 128 with three 0.806
    // against 0.659 (128) and 1.314 against 1.171 (256): on these launches resident workgroups DO hide the latency better than depth)
    case 31: launch_halo<2, 2, 4, 1, 2, 2, 2, 0, 1, 1>(p, st); break;  // 253 x 64
    case 33: launch_halo<2, 2, 4, 2, 1, 2, 2, 0, 1, 1>(p, st); break;  // 253 x 128, 8 waves
    case 34: launch_halo<2, 2, 2, 2, 2, 2, 2, 0, 1, 1>(p, st); break;  // 125 x 128
    case 41: launch_halo<2, 1, 2, 2, 2, 2, 2, 0, 2, 1>(p, st); break;  // stride 2: 127 x 64
    case 42: launch_halo<2, 2, 2, 2, 2, 2, 2, 0, 2, 1>(p, st); break;  // stride 2: 127 x 128
    case 43: launch_halo<2, 2, 2, 2, 2, 2, 2, 0, 2, 1, 1>(p, st); break;  // 42 with de-interleaved staging
    case 44: launch_halo<2, 1, 2, 2, 2, 2, 2, 0, 2, 1, 1>(p, st); break;  // 41 with de-interleaved staging
    case 45: launch_halo<2, 2, 2, 2, 2, 2, 2, 0, 2, 0, 1>(p, st); break;  // 22 with de-interleaved staging
    // round 5 (A/B): the tile of 42 on twice the waves (wave tile 64 x 32 instead of 64 x 64) -- 0.520 / 0.807 ms against 0.519 / 0.802: unlike the
    // trunk's one-wave-per-SIMD launches these kernels are not bound by what a single wave issues
    case 46: launch_halo<2, 1, 2, 4, 2, 2, 2, 0, 2, 1>(p, st); break;  // stride 2: 127 x 128, 8 waves, 80 KB, two per CU
    // z-blocked tiles (conv_wino_zblk_kernel; Wo = 3 or 6 only): 50 / 51 = 8 / 16 waves at Z = 3, 60 at Z = 6.  Measured and not kept as
    // instantiations (profiles/r04_zblk_ab.md): 384 x 64 (8 waves) 1.04, 192 x 64 (4 waves) 1.07, 192 x 256 (16 waves) 1.05, 96 x 128 1.22 ms at
    // Z = 3 (50: 0.99); at Z = 6: 192 x 64 0.675, 192 x 256 1.08, 384 x 128 0.67 (60: 0.63-0.65); z stride 2: 6 -> 3 slices 64 columns 0.874, 12 -> 6 0.578
    // (the halo form 42: 0.785 / 0.507)
    case 50: if (p.Wo != 3) return zblk_wrong_z(3, p.Wo); launch_zblk<3, 2, 4, 4>(p, st); return IVX_OK;   // 192 rows (64 columns) x 128: 72 KB, two per CU
    case 51: if (p.Wo != 3) return zblk_wrong_z(3, p.Wo); launch_zblk<3, 4, 4, 4>(p, st); return IVX_OK;   // 384 rows x 128, 16 waves: 96 KB, one per CU
    case 60: if (p.Wo != 6) return zblk_wrong_z(6, p.Wo); launch_zblk<6, 1, 4, 4>(p, st); return IVX_OK;   // 192 rows (32 columns) x 128
    // z stride 2 (p.W == 2 p.Wo), 6 -> 3 slices: 0.818 ms against 0.785 of the halo form 42 -- an A/B point, not the rule
    case 71: if (p.Wo != 3 || p.W != 6) return zblk_wrong_z(3, p.Wo); launch_zblk<3, 1, 4, 2, 2>(p, st); return IVX_OK;   // 32 columns x 128 (two waves per SIMD)
    default:
      ivx_set_error("ivx_conv_launch_halo: unknown config %d", cfg);
      return IVX_ERR_INVALID_ARG;
  }
  return IVX_OK;
}

// ------------------------------------------------------------------------------------------------
// Round 5: the Winograd-domain GEMM of an F(4x4, 3x3) layer with its OUTPUT TRANSFORM AND EPILOGUE FUSED -- M never reaches HBM.
// (Round-4 verdict item 1; reference layers: mmdet3d/models/necks/imvoxelnet.py:94-123, the ResModule convolutions of the stack necks.)
// A workgroup owns BM - 2 = 62 rows (tile column, z) x 64 output channels for ALL 36 frequency points xi.  It walks xi = (i, j), j fastest;
// for each xi the z-halo K loop of conv_wino_halo_kernel (one staged tile serves the three z-taps; hi*hi + hi*lo + lo*hi per 16-channel pair
// group) leaves M[xi] of its tile in 16 accumulator registers per lane, which are folded into the output domain right away:
//     P[e]      += At[e][j] * M[i][j]          (e = 0 .. 3; after the six j of a row i:)
//     out[a][e] += At[a][i] * P[e]             (a = 0 .. 3)        ->  out = At M A  in 16 + 4 + 1 accumulator tiles instead of 36
// -- an all-xi workgroup that kept M itself would hold 36 tiles (576 registers per lane at this tile size).  The groups of all xi form ONE
// stream of 36 * Cin / 16 iterations through an NBUF-deep LDS ring (16 KB per group: 64 rows of V[xi] and the three taps' 64 filter rows of
// U[xi]), so the pipeline fills once per tile, not once per xi.  The epilogue transposes each of the 16 output tiles through LDS and applies
// the scales of the pair operands, BN scale / shift, residual and ReLU exactly as wino_output_kernel does, writing 16-byte pieces of the
// channels-last output, and leaves the workgroup's max |out| for the next layer's operand scale.
// Cost: the filters of all 36 xi are re-read (from L2) by every row tile: 36 x 12 KB x Cin / 16 per 62 rows; V is read once per 64 output
// channels.  One workgroup per CU (64 KB of LDS, > 256 registers per lane: the accumulators spill into the AGPR half of the file).
__constant__ float kWinoAt4[4][6] = {{1.f, 1.f, 1.f, 1.f, 1.f, 0.f}, {0.f, 1.f, -1.f, 2.f, -2.f, 0.f}, {0.f, 1.f, 1.f, 4.f, 4.f, 0.f}, {0.f, 1.f, -1.f, 8.f, -8.f, 1.f}};

__device__ __forceinline__ float fold4_vscale(const float amax) {      // == winograd.hip wino_pair_vscale: 2^k with 256 amax 2^k in [2^14, 2^15)
  if (!(amax < 3.0e38f)) return 0.00390625f;
  if (!(amax > 0.f)) return 1.0f;
  int e;
  (void)frexpf(amax, &e);
  int k = 15 - 8 - e;
  k = k < -120 ? -120 : (k > 120 ? 120 : k);
  return ldexpf(1.0f, k);
}

template <int PAIR, int NA, int NB>
__global__ __launch_bounds__(256, 1) void conv_wino_fold4_kernel(const ConvParams p, const IvxWinoFold f, const unsigned in_bytes, const unsigned w_bytes) {
  typedef __bf16 T;
  constexpr int BM = 64, BN = 64, NT = 256, BK = 32, EPC = 8, NCH = 4, RP = NT / NCH;
  constexpr int BMO = BM - 2;                       // rows a tile stores (consecutive tiles overlap by two staged rows)
  // Two rings with their own depths: the rows of V[xi] come from HBM (every byte is read once: 2 - 4 us under load), the filter rows of
  // U[xi] from L2 (all 36 banks are re-read by every row tile), and with one wave per SIMD nothing but the ring hides either.  An A slot
  // holds the 64 staged rows and one row that is ALWAYS ZERO (row 64: the DMA never writes it): a lane whose z - 1 / z + 1 neighbour lies
  // outside the column reads that tap's fragments from it -- an address chosen once -- instead of masking 8 registers per group.
  constexpr int ASL = (BM + 1) * BK, BSL = 3 * BN * BK;   // elements per slot of the A ring (4160 B) / of the B ring (12 KB)
  constexpr int NXI = 36;
  static_assert(RP == BM && RP == BN, "one pass of the workgroup's lanes covers the A rows / one tap of the B rows");
  static_assert(NA >= NB && NB >= 3 && (NB - 2) * 4 <= 15, "vmcnt immediates; the A ring runs at least as far ahead as the B ring");
  __shared__ __attribute__((aligned(16))) T smem[NA * ASL + NB * BSL];
  __shared__ float wmax[4];
  static_assert(sizeof(smem) >= 4 * 4096, "4 KB of staging LDS per wave for the transposed epilogue");
  T *const As = smem, *const Bs = smem + NA * ASL;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wr = wid >> 1, wc = wid & 1;
  int mt, nt;
  {
    const int Nt = (p.Cout + BN - 1) / BN;
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const int lt = idx / Nt;
    nt = idx - lt * Nt;
    mt = xcd * p.q_total + lt;
  }
  if (mt * BMO >= p.M) {                            // padding of the last XCD's range
    if (f.pmax && tid == 0) f.pmax[blockIdx.x] = 0.f;
    return;
  }
  const int m0 = mt * BMO, n0 = nt * BN;
  const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc((void *)p.in, 0, in_bytes, 0x00020000);     // all 36 planes of V
  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void *)p.wgt, 0, w_bytes, 0x00020000);      // all 36 filter banks
  const int lr = tid / NCH;
  const int cc = (tid & (NCH - 1)) ^ ((lr >> 2) & 3);
  const int wid_u = __builtin_amdgcn_readfirstlane(wid);
  const unsigned OOB = 0x80000000u;
  typedef __attribute__((address_space(3))) void *lds_ptr_t;
  typedef __attribute__((address_space(3))) T *lds_t;
  // A: LDS row lr of a slot holds plane row m0 - 1 + lr;  B: LDS row t * BN + lr of a slot holds filter row n0 + lr of tap t (chunk-major K).
  // The per-lane part of a request's address never changes (voffset); the group / xi part is a wave-uniform running byte offset (soffset).
  const int arow = m0 - 1 + lr;
  const unsigned a_vo = (arow >= 0 && arow < p.M) ? ((unsigned)arow * (unsigned)p.Cin + cc * EPC) * 2u : OOB;
  unsigned b_vo[3];
#pragma unroll
  for (int t = 0; t < 3; ++t) b_vo[t] = (n0 + lr < p.Cout) ? ((unsigned)(n0 + lr) * (unsigned)p.K + t * 64 + cc * EPC) * 2u : OOB;
  const int G = p.Cin / BK;                                   // 16-channel pair groups per xi
  const unsigned xi_in = (unsigned)p.g_in * 2u, xi_w = (unsigned)p.g_w * 2u;      // byte strides between the xi planes / filter banks
  const int NIT = NXI * G;
  // zero rows of the A slots
  for (int t = tid; t < NA * 16; t += NT) reinterpret_cast<float *>(smem)[(t >> 4) * (ASL / 2) + BM * (BK / 2) + (t & 15)] = 0.f;
  // request cursors (all wave-uniform): byte offset of the next group inside V / U, its group index inside the xi, its ring slot
  unsigned sa = 0, sb = 0;
  int ga = 0, gb = 0, slot_a = 0, slot_b = 0;
  const unsigned a_lds0 = (unsigned)(uintptr_t)(lds_t)As + (unsigned)(wid_u * (64 / NCH) * BK * 2);
  const unsigned b_lds0 = (unsigned)(uintptr_t)(lds_t)Bs + (unsigned)(wid_u * (64 / NCH) * BK * 2);
  auto issue_a = [&]() {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in, (lds_ptr_t)(uintptr_t)(a_lds0 + (unsigned)slot_a * (ASL * 2)), 16, a_vo, sa, 0, 0);
    sa += BK * 2;
    if (++ga == G) { ga = 0; sa += xi_in - (unsigned)G * (BK * 2); }
    slot_a = slot_a + 1 == NA ? 0 : slot_a + 1;
  };
  auto issue_b = [&]() {
    const unsigned l = b_lds0 + (unsigned)slot_b * (BSL * 2);
#pragma unroll
    for (int t = 0; t < 3; ++t) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_ptr_t)(uintptr_t)(l + t * RP * BK * 2), 16, b_vo[t], sb, 0, 0);
    // next group of the chunk-major K order: the second half of a 32-channel chunk lies 64 bytes on, the next chunk 3 * 128 bytes after the chunk's start
    sb += (gb & 1) ? (3 * 128 - 64) : 64;
    if (++gb == G) { gb = 0; sb += xi_w - (unsigned)(G >> 1) * (3 * 128); }
    slot_b = slot_b + 1 == NB ? 0 : slot_b + 1;
  };
  auto skip_a = [&]() {                                       // advance the A cursor by one group without a request
    sa += BK * 2;
    if (++ga == G) { ga = 0; sa += xi_in - (unsigned)G * (BK * 2); }
    slot_a = slot_a + 1 == NA ? 0 : slot_a + 1;
  };
  const int rl = lane & 31, fh = lane >> 5;
  const int Z = p.Wo;
  const int zl = (m0 + wr * 32 + rl) % Z;                     // z of this lane's output row
  const bool ok0 = zl >= 1, ok2 = zl + 1 < Z;
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  // The 16 output-domain tiles live in the AGPR half of the register file (every access below goes through v_accvgpr_read / _write with
  // an "a" operand), the four row accumulators P, the product tile and everything the group loop touches in VGPRs: left to itself the
  // compiler put `out` into VGPRs and P / the product tile into AGPRs, i.e. the moves on the per-xi path instead of the per-row path.
  float outa[4][4][16];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int r = 0; r < 16; ++r) asm volatile("v_accvgpr_write_b32 %0, 0" : "=a"(outa[a][e][r]));
  f32x16 acc, P[4];
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
  for (int e = 0; e < 4; ++e) P[e] = acc;
  // Prologue: requests in the order the steady state leaves behind (one A and three B requests per group, group after group; the A ring
  // runs NA - NB groups further ahead, so its extra requests are the OLDEST): A of groups NB - 1 .. NA - 2 first, then (A, B) of 0 .. NB - 2.
  {
    for (int k = 0; k < NB - 1; ++k) skip_a();
    for (int k = NB - 1; k < NA - 1; ++k) issue_a();           // (NIT >= 36 > NA)
    const unsigned e_sa = sa; const int e_ga = ga, e_slot = slot_a;
    sa = 0; ga = 0; slot_a = 0;
    for (int k = 0; k < NB - 1; ++k) { issue_a(); issue_b(); }
    sa = e_sa; ga = e_ga; slot_a = e_slot;                     // next A request: group NA - 1
  }
  // group 0 (and the zero rows) has landed when only the (A, B) pairs of groups 1 .. NB - 2 are younger.  No __syncthreads() from here to
  // the epilogue: its fence waits for vmcnt(0), i.e. for every request of the rings (the first form of this kernel ran at the full memory
  // latency per group whatever the ring depth -- as every deeper ring tried in this library did).  RAW barriers, bracketed by compiler-level
  // memory clobbers; a wave's fragment reads of a group are complete when it arrives (its MFMAs consumed them).
  __builtin_amdgcn_s_waitcnt(0x0070 | ((NB - 2) * 4));        // vmcnt(..) lgkmcnt(0): the zero-row stores too
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  // Fragment reads as inline asm: the compiler's wait-count pass puts s_waitcnt vmcnt(0) in front of any LDS read it can see while an
  // LDS-DMA request is in flight (it cannot tell the ring slots apart).  Byte offsets of this lane's fragments inside a slot (hi chunk fh,
  // lo chunk 2 + fh, XOR-swizzled by the row as the DMA lanes wrote them); a masked tap points at the slot's zero row.
  const unsigned as_lds = (unsigned)(uintptr_t)(lds_t)As, bs_lds = (unsigned)(uintptr_t)(lds_t)Bs;
  unsigned aoh[3], aol[3];
#pragma unroll
  for (int kz = 0; kz < 3; ++kz) {
    const int ar = rl + kz, asw = (ar >> 2) & 3;              // A rows of tap kz: slot row o + kz (row 0 is plane row m0 - 1)
    const bool ok = kz == 0 ? ok0 : (kz == 2 ? ok2 : true);
    aoh[kz] = ok ? (unsigned)(((wr * 32 + ar) * BK + ((fh ^ asw) * EPC)) * 2) : (unsigned)(BM * BK * 2);
    aol[kz] = ok ? (unsigned)(((wr * 32 + ar) * BK + (((2 + fh) ^ asw) * EPC)) * 2) : (unsigned)(BM * BK * 2);
  }
  const int bsw = (rl >> 2) & 3;
  const unsigned boh = (unsigned)(((wc * 32 + rl) * BK + ((fh ^ bsw) * EPC)) * 2), bol = (unsigned)(((wc * 32 + rl) * BK + (((2 + fh) ^ bsw) * EPC)) * 2);
#define FOLD_DS_READ(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
  unsigned abase = as_lds, bbase = bs_lds;                    // slot bases of the current group
  int it = 0;
  // Three nested loops (i, j, group) over ONE request stream: the innermost touches only the product tile and the fragments.
  for (int i = 0; i < 6; ++i) {
    for (int j = 0; j < 6; ++j) {
      for (int gq = 0; gq < G; ++gq, ++it) {
        f32x4 ah[3], al[3], bh[3], bl[3];
        {
          const unsigned b0 = bbase + boh, b1 = bbase + bol;
          FOLD_DS_READ(ah[0], abase + aoh[0], 0); FOLD_DS_READ(al[0], abase + aol[0], 0);
          FOLD_DS_READ(bh[0], b0, 0);             FOLD_DS_READ(bl[0], b1, 0);
          FOLD_DS_READ(ah[1], abase + aoh[1], 0); FOLD_DS_READ(al[1], abase + aol[1], 0);
          FOLD_DS_READ(bh[1], b0, BN * BK * 2);   FOLD_DS_READ(bl[1], b1, BN * BK * 2);
          FOLD_DS_READ(ah[2], abase + aoh[2], 0); FOLD_DS_READ(al[2], abase + aol[2], 0);
          FOLD_DS_READ(bh[2], b0, 2 * BN * BK * 2); FOLD_DS_READ(bl[2], b1, 2 * BN * BK * 2);
        }
        // LDS returns in order: tap 0's four fragments are there when eight reads are outstanding, and so on (the operands tie the MFMAs
        // to the wait).  The four requests that refill the slots group it - 1 left (A: group it + NA - 1, B: group it + NB - 1) are issued
        // BETWEEN the MFMAs: an LDS-DMA instruction holds the issuing wave for 60 - 180 cycles, which only its own matrix work can cover.
        asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(ah[0]), "+v"(al[0]), "+v"(bh[0]), "+v"(bl[0]));
        acc = pair_mfma<PAIR>(ah[0], bh[0], acc);
        acc = pair_mfma<PAIR>(ah[0], bl[0], acc);
        __builtin_amdgcn_sched_barrier(0);
        if (it + NA - 1 < NIT) issue_a();
        __builtin_amdgcn_sched_barrier(0);
        acc = pair_mfma<PAIR>(al[0], bh[0], acc);
        asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(ah[1]), "+v"(al[1]), "+v"(bh[1]), "+v"(bl[1]));
        acc = pair_mfma<PAIR>(ah[1], bh[1], acc);
        acc = pair_mfma<PAIR>(ah[1], bl[1], acc);
        acc = pair_mfma<PAIR>(al[1], bh[1], acc);
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ah[2]), "+v"(al[2]), "+v"(bh[2]), "+v"(bl[2]));
        acc = pair_mfma<PAIR>(ah[2], bh[2], acc);
        __builtin_amdgcn_sched_barrier(0);
        if (it + NB - 1 < NIT) issue_b();
        __builtin_amdgcn_sched_barrier(0);
        acc = pair_mfma<PAIR>(ah[2], bl[2], acc);
        acc = pair_mfma<PAIR>(al[2], bh[2], acc);
        // group it + 1 must have landed: younger are the requests of groups it + 2 .. it + NB - 1 (B ring) with their A partners while
        // the A ring still issues
        if (it + 1 < NIT) {
          if (it + NA - 1 < NIT) __builtin_amdgcn_s_waitcnt(0x0f70 | ((NB - 2) * 4));         // steady state
          else if (it + NB - 1 < NIT) __builtin_amdgcn_s_waitcnt(0x0f70 | ((NB - 2) * 3));    // the A ring has run dry: only B requests are younger
          else lds_dma_wait_all();
          asm volatile("" ::: "memory");
          __builtin_amdgcn_s_barrier();   // group it + 1 is visible to every wave; every wave is done with the slots of group `it`
          asm volatile("" ::: "memory");
        }
        abase = abase + ASL * 2 == as_lds + NA * ASL * 2 ? as_lds : abase + ASL * 2;
        bbase = bbase + BSL * 2 == bs_lds + NB * BSL * 2 ? bs_lds : bbase + BSL * 2;
      }
      // M[xi] of this tile is complete (xi = 6 i + j): fold it into the row accumulators
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float c = kWinoAt4[e][j];
        if (c != 0.f) P[e] = c * acc + P[e];
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    }
    // the six j of row i are in: out[a][e] += At[a][i] * P[e]
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const float c = kWinoAt4[a][i];
      if (c != 0.f) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            float t;
            asm("v_accvgpr_read_b32 %0, %1" : "=v"(t) : "a"(outa[a][e][r]));
            t = __builtin_fmaf(c, P[e][r], t);
            asm("v_accvgpr_write_b32 %0, %1" : "=a"(outa[a][e][r]) : "v"(t));
          }
      }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int r = 0; r < 16; ++r) P[e][r] = 0.f;
  }
  __syncthreads();                      // the staging area of the epilogue overlaps the rings
  // ---- epilogue: out[a][e] is the (4 tx + a, 4 ty + e) output of the tile rows; as wino_output_kernel: v = act(M * sc + sf [+ res])
  float *stage = reinterpret_cast<float *>(smem) + wid_u * 1024;
  const float mscale = 1.0f / (fold4_vscale(__uint_as_float(f.hdr_v[0])) * f.uscale[0]);
  const int col_l = lane & 31, hh = lane >> 5, rrow = lane >> 3, c4 = (lane & 7) * 4;
  const int nb = n0 + wc * 32 + c4;
  const bool nok = nb < p.Cout;
  f32x4 sc = {mscale, mscale, mscale, mscale}, sf = {0.f, 0.f, 0.f, 0.f};
  if (nok && f.scale) sc = mscale * *reinterpret_cast<const f32x4 *>(f.scale + nb);
  if (nok && f.shift) sf = *reinterpret_cast<const f32x4 *>(f.shift + nb);
  size_t base[4];
  int xlim[4], ylim[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int o = wr * 32 + rrow + 8 * q, m = m0 + o;
    xlim[q] = 0; ylim[q] = 0; base[q] = 0;
    if (o < BMO && m < p.M && nok) {
      const int z = m % Z, col = m / Z;
      const int ty = col % f.TY, t2 = col / f.TY;
      const int tx = t2 % f.TX, b = t2 / f.TX;
      base[q] = ((((size_t)b * f.Xo + 4 * tx) * f.Yo + 4 * ty) * Z + z) * (size_t)f.Co + nb;
      xlim[q] = f.Xo - 4 * tx;
      ylim[q] = f.Yo - 4 * ty;
    }
  }
  const size_t ystep = (size_t)Z * f.Co, xstep = (size_t)f.Yo * ystep;
  float omax = 0.f;
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float t;
        asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(t) : "a"(outa[a][e][r]));
        stage[((r & 3) + 8 * (r >> 2) + 4 * hh) * 32 + col_l] = t;
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        f32x4 v = *reinterpret_cast<const f32x4 *>(stage + (rrow + 8 * q) * 32 + c4);
        if (a < xlim[q] && e < ylim[q]) {
          const size_t o = base[q] + a * xstep + e * ystep;
          v = v * sc + sf;
          f32x4 rr = zero4;
          if (f.res_mode) rr = *reinterpret_cast<const f32x4 *>(f.res + o);
          if (f.res_mode && !f.res_after_act) v += rr;
          if (f.relu) {
            v[0] = v[0] > 0.f ? v[0] : 0.f; v[1] = v[1] > 0.f ? v[1] : 0.f;
            v[2] = v[2] > 0.f ? v[2] : 0.f; v[3] = v[3] > 0.f ? v[3] : 0.f;
          }
          if (f.res_mode && f.res_after_act) v += rr;
          v *= f.post_scale;
          *reinterpret_cast<f32x4 *>(f.out + o) = v;
          omax = fmaxf(omax, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
        }
      }
    }
  if (f.pmax) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) omax = fmaxf(omax, __shfl_xor(omax, o));
    if (lane == 0) wmax[wid_u] = omax;
    __syncthreads();
    if (tid == 0) f.pmax[blockIdx.x] = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
  }
}

// workgroups of the fused launch for M rows and Cout channels (= entries of its partial-maximum array)
int ivx_conv_fold4_blocks(long long M, int Cout) {
  const long long Mt = (M + 61) / 62, Nt = (Cout + 63) / 64;
  return (int)(8 * ((Mt + 7) / 8) * Nt);
}

int ivx_conv_launch_fold4(ConvParams &p, const IvxWinoFold &f, int n2, hipStream_t st) {
  if (p.in_pair != 2 || n2 != 36 || p.kmode != 1 || p.Cin % 64 != 0 || p.K != 3 * p.Cin) {
    ivx_set_error("ivx_conv_launch_fold4: fp16 pair operands, F(4x4,3x3), chunk-major filters, Cin %% 32 == 0 and a 3-tap z kernel only");
    return IVX_ERR_INVALID_ARG;
  }
  const long long in_bytes = (long long)n2 * p.g_in * 2, w_bytes = (long long)n2 * p.g_w * 2;
  if (in_bytes >= (1LL << 31) || w_bytes >= (1LL << 31)) {
    ivx_set_error("ivx_conv_launch_fold4: the %d transformed planes must stay below 2 GiB together", n2);
    return IVX_ERR_UNSUPPORTED;
  }
  const long long Mt = (p.M + 61) / 62, Nt = (p.Cout + 63) / 64;
  p.bm = 62;
  p.q_total = (int)((Mt + 7) / 8); p.q_begin = 0; p.q_count = p.q_total;
  const dim3 grid((unsigned)(8LL * p.q_total * Nt));
  hipLaunchKernelGGL((conv_wino_fold4_kernel<2, 16, 5>), grid, dim3(256), 0, st, p, f, (unsigned)in_bytes, (unsigned)w_bytes);      // A ring 16 slots, B ring 5: 124 KB (8 + 4 and 20 + 5 slots measured the same)
  return IVX_OK;
}
#endif

#if IVX_CONV_TU == 1
int ivx_conv_launch_f32(ConvParams &p, const ConvPlan &pl, hipStream_t st) {
  switch (pl.cfg) {
    case 41: launch_v4<float, 2, 2, 2, 2, 32>(p, st); break;   // LDS-DMA: 128 x 128, 128-byte LDS rows
    case 43: launch_v4<float, 2, 1, 2, 2, 32>(p, st); break;   //          128 x 64
    case 44: launch_v4<float, 1, 1, 4, 1, 32>(p, st); break;   //          128 x 32
    case 46: launch_v4<float, 1, 1, 2, 2, 32>(p, st); break;   //          64 x 64
    case 51: launch_v4<float, 2, 2, 2, 2, 16>(p, st); break;   //          128 x 128, 64-byte rows: 32 KB LDS, 3 workgroups/CU
    case 53: launch_v4<float, 2, 1, 2, 2, 16>(p, st); break;
    case 52: launch_v4<float, 2, 2, 4, 1, 16>(p, st); break;   //          256 x 64, 64-byte rows: 40 KB LDS
    case 54: launch_v4<float, 2, 2, 2, 2, 16, 4>(p, st); break;  // 51 squeezed to 128 registers: 4 workgroups/CU
    case 55: launch_v4<float, 2, 2, 2, 2, 16, 5>(p, st); break;  //    ... to 102 registers: 5 workgroups/CU (all 160 KB of LDS)
    case 56: launch_v4<float, 2, 1, 2, 2, 16, 6>(p, st); break;  // 53 at 6 workgroups/CU
    case 57: launch_v4<float, 2, 2, 4, 1, 16, 4>(p, st); break;  // 52 at 4 workgroups/CU
    case 58: launch_v4<float, 2, 2, 4, 2, 16, 4>(p, st); break;  // 8 waves, 256 x 128: two workgroups per CU, half the tiles per FLOP
    case 59: launch_v4<float, 2, 2, 2, 4, 16, 4>(p, st); break;  // 8 waves, 128 x 256
    case 47: launch_v4<float, 1, 1, 2, 2, 16, 6>(p, st); break;  // 64 x 64, 64-byte rows: 16 KB LDS, 6 workgroups/CU
    case 48: launch_v4<float, 1, 2, 2, 2, 16, 5>(p, st); break;  // 64 x 128
    case 49: launch_v4<float, 2, 1, 2, 2, 16, 5>(p, st); break;  // 128 x 64 at 5 workgroups/CU
    default:
      ivx_set_error("ivx_conv_fwd: tile %d is not an fp32 LDS-DMA tile", pl.cfg);
      return IVX_ERR_INVALID_ARG;
  }
  return IVX_OK;
}
#endif

#if IVX_CONV_TU == 2
int ivx_conv_launch_lowp(ConvParams &p, const ConvPlan &pl, hipStream_t st) {
  switch (pl.cfg) {
    case 74: launch_v4<__bf16, 2, 2, 2, 2, 32, 4>(p, st); break; // 71 at 4 workgroups/CU
    case 91: launch_v4<fp8_t, 2, 2, 2, 2, 64, 4>(p, st); break;  // e4m3 operands, v_mfma_f32_32x32x16_fp8_fp8: 128 x 128, 64-byte rows
    case 92: launch_v4<fp8_t, 2, 1, 2, 2, 64, 5>(p, st); break;  //   128 x 64
    case 93: launch_v4<fp8_t, 1, 1, 2, 2, 64, 6>(p, st); break;  //   64 x 64
    case 94: launch_v4<fp8_t, 2, 2, 2, 2, 128, 2>(p, st); break; //   128 x 128, 128-byte rows
    case 81: launch_v4<__bf16, 2, 2, 4, 2, 32, 4>(p, st); break; // 8 waves, 256 x 128: 25 % less L2->LDS traffic per flop
    case 82: launch_v4<__bf16, 2, 2, 4, 4, 32, 4>(p, st); break; // 16 waves, 256 x 256: half the traffic per flop
    case 83: launch_v4<__bf16, 2, 2, 4, 2, 64, 2>(p, st); break; // 8 waves, 256 x 128, 128-byte rows
    case 61: launch_v4<__bf16, 2, 2, 2, 2, 64>(p, st); break;  // bf16 operands, v_mfma_f32_32x32x16_bf16
    case 63: launch_v4<__bf16, 2, 1, 2, 2, 64>(p, st); break;
    case 64: launch_v4<__bf16, 1, 1, 4, 1, 64>(p, st); break;
    case 66: launch_v4<__bf16, 1, 1, 2, 2, 64>(p, st); break;
    case 67: launch_v4<__bf16, 1, 1, 2, 2, 32, 6>(p, st); break;
    case 71: launch_v4<__bf16, 2, 2, 2, 2, 32>(p, st); break;
    case 73: launch_v4<__bf16, 2, 1, 2, 2, 32>(p, st); break;
    case 72: launch_v4<__bf16, 2, 2, 4, 1, 32>(p, st); break;
    default:
      ivx_set_error("ivx_conv_fwd: tile %d is not a bf16 / e4m3 LDS-DMA tile", pl.cfg);
      return IVX_ERR_INVALID_ARG;
  }
  return IVX_OK;
}
#endif

#if IVX_CONV_TU == 0
static int launch_one(ConvParams &p, const ConvPlan &pl, hipStream_t st) {
  if (p.in_pair) return p.in_pair == 2 ? ivx_conv_launch_pair_f16(p, pl, st) : ivx_conv_launch_pair_bf16(p, pl, st);
  if (pl.cfg >= 40) return (p.in_bf16 || p.in_fp8) ? ivx_conv_launch_lowp(p, pl, st) : ivx_conv_launch_f32(p, pl, st);
  switch (pl.cfg) {
    case 1: launch_cfg<2, 2, 2, 2>(p, st); break;  // 128 x 128, 2 workgroups/CU
    case 2: launch_cfg<2, 2, 4, 1>(p, st); break;  // 256 x 64, 1 workgroup/CU
    case 3: launch_cfg<2, 1, 2, 2>(p, st); break;  // 128 x 64
    case 4: launch_cfg<1, 1, 4, 1>(p, st); break;  // 128 x 32
    case 5: launch_cfg<1, 2, 4, 1>(p, st); break;  // 128 x 64 (wave 32 x 64)
    case 6: launch_cfg<1, 1, 2, 2>(p, st); break;  // 64 x 64
    case 7: launch_cfg<1, 2, 2, 2>(p, st); break;  // 64 x 128
    default:
      ivx_set_error("ivx_conv_fwd: unknown tile override %d", pl.cfg);
      return IVX_ERR_INVALID_ARG;
  }
  return IVX_OK;
}

static void launch_reduce(const ConvParams &p, hipStream_t st) {
  const size_t total = (size_t)8 * p.q_count * p.bm * p.Cout;
  if (p.pio && (p.Cout & 3) == 0 && total < (1ull << 31)) {
    size_t b4 = (total / 4 + 255) / 256;
    if (b4 > 4096) b4 = 4096;
    hipLaunchKernelGGL(conv_splitk_reduce_pio_kernel, dim3((unsigned)b4), dim3(256), 0, st, p);
    return;
  }
  size_t blocks = (total + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(conv_splitk_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, st, p);
}

static int run_conv(ConvParams &p, const ConvPlan &pl, void *workspace, hipStream_t st) {
  TileInfo ti;
  if (p.in_fp8 && (!dma_applicable(p) || pl.cfg < 91 || pl.cfg > 94)) {
    ivx_set_error("ivx_conv_fwd: fp8 input is only implemented by the LDS-DMA kernel (tensor < 2 GiB, kernel extents <= 8)");
    return IVX_ERR_UNSUPPORTED;
  }
  const bool deep_cfg = pl.cfg == 166 || pl.cfg == 174 || pl.cfg == 474 || pl.cfg == 475 || pl.cfg == 574 || pl.cfg == 177 || pl.cfg == 179 || pl.cfg == 183;      // deep-ring pair tiles
  if (p.in_bf16 && (!dma_applicable(p) || !(tile_info(pl.cfg, &ti) && ((pl.cfg >= 61 && pl.cfg < 91) || (deep_cfg && p.in_pair))))) {
    ivx_set_error("ivx_conv_fwd: bf16 input is only implemented by the LDS-DMA kernel (tensor < 2 GiB, kernel extents <= 8)");
    return IVX_ERR_UNSUPPORTED;
  }
  if (p.kmode == 1 && (!dma_applicable(p) || pl.cfg < 40)) {
    ivx_set_error("ivx_conv_fwd: wgt_layout 1 is only implemented by the LDS-DMA kernel (tensor < 2 GiB, kernel extents <= 8)");
    return IVX_ERR_UNSUPPORTED;
  }
  p.stagger_ticks = 0;
  if (p.pio && p.res_mode == 1 && pl.ksplit <= 1 && pl.tail_ks <= 1 && tile_info(pl.cfg, &ti)) {
    // (experiment, default off: IVX_PIO_STAGGER_US = delay in us, IVX_PIO_STAGGER_SHIFT = which bit of the XCD-local workgroup index picks the late half)
    static const int st_us = getenv("IVX_PIO_STAGGER_US") ? atoi(getenv("IVX_PIO_STAGGER_US")) : 0;
    static const int st_sh = getenv("IVX_PIO_STAGGER_SHIFT") ? atoi(getenv("IVX_PIO_STAGGER_SHIFT")) : 0;
    const long long tiles = (long long)((p.M + ti.bm - 1) / ti.bm) * ((p.Cout + ti.bn - 1) / ti.bn);
    if (st_us > 0 && tiles >= 2LL * 256 * ti.wg_per_cu) {
      p.stagger_ticks = st_us * 100;
      p.stagger_shift = st_sh;
      p.stagger_first = 32 * ti.wg_per_cu;
    }
  }
  int rc;
  if (pl.tail_ks > 1) {
    // full rounds ...
    p.q_total = pl.q_total; p.q_begin = 0; p.q_count = pl.qa; p.ksplit = 1; p.partial = nullptr;
    if ((rc = launch_one(p, pl, st)) != IVX_OK) return rc;
    // ... then the remainder, K split so that it fills every workgroup slot once
    p.q_begin = pl.qa; p.q_count = pl.q_total - pl.qa; p.ksplit = pl.tail_ks; p.partial = (float *)workspace;
    if ((rc = launch_one(p, pl, st)) != IVX_OK) return rc;
    launch_reduce(p, st);
    return IVX_OK;
  }
  if (pl.ksplit > 1) {
    p.ksplit = pl.ksplit;
    p.partial = (float *)workspace;
  }
  if ((rc = launch_one(p, pl, st)) != IVX_OK) return rc;
  if (pl.ksplit > 1) launch_reduce(p, st);
  return IVX_OK;
}

// Batch slicing: the LDS-DMA kernel addresses its operands with 31-bit buffer offsets.  When the input of a launch
// reaches 2 GiB (e.g. the first KITTI neck layers from batch 13 up) the batch is cut into the largest slices that fit
// and the slices run back to back on the stream, sharing the workspace; B is the outermost dimension, so a slice is a
// pointer offset.  Returns the number of samples per slice (B = no slicing needed or not possible).
static int conv_batch_slice(const ConvParams &p) {
  if (dma_applicable(p) || p.B == 1 || p.KD > 8 || p.KH > 8 || p.KW > 8) return p.B;
  const int el = p.in_fp8 ? 1 : (p.in_bf16 ? 2 : 4);
  const int64_t in_s = (int64_t)p.D * p.H * p.W * p.Cin * el;
  if ((int64_t)p.Cout * p.K * el >= (1LL << 31) || in_s >= (1LL << 31)) return p.B;
  const int64_t nb = ((1LL << 31) - 1) / in_s;
  return (int)(nb < p.B ? nb : p.B);
}

static ConvParams conv_slice_params(const ConvParams &p, int b0, int nb) {
  ConvParams q = p;
  const size_t in_el = p.in_fp8 ? 1 : (p.in_bf16 ? 2 : 4), out_el = p.out_fp8 ? 1 : (p.out_bf16 ? 2 : 4);
  const size_t in_s = (size_t)p.D * p.H * p.W * p.Cin;
  const size_t out_s = p.out_mode == 1 ? (size_t)8 * p.D * p.H * p.W * p.Cr : (size_t)p.Do * p.Ho * p.Wo * p.Cout;
  q.B = nb;
  q.M = nb * p.Do * p.Ho * p.Wo;
  q.in = (const float *)((const char *)p.in + (size_t)b0 * in_s * in_el);
  q.out = (float *)((char *)p.out + (size_t)b0 * out_s * out_el);
  if (p.res) {
    const size_t res_s = p.res_mode == 2 ? (size_t)p.rH * p.rW * p.Cout : out_s;
    q.res = (const float *)((const char *)p.res + (size_t)b0 * res_s * out_el);
  }
  return q;
}

// Plan (and with `launch`, run) the whole problem: one call, or one call per batch slice.  *need = workspace bytes.
static int conv_dispatch(const ConvParams &p, bool allow_ws, bool launch, void *workspace, int64_t workspace_bytes, hipStream_t st,
                         int64_t *need, const char *who) {
  const int nb = conv_batch_slice(p);
  *need = 0;
  for (int b0 = 0; b0 < p.B; b0 += nb) {
    ConvParams q = nb == p.B ? p : conv_slice_params(p, b0, (p.B - b0) < nb ? (p.B - b0) : nb);
    const ConvPlan pl = plan_conv(q, allow_ws);
    if (pl.ws_bytes > *need) *need = pl.ws_bytes;
    if (!launch) continue;
    if (pl.ws_bytes > 0 && (!workspace || workspace_bytes < pl.ws_bytes)) {
      ivx_set_error("%s: workspace too small (%lld < %lld); size it with ivx_conv_workspace_bytes", who, (long long)workspace_bytes,
                    (long long)pl.ws_bytes);
      return IVX_ERR_WORKSPACE;
    }
    const int rc = run_conv(q, pl, workspace, st);
    if (rc != IVX_OK) return rc;
  }
  return IVX_OK;
}

extern "C" int ivx_conv_fwd(const ivx_conv_desc *d, const void *in, const void *wgt, const float *scale,
                            const float *shift, const void *res, void *out, ivx_stream_t stream) {
  ConvParams p;
  int rc = fill_params(d, in, wgt, scale, shift, res, out, &p);
  if (rc != IVX_OK) return rc;
  int64_t need;
  rc = conv_dispatch(p, false, true, nullptr, 0, (hipStream_t)stream, &need, "ivx_conv_fwd");
  if (rc != IVX_OK) return rc;
  IVX_CHECK_LAUNCH("ivx_conv_fwd");
  return IVX_OK;
}

// Internal (winograd.hip): `groups` independent convolutions of the same shape in ONE launch of the LDS-DMA kernel
// (grid.z = group), identity epilogue.  `d` describes one group; operands of group g start g*stride elements further.
// the default rule takes the z-blocked tile (conv_wino_zblk_kernel, config 50) for 3-slice columns with more than 64 output channels
static bool zblk_rule(const ConvParams &p) { return p.Wo == 3 && p.W == 3 && p.Cout > 64; }

// Fraction of the 3 Z tap-slices a stride-1, pad-1 Winograd-domain GEMM issues under the default rule (for FLOP accounting: the z-blocked
// tile skips the taps outside the column, the halo tile multiplies zeros there).  d: the LAYER descriptor (W = slices, KW = 3).
extern "C" float ivx_conv_winograd_issued_fraction(const ivx_conv_desc *d) {
  if (!d || d->wino_operands != IVX_F16_PAIR || d->KW != 3 || d->sw != 1 || d->pw != 1 || d->W != 3 || d->Cin % 64 || d->Cout <= 64 || d->wgt_layout != 1 ||
      g_halo_mode >= 0 || g_tile_override != 0)
    return 1.0f;
  return 7.0f / 9.0f;
}

int ivx_conv_grouped_launch(const ivx_conv_desc *d, int groups, const float *in, long long g_in, const float *wgt, long long g_w,
                            float *out, long long g_out, hipStream_t st, const unsigned *cp_src, unsigned *cp_dst) {
  ConvParams p;
  int rc = fill_params(d, in, wgt, nullptr, nullptr, nullptr, out, &p);
  if (rc != IVX_OK) return rc;
  IVX_REQUIRE(groups >= 1 && groups <= 65535, "ivx_conv_grouped_launch: bad group count");
  IVX_REQUIRE((d->in_dtype == IVX_F32 || d->in_dtype == IVX_BF16_PAIR || d->in_dtype == IVX_F16_PAIR) && d->out_dtype == IVX_F32 &&
                  d->out_mode == 0 && d->res_mode == 0,
              "ivx_conv_grouped_launch: fp32 or pair operands, plain fp32 output only");
  if (!dma_applicable(p)) {
    ivx_set_error("ivx_conv_grouped_launch: one group must stay below 2 GiB");
    return IVX_ERR_UNSUPPORTED;
  }
  // (after the argument checks: a rejected call has no side effect -- round-5 advisor)
  if (cp_dst && !p.in_pair) {                    // (only the pair-operand kernels carry the word; not reached by the library's own callers)
    if (hipMemcpyAsync(cp_dst, cp_src, 4, hipMemcpyDeviceToDevice, st) != hipSuccess) {
      ivx_set_error("ivx_conv_grouped_launch: hipMemcpyAsync failed");
      return IVX_ERR_HIP;
    }
  } else {
    p.cp_src = cp_src; p.cp_dst = cp_dst;
  }
  p.groups = groups; p.g_in = g_in; p.g_w = g_w; p.g_out = g_out;
  // z-halo kernel (TU 5): 1x1x3 along z, stride 1, pad 1, chunk-major fp16 pairs -- the ResModule layers of the stack necks
  if (p.in_pair == 2 && p.kmode == 1 && d->KD == 1 && d->KH == 1 && d->KW == 3 && d->sw == 2 && d->pw == 1 && p.W == 2 * p.Wo && p.Cin % 64 == 0 &&
      ((g_halo_mode >= 21 && g_halo_mode < 30) || (g_halo_mode >= 40 && g_halo_mode < 50) || g_halo_mode >= 70 || (g_halo_mode < 0 && g_tile_override == 0))) {
    // (round 4, tools/halo_ab.py, profiles/r04_halo_ab.md: the zero-row form 42 vs 22: 0.494 / 0.790 vs 0.492 / 0.811 ms; de-interleaved staging
    // 43 / 45: equal -- neither the masks nor the bank conflicts are what this kernel waits for)
    return ivx_conv_launch_halo(p, g_halo_mode >= 21 ? g_halo_mode : 42, st);
  }
  if (p.in_pair == 2 && p.kmode == 1 && d->KD == 1 && d->KH == 1 && d->KW == 3 && d->sw == 1 && d->pw == 1 && p.Cin % 64 == 0 &&
      ((g_halo_mode > 0 && g_halo_mode < 21) || (g_halo_mode >= 30 && g_halo_mode < 40) || (g_halo_mode >= 50 && g_halo_mode < 70) ||
       (g_halo_mode < 0 && g_tile_override == 0))) {
    // measured (tools/pair_ab.py --halo N, profiles/r03b_pair_ab_halo.log; generic kernel 0.73 / 0.90 / 1.41 ms for Cout 64 / 128 / 256):
    // with the 16-row tail pass: 128 x 64 at three per CU 0.57 / 0.84 / 1.44, 256 x 64 0.60-0.64 / 0.83 / 1.33, 256 x 128 0.99 / 0.84 / 1.37,
    // 256 x 256 (16 waves) 1.40-1.44 / 1.19 / 1.24-1.27; every three-buffer ring is slower than its two-buffer form (resident workgroups
    // hide the load latency better than depth does).  Overlapping tiles without the tail pass: 126 x 64 at FOUR per CU 0.49 / 0.74 / 1.35,
    // 254 x 64 0.51 / 0.72 / 1.23, 254 x 128 (8 waves, two per CU) 0.65 / 0.64 / 1.15, 254 x 256 1.19 / 1.10 / 1.20
    // round 4: the zero-row forms (30 / 33: the fragment ADDRESS of a masked tap selected once per tile instead of 32 v_cndmask per group):
    // 125 x 64 0.487 / 0.756 / 1.410 (= 10), 253 x 128 0.597 / 0.644 / 1.144 (13: 0.607 / 0.653 / 1.167); bit-identical results
    // round 4, z-blocked tiles for 3-slice columns (50: a tap outside the column is skipped instead of multiplied by zeros: 7 of 9
    // tap-slices): 256 -> 256 at 216 x 248 x 3 0.99 vs 1.135-1.15 ms, bit-identical; at 6 slices (60) 0.626-0.651 vs 0.637-0.650: a wash
    // round 6: small volumes (the indoor necks: 200 .. 1600 rows per Winograd position).  The 253 x 128 tile on 8 waves leaves most CUs idle there --
    // 72 workgroups for 256 -> 256 at 20 x 20 x 8 -- and the 125 x 64 tile at four per CU wins for every Cout: 0.073 -> 0.050 ms (12 launches per ScanNet v1
    // step), 512 -> 512 0.208 -> 0.163, 128 -> 128 at 40 x 40 x 16 0.037 -> 0.033 (tools/neck_halo_ab.py, profiles/r06_halo_small_volumes.md); bit-identical
    // results.  Rule: fewer workgroups of the 253 x 128 tile than the chip has slots for them (512).  IVX_HALO_SMALL=0 turns it off (A/B).
    static const int halo_small = getenv("IVX_HALO_SMALL") ? atoi(getenv("IVX_HALO_SMALL")) : 1;
    const bool few = halo_small && p.Cout > 64 && !zblk_rule(p) && (long long)((p.M + 252) / 253) * ((p.Cout + 127) / 128) * groups < 512;
    const int cfg = g_halo_mode > 0 ? g_halo_mode : ((p.Cout <= 64 || few) ? 30 : (zblk_rule(p) ? 50 : 33));
    return ivx_conv_launch_halo(p, cfg, st);
  }
  ConvPlan pl = {g_tile_override, 1, 1, 0, 0, 0, 0};
  if (pl.cfg == 0 && p.in_pair) {
    // pair operands: three bf16-rate products per staged operand pair; 8- / 16-wave workgroups stage the fewest bytes per product
    // (tools/gemm_ab.py --pair, profiles/r03_gemm_ab_pair.log)
    // measured on the KITTI neck (tools/pair_ab.py, profiles/r03_pair_ab.log): Cout 64: 256 x 64 at four per CU 0.69 ms (128 x 64: 0.74-0.84);
    // Cout 128: 256 x 128 8 waves 0.60 / 0.88 (128 x 128: 0.63 / 0.89); Cout 256: 256 x 256 16 waves 0.87 / 1.39 (8 waves: 0.91 / 1.41)
    // round 5: the 256 x 256 tile with a THREE-buffer ring behind raw barriers (183; one workgroup of 16 waves per CU either way): the last KITTI
    // neck layer (256 -> 256, z 3 -> 1: a dense GEMM with K = 3 Cin, 48 slabs of 1.9 us for 0.64 us of matrix work) 0.633 -> 0.579 ms (four
    // buffers 0.582; tools/l9_ab.py); same products in the same order
    pl.cfg = p.Cout <= 64 ? 76 : (p.Cout <= 128 ? 81 : (p.in_pair == 2 && p.kmode == 1 ? 183 : 82));
    // few tiles (the 2-D 3x3 layers of the trunk at KITTI size, the indoor necks): the 8- / 16-wave tiles would leave most CUs idle
    const long long nblk = (long long)groups * ((p.M + 127) / 128) * ((p.Cout + 127) / 128);
    if (nblk < 2500) pl.cfg = 67;
  }
  if (pl.cfg == 0) {
    const long long nblk = (long long)groups * ((p.M + 127) / 128) * ((p.Cout + 127) / 128);
    if (p.Cout <= 32) pl.cfg = 44;
    // measured on the Winograd-domain GEMMs of the KITTI neck (tools/conv_bench.py --winograd --wcfgs ...): K is only
    // KW*Cin here (192 .. 768), so a workgroup's prologue and epilogue weigh more than in the direct form and one more
    // resident workgroup per CU pays: 128 x 64 at six per CU for Cout <= 64 (1.51 vs 1.85 ms at five), 128 x 128 at five
    // per CU up to K = 512 (2.50 vs 2.90 ms at four); at K = 768 four and five tie
    // re-measured with the LDS-transposed epilogue, configs interleaved inside one process (tools/gemm_ab.py,
    // profiles/r02_gemm_ab.log; run-to-run spread exceeds most tile effects otherwise): the shorter epilogue removes the
    // advantage of a fifth / sixth resident workgroup -- 128 x 128 at four per CU wins for every Cout >= 128 layer
    // (K = 192: 1.05 vs 1.12 ms at five; K = 384: 1.79 vs 1.86; K = 768: 3.37 vs 3.40), 128 x 64 at five for Cout <= 64
    // (1.08 vs 1.09 at six, 1.14 for 256 x 64)
    // round 3, same tool with the 8-wave 256 x 128 tile added: it wins only the 128 -> 128 layers (K = 384: 1.838 vs
    // 1.903 ms, spreads disjoint) and ties at K = 192 / 768 (profiles/r03_gemm_ab.log)
    else if (nblk >= 2500) pl.cfg = p.Cout > 64 ? (p.Cout == 128 && p.K == 384 ? 58 : 54) : 49;
    else pl.cfg = p.K <= 640 ? 47 : 46;
  }
  TileInfo t;
  if (!tile_info(pl.cfg, &t) || (pl.cfg >= 61) != (p.in_pair != 0)) {
    ivx_set_error("ivx_conv_grouped_launch: tile %d does not match the operand type", pl.cfg);
    return IVX_ERR_INVALID_ARG;
  }
  return launch_one(p, pl, st);
}

// Internal (winograd.hip): the n2 = 36 Winograd-domain convolutions of an F(4x4,3x3) layer with the output transform and the layer's epilogue
// fused into the launch (conv_wino_fold4_kernel).  `d` describes one xi convolution (as for ivx_conv_grouped_launch).
int ivx_conv_grouped_fold4(const ivx_conv_desc *d, int groups, const float *in, long long g_in, const float *wgt, long long g_w, const IvxWinoFold *f,
                           hipStream_t st) {
  ConvParams p;
  float dummy;
  int rc = fill_params(d, in, wgt, nullptr, nullptr, nullptr, &dummy, &p);
  if (rc != IVX_OK) return rc;
  IVX_REQUIRE(d->in_dtype == IVX_F16_PAIR && d->KD == 1 && d->KH == 1 && d->KW == 3 && d->sw == 1 && d->pw == 1 && d->wgt_layout == 1 && f && f->out,
              "ivx_conv_grouped_fold4: fp16 pair operands, 1x1x3 along z with stride 1 and padding 1, chunk-major filters");
  p.groups = groups; p.g_in = g_in; p.g_w = g_w; p.g_out = 0;
  return ivx_conv_launch_fold4(p, *f, groups, st);
}

extern "C" int64_t ivx_conv_workspace_bytes(const ivx_conv_desc *d) {
  ConvParams p;
  float dummy;
  if (fill_params(d, &dummy, &dummy, nullptr, nullptr, d && d->res_mode ? &dummy : nullptr, &dummy, &p) != IVX_OK) return -1;
  int64_t need;
  if (conv_dispatch(p, true, false, nullptr, 0, nullptr, &need, "ivx_conv_workspace_bytes") != IVX_OK) return -1;
  return need;
}

extern "C" int ivx_conv_fwd_ws(const ivx_conv_desc *d, const void *in, const void *wgt, const float *scale,
                               const float *shift, const void *res, void *out, void *workspace, int64_t workspace_bytes,
                               ivx_stream_t stream) {
  ConvParams p;
  int rc = fill_params(d, in, wgt, scale, shift, res, out, &p);
  if (rc != IVX_OK) return rc;
  int64_t need;
  rc = conv_dispatch(p, true, true, workspace, workspace_bytes, (hipStream_t)stream, &need, "ivx_conv_fwd_ws");
  if (rc != IVX_OK) return rc;
  IVX_CHECK_LAUNCH("ivx_conv_fwd_ws");
  return IVX_OK;
}

extern "C" int64_t ivx_conv_pio_workspace_bytes(const ivx_conv_desc *d, const ivx_pair_io *io) {
  ConvParams p;
  float dummy;
  if (!io) { ivx_set_error("ivx_conv_pio_workspace_bytes: null pair io"); return -1; }
  if (fill_params(d, &dummy, &dummy, nullptr, nullptr, d && d->res_mode ? &dummy : nullptr, &dummy, &p, io) != IVX_OK) return -1;
  int64_t need;
  if (conv_dispatch(p, true, false, nullptr, 0, nullptr, &need, "ivx_conv_pio_workspace_bytes") != IVX_OK) return -1;
  return need;
}

extern "C" int ivx_conv_fwd_pio(const ivx_conv_desc *d, const ivx_pair_io *io, const void *in, const void *wgt, const float *scale, const float *shift,
                                const void *res, void *out, void *workspace, int64_t workspace_bytes, ivx_stream_t stream) {
  ConvParams p;
  IVX_REQUIRE(io, "ivx_conv_fwd_pio: null pair io");
  int rc = fill_params(d, in, wgt, scale, shift, res, out, &p, io);
  if (rc != IVX_OK) return rc;
  int64_t need;
  rc = conv_dispatch(p, true, true, workspace, workspace_bytes, (hipStream_t)stream, &need, "ivx_conv_fwd_pio");
  if (rc != IVX_OK) return rc;
  IVX_CHECK_LAUNCH("ivx_conv_fwd_pio");
  return IVX_OK;
}

extern "C" int ivx_conv_fwd_pio_naive(const ivx_conv_desc *d, const ivx_pair_io *io, const void *in, const void *wgt, const float *scale, const float *shift,
                                      const void *res, void *out, ivx_stream_t stream) {
  ConvParams p;
  IVX_REQUIRE(io, "ivx_conv_fwd_pio_naive: null pair io");
  int rc = fill_params(d, in, wgt, scale, shift, res, out, &p, io);
  if (rc != IVX_OK) return rc;
  const size_t total = (size_t)p.M * p.Cout;
  size_t blocks = (total + 255) / 256;
  if (blocks > 65536 * 8) blocks = 65536 * 8;
  hipLaunchKernelGGL(conv_naive_f32_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p);
  IVX_CHECK_LAUNCH("ivx_conv_fwd_pio_naive");
  return IVX_OK;
}

extern "C" int ivx_conv_pair_supported(const ivx_conv_desc *d) {
  if (!d || d->in_dtype != IVX_F32 || d->out_dtype != IVX_F32 || d->out_mode != 0 || d->Cin % 16 != 0 || d->Cin <= 0) return 0;
  if (d->wgt_layout == 1 && d->Cin % 32 != 0) return 0;
  if (d->KD > 8 || d->KH > 8 || d->KW > 8 || d->KD < 1 || d->KH < 1 || d->KW < 1) return 0;
  const int64_t in_b = (int64_t)d->B * d->D * d->H * d->W * d->Cin * 4, w_b = (int64_t)d->Cout * d->KD * d->KH * d->KW * d->Cin * 4;
  return in_b < (1LL << 31) && w_b < (1LL << 31) ? 1 : 0;   // no batch slicing in this form: the caller falls back to fp32 MFMA
}

extern "C" int ivx_conv_fwd_naive(const ivx_conv_desc *d, const void *in, const void *wgt, const float *scale,
                                  const float *shift, const void *res, void *out, ivx_stream_t stream) {
  ConvParams p;
  int rc = fill_params(d, in, wgt, scale, shift, res, out, &p);
  if (rc != IVX_OK) return rc;
  const size_t total = (size_t)p.M * p.Cout;
  size_t blocks = (total + 255) / 256;
  if (blocks > 65536 * 8) blocks = 65536 * 8;
  hipLaunchKernelGGL(conv_naive_f32_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p);
  IVX_CHECK_LAUNCH("ivx_conv_fwd_naive");
  return IVX_OK;
}
#endif   // IVX_CONV_TU == 0
