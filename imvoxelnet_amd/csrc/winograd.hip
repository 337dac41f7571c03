// Winograd F(2x2, 3x3) over the first two spatial axes of a channels-last 3-D convolution (gfx950).
//
// The 3-D necks (mmdet3d/models/necks/imvoxelnet.py:99-113,181-230) are 3x3x3 convolutions on volumes that are wide in
// (x, y) and shallow in z (12 / 6 / 3 slices), and their 128- and 256-channel layers are bound by the fp32 MFMA rate.
// The minimal-filtering form over (x, y) needs 16 multiplications per 2x2 output tile and z-tap instead of 36:
//
//   V[xi][b,tx,ty,z,:]  = (Bt d B)[i][j]          d = 4x4 input patch at x = 2tx - pd + i, y = 2ty - ph + j   (xi = 4i + j)
//   U[xi][co,kz,:]      = (G g Gt)[i][j]          g = the 3x3 (kd,kh) slice of the filter for z-tap kz
//   M[xi]               = conv_z(V[xi], U[xi])    16 independent 1x1xKW convolutions (stride / padding of the z axis)
//   out[b,2tx+a,2ty+e]  = epilogue((At m A)[a][e])
//
// The 16 convolutions of M run as ONE grouped launch of the LDS-DMA implicit-GEMM kernel (conv_igemm.hip, grid.z = xi);
// the two transforms are streaming kernels (16-byte accesses, one thread per four channels).  The z axis stays a direct
// convolution because it is too shallow to tile.  Arithmetic is fp32 throughout; the result differs from the direct form
// by fp32 rounding only (different summation order).
#include "ivx_common.h"

int ivx_conv_grouped_launch(const ivx_conv_desc *d, int groups, const float *in, long long g_in, const float *wgt, long long g_w,
                            float *out, long long g_out, hipStream_t st);

namespace {

struct WinoP {
  const float *in, *scale, *shift, *res;
  float *out;
  float *V, *Mw;
  int B, X, Y, Z, C;        // input volume
  int Xo, Yo, Zo, Co;       // output volume
  int TX, TY;               // 2x2 output tiles
  int px, py;               // padding of the transformed axes
  int relu, res_mode, res_after_act;
  float post_scale;
};

__device__ __forceinline__ float4 f4add(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 f4sub(float4 a, float4 b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }

// V = Bt d B, Bt = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1].  One thread: four channels of one (b, tx, ty, z).
__global__ __launch_bounds__(256) void wino_input_kernel(const WinoP p) {
  const int C4 = p.C >> 2;
  const long long per_tile = (long long)p.Z * C4;                   // contiguous float4s of one (x, y) column
  const long long total = (long long)p.B * p.TX * p.TY * per_tile;
  const long long plane = total;                                     // float4s per xi plane
  const float4 *in = reinterpret_cast<const float4 *>(p.in);
  float4 *V = reinterpret_cast<float4 *>(p.V);
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
    const long long zc = t % per_tile;
    long long q = t / per_tile;
    const int ty = (int)(q % p.TY);
    q /= p.TY;
    const int tx = (int)(q % p.TX);
    const int b = (int)(q / p.TX);
    const int x0 = 2 * tx - p.px, y0 = 2 * ty - p.py;
    float4 d[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int x = x0 + i;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int y = y0 + j;
        const bool ok = (unsigned)x < (unsigned)p.X && (unsigned)y < (unsigned)p.Y;
        d[i][j] = ok ? in[(((long long)b * p.X + x) * p.Y + y) * per_tile + zc] : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
    float4 w[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {   // rows: Bt d
      w[0][j] = f4sub(d[0][j], d[2][j]);
      w[1][j] = f4add(d[1][j], d[2][j]);
      w[2][j] = f4sub(d[2][j], d[1][j]);
      w[3][j] = f4sub(d[1][j], d[3][j]);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {   // columns: (Bt d) B
      const float4 v0 = f4sub(w[i][0], w[i][2]);
      const float4 v1 = f4add(w[i][1], w[i][2]);
      const float4 v2 = f4sub(w[i][2], w[i][1]);
      const float4 v3 = f4sub(w[i][1], w[i][3]);
      V[(long long)(4 * i + 0) * plane + t] = v0;
      V[(long long)(4 * i + 1) * plane + t] = v1;
      V[(long long)(4 * i + 2) * plane + t] = v2;
      V[(long long)(4 * i + 3) * plane + t] = v3;
    }
  }
}

__device__ __forceinline__ float wino_finish(const WinoP &p, float acc, float sc, float sf, float r) {
  float v = acc * sc + sf;
  if (p.res_mode && !p.res_after_act) v += r;
  if (p.relu) v = v > 0.f ? v : 0.f;
  if (p.res_mode && p.res_after_act) v += r;
  return v * p.post_scale;
}

// out = epilogue(At m A), At = [1 1 1 0; 0 1 -1 -1].  One thread: four channels of one (b, tx, ty, zo) -> 2x2 outputs.
__global__ __launch_bounds__(256) void wino_output_kernel(const WinoP p) {
  const int C4 = p.Co >> 2;
  const long long per_tile = (long long)p.Zo * C4;
  const long long total = (long long)p.B * p.TX * p.TY * per_tile;
  const long long plane = total;
  const float4 *Mw = reinterpret_cast<const float4 *>(p.Mw);
  const float4 *res = reinterpret_cast<const float4 *>(p.res);
  float4 *out = reinterpret_cast<float4 *>(p.out);
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
    const long long zc = t % per_tile;
    const int c4 = (int)(zc % C4);
    long long q = t / per_tile;
    const int ty = (int)(q % p.TY);
    q /= p.TY;
    const int tx = (int)(q % p.TX);
    const int b = (int)(q / p.TX);
    float4 r[2][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {   // At m
      const float4 m0 = Mw[(long long)(0 + j) * plane + t], m1 = Mw[(long long)(4 + j) * plane + t];
      const float4 m2 = Mw[(long long)(8 + j) * plane + t], m3 = Mw[(long long)(12 + j) * plane + t];
      r[0][j] = f4add(f4add(m0, m1), m2);
      r[1][j] = f4sub(f4sub(m1, m2), m3);
    }
    float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sf = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p.scale) sc = reinterpret_cast<const float4 *>(p.scale)[c4];
    if (p.shift) sf = reinterpret_cast<const float4 *>(p.shift)[c4];
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      const int x = 2 * tx + a;
      if (x >= p.Xo) continue;
      const float4 y0 = f4add(f4add(r[a][0], r[a][1]), r[a][2]);
      const float4 y1 = f4sub(f4sub(r[a][1], r[a][2]), r[a][3]);
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int y = 2 * ty + e;
        if (y >= p.Yo) continue;
        const float4 acc = e == 0 ? y0 : y1;
        const long long o = (((long long)b * p.Xo + x) * p.Yo + y) * per_tile + zc;
        float4 rr = make_float4(0.f, 0.f, 0.f, 0.f);
        if (p.res_mode) rr = res[o];
        float4 v;
        v.x = wino_finish(p, acc.x, sc.x, sf.x, rr.x);
        v.y = wino_finish(p, acc.y, sc.y, sf.y, rr.y);
        v.z = wino_finish(p, acc.z, sc.z, sf.z, rr.z);
        v.w = wino_finish(p, acc.w, sc.w, sf.w, rr.w);
        out[o] = v;
      }
    }
  }
}

// U = G g Gt, G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1].  wgt is layout 0 [Co,3,3,KW,Ci]; U is [16][Co][K] with the K order
// of `kmode` (0: k = kz*Ci + ci; 1: k = (ci/32)*KW*32 + kz*32 + ci%32), i.e. each xi holds a packed 1x1xKW filter bank.
__global__ __launch_bounds__(256) void wino_weight_kernel(const float *__restrict__ w, float *__restrict__ U, int Co, int KW, int Ci,
                                                          int kmode) {
  const long long total = (long long)Co * KW * Ci;
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= total) return;
  const int ci = (int)(t % Ci);
  const int kz = (int)((t / Ci) % KW);
  const int co = (int)(t / ((long long)Ci * KW));
  float g[3][3];
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int e = 0; e < 3; ++e) g[a][e] = w[((((long long)co * 3 + a) * 3 + e) * KW + kz) * Ci + ci];
  float h[4][3];
#pragma unroll
  for (int e = 0; e < 3; ++e) {
    h[0][e] = g[0][e];
    h[1][e] = 0.5f * (g[0][e] + g[1][e] + g[2][e]);
    h[2][e] = 0.5f * (g[0][e] - g[1][e] + g[2][e]);
    h[3][e] = g[2][e];
  }
  const long long K = (long long)KW * Ci;
  const long long k = kmode == 1 ? ((long long)(ci >> 5) * KW + kz) * 32 + (ci & 31) : (long long)kz * Ci + ci;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float u0 = h[i][0];
    const float u1 = 0.5f * (h[i][0] + h[i][1] + h[i][2]);
    const float u2 = 0.5f * (h[i][0] - h[i][1] + h[i][2]);
    const float u3 = h[i][2];
    U[((long long)(4 * i + 0) * Co + co) * K + k] = u0;
    U[((long long)(4 * i + 1) * Co + co) * K + k] = u1;
    U[((long long)(4 * i + 2) * Co + co) * K + k] = u2;
    U[((long long)(4 * i + 3) * Co + co) * K + k] = u3;
  }
}

struct WinoDims {
  int Xo, Yo, Zo, TX, TY;
  int64_t v_elems, m_elems;   // elements of one xi plane of V / M
};

int wino_dims(const ivx_conv_desc *d, WinoDims *w, const char *who) {
  IVX_REQUIRE(d, "%s: null descriptor", who);
  IVX_REQUIRE(d->KD == 3 && d->KH == 3 && d->sd == 1 && d->sh == 1, "%s: needs a 3x3 kernel with stride 1 on the first two axes", who);
  IVX_REQUIRE(d->KW >= 1 && d->KW <= 8 && d->sw >= 1 && d->pd >= 0 && d->ph >= 0 && d->pw >= 0, "%s: bad z kernel / stride / padding", who);
  IVX_REQUIRE(d->B > 0 && d->D > 0 && d->H > 0 && d->W > 0, "%s: non-positive dims", who);
  IVX_REQUIRE(d->Cin > 0 && d->Cin % 4 == 0 && d->Cout > 0 && d->Cout % 4 == 0, "%s: Cin and Cout must be multiples of 4", who);
  IVX_REQUIRE(d->in_dtype == IVX_F32 && d->out_dtype == IVX_F32, "%s: fp32 only", who);
  IVX_REQUIRE(d->out_mode == 0 && (d->res_mode == 0 || d->res_mode == 1), "%s: out_mode 0 and res_mode 0/1 only", who);
  IVX_REQUIRE(d->wgt_layout == 0 || (d->wgt_layout == 1 && d->Cin % 32 == 0), "%s: wgt_layout 1 needs Cin %% 32 == 0", who);
  int32_t Xo, Yo, Zo;
  if (ivx_conv_out_dims(d, &Xo, &Yo, &Zo) != IVX_OK) return IVX_ERR_INVALID_ARG;
  w->Xo = Xo; w->Yo = Yo; w->Zo = Zo;
  w->TX = (Xo + 1) / 2; w->TY = (Yo + 1) / 2;
  w->v_elems = (int64_t)d->B * w->TX * w->TY * d->W * d->Cin;
  w->m_elems = (int64_t)d->B * w->TX * w->TY * Zo * d->Cout;
  return IVX_OK;
}

// descriptor of ONE xi convolution: volume [B, TX, TY, Z, Cin], kernel 1x1xKW along z
ivx_conv_desc wino_group_desc(const ivx_conv_desc *d, const WinoDims &w) {
  ivx_conv_desc g = *d;
  g.D = w.TX; g.H = w.TY;
  g.KD = 1; g.KH = 1; g.pd = 0; g.ph = 0; g.sd = 1; g.sh = 1;
  g.relu = 0; g.res_mode = 0; g.res_h = 0; g.res_w = 0; g.res_after_act = 0; g.post_scale = 1.0f;
  return g;
}

unsigned wino_blocks(int64_t items) {
  int64_t b = (items + 255) / 256;
  return (unsigned)(b < 1 ? 1 : (b > (1 << 20) ? (1 << 20) : b));
}

}  // namespace

extern "C" int ivx_conv_winograd_supported(const ivx_conv_desc *d) {
  WinoDims w;
  if (wino_dims(d, &w, "ivx_conv_winograd_supported") != IVX_OK) return 0;
  // one xi plane is one group of the grouped launch: 31-bit buffer offsets
  if (w.v_elems * 4 >= (1LL << 31) || w.m_elems >= (1LL << 31) - 512 * (int64_t)d->Cout) return 0;
  if ((int64_t)d->Cout * d->KW * d->Cin * 4 >= (1LL << 31)) return 0;
  return 1;
}

extern "C" int64_t ivx_conv_winograd_weight_elems(const ivx_conv_desc *d) {
  WinoDims w;
  if (wino_dims(d, &w, "ivx_conv_winograd_weight_elems") != IVX_OK) return -1;
  return (int64_t)16 * d->Cout * d->KW * d->Cin;
}

extern "C" int ivx_conv_winograd_weights(const ivx_conv_desc *d, const float *wgt, float *u, ivx_stream_t stream) {
  WinoDims w;
  int rc = wino_dims(d, &w, "ivx_conv_winograd_weights");
  if (rc != IVX_OK) return rc;
  IVX_REQUIRE(wgt && u, "ivx_conv_winograd_weights: null argument");
  const int64_t total = (int64_t)d->Cout * d->KW * d->Cin;
  hipLaunchKernelGGL(wino_weight_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, wgt, u, d->Cout,
                     d->KW, d->Cin, d->wgt_layout);
  IVX_CHECK_LAUNCH("ivx_conv_winograd_weights");
  return IVX_OK;
}

extern "C" int64_t ivx_conv_winograd_workspace_bytes(const ivx_conv_desc *d) {
  WinoDims w;
  if (wino_dims(d, &w, "ivx_conv_winograd_workspace_bytes") != IVX_OK) return -1;
  return ivx_align_up(16 * w.v_elems * 4, 256) + ivx_align_up(16 * w.m_elems * 4, 256);
}

namespace {
int wino_setup(const ivx_conv_desc *d, const void *in, const float *scale, const float *shift, const void *res, void *out,
               void *workspace, int64_t workspace_bytes, WinoDims *w, WinoP *p, const char *who) {
  int rc = wino_dims(d, w, who);
  if (rc != IVX_OK) return rc;
  IVX_REQUIRE(workspace, "%s: null workspace", who);
  if (!ivx_conv_winograd_supported(d)) {
    ivx_set_error("%s: one transformed plane must stay below 2 GiB (use ivx_conv_fwd)", who);
    return IVX_ERR_UNSUPPORTED;
  }
  const int64_t need = ivx_conv_winograd_workspace_bytes(d);
  if (workspace_bytes < need) {
    ivx_set_error("%s: workspace too small (%lld < %lld); size it with ivx_conv_winograd_workspace_bytes", who,
                  (long long)workspace_bytes, (long long)need);
    return IVX_ERR_WORKSPACE;
  }
  p->in = (const float *)in; p->scale = scale; p->shift = shift; p->res = d->res_mode ? (const float *)res : nullptr;
  p->out = (float *)out;
  p->V = (float *)workspace;
  p->Mw = (float *)((char *)workspace + ivx_align_up(16 * w->v_elems * 4, 256));
  p->B = d->B; p->X = d->D; p->Y = d->H; p->Z = d->W; p->C = d->Cin;
  p->Xo = w->Xo; p->Yo = w->Yo; p->Zo = w->Zo; p->Co = d->Cout;
  p->TX = w->TX; p->TY = w->TY; p->px = d->pd; p->py = d->ph;
  p->relu = d->relu; p->res_mode = d->res_mode; p->res_after_act = d->res_after_act;
  p->post_scale = d->post_scale == 0.f ? 1.0f : d->post_scale;
  return IVX_OK;
}
}  // namespace

// The three stages are separate entry points so that a caller can time them (bench.py); ivx_conv_winograd_fwd runs all.
extern "C" int ivx_conv_winograd_input(const ivx_conv_desc *d, const void *in, void *workspace, int64_t workspace_bytes,
                                       ivx_stream_t stream) {
  WinoDims w;
  WinoP p;
  float dummy;
  IVX_REQUIRE(in, "ivx_conv_winograd_input: null argument");
  int rc = wino_setup(d, in, nullptr, nullptr, &dummy, &dummy, workspace, workspace_bytes, &w, &p, "ivx_conv_winograd_input");
  if (rc != IVX_OK) return rc;
  hipLaunchKernelGGL(wino_input_kernel, dim3(wino_blocks(w.v_elems / 4)), dim3(256), 0, (hipStream_t)stream, p);
  IVX_CHECK_LAUNCH("ivx_conv_winograd_input");
  return IVX_OK;
}

extern "C" int ivx_conv_winograd_gemm(const ivx_conv_desc *d, const float *u, void *workspace, int64_t workspace_bytes,
                                      ivx_stream_t stream) {
  WinoDims w;
  WinoP p;
  float dummy;
  IVX_REQUIRE(u, "ivx_conv_winograd_gemm: null argument");
  int rc = wino_setup(d, &dummy, nullptr, nullptr, &dummy, &dummy, workspace, workspace_bytes, &w, &p, "ivx_conv_winograd_gemm");
  if (rc != IVX_OK) return rc;
  const ivx_conv_desc g = wino_group_desc(d, w);
  rc = ivx_conv_grouped_launch(&g, 16, p.V, w.v_elems, u, (long long)d->Cout * d->KW * d->Cin, p.Mw, w.m_elems, (hipStream_t)stream);
  if (rc != IVX_OK) return rc;
  IVX_CHECK_LAUNCH("ivx_conv_winograd_gemm");
  return IVX_OK;
}

extern "C" int ivx_conv_winograd_output(const ivx_conv_desc *d, const float *scale, const float *shift, const void *res, void *out,
                                        void *workspace, int64_t workspace_bytes, ivx_stream_t stream) {
  WinoDims w;
  WinoP p;
  float dummy;
  IVX_REQUIRE(out, "ivx_conv_winograd_output: null argument");
  IVX_REQUIRE(!d || d->res_mode == 0 || res, "ivx_conv_winograd_output: res_mode set but res is NULL");
  int rc = wino_setup(d, &dummy, scale, shift, res, out, workspace, workspace_bytes, &w, &p, "ivx_conv_winograd_output");
  if (rc != IVX_OK) return rc;
  hipLaunchKernelGGL(wino_output_kernel, dim3(wino_blocks(w.m_elems / 4)), dim3(256), 0, (hipStream_t)stream, p);
  IVX_CHECK_LAUNCH("ivx_conv_winograd_output");
  return IVX_OK;
}

extern "C" int ivx_conv_winograd_fwd(const ivx_conv_desc *d, const void *in, const float *u, const float *scale, const float *shift,
                                     const void *res, void *out, void *workspace, int64_t workspace_bytes, ivx_stream_t stream) {
  int rc = ivx_conv_winograd_input(d, in, workspace, workspace_bytes, stream);
  if (rc != IVX_OK) return rc;
  rc = ivx_conv_winograd_gemm(d, u, workspace, workspace_bytes, stream);
  if (rc != IVX_OK) return rc;
  return ivx_conv_winograd_output(d, scale, shift, res, out, workspace, workspace_bytes, stream);
}
