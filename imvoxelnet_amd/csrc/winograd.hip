// Winograd minimal filtering F(m x m, 3x3), m = 2, 4 or 6, over the first two spatial axes of a channels-last 3-D convolution
// (gfx950).
//
// The 3-D necks (mmdet3d/models/necks/imvoxelnet.py:99-113,181-230) are 3x3x3 convolutions on volumes that are wide in
// (x, y) and shallow in z (12 / 6 / 3 slices), and their 128- and 256-channel layers are bound by the fp32 MFMA rate.
// With n = m + 2, the minimal-filtering form over (x, y) needs n*n multiplications per m x m output tile and z-tap
// instead of 9*m*m (16 vs 36 for m = 2, 36 vs 144 for m = 4, 64 vs 324 for m = 6):
//
//   V[xi][b,tx,ty,z,:]  = (Bt d B)[i][j]          d = n x n input patch at x = m*tx - pd + i, y = m*ty - ph + j   (xi = n*i + j)
//   U[xi][co,kz,:]      = (G g Gt)[i][j]          g = the 3x3 (kd,kh) slice of the filter for z-tap kz
//   M[xi]               = conv_z(V[xi], U[xi])    n*n independent 1x1xKW convolutions (stride / padding of the z axis)
//   out[b,m*tx+a,m*ty+e] = epilogue((At M A)[a][e])
//
// The n*n convolutions of M run as ONE grouped launch of the LDS-DMA implicit-GEMM kernel (conv_igemm.hip, grid.z = xi);
// the two transforms are streaming kernels (one thread per 4 (m = 2) or 2 (m = 4, 6) channels).  The z axis stays a direct
// convolution because it is too shallow to tile.  Arithmetic is fp32 throughout; the result differs from the direct form
// by fp32 rounding only: measured max deviation from the direct MFMA kernel on the KITTI neck layers, as a fraction of the
// output range: m = 2 up to 4e-6, m = 4 up to 2.4e-5, m = 6 up to 4e-5 (profiles/r01_conv_layers.log).
#include "ivx_common.h"
#include <stdlib.h>
int ivx_conv_launch_fold4w(const _Float16 *V, long long vs, const _Float16 *U, long long us, int M, int Cin_stored, int Cout, int Z, const IvxWinoFold &f,
                           hipStream_t st);

int ivx_conv_grouped_launch(const ivx_conv_desc *d, int groups, const float *in, long long g_in, const float *wgt, long long g_w,
                            float *out, long long g_out, hipStream_t st, const unsigned *cp_src, unsigned *cp_dst);
int ivx_conv_grouped_fold4(const ivx_conv_desc *d, int groups, const float *in, long long g_in, const float *wgt, long long g_w, const IvxWinoFold *f,
                           hipStream_t st);
int ivx_conv_fold4_blocks(long long M, int Cout);

namespace {

struct WinoP {
  const float *in, *scale, *shift, *res;
  float *out;
  float *V, *Mw;
  long long vs, ms;         // xi-plane strides of V / M in floats (>= the plane size: see wino_plane_stride)
  int B, X, Y, Z, C;        // input volume
  int Xo, Yo, Zo, Co;       // output volume
  int TX, TY;               // m x m output tiles
  int px, py;               // padding of the transformed axes
  int relu, res_mode, res_after_act;
  float post_scale;
  float *pmax;              // output transform: per-workgroup max |out| goes to pmax[blockIdx.x] (the next layer's input scale), or NULL
  const unsigned *hdr;      // pair operands: {bits of max |input|, bits of the filter scale} (device; see wino_pair_vscale), else NULL
#ifdef IVX_CONV_TIMELINE
  unsigned long long *tl;   // debug build (tools/neck_timeline.py): 8 words per workgroup -- s_memrealtime at entry and at the end, HW_ID, XCC_ID
#endif
};
#ifdef IVX_CONV_TIMELINE
#define WINO_TL_BEGIN const unsigned long long wtl0 = __builtin_amdgcn_s_memrealtime();
#define WINO_TL_END(p)                                                                                              \
  if ((p).tl && threadIdx.x == 0) {                                                                                \
    unsigned long long *t_ = (p).tl + (size_t)blockIdx.x * 8;                                                       \
    t_[0] = wtl0; t_[1] = wtl0; t_[2] = wtl0; t_[3] = __builtin_amdgcn_s_memrealtime();                             \
    t_[4] = __builtin_amdgcn_s_getreg(63492); t_[5] = __builtin_amdgcn_s_getreg(63508);                             \
  }
#else
#define WINO_TL_BEGIN
#define WINO_TL_END(p)
#endif

// Power-of-two scales of the fp16-pair operands (ivx_conv_desc.wino_operands = IVX_F16_PAIR), chosen on the device from the data so
// that the largest value lands in [2^14, 2^15) -- below fp16's 65504 with room for the transform's growth -- and everything down to
// 2^-18 of it keeps a normal lo half (fp16 subnormals have a fixed spacing of 2^-24, so a lo half below 2^-14 loses relative precision:
// measured 1.8e-4 rms with a fixed scale at activation scale 1e-3, reproduced by tools/wino_pair_sim.py).  Activations: |V| = |Bt d B| <= 225 max|d| for F(6,3) (|Bt| row sums <= 15), 100 for F(4,3): bounded by 256.
// Filters: the exact max |U| is reduced while they are transformed.  Both scales are undone in the output transform (exact).
// A non-finite maximum (an Inf / NaN somewhere in the tensor, or values beyond 3e38) carries no information about the finite values: the
// operands then get the fixed scale 2^-8 and the conversion SATURATES finite values at +-65504 (wino_pair2's `sat`), so that the damage
// stays where fp32 arithmetic would keep it -- the tiles that touch the non-finite element -- instead of every V overflowing to Inf.
__device__ __forceinline__ bool wino_amax_bad(const float amax) { return !(amax < 3.0e38f); }      // NaN / Inf / huge
__device__ __forceinline__ float wino_pow2_scale(const float amax, const float gain_log2) {   // s = 2^k with gain * amax * s in [2^14, 2^15)
  if (wino_amax_bad(amax)) return 0.00390625f;
  if (!(amax > 0.f)) return 1.0f;
  int e;
  (void)frexpf(amax, &e);                        // amax in [2^(e-1), 2^e)
  int k = 15 - (int)gain_log2 - e;
  k = k < -120 ? -120 : (k > 120 ? 120 : k);
  return ldexpf(1.0f, k);
}
__device__ __forceinline__ float wino_pair_vscale(const unsigned *hdr) { return wino_pow2_scale(__uint_as_float(hdr[0]), 8.f); }
// multiplier of M in the output transform: 1 / (activation scale * filter scale); 1 for fp32 operands (no header)
__device__ __forceinline__ float wino_mscale(const unsigned *hdr) { return hdr ? 1.0f / (wino_pair_vscale(hdr) * __uint_as_float(hdr[1])) : 1.0f; }

__device__ __forceinline__ float vabsmax(const float v) { return fabsf(v); }
__device__ __forceinline__ float vabsmax(const float2 v) { return fmaxf(fabsf(v.x), fabsf(v.y)); }
__device__ __forceinline__ float vabsmax(const float4 v) { return fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))); }
// max over the 256 threads of a workgroup -> pmax[blockIdx.x] (every thread must call it)
__device__ __forceinline__ void wino_block_max_store(float m, float *pmax) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  __shared__ float wmax[4];
  if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) pmax[blockIdx.x] = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
}
// the per-workgroup maxima of a producer (wino_block_max_store) -> hdr[0] (zeroed by the caller)
__global__ __launch_bounds__(256) void wino_amax_partials_kernel(const float *__restrict__ part, int n, unsigned *out) {
  float m = 0.f;
  for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < n; t += gridDim.x * blockDim.x) m = fmaxf(m, part[t]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  if ((threadIdx.x & 63) == 0) atomicMax(out, __float_as_uint(m));
}

// The same reduction by ONE workgroup that WRITES hdr[0]: no zeroing launch in front of it (round 6: the 4-byte hipMemsetAsync + the 64-block
// reduction were two ~5 us launches in front of every input transform of the chained neck; up to 65536 maxima = 256 KB, a few us from L2).
__global__ __launch_bounds__(1024) void wino_amax_partials_write_kernel(const float *__restrict__ part, int n, unsigned *out) {
  __shared__ float red[16];
  float m = 0.f;
  for (int t = threadIdx.x; t < n; t += 1024) m = fmaxf(m, part[t]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x < 64) {
    m = threadIdx.x < 16 ? red[threadIdx.x] : 0.f;
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if (threadIdx.x == 0) *out = __float_as_uint(m);
  }
}

// max |x| over a tensor into hdr[0] (zeroed by the caller): non-negative floats order like their bit patterns.  A read-only stream:
// four independent 16-byte loads per lane and iteration keep enough bytes in flight for HBM (one load per iteration ran at 2.6 TB/s).
typedef float wf32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float wino_amax4(const wf32x4 v, const float m) {
  return fmaxf(fmaxf(m, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
}
__global__ __launch_bounds__(256) void wino_amax_kernel(const wf32x4 *__restrict__ x, size_t n4, unsigned *out) {
  float m = 0.f;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; t + 3 * stride < n4; t += 4 * stride) {
    const wf32x4 a = __builtin_nontemporal_load(x + t), b = __builtin_nontemporal_load(x + t + stride);
    const wf32x4 c = __builtin_nontemporal_load(x + t + 2 * stride), d = __builtin_nontemporal_load(x + t + 3 * stride);
    m = wino_amax4(d, wino_amax4(c, wino_amax4(b, wino_amax4(a, m))));
  }
  for (; t < n4; t += stride) m = wino_amax4(__builtin_nontemporal_load(x + t), m);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  // one atomic per workgroup: same-address device-scope atomics serialise (8192 per-wave atomics cost more than the 658 MB read)
  __shared__ float wmax[4];
  if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) atomicMax(out, __float_as_uint(fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]))));
}

// In place: n fp32 values -> fp16 (hi, lo) pairs of s * x in the IVX_F16_PAIR order, s from hdr[0] = bits of max |x| (one thread per 16
// values: it reads its 64 bytes before it writes them); thread 0 leaves s in hdr[1].
__global__ __launch_bounds__(256) void wino_pair_inplace_kernel(float *buf, size_t n16, unsigned *hdr) {
  const float s = wino_pow2_scale(__uint_as_float(hdr[0]), 0.f);
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t == 0) hdr[1] = __float_as_uint(s);
  if (t >= n16) return;
  float4 v[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) v[q] = reinterpret_cast<const float4 *>(buf)[4 * t + q];
  typedef _Float16 h8 __attribute__((ext_vector_type(8)));
  h8 hi[2], lo[2];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const float x[4] = {v[q].x, v[q].y, v[q].z, v[q].w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float y = __builtin_fminf(__builtin_fmaxf(x[e] * s, -65504.f), 65504.f);
      const _Float16 h = (_Float16)y;
      hi[q >> 1][(q & 1) * 4 + e] = h;
      lo[q >> 1][(q & 1) * 4 + e] = (_Float16)(y - (float)h);
    }
  }
  h8 *o = reinterpret_cast<h8 *>(buf + 16 * t);
  o[0] = hi[0]; o[1] = hi[1]; o[2] = lo[0]; o[3] = lo[1];
}

// channel vectors: 4 floats per thread for m = 2, 2 for m = 4 (36 live values per thread instead of 16)
template <int W> struct VecT;
template <> struct VecT<4> { typedef float4 T; };
template <> struct VecT<2> { typedef float2 T; };
template <> struct VecT<1> { typedef float T; };
__device__ __forceinline__ float vzero(float *) { return 0.f; }
__device__ __forceinline__ float vone(float *) { return 1.f; }
__device__ __forceinline__ float4 vzero(float4 *) { return make_float4(0.f, 0.f, 0.f, 0.f); }
__device__ __forceinline__ float2 vzero(float2 *) { return make_float2(0.f, 0.f); }
__device__ __forceinline__ float4 vone(float4 *) { return make_float4(1.f, 1.f, 1.f, 1.f); }
__device__ __forceinline__ float2 vone(float2 *) { return make_float2(1.f, 1.f); }
__device__ __forceinline__ float4 operator+(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 operator-(float4 a, float4 b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }
__device__ __forceinline__ float4 operator*(float s, float4 a) { return make_float4(s * a.x, s * a.y, s * a.z, s * a.w); }
__device__ __forceinline__ float2 operator+(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 operator-(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ float2 operator*(float s, float2 a) { return make_float2(s * a.x, s * a.y); }

// 1-D transforms.  m = 2 (points 0, +-1, inf):  Bt = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1]   At = [1 1 1 0; 0 1 -1 -1]
//                                               G  = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1]
// m = 4 (points 0, +-1, +-2, inf):  Bt = [4 0 -5 0 1 0; 0 -4 -4 1 1 0; 0 4 -4 -1 1 0; 0 -2 -1 2 1 0; 0 2 -1 -2 1 0; 0 4 0 -5 0 1]
//                                   At = [1 1 1 1 1 0; 0 1 -1 2 -2 0; 0 1 1 4 4 0; 0 1 -1 8 -8 1]
//                                   G  = [1/4 0 0; -1/6 -1/6 -1/6; -1/6 1/6 -1/6; 1/24 1/12 1/6; 1/24 -1/12 1/6; 0 0 1]
template <int MT, typename V> struct Wino1D;
template <typename V> struct Wino1D<2, V> {
  static __device__ __forceinline__ void in(const V (&d)[4], V (&t)[4]) {
    t[0] = d[0] - d[2];
    t[1] = d[1] + d[2];
    t[2] = d[2] - d[1];
    t[3] = d[1] - d[3];
  }
  static __device__ __forceinline__ void out(const V (&m)[4], V (&y)[2]) {
    y[0] = (m[0] + m[1]) + m[2];
    y[1] = (m[1] - m[2]) - m[3];
  }
  static __device__ __forceinline__ void wgt(const float (&g)[3], float (&u)[4]) {
    u[0] = g[0];
    u[1] = 0.5f * (g[0] + g[1] + g[2]);
    u[2] = 0.5f * (g[0] - g[1] + g[2]);
    u[3] = g[2];
  }
};
template <typename V> struct Wino1D<4, V> {
  static __device__ __forceinline__ void in(const V (&d)[6], V (&t)[6]) {
    const V a = d[4] - 4.0f * d[2], b = d[3] - 4.0f * d[1];
    const V c = d[4] - d[2], e = 2.0f * (d[3] - d[1]);
    t[0] = (4.0f * d[0] - 5.0f * d[2]) + d[4];
    t[1] = a + b;
    t[2] = a - b;
    t[3] = c + e;
    t[4] = c - e;
    t[5] = (4.0f * d[1] - 5.0f * d[3]) + d[5];
  }
  static __device__ __forceinline__ void out(const V (&m)[6], V (&y)[4]) {
    const V s1 = m[1] + m[2], d1 = m[1] - m[2], s2 = m[3] + m[4], d2 = m[3] - m[4];
    y[0] = (m[0] + s1) + s2;
    y[1] = d1 + 2.0f * d2;
    y[2] = s1 + 4.0f * s2;
    y[3] = (d1 + 8.0f * d2) + m[5];
  }
  static __device__ __forceinline__ void wgt(const float (&g)[3], float (&u)[6]) {
    u[0] = 0.25f * g[0];
    u[1] = (-1.0f / 6.0f) * (g[0] + g[1] + g[2]);
    u[2] = (-1.0f / 6.0f) * (g[0] - g[1] + g[2]);
    u[3] = (1.0f / 24.0f) * g[0] + (1.0f / 12.0f) * g[1] + (1.0f / 6.0f) * g[2];
    u[4] = (1.0f / 24.0f) * g[0] - (1.0f / 12.0f) * g[1] + (1.0f / 6.0f) * g[2];
    u[5] = g[2];
  }
};

// m = 6 (points 0, +-1, +-2, +-1/2, inf): the matrices of the widely used F(6,3) construction
//   Bt = [1 0 -21/4 0 21/4 0 -1 0; 0 1 1 -17/4 -17/4 1 1 0; 0 -1 1 17/4 -17/4 -1 1 0; 0 1/2 1/4 -5/2 -5/4 2 1 0;
//         0 -1/2 1/4 5/2 -5/4 -2 1 0; 0 2 4 -5/2 -5 1/2 1 0; 0 -2 4 5/2 -5 -1/2 1 0; 0 -1 0 21/4 0 -21/4 0 1]
//   At = [1 1 1 1 1 1 1 0; 0 1 -1 2 -2 1/2 -1/2 0; 0 1 1 4 4 1/4 1/4 0; 0 1 -1 8 -8 1/8 -1/8 0; 0 1 1 16 16 1/16 1/16 0;
//         0 1 -1 32 -32 1/32 -1/32 1]
//   G  = [1 0 0; -2/9 -2/9 -2/9; -2/9 2/9 -2/9; 1/90 1/45 2/45; 1/90 -1/45 2/45; 32/45 16/45 8/45; 32/45 -16/45 8/45; 0 0 1]
template <typename V> struct Wino1D<6, V> {
  static __device__ __forceinline__ void in(const V (&d)[8], V (&t)[8]) {
    t[0] = (d[0] - d[6]) + 5.25f * (d[4] - d[2]);
    t[7] = (d[7] - d[1]) + 5.25f * (d[3] - d[5]);
    const V a12 = (d[2] + d[6]) - 4.25f * d[4], b12 = (d[1] + d[5]) - 4.25f * d[3];
    t[1] = a12 + b12;
    t[2] = a12 - b12;
    const V a34 = (d[6] + 0.25f * d[2]) - 1.25f * d[4], b34 = (0.5f * d[1] - 2.5f * d[3]) + 2.0f * d[5];
    t[3] = a34 + b34;
    t[4] = a34 - b34;
    const V a56 = d[6] + 4.0f * (d[2] - 1.25f * d[4]), b56 = (2.0f * d[1] - 2.5f * d[3]) + 0.5f * d[5];
    t[5] = a56 + b56;
    t[6] = a56 - b56;
  }
  static __device__ __forceinline__ void out(const V (&m)[8], V (&y)[6]) {
    const V sa = m[1] + m[2], da = m[1] - m[2], sb = m[3] + m[4], db = m[3] - m[4], sc = m[5] + m[6], dc = m[5] - m[6];
    y[0] = ((m[0] + sa) + sb) + sc;
    y[1] = (da + 2.0f * db) + 0.5f * dc;
    y[2] = (sa + 4.0f * sb) + 0.25f * sc;
    y[3] = (da + 8.0f * db) + 0.125f * dc;
    y[4] = (sa + 16.0f * sb) + 0.0625f * sc;
    y[5] = ((da + 32.0f * db) + 0.03125f * dc) + m[7];
  }
  static __device__ __forceinline__ void wgt(const float (&g)[3], float (&u)[8]) {
    u[0] = g[0];
    u[1] = (-2.0f / 9.0f) * (g[0] + g[1] + g[2]);
    u[2] = (-2.0f / 9.0f) * (g[0] - g[1] + g[2]);
    u[3] = (1.0f / 90.0f) * g[0] + (1.0f / 45.0f) * g[1] + (2.0f / 45.0f) * g[2];
    u[4] = (1.0f / 90.0f) * g[0] - (1.0f / 45.0f) * g[1] + (2.0f / 45.0f) * g[2];
    u[5] = (32.0f / 45.0f) * g[0] + (16.0f / 45.0f) * g[1] + (8.0f / 45.0f) * g[2];
    u[6] = (32.0f / 45.0f) * g[0] - (16.0f / 45.0f) * g[1] + (8.0f / 45.0f) * g[2];
    u[7] = g[2];
  }
};

// V = Bt d B.  One thread: VW channels of one (b, tx, ty, z); xi plane stride = all threads.
typedef _Float16 wf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x2w __attribute__((ext_vector_type(2)));
// (hi, lo) fp16 halves of two scaled values: hi = fp16(x), lo = fp16(x - hi).  No saturation needed: the scale puts 256 max|in| below
// 2^15 and |V| <= 225 max|in| (wino_pair_vscale), so |x| < 65504 for every finite input.
__device__ __forceinline__ float wino_sat16(const float x) {      // finite overflow -> +-65504; Inf and NaN pass through
  return (x > 65504.f && x < __builtin_inff()) ? 65504.f : ((x < -65504.f && x > -__builtin_inff()) ? -65504.f : x);
}
__device__ __forceinline__ void wino_pair2(const float2 v, const float s, unsigned *hi, unsigned *lo, const bool sat = false) {
  float x0 = v.x * s, x1 = v.y * s;
  if (sat) { x0 = wino_sat16(x0); x1 = wino_sat16(x1); }      // (wave-uniform: only when the tensor's maximum is not finite)
  wf16x2 h, l;
  h[0] = (_Float16)x0;
  h[1] = (_Float16)x1;
  l[0] = (_Float16)(x0 - (float)h[0]);
  l[1] = (_Float16)(x1 - (float)h[1]);
  *hi = __builtin_bit_cast(unsigned, h);
  *lo = __builtin_bit_cast(unsigned, l);
}

// (Round 4, an ablation that skips the loads of the two halo columns / of all halo rows and columns -- wrong results, valid timing: 0.357 ->
// 0.347 / 0.345 ms: the 1.78x re-read of the input through L2 / the Infinity Cache costs this kernel 3 %, not the 20 % its share of the
// fabric traffic suggests; a tile order or an LDS exchange that removes it has nothing to win.)
template <int MT, int VW, int PAIR = 0>
__global__ __launch_bounds__(256) void wino_input_kernel(const WinoP p) {
  WINO_TL_BEGIN
  typedef typename VecT<VW>::T V;
  constexpr int N = MT + 2;
  static_assert(!PAIR || VW == 2, "pair operands: two channels per lane (one dword of hi, one of lo)");
  const int CV = p.C / VW;
  const long long per_tile = (long long)p.Z * CV;                    // contiguous vectors of one (x, y) column
  const long long total = (long long)p.B * p.TX * p.TY * per_tile;   // = vectors per xi plane
  const V *in = reinterpret_cast<const V *>(p.in);
  V *Vw = reinterpret_cast<V *>(p.V);
  const float vscale = PAIR ? wino_pair_vscale(p.hdr) : 1.0f;
  const bool sat = PAIR ? wino_amax_bad(__uint_as_float(p.hdr[0])) : false;
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
    const long long zc = t % per_tile;
    long long q = t / per_tile;
    const int ty = (int)(q % p.TY);
    q /= p.TY;
    const int tx = (int)(q % p.TX);
    const int b = (int)(q / p.TX);
    const int x0 = MT * tx - p.px, y0 = MT * ty - p.py;
    V w[N][N];   // w[i][j]: column transform (Bt d) of input column j
#pragma unroll
    for (int j = 0; j < N; ++j) {
      const int y = y0 + j;
      V d[N];
#pragma unroll
      for (int i = 0; i < N; ++i) {
        const int x = x0 + i;
        const bool ok = (unsigned)x < (unsigned)p.X && (unsigned)y < (unsigned)p.Y;
        d[i] = ok ? in[(((long long)b * p.X + x) * p.Y + y) * per_tile + zc] : vzero((V *)nullptr);
      }
      V c[N];
      Wino1D<MT, V>::in(d, c);
#pragma unroll
      for (int i = 0; i < N; ++i) w[i][j] = c[i];
    }
#pragma unroll
    for (int i = 0; i < N; ++i) {   // rows: (Bt d) B
      V v[N];
      Wino1D<MT, V>::in(w[i], v);
      if constexpr (PAIR) {
        // IVX_F16_PAIR order inside the plane: per voxel and 16 channels [hi x16 | lo x16]; this lane owns channels 2cv, 2cv + 1, i.e.
        // dword (cv / 8) * 16 + cv % 8 of the voxel's C dwords for hi and 8 further for lo (same bytes per plane as fp32)
        // Neighbouring lanes (cv even / odd: CV is even, so they are the same voxel) trade halves so that each issues ONE 8-byte store
        // per plane, as the fp32 form does: the even lane writes both lanes' hi dwords, the odd lane both lo dwords.
        const int cv = (int)(zc % CV);
        const bool odd = cv & 1;
        const long long dw = 2 * (t - cv) + ((cv >> 3) << 4) + (cv & 6) + (odd ? 8 : 0);
        u32x2w *Vp = reinterpret_cast<u32x2w *>(reinterpret_cast<unsigned *>(p.V) + dw);
        const long long pstride = p.vs / 2;      // plane stride in 8-byte units (plane sizes are multiples of 16 dwords)
#pragma unroll
        for (int j = 0; j < N; ++j) {
          unsigned hi, lo;
          wino_pair2(v[j], vscale, &hi, &lo, sat);
          const unsigned give = odd ? hi : lo, got = (unsigned)__shfl_xor((int)give, 1);
          u32x2w o;
          o.x = odd ? got : hi;      // even lane: (hi_even, hi_odd); odd lane: (lo_even, lo_odd)
          o.y = odd ? lo : got;
          Vp[(long long)(N * i + j) * pstride] = o;
        }
      } else {
#pragma unroll
        for (int j = 0; j < N; ++j) Vw[(long long)(N * i + j) * (p.vs / VW) + t] = v[j];
      }
    }
  }
  WINO_TL_END(p)
}

__device__ __forceinline__ float wino_finish(const WinoP &p, float acc, float sc, float sf, float r) {
  float v = acc * sc + sf;
  if (p.res_mode && !p.res_after_act) v += r;
  if (p.relu) v = v > 0.f ? v : 0.f;
  if (p.res_mode && p.res_after_act) v += r;
  return v * p.post_scale;
}
__device__ __forceinline__ float4 wino_finish_v(const WinoP &p, float4 a, float4 sc, float4 sf, float4 r) {
  return make_float4(wino_finish(p, a.x, sc.x, sf.x, r.x), wino_finish(p, a.y, sc.y, sf.y, r.y), wino_finish(p, a.z, sc.z, sf.z, r.z),
                     wino_finish(p, a.w, sc.w, sf.w, r.w));
}
__device__ __forceinline__ float wino_finish_v(const WinoP &p, float a, float sc, float sf, float r) { return wino_finish(p, a, sc, sf, r); }
__device__ __forceinline__ float2 wino_finish_v(const WinoP &p, float2 a, float2 sc, float2 sf, float2 r) {
  return make_float2(wino_finish(p, a.x, sc.x, sf.x, r.x), wino_finish(p, a.y, sc.y, sf.y, r.y));
}

// out = epilogue(At M A).  One thread: VW channels of one (b, tx, ty, zo) -> m x m outputs.
template <int MT, int VW>
__global__ __launch_bounds__(256) void wino_output_kernel(const WinoP p) {
  WINO_TL_BEGIN
  typedef typename VecT<VW>::T V;
  constexpr int N = MT + 2;
  const int CV = p.Co / VW;
  const long long per_tile = (long long)p.Zo * CV;
  const long long total = (long long)p.B * p.TX * p.TY * per_tile;
  const V *Mw = reinterpret_cast<const V *>(p.Mw);
  const V *res = reinterpret_cast<const V *>(p.res);
  V *out = reinterpret_cast<V *>(p.out);
  const float mscale = wino_mscale(p.hdr);
  float omax = 0.f;
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
    const long long zc = t % per_tile;
    const int cv = (int)(zc % CV);
    long long q = t / per_tile;
    const int ty = (int)(q % p.TY);
    q /= p.TY;
    const int tx = (int)(q % p.TX);
    const int b = (int)(q / p.TX);
    V r[MT][N];   // r[a][j] = (At M)[a][j]
#pragma unroll
    for (int j = 0; j < N; ++j) {
      V m[N];
#pragma unroll
      for (int i = 0; i < N; ++i) m[i] = Mw[(long long)(N * i + j) * (p.ms / VW) + t];
      V y[MT];
      Wino1D<MT, V>::out(m, y);
#pragma unroll
      for (int a = 0; a < MT; ++a) r[a][j] = y[a];
    }
    const V sc = mscale * (p.scale ? reinterpret_cast<const V *>(p.scale)[cv] : vone((V *)nullptr));   // mscale 1 (fp32 operands): exact
    const V sf = p.shift ? reinterpret_cast<const V *>(p.shift)[cv] : vzero((V *)nullptr);
#pragma unroll
    for (int a = 0; a < MT; ++a) {
      const int x = MT * tx + a;
      if (x >= p.Xo) continue;
      V yy[MT];
      Wino1D<MT, V>::out(r[a], yy);
#pragma unroll
      for (int e = 0; e < MT; ++e) {
        const int y = MT * ty + e;
        if (y >= p.Yo) continue;
        const long long o = (((long long)b * p.Xo + x) * p.Yo + y) * per_tile + zc;
        V rr = vzero((V *)nullptr);
        if (p.res_mode) rr = res[o];
        const V val = wino_finish_v(p, yy[e], sc, sf, rr);
        out[o] = val;
        if (p.pmax) omax = fmaxf(omax, vabsmax(val));
      }
    }
  }
  if (p.pmax) wino_block_max_store(omax, p.pmax);
  WINO_TL_END(p)
}

// Coefficients of At for the column-accumulate form of the output transform (wino_output_buf_kernel below).
template <int MT> struct WinoAt;
template <> struct WinoAt<6> {
  static __device__ __forceinline__ float c(int e, int j) {
    constexpr float A[6][8] = {{1, 1, 1, 1, 1, 1, 1, 0},          {0, 1, -1, 2, -2, 0.5f, -0.5f, 0},
                               {0, 1, 1, 4, 4, 0.25f, 0.25f, 0},   {0, 1, -1, 8, -8, 0.125f, -0.125f, 0},
                               {0, 1, 1, 16, 16, 0.0625f, 0.0625f, 0}, {0, 1, -1, 32, -32, 0.03125f, -0.03125f, 1}};
    return A[e][j];
  }
};
template <> struct WinoAt<4> {
  static __device__ __forceinline__ float c(int e, int j) {
    constexpr float A[4][6] = {{1, 1, 1, 1, 1, 0}, {0, 1, -1, 2, -2, 0}, {0, 1, 1, 4, 4, 0}, {0, 1, -1, 8, -8, 1}};
    return A[e][j];
  }
};
__device__ __forceinline__ float vfma(float s, float a, float c) { return fmaf(s, a, c); }
__device__ __forceinline__ float2 vfma(float s, float2 a, float2 c) { return make_float2(fmaf(s, a.x, c.x), fmaf(s, a.y, c.y)); }
__device__ __forceinline__ float4 vfma(float s, float4 a, float4 c) {
  return make_float4(fmaf(s, a.x, c.x), fmaf(s, a.y, c.y), fmaf(s, a.z, c.z), fmaf(s, a.w, c.w));
}

// Output transform, default form for F(6x6, 3x3): buffer addressing + column accumulation.
//  * ONE 32-bit lane offset per thread; the plane / output-position offsets are wave-uniform SGPR operands of buffer_load /
//    buffer_store (the pointer form above spends two VGPRs and a 64-bit add per access on 64 + 36 + 36 addresses);
//  * the columns of M are consumed one at a time -- 8 loads, the 1-D transform down the column, its contribution At[e][j] * r[a]
//    added to the m x m accumulators -- with the next column's loads issued one step ahead and fenced (sched_barrier), so a thread
//    holds 36 accumulators + two columns instead of the whole 8 x 8 tile: 164 VGPRs / 3 waves per SIMD with 2 channels per lane
//    (the whole-tile form: 211 / 2) -- and the residual loads of one output row overlap the stores of the previous one;
//  * the epilogue is branch-free (border positions get an out-of-range lane offset: the buffer hardware drops those stores and
//    returns 0 for those loads), which keeps the loop body one basic block -- otherwise LLVM sinks the column arithmetic into the
//    conditional store blocks and every load is hoisted to the top again.
// Round 4 (RPF, profiles/r04_wino_ab.log): with the residual requested before the column loop 0.537 -> 0.454 ms (4.67 -> 5.52 TB/s).
// Measured on the KITTI neck at batch 4 (tools/wino_ab.py, profiles/r03_wino_ab.log): with residual 0.63 -> 0.52 ms per launch
// (3.95 -> 4.8 TB/s), without 0.37 -> 0.35.  Needs the (m+2)^2 planes of M below 4 GiB and the output tensor below 2 GiB (the
// launcher falls back to the pointer kernel otherwise).  The sums are accumulated in another order than in the whole-tile form:
// same fp32 arithmetic, results differ by rounding only.
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
template <int VW> struct BufIO;
template <> struct BufIO<2> {
  static __device__ __forceinline__ float2 load(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0);
    return make_float2(__uint_as_float(v.x), __uint_as_float(v.y));
  }
  static __device__ __forceinline__ void store(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, float2 f) {
    u32x2 v;
    v.x = __float_as_uint(f.x);
    v.y = __float_as_uint(f.y);
    __builtin_amdgcn_raw_buffer_store_b64(v, r, voff, soff, 0);
  }
};
template <> struct BufIO<1> {
  static __device__ __forceinline__ float load(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
  }
  static __device__ __forceinline__ void store(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, float f) {
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(f), r, voff, soff, 0);
  }
};

// acc[e] += At[e][J] * y for one column J of F(6,3)'s At (coefficients 0 / +-1 become nothing / plain adds).
template <int J, typename V>
__device__ __forceinline__ void wino6_acc_column(V (&acc)[6], const V y) {
#pragma unroll
  for (int e = 0; e < 6; ++e) {
    const float cf = WinoAt<6>::c(e, J);
    if (cf == 1.0f) acc[e] = acc[e] + y;
    else if (cf == -1.0f) acc[e] = acc[e] - y;
    else if (cf != 0.0f) acc[e] = vfma(cf, y, acc[e]);
  }
}

// The column loop as a compile-time recursion (J is a template constant, so every array index is static): consume column J
// (already loaded), issue column J + 1, fence, transform + accumulate, fence.
template <int J, int VW> struct Wino6Columns {
  typedef typename VecT<VW>::T V;
  static __device__ __forceinline__ void run(V (&acc)[6][6], const V (&col)[8], const __amdgpu_buffer_rsrc_t rm, const unsigned vo,
                                             const unsigned ps) {
    V nxt[8];
    if (J + 1 < 8) {
#pragma unroll
      for (int i = 0; i < 8; ++i) nxt[i] = BufIO<VW>::load(rm, vo, (unsigned)(8 * i + J + 1) * ps);
    }
    __builtin_amdgcn_sched_barrier(0);
    V y[6];
    Wino1D<6, V>::out(col, y);
#pragma unroll
    for (int a = 0; a < 6; ++a) wino6_acc_column<J>(acc[a], y[a]);
    __builtin_amdgcn_sched_barrier(0);
    Wino6Columns<J + 1, VW>::run(acc, nxt, rm, vo, ps);
  }
};
template <int VW> struct Wino6Columns<8, VW> {
  typedef typename VecT<VW>::T V;
  static __device__ __forceinline__ void run(V (&)[6][6], const V (&)[8], const __amdgpu_buffer_rsrc_t, const unsigned, const unsigned) {}
};

// RPF 1 (residual layers): the 36 residual values of the tile are requested BEFORE the column loop instead of row by row in the epilogue.
// A wave of the row-by-row form goes through 8 + 6 dependent memory round trips (eight columns of M, then six rows of residual loads ->
// stores); here the epilogue has no load left to wait for, at the price of 36 * VW registers held through the column loop (two waves per
// SIMD instead of three at VW = 2).
template <int VW, int WPE, int RPF = 0>
__global__ __launch_bounds__(256, WPE) void wino_output_buf_kernel(const WinoP p, const unsigned m_bytes, const unsigned out_bytes) {
  WINO_TL_BEGIN
  typedef typename VecT<VW>::T V;
  constexpr int MT = 6, N = 8, EB = VW * 4;     // bytes per lane item
  const int CV = p.Co / VW;
  const long long per_tile = (long long)p.Zo * CV;
  const long long total = (long long)p.B * p.TX * p.TY * per_tile;
  const __amdgpu_buffer_rsrc_t rm = __builtin_amdgcn_make_buffer_rsrc((void *)p.Mw, 0, m_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc((void *)p.out, 0, out_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc((void *)(p.res ? p.res : p.out), 0, p.res ? out_bytes : 0u, 0x00020000);
  const unsigned ps = (unsigned)(p.ms * 4);                 // plane stride in bytes
  const float mscale = wino_mscale(p.hdr);
  float omax = 0.f;
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
    const long long zc = t % per_tile;
    const int cv = (int)(zc % CV);
    long long q = t / per_tile;
    const int ty = (int)(q % p.TY);
    q /= p.TY;
    const int tx = (int)(q % p.TX);
    const int b = (int)(q / p.TX);
    const unsigned vo = (unsigned)(t * EB);
    const V sc = mscale * (p.scale ? reinterpret_cast<const V *>(p.scale)[cv] : vone((V *)nullptr));     // (uniform branches BEFORE the column loop)
    const V sf = p.shift ? reinterpret_cast<const V *>(p.shift)[cv] : vzero((V *)nullptr);
    const int xb = MT * tx, yb = MT * ty;
    const unsigned o00 = (unsigned)(((((long long)b * p.Xo + xb) * p.Yo + yb) * per_tile + zc) * EB);
    const unsigned row = (unsigned)(p.Yo * per_tile * EB), colb = (unsigned)(per_tile * EB);
    V rres[RPF ? MT : 1][RPF ? MT : 1];
    if constexpr (RPF) {
#pragma unroll
      for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int e = 0; e < MT; ++e)
          rres[a][e] = BufIO<VW>::load(rr, (xb + a < p.Xo && yb + e < p.Yo) ? o00 : 0x80000000u, (unsigned)a * row + (unsigned)e * colb);
      __builtin_amdgcn_sched_barrier(0);
    }
    V acc[MT][MT];
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
      for (int e = 0; e < MT; ++e) acc[a][e] = vzero((V *)nullptr);
    V col[N];
#pragma unroll
    for (int i = 0; i < N; ++i) col[i] = BufIO<VW>::load(rm, vo, (unsigned)(N * i) * ps);
    Wino6Columns<0, VW>::run(acc, col, rm, vo, ps);
#pragma unroll
    for (int a = 0; a < MT; ++a) {
#pragma unroll
      for (int e = 0; e < MT; ++e) {
        const unsigned vo_ae = (xb + a < p.Xo && yb + e < p.Yo) ? o00 : 0x80000000u;
        const unsigned so = (unsigned)a * row + (unsigned)e * colb;
        V rv;
        if constexpr (RPF) rv = rres[a][e];
        else rv = BufIO<VW>::load(rr, vo_ae, so);           // no residual: rr has zero records -> 0
        const V val = wino_finish_v(p, acc[a][e], sc, sf, rv);
        BufIO<VW>::store(ro, vo_ae, so, val);
        omax = fmaxf(omax, (vo_ae >> 31) ? 0.f : vabsmax(val));     // (unconditional: the epilogue stays one basic block) border
      }                                                             // positions are not part of the tensor
      __builtin_amdgcn_sched_barrier(0);       // one output row's residual loads in flight at a time (else all 36 are hoisted)
    }
  }
  if (p.pmax) wino_block_max_store(omax, p.pmax);
  WINO_TL_END(p)
}

// U = G g Gt.  wgt is layout 0 [Co,3,3,KW,Ci]; U is [n*n][Co][K] with the K order of `kmode` (0: k = kz*Ci + ci;
// 1: k = (ci/32)*KW*32 + kz*32 + ci%32), i.e. each xi holds a packed 1x1xKW filter bank.
template <int MT>
__global__ __launch_bounds__(256) void wino_weight_kernel(const float *__restrict__ w, float *__restrict__ U, int Co, int KW, int Ci,
                                                          int kmode, unsigned *amax) {
  constexpr int N = MT + 2;
  const long long total = (long long)Co * KW * Ci;
  long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= total) t = total - 1;   // surplus threads of the last block repeat its last item (same values stored twice): the wave
                                   // reduction below needs every lane
  const int ci = (int)(t % Ci);
  const int kz = (int)((t / Ci) % KW);
  const int co = (int)(t / ((long long)Ci * KW));
  float h[N][3];   // h[i][e] = (G g)[i][e]
#pragma unroll
  for (int e = 0; e < 3; ++e) {
    float g[3], c[N];
#pragma unroll
    for (int a = 0; a < 3; ++a) g[a] = w[((((long long)co * 3 + a) * 3 + e) * KW + kz) * Ci + ci];
    Wino1D<MT, float>::wgt(g, c);
#pragma unroll
    for (int i = 0; i < N; ++i) h[i][e] = c[i];
  }
  const long long K = (long long)KW * Ci;
  float umax = 0.f;
  const long long k = kmode == 1 ? ((long long)(ci >> 5) * KW + kz) * 32 + (ci & 31) : (long long)kz * Ci + ci;
#pragma unroll
  for (int i = 0; i < N; ++i) {
    float u[N];
    Wino1D<MT, float>::wgt(h[i], u);
#pragma unroll
    for (int j = 0; j < N; ++j) {
      U[((long long)(N * i + j) * Co + co) * K + k] = u[j];
      umax = fmaxf(umax, fabsf(u[j]));
    }
  }
  if (amax) {   // pair operands: max |U| for the filter scale (wino_pair_inplace_kernel converts the planes afterwards)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) umax = fmaxf(umax, __shfl_xor(umax, o));
    if ((threadIdx.x & 63) == 0) atomicMax(amax, __float_as_uint(umax));
  }
}

struct WinoDims {
  int Xo, Yo, Zo, TX, TY, n2;
  int64_t v_elems, m_elems;   // elements of one xi plane of V / M
  int64_t v_stride, m_stride; // distance between consecutive xi planes in floats (= the plane size; padding the stride to break
                              // a possible channel-interleave alignment of the 64 concurrent plane streams measured nothing: r03_wino_ab.log)
};

// A/B knob (per calling thread; ivx_conv_winograd_set_variant): the F(6x6,3x3) output transform kernel.
//  -1 default rule (2 with a residual, 1 without) | 0 whole tile, 2 channels per lane (the round-2 kernel) | 1 whole tile, 1 channel
//  per lane | 2 buffer addressing + column accumulation, 2 channels per lane | 3 the same, 1 channel per lane | 4 / 5 = 2 / 3 with the
//  residual prefetched before the column loop (residual layers only)
thread_local int g_wino_out_variant = -1;
thread_local int g_wino_in_variant = -1;     // input transform of F(6x6,3x3): -1 default (0) | 0: 2 channels per lane | 1: 1 channel per lane

int wino_dims(const ivx_conv_desc *d, int tile, WinoDims *w, const char *who) {
  IVX_REQUIRE(d, "%s: null descriptor", who);
  IVX_REQUIRE(tile == 2 || tile == 4 || tile == 6, "%s: tile must be 2, 4 or 6 (F(m x m, 3x3)), got %d", who, tile);
  IVX_REQUIRE(d->KD == 3 && d->KH == 3 && d->sd == 1 && d->sh == 1, "%s: needs a 3x3 kernel with stride 1 on the first two axes", who);
  IVX_REQUIRE(d->KW >= 1 && d->KW <= 8 && d->sw >= 1 && d->pd >= 0 && d->ph >= 0 && d->pw >= 0, "%s: bad z kernel / stride / padding", who);
  IVX_REQUIRE(d->B > 0 && d->D > 0 && d->H > 0 && d->W > 0, "%s: non-positive dims", who);
  IVX_REQUIRE(d->Cin > 0 && d->Cin % 4 == 0 && d->Cout > 0 && d->Cout % 4 == 0, "%s: Cin and Cout must be multiples of 4", who);
  IVX_REQUIRE(d->in_dtype == IVX_F32 && d->out_dtype == IVX_F32, "%s: fp32 only", who);
  IVX_REQUIRE(d->wino_operands == IVX_F32 || (d->wino_operands == IVX_F16_PAIR && tile >= 4 && d->Cin % 16 == 0 &&
                                               (d->wgt_layout == 0 || d->Cin % 32 == 0)),
              "%s: wino_operands is IVX_F32 or IVX_F16_PAIR (tile 4 / 6, Cin %% 16 == 0; wgt_layout 1: Cin %% 32 == 0)", who);
  IVX_REQUIRE(d->out_mode == 0 && (d->res_mode == 0 || d->res_mode == 1), "%s: out_mode 0 and res_mode 0/1 only", who);
  IVX_REQUIRE(d->wgt_layout == 0 || (d->wgt_layout == 1 && d->Cin % 32 == 0), "%s: wgt_layout 1 needs Cin %% 32 == 0", who);
  int32_t Xo, Yo, Zo;
  if (ivx_conv_out_dims(d, &Xo, &Yo, &Zo) != IVX_OK) return IVX_ERR_INVALID_ARG;
  w->Xo = Xo; w->Yo = Yo; w->Zo = Zo;
  w->TX = (Xo + tile - 1) / tile; w->TY = (Yo + tile - 1) / tile;
  w->n2 = (tile + 2) * (tile + 2);
  w->v_elems = (int64_t)d->B * w->TX * w->TY * d->W * d->Cin;
  w->m_elems = (int64_t)d->B * w->TX * w->TY * Zo * d->Cout;
  w->v_stride = w->v_elems;
  w->m_stride = w->m_elems;
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
  const int64_t b = (items + 255) / 256;
  return (unsigned)(b < 1 ? 1 : (b > (1 << 20) ? (1 << 20) : b));
}

}  // namespace

extern "C" int ivx_conv_winograd_set_variant(int32_t output_variant, int32_t input_variant) {
  g_wino_out_variant = output_variant;
  g_wino_in_variant = input_variant;
  return IVX_OK;
}

extern "C" int ivx_conv_winograd_supported(const ivx_conv_desc *d, int32_t tile) {
  WinoDims w;
  if (wino_dims(d, tile, &w, "ivx_conv_winograd_supported") != IVX_OK) return 0;
  // one xi plane is one group of the grouped launch: 31-bit buffer offsets
  if (w.v_stride * 4 >= (1LL << 31) || w.m_stride >= (1LL << 31) - 512 * (int64_t)d->Cout) return 0;
  if ((int64_t)d->Cout * d->KW * d->Cin * 4 >= (1LL << 31)) return 0;
  return 1;
}

extern "C" int64_t ivx_conv_winograd_weight_elems(const ivx_conv_desc *d, int32_t tile) {
  WinoDims w;
  if (wino_dims(d, tile, &w, "ivx_conv_winograd_weight_elems") != IVX_OK) return -1;
  // pair operands: one more plane whose first two words hold {bits of max |U|, the filter scale}
  return (int64_t)(w.n2 + (d->wino_operands == IVX_F16_PAIR ? 1 : 0)) * d->Cout * d->KW * d->Cin;
}

extern "C" int ivx_conv_winograd_weights(const ivx_conv_desc *d, int32_t tile, const float *wgt, float *u, ivx_stream_t stream) {
  WinoDims w;
  int rc = wino_dims(d, tile, &w, "ivx_conv_winograd_weights");
  if (rc != IVX_OK) return rc;
  IVX_REQUIRE(wgt && u, "ivx_conv_winograd_weights: null argument");
  const int64_t total = (int64_t)d->Cout * d->KW * d->Cin;
  const dim3 grid((unsigned)((total + 255) / 256));
  const bool pair = d->wino_operands == IVX_F16_PAIR;
  unsigned *hdr = pair ? reinterpret_cast<unsigned *>(u + (int64_t)w.n2 * total) : nullptr;
  if (pair && hipMemsetAsync(hdr, 0, 8, (hipStream_t)stream) != hipSuccess) {
    ivx_set_error("ivx_conv_winograd_weights: hipMemsetAsync failed");
    return IVX_ERR_HIP;
  }
  if (tile == 2)
    hipLaunchKernelGGL(wino_weight_kernel<2>, grid, dim3(256), 0, (hipStream_t)stream, wgt, u, d->Cout, d->KW, d->Cin, d->wgt_layout, hdr);
  else if (tile == 4)
    hipLaunchKernelGGL(wino_weight_kernel<4>, grid, dim3(256), 0, (hipStream_t)stream, wgt, u, d->Cout, d->KW, d->Cin, d->wgt_layout, hdr);
  else
    hipLaunchKernelGGL(wino_weight_kernel<6>, grid, dim3(256), 0, (hipStream_t)stream, wgt, u, d->Cout, d->KW, d->Cin, d->wgt_layout, hdr);
  if (pair) {
    const size_t n16 = (size_t)w.n2 * total / 16;
    hipLaunchKernelGGL(wino_pair_inplace_kernel, dim3((unsigned)((n16 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, u, n16, hdr);
  }
  IVX_CHECK_LAUNCH("ivx_conv_winograd_weights");
  return IVX_OK;
}

extern "C" int64_t ivx_conv_winograd_workspace_bytes(const ivx_conv_desc *d, int32_t tile) {
  WinoDims w;
  if (wino_dims(d, tile, &w, "ivx_conv_winograd_workspace_bytes") != IVX_OK) return -1;
  return ivx_align_up(w.n2 * w.v_stride * 4, 256) + ivx_align_up(w.n2 * w.m_stride * 4, 256) + 256;   // + the pair-operand header
}

namespace {
#ifdef IVX_CONV_TIMELINE
unsigned long long *g_wino_timeline = nullptr;
#endif
int wino_setup(const ivx_conv_desc *d, int tile, const void *in, const float *scale, const float *shift, const void *res, void *out,
               void *workspace, int64_t workspace_bytes, WinoDims *w, WinoP *p, const char *who) {
  int rc = wino_dims(d, tile, w, who);
  if (rc != IVX_OK) return rc;
  IVX_REQUIRE(workspace, "%s: null workspace", who);
  if (!ivx_conv_winograd_supported(d, tile)) {
    ivx_set_error("%s: one transformed plane must stay below 2 GiB (use ivx_conv_fwd)", who);
    return IVX_ERR_UNSUPPORTED;
  }
  const int64_t need = ivx_conv_winograd_workspace_bytes(d, tile);
  if (workspace_bytes < need) {
    ivx_set_error("%s: workspace too small (%lld < %lld); size it with ivx_conv_winograd_workspace_bytes", who,
                  (long long)workspace_bytes, (long long)need);
    return IVX_ERR_WORKSPACE;
  }
  p->in = (const float *)in; p->scale = scale; p->shift = shift; p->res = d->res_mode ? (const float *)res : nullptr;
  p->out = (float *)out;
  p->V = (float *)workspace;
  p->Mw = (float *)((char *)workspace + ivx_align_up(w->n2 * w->v_stride * 4, 256));
  p->vs = w->v_stride; p->ms = w->m_stride;
  p->B = d->B; p->X = d->D; p->Y = d->H; p->Z = d->W; p->C = d->Cin;
  p->Xo = w->Xo; p->Yo = w->Yo; p->Zo = w->Zo; p->Co = d->Cout;
  p->TX = w->TX; p->TY = w->TY; p->px = d->pd; p->py = d->ph;
  p->relu = d->relu; p->res_mode = d->res_mode; p->res_after_act = d->res_after_act;
  p->post_scale = d->post_scale == 0.f ? 1.0f : d->post_scale;
  p->pmax = nullptr;
#ifdef IVX_CONV_TIMELINE
  p->tl = g_wino_timeline;
#endif
  p->hdr = d->wino_operands == IVX_F16_PAIR
               ? (const unsigned *)((char *)workspace + ivx_align_up(w->n2 * w->v_stride * 4, 256) + ivx_align_up(w->n2 * w->m_stride * 4, 256))
               : nullptr;
  return IVX_OK;
}
}  // namespace

// The three stages are separate entry points so that a caller can time them (bench.py); ivx_conv_winograd_fwd runs all.
static int wino_input_impl(const ivx_conv_desc *d, int32_t tile, const void *in, void *workspace, int64_t workspace_bytes,
                           const float *partials, int32_t n_partials, ivx_stream_t stream) {
  WinoDims w;
  WinoP p;
  float dummy;
  IVX_REQUIRE(in, "ivx_conv_winograd_input: null argument");
  int rc = wino_setup(d, tile, in, nullptr, nullptr, &dummy, &dummy, workspace, workspace_bytes, &w, &p, "ivx_conv_winograd_input");
  if (rc != IVX_OK) return rc;
  if (d->wino_operands == IVX_F16_PAIR) {
    // max |input| -> header word 0 (the scale of V is derived from it on the device: no host round trip)
    static const bool one_wg = !(getenv("IVX_WINO_AMAX_ONE_WG") && atoi(getenv("IVX_WINO_AMAX_ONE_WG")) == 0);      // =0: the round-5 pair of launches (A/B)
    const bool write_form = partials && n_partials <= 65536 && one_wg;
    if (!write_form && hipMemsetAsync((void *)p.hdr, 0, 4, (hipStream_t)stream) != hipSuccess) {
      ivx_set_error("ivx_conv_winograd_input: hipMemsetAsync failed");
      return IVX_ERR_HIP;
    }
    if (write_form) {   // the producer's per-workgroup maxima, reduced by one workgroup that writes the word
      hipLaunchKernelGGL(wino_amax_partials_write_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, partials, n_partials, (unsigned *)p.hdr);
    } else if (partials) {   // the producer left per-workgroup maxima of this tensor (ivx_conv_winograd_output_amax): reduce those (1 - 2 MB at KITTI size)
      const int pb = (n_partials + 255) / 256;
      hipLaunchKernelGGL(wino_amax_partials_kernel, dim3((unsigned)(pb > 64 ? 64 : pb)), dim3(256), 0, (hipStream_t)stream, partials, n_partials,
                         (unsigned *)p.hdr);
    } else {
      const size_t n4 = (size_t)d->B * d->D * d->H * d->W * d->Cin / 4;
      const size_t ab = (n4 + 1023) / 1024;
      hipLaunchKernelGGL(wino_amax_kernel, dim3((unsigned)(ab > 1536 ? 1536 : (ab < 1 ? 1 : ab))), dim3(256), 0, (hipStream_t)stream,
                         (const wf32x4 *)in, n4, (unsigned *)p.hdr);
    }
    if (tile == 4)
      hipLaunchKernelGGL((wino_input_kernel<4, 2, 1>), dim3(wino_blocks(w.v_elems / 2)), dim3(256), 0, (hipStream_t)stream, p);
    else
      hipLaunchKernelGGL((wino_input_kernel<6, 2, 1>), dim3(wino_blocks(w.v_elems / 2)), dim3(256), 0, (hipStream_t)stream, p);
  } else if (tile == 2)
    hipLaunchKernelGGL((wino_input_kernel<2, 4>), dim3(wino_blocks(w.v_elems / 4)), dim3(256), 0, (hipStream_t)stream, p);
  else if (tile == 4)
    hipLaunchKernelGGL((wino_input_kernel<4, 2>), dim3(wino_blocks(w.v_elems / 2)), dim3(256), 0, (hipStream_t)stream, p);
  else if (g_wino_in_variant == 1)
    hipLaunchKernelGGL((wino_input_kernel<6, 1>), dim3(wino_blocks(w.v_elems)), dim3(256), 0, (hipStream_t)stream, p);
  else
    hipLaunchKernelGGL((wino_input_kernel<6, 2>), dim3(wino_blocks(w.v_elems / 2)), dim3(256), 0, (hipStream_t)stream, p);
  IVX_CHECK_LAUNCH("ivx_conv_winograd_input");
  return IVX_OK;
}

extern "C" int ivx_conv_winograd_input(const ivx_conv_desc *d, int32_t tile, const void *in, void *workspace, int64_t workspace_bytes,
                                       ivx_stream_t stream) {
  return wino_input_impl(d, tile, in, workspace, workspace_bytes, nullptr, 0, stream);
}

extern "C" int ivx_conv_winograd_input_amax(const ivx_conv_desc *d, int32_t tile, const void *in, void *workspace, int64_t workspace_bytes,
                                            const float *partials, int32_t n_partials, ivx_stream_t stream) {
  IVX_REQUIRE(!partials || n_partials > 0, "ivx_conv_winograd_input_amax: partials without a count");
  return wino_input_impl(d, tile, in, workspace, workspace_bytes, partials, n_partials, stream);
}

extern "C" int ivx_conv_winograd_gemm(const ivx_conv_desc *d, int32_t tile, const float *u, void *workspace, int64_t workspace_bytes,
                                      ivx_stream_t stream) {
  WinoDims w;
  WinoP p;
  float dummy;
  IVX_REQUIRE(u, "ivx_conv_winograd_gemm: null argument");
  int rc = wino_setup(d, tile, &dummy, nullptr, nullptr, &dummy, &dummy, workspace, workspace_bytes, &w, &p, "ivx_conv_winograd_gemm");
  if (rc != IVX_OK) return rc;
  ivx_conv_desc g = wino_group_desc(d, w);
  const long long el = d->wino_operands == IVX_F16_PAIR ? 2 : 1;   // operand strides count stored elements (two fp16 per value)
  g.in_dtype = d->wino_operands;
  // the filter scale travels with the filters; the output transform reads it from the workspace header: workgroup 0 of the GEMM launch copies
  // the word (a separate 4-byte hipMemcpyAsync was one more launch per layer)
  const unsigned *us = p.hdr ? reinterpret_cast<const unsigned *>(u + (int64_t)w.n2 * d->Cout * d->KW * d->Cin + 1) : nullptr;
  rc = ivx_conv_grouped_launch(&g, w.n2, p.V, el * w.v_stride, u, el * d->Cout * d->KW * d->Cin, p.Mw, w.m_stride, (hipStream_t)stream, us,
                               p.hdr ? const_cast<unsigned *>(p.hdr) + 1 : nullptr);
  if (rc != IVX_OK) return rc;
  IVX_CHECK_LAUNCH("ivx_conv_winograd_gemm");
  return IVX_OK;
}

// ---- F(4x4,3x3) on fp16-pair operands with the output transform fused into the GEMM launch (conv_igemm.hip conv_wino_fold4_kernel): the
// second and third stage of ivx_conv_winograd_fwd in one kernel, M stays on chip.  Same workspace as the three-stage form (V + header; the
// M region is not touched), so ivx_conv_winograd_input[_amax] is its first stage unchanged.
extern "C" int ivx_conv_winograd_fused_supported(const ivx_conv_desc *d, int32_t tile) {
  WinoDims w;
  if (tile != 4 || !d || d->wino_operands != IVX_F16_PAIR || d->KW != 3 || d->sw != 1 || d->pw != 1 || d->wgt_layout != 1 || d->Cin % 32 || d->Cout % 4)
    return 0;
  if (!ivx_conv_winograd_supported(d, tile) || wino_dims(d, tile, &w, "ivx_conv_winograd_fused_supported") != IVX_OK) return 0;
  if ((int64_t)w.n2 * w.v_stride * 4 >= (1LL << 31) || (int64_t)w.n2 * d->Cout * d->KW * d->Cin * 4 >= (1LL << 31)) return 0;   // one buffer resource over all planes
  return 1;
}

extern "C" int32_t ivx_conv_winograd_fused_blocks(const ivx_conv_desc *d, int32_t tile) {
  WinoDims w;
  if (!ivx_conv_winograd_fused_supported(d, tile) || wino_dims(d, tile, &w, "ivx_conv_winograd_fused_blocks") != IVX_OK) return -1;
  return ivx_conv_fold4_blocks((long long)d->B * w.TX * w.TY * w.Zo, d->Cout);
}

extern "C" int ivx_conv_winograd_gemm_output_amax(const ivx_conv_desc *d, int32_t tile, const float *u, const float *scale, const float *shift,
                                                  const void *res, void *out, void *workspace, int64_t workspace_bytes, float *partials,
                                                  ivx_stream_t stream) {
  WinoDims w;
  WinoP p;
  float dummy;
  IVX_REQUIRE(u && out, "ivx_conv_winograd_gemm_output: null argument");
  IVX_REQUIRE(!d || d->res_mode == 0 || res, "ivx_conv_winograd_gemm_output: res_mode set but res is NULL");
  if (!ivx_conv_winograd_fused_supported(d, tile)) {
    ivx_set_error("ivx_conv_winograd_gemm_output: the fused form takes F(4x4,3x3) on IVX_F16_PAIR operands, a 3-tap z kernel with stride 1 and padding 1, "
                  "wgt_layout 1, Cin %% 32 == 0, and all 36 planes of V below 2 GiB (ivx_conv_winograd_fused_supported)");
    return IVX_ERR_UNSUPPORTED;
  }
  int rc = wino_setup(d, tile, &dummy, scale, shift, res, out, workspace, workspace_bytes, &w, &p, "ivx_conv_winograd_gemm_output");
  if (rc != IVX_OK) return rc;
  ivx_conv_desc g = wino_group_desc(d, w);
  g.in_dtype = d->wino_operands;
  IvxWinoFold f;
  f.out = p.out; f.res = p.res; f.scale = p.scale; f.shift = p.shift;
  f.hdr_v = p.hdr;
  f.uscale = u + (int64_t)w.n2 * d->Cout * d->KW * d->Cin + 1;        // the filter scale travels with the filters (ivx_conv_winograd_weights)
  f.pmax = partials;
  f.B = d->B; f.TX = w.TX; f.TY = w.TY; f.Z = w.Zo; f.Xo = w.Xo; f.Yo = w.Yo; f.Co = d->Cout;
  f.relu = p.relu; f.res_mode = p.res_mode; f.res_after_act = p.res_after_act; f.post_scale = p.post_scale;
  // IVX_FOLD4_WAVES=2: the two-waves-per-SIMD form on v_mfma_f32_16x16x32_f16 (csrc/fold4w.hip, round 6: measured slower, profiles/r06_fused_two_wave.md); default 1: the round-5 one-wave kernel
  // (conv_wino_fold4_kernel) -- same tile geometry, same partial-maximum entries
  static const int fold_waves = getenv("IVX_FOLD4_WAVES") ? atoi(getenv("IVX_FOLD4_WAVES")) : 1;
  if (fold_waves == 2 && d->Cout % 8 == 0)
    rc = ivx_conv_launch_fold4w(reinterpret_cast<const _Float16 *>(p.V), 2 * w.v_stride, reinterpret_cast<const _Float16 *>(u), 2LL * d->Cout * d->KW * d->Cin,
                                d->B * w.TX * w.TY * w.Zo, 2 * d->Cin, d->Cout, w.Zo, f, (hipStream_t)stream);
  else
    rc = ivx_conv_grouped_fold4(&g, w.n2, p.V, 2 * w.v_stride, u, 2LL * d->Cout * d->KW * d->Cin, &f, (hipStream_t)stream);
  if (rc != IVX_OK) return rc;
  IVX_CHECK_LAUNCH("ivx_conv_winograd_gemm_output");
  return IVX_OK;
}

namespace {
// workgroups of the output transform's launch (= entries of the partial-maximum array of ivx_conv_winograd_output_amax)
// The F(6x6,3x3) kernel variant of a launch.  A launch that leaves per-workgroup maxima (`with_partials`) always takes the library's rule,
// whatever the A/B knob of the calling thread says: the caller sized its partials array with ivx_conv_winograd_output_blocks -- possibly
// at another time and on another thread -- and the number of workgroups must be the one that call returned.
int wino_output_variant(const ivx_conv_desc *d, const WinoDims &w, bool with_partials) {
  const bool buf_ok = (int64_t)w.n2 * w.m_stride * 4 < (1LL << 32) && (int64_t)d->B * w.Xo * w.Yo * w.Zo * d->Cout * 4 < (1LL << 31);
  int v = (g_wino_out_variant >= 0 && !with_partials) ? g_wino_out_variant : (d->res_mode ? 4 : 1);      // (round 4: 4 instead of 2, 0.54 -> 0.45 ms per launch)
  if (v >= 2 && !buf_ok) v = 0;
  if (v >= 4 && !d->res_mode) v = v == 4 ? 2 : 3;      // the prefetching forms are the residual layers' 
  return v;
}
unsigned wino_output_grid(const ivx_conv_desc *d, int tile, const WinoDims &w, bool with_partials) {
  if (tile == 2) return wino_blocks(w.m_elems / 4);
  if (tile == 4) return wino_blocks(w.m_elems / 2);
  const int v = wino_output_variant(d, w, with_partials);
  return (v == 2 || v == 0 || v == 4) ? wino_blocks(w.m_elems / 2) : wino_blocks(w.m_elems);
}
}  // namespace

extern "C" int32_t ivx_conv_winograd_output_blocks(const ivx_conv_desc *d, int32_t tile) {
  WinoDims w;
  if (wino_dims(d, tile, &w, "ivx_conv_winograd_output_blocks") != IVX_OK) return -1;
  return (int32_t)wino_output_grid(d, tile, w, true);
}

static int wino_output_impl(const ivx_conv_desc *d, int32_t tile, const float *scale, const float *shift, const void *res, void *out,
                            void *workspace, int64_t workspace_bytes, float *partials, ivx_stream_t stream) {
  WinoDims w;
  WinoP p;
  float dummy;
  IVX_REQUIRE(out, "ivx_conv_winograd_output: null argument");
  IVX_REQUIRE(!d || d->res_mode == 0 || res, "ivx_conv_winograd_output: res_mode set but res is NULL");
  int rc = wino_setup(d, tile, &dummy, scale, shift, res, out, workspace, workspace_bytes, &w, &p, "ivx_conv_winograd_output");
  if (rc != IVX_OK) return rc;
  p.pmax = partials;
  if (tile == 2)
    hipLaunchKernelGGL((wino_output_kernel<2, 4>), dim3(wino_blocks(w.m_elems / 4)), dim3(256), 0, (hipStream_t)stream, p);
  else if (tile == 4)
    hipLaunchKernelGGL((wino_output_kernel<4, 2>), dim3(wino_blocks(w.m_elems / 2)), dim3(256), 0, (hipStream_t)stream, p);
  else {
    const int v = wino_output_variant(d, w, partials != nullptr);
    const unsigned mb = (unsigned)((int64_t)w.n2 * w.m_stride * 4), ob = (unsigned)((int64_t)d->B * w.Xo * w.Yo * w.Zo * d->Cout * 4);
    if (v == 2)
      hipLaunchKernelGGL((wino_output_buf_kernel<2, 3>), dim3(wino_blocks(w.m_elems / 2)), dim3(256), 0, (hipStream_t)stream, p, mb, ob);
    else if (v == 3)
      hipLaunchKernelGGL((wino_output_buf_kernel<1, 5>), dim3(wino_blocks(w.m_elems)), dim3(256), 0, (hipStream_t)stream, p, mb, ob);
    else if (v == 4)
      hipLaunchKernelGGL((wino_output_buf_kernel<2, 2, 1>), dim3(wino_blocks(w.m_elems / 2)), dim3(256), 0, (hipStream_t)stream, p, mb, ob);
    else if (v == 5)
      hipLaunchKernelGGL((wino_output_buf_kernel<1, 4, 1>), dim3(wino_blocks(w.m_elems)), dim3(256), 0, (hipStream_t)stream, p, mb, ob);
    else if (v == 1)
      hipLaunchKernelGGL((wino_output_kernel<6, 1>), dim3(wino_blocks(w.m_elems)), dim3(256), 0, (hipStream_t)stream, p);
    else
      hipLaunchKernelGGL((wino_output_kernel<6, 2>), dim3(wino_blocks(w.m_elems / 2)), dim3(256), 0, (hipStream_t)stream, p);
  }
  IVX_CHECK_LAUNCH("ivx_conv_winograd_output");
  return IVX_OK;
}

extern "C" int ivx_conv_winograd_output(const ivx_conv_desc *d, int32_t tile, const float *scale, const float *shift, const void *res,
                                        void *out, void *workspace, int64_t workspace_bytes, ivx_stream_t stream) {
  return wino_output_impl(d, tile, scale, shift, res, out, workspace, workspace_bytes, nullptr, stream);
}

extern "C" int ivx_conv_winograd_output_amax(const ivx_conv_desc *d, int32_t tile, const float *scale, const float *shift, const void *res,
                                             void *out, void *workspace, int64_t workspace_bytes, float *partials, ivx_stream_t stream) {
  return wino_output_impl(d, tile, scale, shift, res, out, workspace, workspace_bytes, partials, stream);
}

extern "C" int ivx_conv_winograd_fwd(const ivx_conv_desc *d, int32_t tile, const void *in, const float *u, const float *scale,
                                     const float *shift, const void *res, void *out, void *workspace, int64_t workspace_bytes,
                                     ivx_stream_t stream) {
  int rc = ivx_conv_winograd_input(d, tile, in, workspace, workspace_bytes, stream);
  if (rc != IVX_OK) return rc;
  rc = ivx_conv_winograd_gemm(d, tile, u, workspace, workspace_bytes, stream);
  if (rc != IVX_OK) return rc;
  return ivx_conv_winograd_output(d, tile, scale, shift, res, out, workspace, workspace_bytes, stream);
}

#ifdef IVX_CONV_TIMELINE
extern "C" int ivx_wino_set_timeline(void *buf) { g_wino_timeline = (unsigned long long *)buf; return 0; }
#endif
