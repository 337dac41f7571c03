// The head of the 2-D trunk in ONE launch on the 16-bit matrix cores: conv 7x7 stride 2 pad 3 (3 -> 64) + BatchNorm + ReLU + MaxPool2d(3, 2, 1),
// straight from the fp32 NCHW image to the fp16 (hi, lo) pair tensor the first bottleneck reads.  Replaces three launches (layout change of
// the image, the stem on the fp32 matrix cores, the max-pool) and their tensors: the 64-channel half-resolution map (126 MB at KITTI's batch
// of 4, 983 MB at 50 ScanNet views) is never written -- HBM carries the image once and the quarter-resolution pair map once.
// Reference: mmdet ResNet stem (conv1 / bn1 / relu / maxpool), built at mmdet3d/models/detectors/imvoxelnet.py:22 from
// configs/imvoxelnet/imvoxelnet_kitti.py:4-12, run at :48.
//
//   Workgroup   a 3 x 16 tile of POOLED pixels of one image = a 7 x 33 tile of conv outputs (231 GEMM rows, 256 with padding) from a 19 x 71
//               image patch per colour (zero outside the image = the conv's padding), 4 waves: wave w owns GEMM row tiles 2w, 2w + 1 and
//               both 32-column tiles.  (A 4 x 16 tile on five waves was measured first: five-wave workgroups got ONE workgroup per CU where
//               LDS and registers allow two -- 256 of 1920 workgroups started together in the workgroup timeline.)
//   K order     k = (colour c, filter row ky) x 8 filter columns (kx = 7: zero filter), 21 (c, ky) pairs + 1 zero pair = 11 steps of 16: a
//               lane's 8 k of one step are 8 consecutive pixels of one patch row -- four 8-byte LDS reads, no per-element address arithmetic.
//   Operands    A: fp32 patch values -> (hi, lo) halves of x * s_in in registers (s_in: the power of two from the image's recorded maximum);
//               B: pair filters packed on the host in fragment order (ivx_stem_pool_pack_filters), copied to LDS once per workgroup.
//               Three fp16 MFMA products per multiply-add, fp32 accumulation -- as every other layer of the pair chain.
//   Epilogue    BN + ReLU -> fp32 conv tile in LDS (zero outside the conv map: every pool window holds a valid pixel and ReLU outputs are
//               >= 0, so this equals the pool's -inf padding) -> 3 x 3 / 2 window maxima (exact: maxima of fp32 values) -> pair split with the
//               scale of the bound max|image| * wbound + sbound (the rule of ivx_maxpool2d_fwd_pair) -> 8-byte stores.
#include "ivx_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

struct StemParams {
  const float *img;
  const void *wfrag;                 // 45056 bytes: [column tile 2][step 11][hi, lo][lane 64][8 halves]
  const float *scale_p, *shift;      // [64]: scale / s_w, shift
  const unsigned *amax_img;          // IVX_AMAX_SLOTS words: bits of max |image|
  _Float16 *out;
  float *out_scale_p;
  unsigned *amax_out;
  float wbound, sbound;
  int B, H, W, Hc, Wc, Hp, Wp;
  int tiles_x, tiles_y, n_tiles, q_total;
#ifdef IVX_CONV_TIMELINE
  unsigned long long *tl;
#endif
};

#define STEM_WBYTES 45056
#define STEM_PATCH_OFF STEM_WBYTES
#define STEM_PITCH 72
#define STEM_TPH 3                                  /* pooled rows per tile */
#define STEM_CTH (2 * STEM_TPH + 1)                 /* conv rows per tile */
#define STEM_ROWS (STEM_CTH * 33)                   /* GEMM rows in use (231) */
#define STEM_PROWS (2 * STEM_CTH + 5)               /* patch rows (19) */
#define STEM_PATCH_FLOATS (3 * STEM_PROWS * STEM_PITCH)
#define STEM_STAGE_BYTES (STEM_ROWS * 64 * 4)
#define STEM_LDS (STEM_STAGE_BYTES > STEM_PATCH_OFF + STEM_PATCH_FLOATS * 4 ? STEM_STAGE_BYTES : STEM_PATCH_OFF + STEM_PATCH_FLOATS * 4)

__global__ __launch_bounds__(256, 2) void stem_pool_pair_kernel(const StemParams p) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[STEM_LDS + 32];
  typedef __attribute__((address_space(3))) void *lds_ptr_t;
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int rr = lane & 31, hh = lane >> 5;
  const int tile = (int)(blockIdx.x & 7) * p.q_total + (int)(blockIdx.x >> 3);
  if (tile >= p.n_tiles) return;
#ifdef IVX_CONV_TIMELINE
  unsigned long long tls[6];
  tls[0] = __builtin_amdgcn_s_memrealtime();
#endif
  const int tpi = p.tiles_x * p.tiles_y;
  const int b = tile / tpi;
  const int trem = tile - b * tpi;
  const int ty = trem / p.tiles_x, tx = trem - ty * p.tiles_x;
  const int py0 = ty * STEM_TPH, px0 = tx * 16;
  const int cy0 = 2 * py0 - 1, cx0 = 2 * px0 - 1;       // conv tile origin
  const int iy0 = 2 * cy0 - 3, ix0 = 2 * cx0 - 3;       // image patch origin

  // scales (every wave computes the same values)
  float s_in, s_out;
  bool sat;
  {
    float a = __uint_as_float(p.amax_img[lane]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) a = fmaxf(a, __shfl_xor(a, o));
    const float bound = (a * p.wbound + p.sbound) * 1.001f;
    sat = !(bound < 3.0e38f);
    s_in = sat ? 0.00390625f : ivx_pow2_scale(a);
    s_out = sat ? 0.00390625f : ivx_pow2_scale(bound);
    s_in = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, s_in)));
    s_out = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, s_out)));
  }
  if (p.out_scale_p && blockIdx.x == 0 && tid == 0) *p.out_scale_p = s_out;

  // ---- filters: 44 KB straight copy global -> LDS (LDS-DMA, 1 KB per wave instruction)
  {
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void *)p.wfrag, 0, STEM_WBYTES, 0x00020000);
#pragma unroll
    for (int it = 0; it < STEM_WBYTES / 4096; ++it) {
      const int c = it * 4 + w;                          // wave-uniform 1 KB chunk
      const unsigned vo = (unsigned)(c * 1024 + lane * 16);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_ptr_t)(smem + c * 1024), 16, vo, 0, 0, 0);
    }
  }
  // ---- image patch: 3 x 19 x 71 fp32, zero outside the image, rows of 72 floats
  {
    float *patch = reinterpret_cast<float *>(smem + STEM_PATCH_OFF);
    constexpr int NIT = (STEM_PATCH_FLOATS + 255) / 256;
    float v[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int e = tid + 256 * it;
      const int c = e / (STEM_PROWS * STEM_PITCH), rem = e - c * (STEM_PROWS * STEM_PITCH);
      const int py = rem / STEM_PITCH, px = rem - py * STEM_PITCH;
      const int gy = iy0 + py, gx = ix0 + px;
      v[it] = 0.f;
      if (e < STEM_PATCH_FLOATS && px < 71 && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W)
        v[it] = p.img[(((size_t)b * 3 + c) * p.H + gy) * p.W + gx];
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int e = tid + 256 * it;
      if (e < STEM_PATCH_FLOATS) patch[e] = v[it];
    }
  }
  __syncthreads();                                       // (the fence waits for the DMA too: vmcnt(0))
#ifdef IVX_CONV_TIMELINE
  tls[1] = __builtin_amdgcn_s_memrealtime();
#endif

  // ---- GEMM: 256 x 64 x 176, three fp16 products per multiply-add
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  {
    const float *patch = reinterpret_cast<const float *>(smem + STEM_PATCH_OFF);
    int abase[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      int r = (2 * w + i) * 32 + rr;
      r = r > STEM_ROWS - 1 ? STEM_ROWS - 1 : r;
      const int ly = (r * 1986) >> 16, lx = r - ly * 33;   // r / 33, r % 33 (exact for r < 320)
      abase[i] = (2 * ly) * STEM_PITCH + 2 * lx;
    }
    const unsigned char *bf = smem + lane * 16;
#pragma unroll 1
    for (int kk = 0; kk < 11; ++kk) {
      int pc = 2 * kk + hh;
      pc = pc > 20 ? 20 : pc;                            // the 22nd (c, ky) pair has zero filters
      const int c = (pc * 37) >> 8, ky = pc - 7 * c;     // pc / 7, pc % 7 (exact for pc < 21)
      const int koff = (c * STEM_PROWS + ky) * STEM_PITCH;
      f32x4 ah[2], al[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const float *src = patch + abase[i] + koff;
        f16x8 h8, l8;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x2 x = *reinterpret_cast<const f32x2 *>(src + 2 * q) * s_in;
          f32x2 xs = x;
          if (sat) { xs[0] = __builtin_amdgcn_fmed3f(x[0], -65504.f, 65504.f); xs[1] = __builtin_amdgcn_fmed3f(x[1], -65504.f, 65504.f); }
          const f16x2 h = __builtin_convertvector(xs, f16x2);
          const f16x2 l = __builtin_convertvector(xs - __builtin_convertvector(h, f32x2), f16x2);
          h8[2 * q] = h[0]; h8[2 * q + 1] = h[1];
          l8[2 * q] = l[0]; l8[2 * q + 1] = l[1];
        }
        ah[i] = __builtin_bit_cast(f32x4, h8);
        al[i] = __builtin_bit_cast(f32x4, l8);
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const f32x4 bh = *reinterpret_cast<const f32x4 *>(bf + ((j * 11 + kk) * 2 + 0) * 1024);
        const f32x4 bl = *reinterpret_cast<const f32x4 *>(bf + ((j * 11 + kk) * 2 + 1) * 1024);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, ah[i]), __builtin_bit_cast(f16x8, bh), acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, ah[i]), __builtin_bit_cast(f16x8, bl), acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, al[i]), __builtin_bit_cast(f16x8, bh), acc[i][j], 0, 0, 0);
        }
      }
    }
  }
#ifdef IVX_CONV_TIMELINE
  tls[2] = __builtin_amdgcn_s_memrealtime();
#endif
  __syncthreads();                                       // patch and filters are dead: the fp32 conv tile takes their place

  // ---- BN + ReLU -> stage [231][64] fp32 (zero outside the conv map)
  {
    float *stage = reinterpret_cast<float *>(smem);
    const float inv_in = 1.0f / s_in;
    float scv[2], shv[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) { scv[j] = p.scale_p[j * 32 + rr] * inv_in; shv[j] = p.shift[j * 32 + rr]; }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (2 * w + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
        const int ly = (row * 1986) >> 16, lx = row - ly * 33;
        const int cy = cy0 + ly, cx = cx0 + lx;
        const float ok = ((unsigned)cy < (unsigned)p.Hc && (unsigned)cx < (unsigned)p.Wc) ? 1.0f : 0.f;
        if (row < STEM_ROWS) {
#pragma unroll
          for (int j = 0; j < 2; ++j) stage[row * 64 + j * 32 + rr] = fmaxf(acc[i][j][r] * scv[j] + shv[j], 0.f) * ok;
        }
      }
  }
#ifdef IVX_CONV_TIMELINE
  tls[3] = __builtin_amdgcn_s_memrealtime();
#endif
  __syncthreads();

  // ---- 3 x 3 / 2 max-pool + pair split: one thread per (pooled pixel, 4 channels)
  float omax = 0.f;
  {
    const float *stage = reinterpret_cast<const float *>(smem);
#pragma unroll 1
    for (int it = 0; it < STEM_TPH; ++it) {             // 16 * STEM_TPH pooled pixels x 16 channel quads = 256 per pass
      const int idx = tid + 256 * it;
      const int pp = idx >> 4, c4 = (idx & 15) * 4;
      const int ppy = pp >> 4, ppx = pp & 15;
      const int py = py0 + ppy, px = px0 + ppx;
      if (py >= p.Hp || px >= p.Wp) continue;
      f32x4 m = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
          const f32x4 v = *reinterpret_cast<const f32x4 *>(stage + ((2 * ppy + dy) * 33 + 2 * ppx + dx) * 64 + c4);
#pragma unroll
          for (int e = 0; e < 4; ++e) m[e] = fmaxf(m[e], v[e]);
        }
      f16x4 hi, lo;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        omax = fmaxf(omax, m[e]);
        const float y = fminf(m[e] * s_out, 65504.f);
        hi[e] = (_Float16)y;
        lo[e] = (_Float16)(y - (float)hi[e]);
      }
      _Float16 *o = p.out + ((((size_t)b * p.Hp + py) * p.Wp + px) * 128) + (size_t)((c4 >> 4) * 32 + (c4 & 15));
      *reinterpret_cast<f16x4 *>(o) = hi;
      *reinterpret_cast<f16x4 *>(o + 16) = lo;
    }
  }
#ifdef IVX_CONV_TIMELINE
  tls[4] = __builtin_amdgcn_s_memrealtime();
#endif
  if (p.amax_out) {
    __syncthreads();                                     // the stage is read-only above; its last 32 bytes of padding take the wave maxima
    float *red = reinterpret_cast<float *>(smem + STEM_LDS);
    ivx_amax_commit_wg(p.amax_out, omax, red, (int)blockIdx.x);
  }
#ifdef IVX_CONV_TIMELINE
  if (p.tl && tid == 0) {
    unsigned long long *t = p.tl + (size_t)blockIdx.x * 8;
    for (int i = 0; i < 5; ++i) t[i] = tls[i];
    t[5] = __builtin_amdgcn_s_memrealtime();
  }
#endif
}

#ifdef IVX_CONV_TIMELINE
static unsigned long long *g_stem_timeline = nullptr;
extern "C" int ivx_stem_set_timeline(void *buf) { g_stem_timeline = (unsigned long long *)buf; return 0; }
#endif

// max |x| of an fp32 buffer into IVX_AMAX_SLOTS words (atomic max on the bits; the caller zeroes them): the image's maximum for the stem's scales
__global__ __launch_bounds__(256) void amax_f32_kernel(const float *x, size_t n, unsigned *slots) {
  float m = 0.f;
  const size_t n4 = n >> 2;
  for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < n4; t += (size_t)gridDim.x * blockDim.x) {
    const f32x4 v = reinterpret_cast<const f32x4 *>(x)[t];
    m = fmaxf(m, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
    if (v[0] != v[0] || v[1] != v[1] || v[2] != v[2] || v[3] != v[3]) m = __builtin_inff();      // a NaN makes the bound non-finite, as an Inf does
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const float v = x[n4 * 4 + threadIdx.x];
    m = fmaxf(m, v != v ? __builtin_inff() : fabsf(v));
  }
  __shared__ float red[4];
  ivx_amax_commit_wg(slots, m, red, (int)blockIdx.x);
}

extern "C" int ivx_amax_f32(const float *x, int64_t n, uint32_t *amax, ivx_stream_t stream) {
  IVX_REQUIRE(x && amax && n >= 0 && ((uintptr_t)x & 15) == 0, "ivx_amax_f32: null / misaligned argument");
  if (n == 0) return IVX_OK;
  size_t blocks = ((size_t)n / 4 + 255) / 256;
  blocks = blocks < 1 ? 1 : (blocks > 2048 ? 2048 : blocks);
  hipLaunchKernelGGL(amax_f32_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, (size_t)n, amax);
  IVX_CHECK_LAUNCH("ivx_amax_f32");
  return IVX_OK;
}

extern "C" int ivx_stem_pool_out_dims(int32_t H, int32_t W, int32_t *Hp, int32_t *Wp) {
  IVX_REQUIRE(Hp && Wp && H > 0 && W > 0, "ivx_stem_pool_out_dims: bad argument");
  const int Hc = (H + 6 - 7) / 2 + 1, Wc = (W + 6 - 7) / 2 + 1;
  *Hp = (Hc + 2 - 3) / 2 + 1;
  *Wp = (Wc + 2 - 3) / 2 + 1;
  return IVX_OK;
}

extern "C" int ivx_stem_pool_fwd_pair(const float *img, int32_t B, int32_t H, int32_t W, const void *wfrag, const float *scale_p, const float *shift,
                                      float wbound, float sbound, const uint32_t *amax_img, void *out, float *out_scale, uint32_t *amax_out,
                                      ivx_stream_t stream) {
  IVX_REQUIRE(img && wfrag && scale_p && shift && amax_img && out && out_scale, "ivx_stem_pool_fwd_pair: null argument");
  IVX_REQUIRE(B > 0 && H >= 7 && W >= 7, "ivx_stem_pool_fwd_pair: bad dims (B %d H %d W %d)", B, H, W);
  IVX_REQUIRE(wbound >= 0.f && sbound >= 0.f, "ivx_stem_pool_fwd_pair: negative bound terms");
  StemParams p;
  p.img = img; p.wfrag = wfrag; p.scale_p = scale_p; p.shift = shift; p.amax_img = amax_img;
  p.out = (_Float16 *)out; p.out_scale_p = out_scale; p.amax_out = amax_out; p.wbound = wbound; p.sbound = sbound;
  p.B = B; p.H = H; p.W = W;
  p.Hc = (H + 6 - 7) / 2 + 1; p.Wc = (W + 6 - 7) / 2 + 1;
  p.Hp = (p.Hc + 2 - 3) / 2 + 1; p.Wp = (p.Wc + 2 - 3) / 2 + 1;
  IVX_REQUIRE((int64_t)B * p.Hp * p.Wp * 64 * 4 < (1LL << 40), "ivx_stem_pool_fwd_pair: output too large");
  p.tiles_x = (p.Wp + 15) / 16; p.tiles_y = (p.Hp + STEM_TPH - 1) / STEM_TPH;
  const int64_t nt = (int64_t)B * p.tiles_x * p.tiles_y;
  IVX_REQUIRE(nt < (1LL << 28), "ivx_stem_pool_fwd_pair: too many tiles");
  p.n_tiles = (int)nt;
  p.q_total = (p.n_tiles + 7) / 8;
#ifdef IVX_CONV_TIMELINE
  p.tl = g_stem_timeline;
#endif
  hipLaunchKernelGGL(stem_pool_pair_kernel, dim3((unsigned)(8 * p.q_total)), dim3(256), 0, (hipStream_t)stream, p);
  IVX_CHECK_LAUNCH("ivx_stem_pool_fwd_pair");
  return IVX_OK;
}
