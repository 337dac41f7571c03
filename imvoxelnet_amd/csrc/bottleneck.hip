// One ResNet bottleneck (identity shortcut) in ONE launch on the 16-bit matrix cores: conv1 1x1 (4P -> P) + BN + ReLU, conv2 3x3 (P -> P) + BN +
// ReLU, conv3 1x1 (P -> 4P) + BN + shortcut + ReLU, on fp16 (hi, lo) pair tensors (IVX_F16_PAIR; three fp16 MFMA products per multiply-add as
// ivx_conv_fwd_pio).  Replaces three launches of conv_igemm_v4_kernel per block of ResNet-50's stages 1 and 2 (reference call site
// mmdet3d/models/detectors/imvoxelnet.py:48, configs/imvoxelnet/imvoxelnet_kitti.py:4-12: the mmdet ResNet(depth=50, style='pytorch') blocks):
// the P-channel intermediates never leave the CU -- per block HBM carries the 4P-channel input once (+ a one-pixel halo, mostly from L2) and
// the 4P-channel output once, instead of (4P + P) + (P + P) + (P + 4P + 4P) channels per pixel.
//
//   Workgroup = an 8 x 16 tile of output pixels of one image; P / 16 waves (P = 64: 4 waves, two workgroups per CU; P = 128: 8 waves, one).
//   Phase 1   conv1 over the tile + its one-pixel ring (10 x 18 = 180 pixels, 192 GEMM rows): A (pixels) and B (filters) stream global -> LDS by
//             LDS-DMA in 32-channel slabs (128-byte XOR-swizzled rows, hardware zero fill outside the image), ring of NB1 slabs; the epilogue
//             (BN, ReLU, zero outside the image = conv2's padding, pair split with the a-priori scale) leaves mid1 in LDS as a pair tile.
//   Phase 2   conv2: A fragments straight from the mid1 tile (im2col by addressing: tap (dy, dx) shifts the pixel row), the 9 x P/32 filter
//             slabs through a ring of NB2; epilogue -> mid2 (pair tile, 128 pixels) over mid1's LDS.
//   Phase 3   conv3 in units of 128 output channels: A from mid2, filter slabs through a ring of NB3; epilogue through a per-wave transposing
//             stage (4 consecutive channels per lane): BN + the block's input re-read as the shortcut (L2) + ReLU + max |out| + pair split,
//             8-byte stores of hi and lo halves.
//   Scales    powers of two from BOUNDS, as conv_pair_io (conv_igemm.hip): b1 = max|in| * wbound1 + sbound1, b2 = b1 * wbound2 + sbound2,
//             b3 = b2 * wbound3 + sbound3 + max|in|; every wave derives them from the input's amax slots -- no pass over a tensor, no
//             communication.  The looseness compounds over the three layers (the layer-wise chain measures max |mid|): a bound loose by L
//             costs nothing up to L = 2^18 (ivx_common.h), ResNet-50's are ~2^10 after three layers.
//   LDS swizzle of the pair tiles: a pixel row is 4P bytes (16-byte chunks [hi8 hi8 lo8 lo8] per 16 channels); chunk c of pixel (y, x) lies at
//             c ^ (x & 15): the 16 lanes of one ds_read_b128 group hold 16 consecutive x (ds_read_b128 lane groups, MI355X guide), so every
//             group reads 16 different 16-byte positions of the 256-byte bank window -- for every tap.
//
// CIN != 4P (round 6, second form): the FIRST block of ResNet stage 1 (layer1.0: CIN = P = 64, stride 1) whose shortcut is a 1x1 convolution + BN of the
// block's input.  conv3 and the shortcut conv are ONE GEMM over K = CIN + P: the filter bank is [4P][x channels | mid2 channels] with both
// BatchNorm scales folded into the filters on the host (ivx_bottleneck_proj_pack; the epilogue's multiplier is 1 / s_w, its shift shift3 + shiftd).
// The A operand of the x part are the block's input at the tile's own pixels, read into registers from conv1's staged slabs at the end of phase 1
// (K = 64: both slabs are resident); the two operand scales differ by a power of two: the accumulator is multiplied by rho = s2 / s_in between the
// x slabs and the mid2 slabs.  No shortcut tensor exists: HBM carries the CIN-channel input (+ halo) and the 4P-channel output, once.
#include "ivx_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

struct BnkParams {
  const _Float16 *in;
  _Float16 *out;
  const _Float16 *w1, *w2, *w3;              // pair filters of ivx_pair_pack_filters: [Cout][Cin/32][taps][hi16 lo16 hi16 lo16]
  const float *sc1, *sh1, *sc2, *sh2, *sc3, *sh3;   // scale / s_w and shift of the three BatchNorms
  const float *in_scale_p;                   // device: the scale the input was written with
  const unsigned *amax_in;                   // device, IVX_AMAX_SLOTS words: bits of max |in|
  float *out_scale_p;                        // device: receives the scale of the output
  unsigned *amax_out;                        // device, IVX_AMAX_SLOTS words: max |out| (atomic max), or NULL
  float wb1, sb1, wb2, sb2, wb3, sb3;
  float wbd;                                 // bound factor of the shortcut: 1 (identity) or max_n sum_k |scaled w_d| (projection)
  int B, H, W;
  int tiles_x, tiles_y, n_tiles, q_total;    // q_total: tiles per XCD (workgroup b runs on XCD b % 8 and owns tile (b % 8) * q_total + b / 8)
#ifdef IVX_CONV_TIMELINE
  unsigned long long *tl;
#endif
};


template <int N>
__device__ __forceinline__ void bnk_wait_vm() {           // s_waitcnt vmcnt(N) only (gfx9 encoding: vmcnt[3:0] bits 3:0, vmcnt[5:4] bits 15:14)
  constexpr int n = N > 63 ? 63 : N;
  __builtin_amdgcn_s_waitcnt(0x0f70 | (n & 15) | ((n >> 4) << 14));
}
// wait until at most PER * newer of this lane's vector-memory requests are outstanding (newer: slabs issued after the one needed)
template <int PER>
__device__ __forceinline__ void bnk_wait_slabs(const int newer) {
  switch (newer) {
    case 1: bnk_wait_vm<PER>(); break;
    case 2: bnk_wait_vm<2 * PER>(); break;
    case 3: bnk_wait_vm<3 * PER>(); break;
    default: bnk_wait_vm<0>(); break;
  }
}
// workgroup barrier without the fence of __syncthreads() (that fence waits for vmcnt(0): every LDS-DMA request of a ring); the caller has
// waited for what must be visible -- lgkmcnt(0) here covers this wave's own LDS writes / fragment reads
__device__ __forceinline__ void bnk_barrier() {
  __builtin_amdgcn_s_waitcnt(0xc07f);      // lgkmcnt(0)
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}
__device__ __forceinline__ f32x16 bnk_mfma(const f32x4 a, const f32x4 b, const f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
// Two channels (ne, ne + 1) of one pixel of a pair tile: e0, e1 are the scaled values in [0, 65504]; hi halves go to 16-byte chunk `ck` of the
// pixel row (byte `inchunk` inside it), lo halves to chunk ck ^ 2.  Packed conversions (v_cvt_pk_f16_f32, round to nearest even).
__device__ __forceinline__ void bnk_put2(unsigned char *px, const int ck, const int inchunk, const float e0, const float e1, const bool ok) {
  const f32x2 e = {e0, e1};
  const f16x2 h = __builtin_convertvector(e, f16x2);
  const f16x2 l = __builtin_convertvector(e - __builtin_convertvector(h, f32x2), f16x2);
  unsigned hi = __builtin_bit_cast(unsigned, h), lo = __builtin_bit_cast(unsigned, l);
  if (!ok) { hi = 0u; lo = 0u; }
  *reinterpret_cast<unsigned *>(px + ck * 16 + inchunk) = hi;
  *reinterpret_cast<unsigned *>(px + (ck ^ 2) * 16 + inchunk) = lo;
}
// Lanes l, l ^ 1 hold channels n, n ^ 1 of the same accumulator rows: ya / yb = this lane's values of rows a / a + 1.  After the exchange the
// even lane owns both channels of row a, the odd lane both channels of row a + 1: (e0, e1) = channels (n & ~1, n | 1).
__device__ __forceinline__ void bnk_swap2(const float ya, const float yb, const bool odd, float &e0, float &e1) {
  const float pa = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, ya), 0xB1, 0xf, 0xf, true));   // quad_perm [1, 0, 3, 2]
  const float pb = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, yb), 0xB1, 0xf, 0xf, true));
  e0 = odd ? pb : ya;
  e1 = odd ? yb : pa;
}

typedef __attribute__((address_space(3))) void *lds_ptr_t;

template <int P, int CIN, int NB1, int NB2, int NB3>
struct BnkCfg {
  static constexpr int C = 4 * P, NW = P / 16, NT = 64 * NW;
  static constexpr bool PROJ = CIN != C;       // the shortcut is a 1x1 conv of the CIN-channel input (folded into conv3's filter bank)
  static constexpr int NQ = P / 32;            // 32-channel chunks of a P-channel tensor (K slabs of conv2 per tap / of conv3; 128-column units of conv3)
  static constexpr int NQ1 = CIN / 32;         // K slabs of conv1
  static constexpr int NQX = PROJ ? CIN / 32 : 0;   // K slabs of the shortcut conv in front of conv3's
  static constexpr int NTN = P / 32;           // 32-column tiles of a P-column GEMM
  static constexpr int PXB = 4 * P;            // bytes of one pixel of a pair tile
  static constexpr int M1 = 180, M1P = 192;
  static constexpr int A1 = M1P * 128, B1 = P * 128, SL1 = A1 + B1;
  static constexpr int MID = M1P * PXB;         // (192 rows: the epilogue writes the 12 padding rows too instead of branching around them)
  static constexpr int SL2 = P * 128, R2 = MID;
  static constexpr int MID2 = 128 * PXB, ST3 = MID2, STG = NW * 2048, R3 = ST3 + STG, SL3 = 128 * 128;
  static constexpr int L1 = NB1 * SL1, L2 = R2 + NB2 * SL2, L3 = R3 + NB3 * SL3;
  static constexpr int LBUF = L1 > L2 ? (L1 > L3 ? L1 : L3) : (L2 > L3 ? L2 : L3);
  static constexpr int PRM = LBUF;             // 12 P floats: scale1 | shift1 | scale2 | shift2 | scale3 | shift3 (read by the epilogues from LDS, not L2)
  static constexpr int LDS = LBUF + 12 * P * 4;
  static constexpr int RP = NT / 8;            // rows of 128 bytes one pass of the workgroup's DMA covers
  static constexpr int AR = M1P / RP, BR1 = P / RP, BR2 = P / RP, BR3 = 128 / RP;
};



template <int P, int CIN, int NB1, int NB2, int NB3>
__global__ __launch_bounds__(P * 4, 2) void bottleneck_pio_kernel(const BnkParams p, const unsigned in_bytes, const unsigned w1_bytes,
                                                                                 const unsigned w2_bytes, const unsigned w3_bytes) {
  typedef BnkCfg<P, CIN, NB1, NB2, NB3> G;
  constexpr int C = G::C, NW = G::NW, NQ = G::NQ, NQ1 = G::NQ1, NTN = G::NTN, PXB = G::PXB, RP = G::RP;
  constexpr bool PROJ = G::PROJ;
  constexpr int NQX = G::NQX;
  static_assert(!PROJ || (P == 64 && NQ1 == NB1 && NB3 == 2), "projection form: P = 64, both conv1 slabs resident, two-slot conv3 ring");
  constexpr int AR = G::AR, BR1 = G::BR1, BR2 = G::BR2, BR3 = G::BR3;      // (local: arrays with these bounds are captured by the DMA lambdas)
  static_assert(P == 64 || P == 128, "planes");
  static_assert(G::LDS <= (P == 64 ? 81920 : 163840), "LDS budget");
  static_assert(G::NT == 4 * P, "the parameter block is loaded three floats per thread");
  __shared__ __attribute__((aligned(16))) unsigned char smem[G::LDS];

  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int rr = lane & 31, hh = lane >> 5;
  // tile of this workgroup
  const int tile = (int)(blockIdx.x & 7) * p.q_total + (int)(blockIdx.x >> 3);
  if (tile >= p.n_tiles) return;
#ifdef IVX_CONV_TIMELINE
  const unsigned long long tl0 = __builtin_amdgcn_s_memrealtime();
  unsigned long long tl1 = 0, tl1b = 0, tl2 = 0, tl2b = 0, tl3 = 0;
#endif
  const int tpi = p.tiles_x * p.tiles_y;
  const int b = tile / tpi;
  const int trem = tile - b * tpi;
  const int ty = trem / p.tiles_x, tx = trem - ty * p.tiles_x;
  const int y0 = ty * 8, x0 = tx * 16;

  // ---- scales (every wave computes the same values from the same device words)
  float inv_in, s1, s2, s_out;
  bool sat;
  {
    float a = __uint_as_float(p.amax_in[lane]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) a = fmaxf(a, __shfl_xor(a, o));
    const float b1 = (a * p.wb1 + p.sb1) * 1.001f;
    const float b2 = (b1 * p.wb2 + p.sb2) * 1.001f;
    const float b3 = (b2 * p.wb3 + p.sb3 + a * p.wbd) * 1.001f;
    sat = !(b3 < 3.0e38f);
    s1 = sat ? 0.00390625f : ivx_pow2_scale(b1);
    s2 = sat ? 0.00390625f : ivx_pow2_scale(b2);
    s_out = sat ? 0.00390625f : ivx_pow2_scale(b3);
    inv_in = 1.0f / *p.in_scale_p;
    inv_in = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, inv_in)));
    s1 = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, s1)));
    s2 = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, s2)));
    s_out = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, s_out)));
  }
  if (p.out_scale_p && blockIdx.x == 0 && tid == 0) *p.out_scale_p = s_out;
  float *prm = reinterpret_cast<float *>(smem + G::PRM);
  {   // 12 P floats = 3 per thread; first read after conv1's K loop (a __syncthreads() lies between)
    const float *src[6] = {p.sc1, p.sh1, p.sc2, p.sh2, p.sc3, p.sh3};
    prm[tid] = tid < P ? src[0][tid] : (tid < 2 * P ? src[1][tid - P] : (tid < 3 * P ? src[2][tid - 2 * P] : src[3][tid - 3 * P]));
    prm[4 * P + tid] = p.sc3[tid];
    prm[8 * P + tid] = p.sh3[tid];
  }

  const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc((void *)p.in, 0, in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_w1 = __builtin_amdgcn_make_buffer_rsrc((void *)p.w1, 0, w1_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_w2 = __builtin_amdgcn_make_buffer_rsrc((void *)p.w2, 0, w2_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_w3 = __builtin_amdgcn_make_buffer_rsrc((void *)p.w3, 0, w3_bytes, 0x00020000);
  const unsigned OOB = 0x80000000u;

  // DMA lane geometry: 8 lanes per 128-byte row; slot tid & 7 of row lr receives k-chunk slot ^ ((lr >> 1) & 7)
  const int lr = tid >> 3;
  const int cc = (tid & 7) ^ ((lr >> 1) & 7);
  const int fsw = (rr >> 1) & 7;                        // fragment reads of the DMA-staged slabs undo it

  // projection form: A fragments of the block's input at the tile's own pixels (row tiles 2 wm, 2 wm + 1 of phase 3), all K = CIN
  f32x4 xf[PROJ ? 2 : 1][PROJ ? NQX : 1][4];
  // =========================================================================================== phase 1: conv1 over the haloed tile
  {
    unsigned a_off[AR];
#pragma unroll
    for (int j = 0; j < AR; ++j) {
      const int r = lr + RP * j;
      const int hy = (r * 3641) >> 16, hx = r - hy * 18;     // r / 18, r % 18 (exact for r < 192)
      const int gy = y0 - 1 + hy, gx = x0 - 1 + hx;
      const bool ok = r < G::M1 && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
      a_off[j] = ok ? (unsigned)(((b * p.H + gy) * p.W + gx) * (4 * CIN) + cc * 16) : OOB;
    }
    unsigned b_off[BR1];
#pragma unroll
    for (int j = 0; j < BR1; ++j) b_off[j] = (unsigned)((lr + RP * j) * NQ1 * 128 + cc * 16);
    auto load1 = [&](const int k, const int buf) {
      unsigned char *Ab = smem + buf * G::SL1 + w * 1024;
      unsigned char *Bb = Ab + G::A1;
#pragma unroll
      for (int j = 0; j < AR; ++j) {       // (the offset goes through a local: with the captured array element as the builtin's argument clang drops
        const unsigned vo = a_off[j] == OOB ? OOB : a_off[j] + (unsigned)k * 128u;       //  the kernel's host stub without a diagnostic)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in, (lds_ptr_t)(Ab + RP * j * 128), 16, vo, 0, 0, 0);
      }
#pragma unroll
      for (int j = 0; j < BR1; ++j) {
        const unsigned vo = b_off[j] + (unsigned)k * 128u;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w1, (lds_ptr_t)(Bb + RP * j * 128), 16, vo, 0, 0, 0);
      }
    };
    const int nt = w % NTN, mg = w / NTN;                // this wave: column tile nt, row tiles mg * 3 .. + 2
    f32x16 acc[3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
#pragma unroll
    for (int k = 0; k < NB1 - 1; ++k) load1(k, k);
    int cur = 0;
    for (int k = 0; k < NQ1; ++k) {
      int newer = NQ1 - 1 - k;
      newer = newer > NB1 - 2 ? NB1 - 2 : newer;
      bnk_wait_slabs<AR + BR1>(newer);
      bnk_barrier();                                     // slab k visible to every wave; every wave has finished slab k - 1
      if (k + NB1 - 1 < NQ1) load1(k + NB1 - 1, cur == 0 ? NB1 - 1 : cur - 1);
      const unsigned char *Ac = smem + cur * G::SL1 + (mg * 96 + rr) * 128;
      const unsigned char *Bc = smem + cur * G::SL1 + G::A1 + (nt * 32 + rr) * 128;
      f32x4 fa[2][3], fb[2];
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        const int sl = kk & 1;
        const int ch = ((2 * kk + hh) ^ fsw) * 16;
#pragma unroll
        for (int i = 0; i < 3; ++i) fa[sl][i] = *reinterpret_cast<const f32x4 *>(Ac + i * 4096 + ch);
        fb[sl] = *reinterpret_cast<const f32x4 *>(Bc + ch);
        if (sl == 0) {
#pragma unroll
          for (int i = 0; i < 3; ++i) acc[i] = bnk_mfma(fa[0][i], fb[0], acc[i]);          // hi * hi
        } else {
#pragma unroll
          for (int i = 0; i < 3; ++i) {
            acc[i] = bnk_mfma(fa[0][i], fb[1], acc[i]);                                    // hi * lo
            acc[i] = bnk_mfma(fa[1][i], fb[0], acc[i]);                                    // lo * hi
          }
        }
      }
      cur = cur + 1 == NB1 ? 0 : cur + 1;
    }
    if constexpr (PROJ) {                                // slab q still lies in ring slot q (NQ1 == NB1: nothing was loaded over it)
      const int wm3 = w >> 1;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int m = (2 * wm3 + i) * 32 + rr;           // tile pixel (m >> 4, m & 15) = halo row ((m >> 4) + 1) * 18 + (m & 15) + 1
        const int hr = ((m >> 4) + 1) * 18 + (m & 15) + 1;
        const int key = (hr >> 1) & 7;
#pragma unroll
        for (int q = 0; q < NQX; ++q)
#pragma unroll
          for (int kk = 0; kk < 4; ++kk)
            xf[i][q][kk] = *reinterpret_cast<const f32x4 *>(smem + q * G::SL1 + hr * 128 + (((2 * kk + hh) ^ key) * 16));
      }
    }
    __syncthreads();                                     // the conv1 ring is dead: mid1 and the conv2 ring take its place
#ifdef IVX_CONV_TIMELINE
    tl1 = __builtin_amdgcn_s_memrealtime();
#endif
    // ---- epilogue: mid1 = s1 * relu(bn1(conv1)) as a pair tile, zero outside the image (conv2's padding)
    const int n = nt * 32 + rr;
    const float sA = prm[n] * (inv_in * s1), tA = prm[P + n] * s1;
    const int ne = n & ~1;
    const int chunk_hi = (ne >> 4) * 4 + ((ne & 15) >> 3);
    const int inchunk = (ne & 7) * 2;
    const bool odd = lane & 1;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
      for (int pr = 0; pr < 8; ++pr) {
        const int ra = 2 * pr;
        const int row = (mg * 3 + i) * 32 + (ra & 3) + 8 * (ra >> 2) + 4 * hh + (odd ? 1 : 0);
        const int hy = (row * 3641) >> 16, hx = row - hy * 18;
        const int gy = y0 - 1 + hy, gx = x0 - 1 + hx;
        const bool ok = (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
        const float ya = __builtin_amdgcn_fmed3f(acc[i][ra] * sA + tA, 0.f, 65504.f), yb = __builtin_amdgcn_fmed3f(acc[i][ra + 1] * sA + tA, 0.f, 65504.f);
        float e0, e1;
        bnk_swap2(ya, yb, odd, e0, e1);
        bnk_put2(smem + row * PXB, chunk_hi ^ (hx & 15), inchunk, e0, e1, ok);
      }
    }
  }

  // =========================================================================================== phase 2: conv2 (3x3) from the mid1 tile
  {
#ifdef IVX_CONV_TIMELINE
    tl1b = __builtin_amdgcn_s_memrealtime();
#endif
    constexpr int S2 = 9 * NQ;
    const int nt = w % NTN, g = w / NTN;                 // column tile nt, row tiles 2g, 2g + 1
    f32x16 acc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    const int ox = rr & 15;
    int prow[2];                                          // halo-pixel index of tap (0, 0) of this lane's row in tile i
#pragma unroll
    for (int i = 0; i < 2; ++i) prow[i] = (2 * (2 * g + i) + (rr >> 4)) * 18 + ox;
    {
    unsigned b_off[BR2];
#pragma unroll
    for (int j = 0; j < BR2; ++j) b_off[j] = (unsigned)((lr + RP * j) * S2 * 128 + cc * 16);
    auto load2 = [&](const int s, const int buf) {      // slab s = (chunk q, tap t) = s-th 128-byte block of every filter row
      unsigned char *Bb = smem + G::R2 + buf * G::SL2 + w * 1024;
#pragma unroll
      for (int j = 0; j < BR2; ++j) {
        const unsigned vo = b_off[j] + (unsigned)s * 128u;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w2, (lds_ptr_t)(Bb + RP * j * 128), 16, vo, 0, 0, 0);
      }
    };
#pragma unroll
    for (int s = 0; s < NB2 - 1; ++s) load2(s, s);
    int cur = 0, q = 0, t = 0;
    for (int s = 0; s < S2; ++s) {
      int newer = S2 - 1 - s;
      newer = newer > NB2 - 2 ? NB2 - 2 : newer;
      bnk_wait_slabs<BR2>(newer);
      bnk_barrier();                                     // (first pass: also publishes mid1)
      if (s + NB2 - 1 < S2) load2(s + NB2 - 1, cur == 0 ? NB2 - 1 : cur - 1);
      const int dy = t / 3, dx = t - dy * 3;
      const int key = (ox + dx) & 15;
      const unsigned char *Bc = smem + G::R2 + cur * G::SL2 + (nt * 32 + rr) * 128;
      const unsigned char *Ar[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) Ar[i] = smem + (prow[i] + dy * 18 + dx) * PXB;
      f32x4 fa[2][2], fb[2];
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        const int sl = kk & 1;
#pragma unroll
        for (int i = 0; i < 2; ++i) fa[sl][i] = *reinterpret_cast<const f32x4 *>(Ar[i] + (((q * 8 + 2 * kk + hh) ^ key) * 16));
        fb[sl] = *reinterpret_cast<const f32x4 *>(Bc + (((2 * kk + hh) ^ fsw) * 16));
        if (sl == 0) {
#pragma unroll
          for (int i = 0; i < 2; ++i) acc[i] = bnk_mfma(fa[0][i], fb[0], acc[i]);
        } else {
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            acc[i] = bnk_mfma(fa[0][i], fb[1], acc[i]);
            acc[i] = bnk_mfma(fa[1][i], fb[0], acc[i]);
          }
        }
      }
      cur = cur + 1 == NB2 ? 0 : cur + 1;
      if (++t == 9) { t = 0; ++q; }
    }
    }
    __syncthreads();                                     // mid1 and the conv2 ring are dead
#ifdef IVX_CONV_TIMELINE
    tl2 = __builtin_amdgcn_s_memrealtime();
#endif
    // ---- epilogue: mid2 = s2 * relu(bn2(conv2)), pixel row m = 16 * oy + ox, swizzle key ox
    const int n = nt * 32 + rr;
    const float sA = prm[2 * P + n] * (s2 / s1), tA = prm[3 * P + n] * s2;
    const int ne = n & ~1;
    const int chunk_hi = (ne >> 4) * 4 + ((ne & 15) >> 3);
    const int inchunk = (ne & 7) * 2;
    const bool odd = lane & 1;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int pr = 0; pr < 8; ++pr) {
        const int ra = 2 * pr;
        const int row = (2 * g + i) * 32 + (ra & 3) + 8 * (ra >> 2) + 4 * hh + (odd ? 1 : 0);
        const float ya = __builtin_amdgcn_fmed3f(acc[i][ra] * sA + tA, 0.f, 65504.f), yb = __builtin_amdgcn_fmed3f(acc[i][ra + 1] * sA + tA, 0.f, 65504.f);
        float e0, e1;
        bnk_swap2(ya, yb, odd, e0, e1);
        bnk_put2(smem + row * PXB, chunk_hi ^ (row & 15), inchunk, e0, e1, true);
      }
    }
  }

  // =========================================================================================== phase 3: conv3 + shortcut, 128 columns at a time
  if constexpr (PROJ) {
    // conv3 and the shortcut conv as one GEMM: per unit of 128 output columns the slabs (x chunk 0 .. NQX - 1, mid2 chunk 0 .. NQ - 1) of the joint
    // filter bank go through the two-slot ring; the x part's A fragments are registers (xf), the mid2 part's come from the LDS tile.
#ifdef IVX_CONV_TIMELINE
    tl2b = __builtin_amdgcn_s_memrealtime();
#endif
    constexpr int NQ3 = NQX + NQ, NU = C / 128, S3 = NU * NQ3, TN3 = 2;
    static_assert(NQ3 % 2 == 0, "a unit's first slab lies in ring slot 0");
    unsigned b_off[BR3];
#pragma unroll
    for (int j = 0; j < BR3; ++j) b_off[j] = (unsigned)((lr + RP * j) * NQ3 * 128 + cc * 16);
    auto load3 = [&](const int s, const int buf) {
      const int u = s / NQ3, q = s - u * NQ3;
      unsigned char *Bb = smem + G::R3 + buf * G::SL3 + w * 1024;
#pragma unroll
      for (int j = 0; j < BR3; ++j) {
        const unsigned vo = b_off[j] + (unsigned)((u * 128 * NQ3 + q) * 128);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w3, (lds_ptr_t)(Bb + RP * j * 128), 16, vo, 0, 0, 0);
      }
    };
    const int wm = w >> 1, wn = w & 1;
    load3(0, 0);
    f32x16 acc[2][TN3];
    float *stage = reinterpret_cast<float *>(smem + G::ST3 + w * 2048);
    const int rrow = lane >> 2, c8 = (lane & 3) * 8;
    float omax = 0.f;
    size_t poff[2][2];
    bool pok[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        const int pr = (2 * wm + i) * 32 + hf * 16 + rrow;
        const int gy = y0 + (pr >> 4), gx = x0 + (pr & 15);
        pok[i][hf] = gy < p.H && gx < p.W;
        poff[i][hf] = pok[i][hf] ? (((size_t)b * p.H + gy) * p.W + gx) * (size_t)(2 * C) : (size_t)0;
      }
    const float rho = s2 * inv_in;                       // x slabs accumulate in units of s_in s_w, mid2 slabs in units of s2 s_w
    const float k_acc = s_out / s2;
    auto epilogue3 = [&](const int u) {
#pragma unroll
      for (int j = 0; j < TN3; ++j) {
        const int nb = u * 128 + (wn * TN3 + j) * 32 + c8;
        const int coff = (nb >> 4) * 32 + (nb & 15);
        const f32x4 sc0 = *reinterpret_cast<const f32x4 *>(prm + 4 * P + nb) * k_acc, sc1 = *reinterpret_cast<const f32x4 *>(prm + 4 * P + nb + 4) * k_acc;
        const f32x4 sf0 = *reinterpret_cast<const f32x4 *>(prm + 8 * P + nb) * s_out, sf1 = *reinterpret_cast<const f32x4 *>(prm + 8 * P + nb + 4) * s_out;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
          for (int hf = 0; hf < 2; ++hf) {
#pragma unroll
            for (int r8 = 0; r8 < 8; ++r8) {
              const int rowh = (r8 & 3) + 8 * (r8 >> 2) + 4 * hh;
              stage[rowh * 32 + ((((rr >> 2) ^ ((rowh >> 1) & 1)) << 2) | (rr & 3))] = acc[i][j][hf * 8 + r8];
            }
            const int sw = (rrow >> 1) & 1;
            f32x4 v0 = *reinterpret_cast<const f32x4 *>(stage + rrow * 32 + (((c8 >> 2) ^ sw) << 2));
            f32x4 v1 = *reinterpret_cast<const f32x4 *>(stage + rrow * 32 + ((((c8 >> 2) + 1) ^ sw) << 2));
            if (pok[i][hf]) {
              v0 = v0 * sc0 + sf0;
              v1 = v1 * sc1 + sf1;
              f16x8 oh, ol;
#pragma unroll
              for (int e = 0; e < 8; e += 2) {
                f32x2 x = {e < 4 ? v0[e & 3] : v1[e & 3], e < 4 ? v0[(e & 3) + 1] : v1[(e & 3) + 1]};
                x[0] = __builtin_amdgcn_fmed3f(x[0], 0.f, 65504.f);
                x[1] = __builtin_amdgcn_fmed3f(x[1], 0.f, 65504.f);
                omax = fmaxf(omax, fmaxf(x[0], x[1]));
                const f16x2 h = __builtin_convertvector(x, f16x2);
                const f16x2 l = __builtin_convertvector(x - __builtin_convertvector(h, f32x2), f16x2);
                oh[e] = h[0]; oh[e + 1] = h[1];
                ol[e] = l[0]; ol[e + 1] = l[1];
              }
              *reinterpret_cast<f16x8 *>(p.out + poff[i][hf] + coff) = oh;
              *reinterpret_cast<f16x8 *>(p.out + poff[i][hf] + coff + 16) = ol;
            }
          }
        }
      }
    };
    int s = 0;
    for (int u = 0; u < NU; ++u) {
#pragma unroll
      for (int q = 0; q < NQ3; ++q, ++s) {
        bnk_wait_vm<0>();                                // slab s (and the stores of the previous unit's epilogue)
        bnk_barrier();                                   // (first pass: also publishes mid2)
        if (s + 1 < S3) load3(s + 1, (q & 1) ^ 1);
        if (q == 0) {
          if (u > 0) epilogue3(u - 1);
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < TN3; ++j)
#pragma unroll
              for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        }
        if (q == NQX) {
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < TN3; ++j) acc[i][j] = acc[i][j] * rho;
        }
        const unsigned char *Bc = smem + G::R3 + (q & 1) * G::SL3 + (wn * TN3 * 32 + rr) * 128;
        const unsigned char *Ac = smem + ((2 * wm) * 32 + rr) * PXB;
        const int key = rr & 15;
        f32x4 fa[2][2], fb[2][TN3];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
          const int sl = kk & 1;
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            if (q < NQX) fa[sl][i] = xf[i][q < NQX ? q : 0][kk];
            else fa[sl][i] = *reinterpret_cast<const f32x4 *>(Ac + i * 32 * PXB + ((((q - NQX) * 8 + 2 * kk + hh) ^ key) * 16));
          }
#pragma unroll
          for (int j = 0; j < TN3; ++j) fb[sl][j] = *reinterpret_cast<const f32x4 *>(Bc + j * 4096 + (((2 * kk + hh) ^ fsw) * 16));
          if (sl == 0) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
              for (int j = 0; j < TN3; ++j) acc[i][j] = bnk_mfma(fa[0][i], fb[0][j], acc[i][j]);
          } else {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
              for (int j = 0; j < TN3; ++j) {
                acc[i][j] = bnk_mfma(fa[0][i], fb[1][j], acc[i][j]);
                acc[i][j] = bnk_mfma(fa[1][i], fb[0][j], acc[i][j]);
              }
          }
        }
      }
    }
#ifdef IVX_CONV_TIMELINE
    tl3 = __builtin_amdgcn_s_memrealtime();
#endif
    epilogue3(NU - 1);
    if (p.amax_out) {
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) omax = fmaxf(omax, __shfl_xor(omax, o));
      __syncthreads();
      if (lane == 0) stage[0] = omax;
      __syncthreads();
      if (w == 0) ivx_amax_commit(p.amax_out, lane < NW ? reinterpret_cast<const float *>(smem + G::ST3)[lane * 512] * (1.0f / s_out) : 0.f, (int)blockIdx.x);
    }
  } else
  {
#ifdef IVX_CONV_TIMELINE
    tl2b = __builtin_amdgcn_s_memrealtime();
#endif
    constexpr int S3 = NQ * NQ;                          // (unit u, chunk q)
    constexpr int TN3 = P == 64 ? 2 : 1;
    unsigned b_off[BR3];
#pragma unroll
    for (int j = 0; j < BR3; ++j) b_off[j] = (unsigned)((lr + RP * j) * NQ * 128 + cc * 16);
    auto load3 = [&](const int s, const int buf) {
      const int u = s / NQ, q = s - u * NQ;
      unsigned char *Bb = smem + G::R3 + buf * G::SL3 + w * 1024;
#pragma unroll
      for (int j = 0; j < BR3; ++j) {
        const unsigned vo = b_off[j] + (unsigned)((u * 128 * NQ + q) * 128);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w3, (lds_ptr_t)(Bb + RP * j * 128), 16, vo, 0, 0, 0);
      }
    };
    const int wm = P == 64 ? (w >> 1) : (w >> 2), wn = P == 64 ? (w & 1) : (w & 3);
#pragma unroll
    for (int s = 0; s < NB3 - 1; ++s) load3(s, s);
    f32x16 acc[2][TN3];
    // Epilogue: a wave transposes one half tile (16 rows x 32 channels) at a time through its 2 KB stage so that a lane owns 8 consecutive
    // channels of a row: 16 bytes of hi halves and 16 bytes of lo halves -- one 16-byte load each for the shortcut, one 16-byte store each for
    // the output.  Stage rows are 128 bytes; 16-byte slot c of row r lies at c ^ ((r >> 1) & 1) (ds_read_b128 lane groups hit 4 rows).
    // The shortcut (the block's input at the tile's own pixels) of unit u is REQUESTED before the unit's K loop and decoded after it: with
    // the request in the epilogue a wave ran 16 load -> store round trips in series per unit.
    float *stage = reinterpret_cast<float *>(smem + G::ST3 + w * 2048);
    const int rrow = lane >> 2, c8 = (lane & 3) * 8;
    float omax = 0.f;
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    u32x4 rhi[2][TN3][2], rlo[2][TN3][2];
    size_t poff[2][2];                                  // element offset of the pixel row of (tile i, half hf) in the pair tensor
    bool pok[2][2];                                     // ... inside the image (outside: the shortcut is requested from pixel 0 -- the NUMBER of
                                                        // requests per unit is what the counted wait below relies on -- and nothing is stored)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        const int pr = (2 * wm + i) * 32 + hf * 16 + rrow;
        const int gy = y0 + (pr >> 4), gx = x0 + (pr & 15);
        pok[i][hf] = gy < p.H && gx < p.W;
        poff[i][hf] = pok[i][hf] ? (((size_t)b * p.H + gy) * p.W + gx) * (size_t)(2 * C) : (size_t)0;
      }
    auto res_request = [&](const int u) {
#pragma unroll
      for (int j = 0; j < TN3; ++j) {
        const int nb = u * 128 + (wn * TN3 + j) * 32 + c8;
        const int coff = (nb >> 4) * 32 + (nb & 15);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int hf = 0; hf < 2; ++hf) {
            rhi[i][j][hf] = *reinterpret_cast<const u32x4 *>(p.in + poff[i][hf] + coff);
            rlo[i][j][hf] = *reinterpret_cast<const u32x4 *>(p.in + poff[i][hf] + coff + 16);
          }
      }
    };
    const float k_acc = s_out / s2, k_res = inv_in * s_out;        // powers of two: the output scale is folded into the epilogue's multipliers
    auto epilogue3 = [&](const int u) {
#pragma unroll
      for (int j = 0; j < TN3; ++j) {
        const int nb = u * 128 + (wn * TN3 + j) * 32 + c8;
        const int coff = (nb >> 4) * 32 + (nb & 15);
        const f32x4 sc0 = *reinterpret_cast<const f32x4 *>(prm + 4 * P + nb) * k_acc, sc1 = *reinterpret_cast<const f32x4 *>(prm + 4 * P + nb + 4) * k_acc;
        const f32x4 sf0 = *reinterpret_cast<const f32x4 *>(prm + 8 * P + nb) * s_out, sf1 = *reinterpret_cast<const f32x4 *>(prm + 8 * P + nb + 4) * s_out;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
          for (int hf = 0; hf < 2; ++hf) {
#pragma unroll
            for (int r8 = 0; r8 < 8; ++r8) {
              const int rowh = (r8 & 3) + 8 * (r8 >> 2) + 4 * hh;
              stage[rowh * 32 + ((((rr >> 2) ^ ((rowh >> 1) & 1)) << 2) | (rr & 3))] = acc[i][j][hf * 8 + r8];
            }
            const int sw = (rrow >> 1) & 1;
            f32x4 v0 = *reinterpret_cast<const f32x4 *>(stage + rrow * 32 + (((c8 >> 2) ^ sw) << 2));
            f32x4 v1 = *reinterpret_cast<const f32x4 *>(stage + rrow * 32 + ((((c8 >> 2) + 1) ^ sw) << 2));
            if (pok[i][hf]) {
              const f16x8 h8 = __builtin_bit_cast(f16x8, rhi[i][j][hf]), l8 = __builtin_bit_cast(f16x8, rlo[i][j][hf]);
              v0 = v0 * sc0 + sf0;
              v1 = v1 * sc1 + sf1;
              f16x8 oh, ol;
#pragma unroll
              for (int e = 0; e < 8; e += 2) {
                f32x2 x = {e < 4 ? v0[e & 3] : v1[e & 3], e < 4 ? v0[(e & 3) + 1] : v1[(e & 3) + 1]};
                x = f32x2{(float)h8[e], (float)h8[e + 1]} * k_res + x;
                x = f32x2{(float)l8[e], (float)l8[e + 1]} * k_res + x;
                x[0] = __builtin_amdgcn_fmed3f(x[0], 0.f, 65504.f);
                x[1] = __builtin_amdgcn_fmed3f(x[1], 0.f, 65504.f);
                omax = fmaxf(omax, fmaxf(x[0], x[1]));
                const f16x2 h = __builtin_convertvector(x, f16x2);
                const f16x2 l = __builtin_convertvector(x - __builtin_convertvector(h, f32x2), f16x2);
                oh[e] = h[0]; oh[e + 1] = h[1];
                ol[e] = l[0]; ol[e + 1] = l[1];
              }
              *reinterpret_cast<f16x8 *>(p.out + poff[i][hf] + coff) = oh;
              *reinterpret_cast<f16x8 *>(p.out + poff[i][hf] + coff + 16) = ol;
            }
          }
        }
      }
    };
    constexpr int NRES = 2 * TN3 * 2 * 2;                // shortcut requests of one unit per lane
    int cur = 0, u = 0, q = 0;
    for (int s = 0; s < S3; ++s) {
      // slab s was requested one pass ago, right after that pass's barrier; only a unit's shortcut requests can be younger (loads return in
      // order, so "at most NRES outstanding" implies the slab has landed whatever the epilogue's stores are doing)
      if (NB3 == 2 && q == 1) bnk_wait_vm<NRES>();
      else bnk_wait_vm<0>();
      bnk_barrier();                                     // (first pass: also publishes mid2)
      if (s + NB3 - 1 < S3) load3(s + NB3 - 1, cur == 0 ? NB3 - 1 : cur - 1);
      if (q == 0) {
        if (u > 0) epilogue3(u - 1);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < TN3; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        res_request(u);
      }
      const unsigned char *Bc = smem + G::R3 + cur * G::SL3 + (wn * TN3 * 32 + rr) * 128;
      const unsigned char *Ac = smem + ((2 * wm) * 32 + rr) * PXB;
      const int key = rr & 15;
      f32x4 fa[2][2], fb[2][TN3];
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        const int sl = kk & 1;
#pragma unroll
        for (int i = 0; i < 2; ++i) fa[sl][i] = *reinterpret_cast<const f32x4 *>(Ac + i * 32 * PXB + (((q * 8 + 2 * kk + hh) ^ key) * 16));
#pragma unroll
        for (int j = 0; j < TN3; ++j) fb[sl][j] = *reinterpret_cast<const f32x4 *>(Bc + j * 4096 + (((2 * kk + hh) ^ fsw) * 16));
        if (sl == 0) {
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < TN3; ++j) acc[i][j] = bnk_mfma(fa[0][i], fb[0][j], acc[i][j]);
        } else {
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < TN3; ++j) {
              acc[i][j] = bnk_mfma(fa[0][i], fb[1][j], acc[i][j]);
              acc[i][j] = bnk_mfma(fa[1][i], fb[0][j], acc[i][j]);
            }
        }
      }
      cur = cur + 1 == NB3 ? 0 : cur + 1;
      if (++q == NQ) { q = 0; ++u; }
    }
#ifdef IVX_CONV_TIMELINE
    tl3 = __builtin_amdgcn_s_memrealtime();
#endif
    epilogue3(NQ - 1);
    if (p.amax_out) {        // one atomic per workgroup (ivx_common.h); the stage slices are per wave, word 0 of each is free now
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) omax = fmaxf(omax, __shfl_xor(omax, o));
      __syncthreads();
      if (lane == 0) stage[0] = omax;
      __syncthreads();
      if (w == 0) ivx_amax_commit(p.amax_out, lane < NW ? reinterpret_cast<const float *>(smem + G::ST3)[lane * 512] * (1.0f / s_out) : 0.f, (int)blockIdx.x);
    }
  }
#ifdef IVX_CONV_TIMELINE
  if (p.tl && tid == 0) {
    unsigned long long *t = p.tl + (size_t)blockIdx.x * 8;
    t[0] = tl0; t[1] = tl1; t[2] = tl1b; t[3] = tl2; t[4] = tl2b; t[5] = tl3; t[6] = __builtin_amdgcn_s_memrealtime();
    t[7] = (unsigned long long)__builtin_amdgcn_s_getreg(63492) | ((unsigned long long)__builtin_amdgcn_s_getreg(63508) << 32);
  }
#endif
}

#ifdef IVX_CONV_TIMELINE
static unsigned long long *g_bnk_timeline = nullptr;
extern "C" int ivx_bottleneck_set_timeline(void *buf) { g_bnk_timeline = (unsigned long long *)buf; return 0; }
#endif

extern "C" int ivx_bottleneck_supported(const ivx_bottleneck_desc *d) {
  if (!d || d->B <= 0 || d->H <= 0 || d->W <= 0) return 0;
  if (d->P != 64 && d->P != 128) return 0;
  if ((int64_t)d->B * d->H * d->W * d->P * 16 >= (1LL << 31)) return 0;      // the 4P-channel pair tensor below 2 GiB (32-bit buffer offsets)
  return 1;
}

extern "C" int ivx_bottleneck_fwd_pio(const ivx_bottleneck_desc *d, const ivx_bottleneck_io *io, const void *in, const void *w1, const float *scale1,
                                      const float *shift1, const void *w2, const float *scale2, const float *shift2, const void *w3,
                                      const float *scale3, const float *shift3, void *out, ivx_stream_t stream) {
  IVX_REQUIRE(d && io && in && w1 && w2 && w3 && out && scale1 && shift1 && scale2 && shift2 && scale3 && shift3, "ivx_bottleneck_fwd_pio: null argument");
  IVX_REQUIRE(ivx_bottleneck_supported(d), "ivx_bottleneck_fwd_pio: planes must be 64 or 128 and the [B, H, W, 4 * planes] pair tensor below 2 GiB "
              "(B %d H %d W %d planes %d)", d->B, d->H, d->W, d->P);
  IVX_REQUIRE(io->in_scale && io->amax_in && io->out_scale, "ivx_bottleneck_fwd_pio: in_scale, amax_in and out_scale are required");
  IVX_REQUIRE(in != out, "ivx_bottleneck_fwd_pio: the block cannot run in place (the shortcut re-reads the input)");
  BnkParams p;
  p.in = (const _Float16 *)in; p.out = (_Float16 *)out;
  p.w1 = (const _Float16 *)w1; p.w2 = (const _Float16 *)w2; p.w3 = (const _Float16 *)w3;
  p.sc1 = scale1; p.sh1 = shift1; p.sc2 = scale2; p.sh2 = shift2; p.sc3 = scale3; p.sh3 = shift3;
  p.in_scale_p = io->in_scale; p.amax_in = io->amax_in; p.out_scale_p = io->out_scale; p.amax_out = io->amax_out;
  p.wb1 = io->wbound[0]; p.sb1 = io->sbound[0]; p.wb2 = io->wbound[1]; p.sb2 = io->sbound[1]; p.wb3 = io->wbound[2]; p.sb3 = io->sbound[2];
  p.wbd = 1.0f;
  p.B = d->B; p.H = d->H; p.W = d->W;
  p.tiles_x = (d->W + 15) / 16; p.tiles_y = (d->H + 7) / 8;
  p.n_tiles = d->B * p.tiles_x * p.tiles_y;
  p.q_total = (p.n_tiles + 7) / 8;
#ifdef IVX_CONV_TIMELINE
  p.tl = g_bnk_timeline;
#endif
  const int P = d->P, C = 4 * P;
  const unsigned in_bytes = (unsigned)((int64_t)d->B * d->H * d->W * C * 4);
  const unsigned w1_bytes = (unsigned)(P * C * 4), w2_bytes = (unsigned)(P * 9 * P * 4), w3_bytes = (unsigned)(C * P * 4);
  const dim3 grid((unsigned)(8 * p.q_total));
  hipStream_t st = (hipStream_t)stream;
  // (measured and removed, profiles/r06_fused_bottleneck.md (c): conv2's filters straight from L2 into registers, in the chain's layout and in
  // fragment order -- no gain over the LDS ring)
  if (P == 64) hipLaunchKernelGGL((bottleneck_pio_kernel<64, 256, 2, 3, 2>), grid, dim3(256), 0, st, p, in_bytes, w1_bytes, w2_bytes, w3_bytes);
  else hipLaunchKernelGGL((bottleneck_pio_kernel<128, 512, 3, 3, 2>), grid, dim3(512), 0, st, p, in_bytes, w1_bytes, w2_bytes, w3_bytes);
  IVX_CHECK_LAUNCH("ivx_bottleneck_fwd_pio");
  return IVX_OK;
}

// ---- the first block of stage 1 (shortcut = 1x1 conv + BN of the input, stride 1): Cin = P = 64 -> 4P
extern "C" int ivx_bottleneck_proj_supported(const ivx_bottleneck_desc *d, int32_t Cin) {
  if (!d || d->B <= 0 || d->H <= 0 || d->W <= 0) return 0;
  if (d->P != 64 || Cin != 64) return 0;
  if ((int64_t)d->B * d->H * d->W * d->P * 16 >= (1LL << 31)) return 0;      // the 4P-channel pair output below 2 GiB
  return 1;
}

extern "C" int ivx_bottleneck_proj_fwd_pio(const ivx_bottleneck_desc *d, int32_t Cin, const ivx_bottleneck_io *io, float wbound_shortcut, const void *in,
                                           const void *w1, const float *scale1, const float *shift1, const void *w2, const float *scale2,
                                           const float *shift2, const void *w3d, const float *scale3d, const float *shift3d, void *out,
                                           ivx_stream_t stream) {
  IVX_REQUIRE(d && io && in && w1 && w2 && w3d && out && scale1 && shift1 && scale2 && shift2 && scale3d && shift3d, "ivx_bottleneck_proj_fwd_pio: null argument");
  IVX_REQUIRE(ivx_bottleneck_proj_supported(d, Cin), "ivx_bottleneck_proj_fwd_pio: built for planes = Cin = 64 and a [B, H, W, 256] pair output below 2 GiB "
              "(B %d H %d W %d planes %d Cin %d)", d->B, d->H, d->W, d->P, Cin);
  IVX_REQUIRE(io->in_scale && io->amax_in && io->out_scale, "ivx_bottleneck_proj_fwd_pio: in_scale, amax_in and out_scale are required");
  IVX_REQUIRE(wbound_shortcut >= 0.f, "ivx_bottleneck_proj_fwd_pio: negative shortcut bound");
  BnkParams p;
  p.in = (const _Float16 *)in; p.out = (_Float16 *)out;
  p.w1 = (const _Float16 *)w1; p.w2 = (const _Float16 *)w2; p.w3 = (const _Float16 *)w3d;
  p.sc1 = scale1; p.sh1 = shift1; p.sc2 = scale2; p.sh2 = shift2; p.sc3 = scale3d; p.sh3 = shift3d;
  p.in_scale_p = io->in_scale; p.amax_in = io->amax_in; p.out_scale_p = io->out_scale; p.amax_out = io->amax_out;
  p.wb1 = io->wbound[0]; p.sb1 = io->sbound[0]; p.wb2 = io->wbound[1]; p.sb2 = io->sbound[1]; p.wb3 = io->wbound[2]; p.sb3 = io->sbound[2];
  p.wbd = wbound_shortcut;
  p.B = d->B; p.H = d->H; p.W = d->W;
  p.tiles_x = (d->W + 15) / 16; p.tiles_y = (d->H + 7) / 8;
  p.n_tiles = d->B * p.tiles_x * p.tiles_y;
  p.q_total = (p.n_tiles + 7) / 8;
#ifdef IVX_CONV_TIMELINE
  p.tl = g_bnk_timeline;
#endif
  const int P = d->P, C = 4 * P;
  const unsigned in_bytes = (unsigned)((int64_t)d->B * d->H * d->W * Cin * 4);
  const unsigned w1_bytes = (unsigned)(P * Cin * 4), w2_bytes = (unsigned)(P * 9 * P * 4), w3_bytes = (unsigned)(C * (Cin + P) * 4);
  const dim3 grid((unsigned)(8 * p.q_total));
  hipLaunchKernelGGL((bottleneck_pio_kernel<64, 64, 2, 3, 2>), grid, dim3(256), 0, (hipStream_t)stream, p, in_bytes, w1_bytes, w2_bytes, w3_bytes);
  IVX_CHECK_LAUNCH("ivx_bottleneck_proj_fwd_pio");
  return IVX_OK;
}
