// Two-waves-per-SIMD form of the fused Winograd GEMM + output transform of an F(4x4,3x3) layer (round-5 verdict, item 3; the one-wave form is
// conv_wino_fold4_kernel in conv_igemm.hip, same contract: ivx_conv_winograd_gemm_output_amax).  Replaces winograd.hip's GEMM + output stages
// for the stride-1, pad-1 ResModule convolutions of the stack necks (reference mmdet3d/models/necks/imvoxelnet.py:94-123,191-230).
//
//   Tile       64 staged rows (tile column, z) x 64 output channels per workgroup, 62 rows stored (two rows of z-halo overlap), ALL 36 frequency
//              points: for every xi the z-halo K loop leaves M[xi] of the tile on chip, folded into the output domain right away
//              (P[e] += At[e][j] M;  out[a][e] += At[a][i] P[e]) -- M never reaches HBM.
//   Waves      8 per workgroup (two per SIMD): wave (wr, wc) owns rows 16 wr .. + 15 and columns 32 wc .. + 31 as two 16 x 16 tiles of
//              v_mfma_f32_16x16x32_f16: 4 accumulator registers per tile, so the 16 + 4 + 1 output-domain / row / product tile sets are
//              168 registers per lane instead of the 320 that pinned the one-wave form to one wave per SIMD (instruction-bound: 1.26 ms).
//   Operands   one MFMA covers a whole 16-channel pair group (K = 32 halves = [hi16 | lo16]):  A = [a_hi | a_lo] (ONE 16-byte LDS read per
//              lane), B1 = [b_hi | b_hi] -> a_hi b_hi + a_lo b_hi;  B2 = [b_lo | 0] -> a_hi b_lo.  Two MFMAs instead of three products of
//              half the K: a quarter of the second MFMA multiplies zeros (read from the slot's always-zero row), but the LDS read port, not the
//              matrix pipe, bounds this kernel (15 reads of 1 KB per 12 MFMAs of 16 cycles).
//   Staging    one ring of NBUF slots, a slot = one (xi, 16-channel group): 64 rows of V[xi] + the three taps' 64 filter rows of U[xi] (16 KB),
//              LDS-DMA, 2 requests per lane and group, raw barriers with counted vmcnt waits (one barrier per group).
#include "ivx_common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

struct Fold4wParams {
  const _Float16 *V, *U;       // 36 planes of V [M][Cin stored halves] (plane stride vs halves), 36 filter banks [Cout][K = 3 Cin halves] (bank stride us)
  long long vs, us;
  int M, Cin, Cout, K, Z;      // Cin: stored halves per row (2 x real channels); Z: rows per tile column
  int q_total;
  IvxWinoFold f;
};

__constant__ float kFold4wAt[4][6] = {{1.f, 1.f, 1.f, 1.f, 1.f, 0.f}, {0.f, 1.f, -1.f, 2.f, -2.f, 0.f}, {0.f, 1.f, 1.f, 4.f, 4.f, 0.f}, {0.f, 1.f, -1.f, 8.f, -8.f, 1.f}};

__device__ __forceinline__ float fold4w_vscale(const float amax) {      // == winograd.hip wino_pair_vscale
  if (!(amax < 3.0e38f)) return 0.00390625f;
  if (!(amax > 0.f)) return 1.0f;
  int e;
  (void)frexpf(amax, &e);
  int k = 15 - 8 - e;
  k = k < -120 ? -120 : (k > 120 ? 120 : k);
  return ldexpf(1.0f, k);
}

template <int N>
__device__ __forceinline__ void fold4w_wait_vm() { __builtin_amdgcn_s_waitcnt(0x0f70 | (N & 15) | ((N >> 4) << 14)); }

template <int NBUF>
__global__ __launch_bounds__(512, 2) void conv_wino_fold4w_kernel(const Fold4wParams p, const unsigned in_bytes, const unsigned w_bytes) {
  constexpr int BM = 64, BN = 64, BMO = BM - 2;
  constexpr int ABYTES = (BM + 1) * 64, BBYTES = 3 * BN * 64, SLOT = ABYTES + BBYTES;      // 4160 + 12288
  constexpr int ZROW = BM * 64;                     // byte offset of the slot's zero row
  constexpr int NXI = 36;
  static_assert(NBUF >= 3 && NBUF <= 9, "ring depth (vmcnt immediates)");
  __shared__ __attribute__((aligned(16))) unsigned char smem[NBUF * SLOT];
  __shared__ float wmax[8];
  static_assert(sizeof(smem) >= 8 * 2048, "2 KB of staging LDS per wave for the transposed epilogue");
  typedef __attribute__((address_space(3))) void *lds_ptr_t;
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = w >> 1, wc = w & 1;
  int mt, nt;
  {
    const int Nt = (p.Cout + BN - 1) / BN;
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const int lt = idx / Nt;
    nt = idx - lt * Nt;
    mt = xcd * p.q_total + lt;
  }
  const IvxWinoFold &f = p.f;
  if (mt * BMO >= p.M) {
    if (f.pmax && tid == 0) f.pmax[blockIdx.x] = 0.f;
    return;
  }
  const int m0 = mt * BMO, n0 = nt * BN;
  const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc((void *)p.V, 0, in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void *)p.U, 0, w_bytes, 0x00020000);
  const unsigned OOB = 0x80000000u;
  const int G = p.Cin / 32;                         // 16-channel pair groups per xi
  const int NIT = NXI * G;
  // ---- DMA lanes: two requests per lane and group.  Request 0: lanes 0 .. 255 the 64 rows of A, lanes 256 .. 511 tap 0 of B; request 1: taps 1, 2.
  unsigned vo0, vo1;                                // per-lane byte offsets inside plane 0 / bank 0, group 0 (or OOB)
  bool r0_is_a;
  {
    const int c = tid & 255, row = c >> 2, s = c & 3, kc = s ^ ((row >> 2) & 3);
    r0_is_a = w < 4;                                // (wave-uniform: lanes 0 .. 255 = waves 0 .. 3)
    const int arow = m0 - 1 + row;
    const unsigned a_vo = (arow >= 0 && arow < p.M) ? ((unsigned)arow * (unsigned)p.Cin + kc * 8) * 2u : OOB;
    const bool nok = n0 + row < p.Cout;
    const unsigned b_row = ((unsigned)(n0 + row) * (unsigned)p.K + kc * 8) * 2u;
    vo0 = r0_is_a ? a_vo : (nok ? b_row : OOB);                                     // tap 0
    vo1 = nok ? b_row + (unsigned)((tid < 256 ? 1 : 2) * 64 * 2) : OOB;            // tap 1 (lanes 0 .. 255) / tap 2
  }
  const unsigned xi_in = (unsigned)(p.vs * 2), xi_w = (unsigned)(p.us * 2);
  auto issue = [&](const int it, const int slot) {
    const int xi = it / G, g = it - xi * G;
    const unsigned ka = (unsigned)xi * xi_in + (unsigned)g * 64u;
    const unsigned kb = (unsigned)xi * xi_w + (unsigned)(((g >> 1) * 3 * 64 + (g & 1) * 32) * 2);
    unsigned char *base = smem + slot * SLOT;
    const unsigned v0 = vo0 == OOB ? OOB : vo0 + (r0_is_a ? ka : kb);
    const unsigned v1 = vo1 == OOB ? OOB : vo1 + kb;
    // request 0 lands at chunk tid of [A rows 0 .. 63 | B tap 0], request 1 at chunk 256 + tid of the B area (taps 1, 2)
    unsigned char *d0 = r0_is_a ? base + w * 1024 : base + ABYTES + (w - 4) * 1024;
    unsigned char *d1 = base + ABYTES + (4 + w) * 1024;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r0_is_a ? rs_in : rs_w, (lds_ptr_t)d0, 16, v0, 0, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_ptr_t)d1, 16, v1, 0, 0, 0);
  };
  // zero rows of the slots
  for (int t = tid; t < NBUF * 16; t += 512) reinterpret_cast<float *>(smem + (t >> 4) * SLOT + ZROW)[t & 15] = 0.f;
  // ---- fragment addresses (bytes inside a slot)
  const int rl = lane & 15, kq = lane >> 4;
  const int Z = p.Z;
  const int orow = wr * 16 + rl;                    // this lane's A row of tap 1 = its output row of the tile
  const int zl = (m0 + orow) % Z;
  unsigned aoff[3];
#pragma unroll
  for (int kz = 0; kz < 3; ++kz) {
    const int ar = orow + kz;                       // slot row (row 0 is plane row m0 - 1)
    const bool ok = ar < BM && (kz == 0 ? zl >= 1 : (kz == 2 ? zl + 1 < Z : true));
    aoff[kz] = ok ? (unsigned)(ar * 64 + ((kq ^ ((ar >> 2) & 3)) * 16)) : (unsigned)(ZROW + kq * 16);
  }
  unsigned b1off[2], b2off[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int n = wc * 32 + j * 16 + rl, sw = (n >> 2) & 3;
    b1off[j] = (unsigned)(ABYTES + n * 64 + (((kq & 1) ^ sw) * 16));                       // [b_hi | b_hi]
    b2off[j] = kq < 2 ? (unsigned)(ABYTES + n * 64 + (((2 + kq) ^ sw) * 16)) : 0xffffffffu;   // [b_lo | 0]: the upper k half reads the zero row
  }
  f32x4 acc[2], P[4][2], out[4][4][2];
  const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int j = 0; j < 2; ++j) acc[j] = z4;
#pragma unroll
  for (int e = 0; e < 4; ++e)
#pragma unroll
    for (int j = 0; j < 2; ++j) P[e][j] = z4;
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int j = 0; j < 2; ++j) out[a][e][j] = z4;
  // ---- ring: groups it .. it + NBUF - 2 in flight while group it is multiplied
#pragma unroll
  for (int k = 0; k < NBUF - 1; ++k) issue(k, k);        // (NIT >= 36 * 2 > NBUF)
  int cur = 0, it = 0;
  for (int i = 0; i < 6; ++i) {
    for (int jj = 0; jj < 6; ++jj) {
      for (int gq = 0; gq < G; ++gq, ++it) {
        int newer = NIT - 1 - it;
        newer = newer > NBUF - 2 ? NBUF - 2 : newer;
        switch (newer) {                            // this lane's two requests of group `it` have landed when only the younger groups' are outstanding
          case 7: fold4w_wait_vm<14>(); break;
          case 6: fold4w_wait_vm<12>(); break;
          case 5: fold4w_wait_vm<10>(); break;
          case 4: fold4w_wait_vm<8>(); break;
          case 3: fold4w_wait_vm<6>(); break;
          case 2: fold4w_wait_vm<4>(); break;
          case 1: fold4w_wait_vm<2>(); break;
          default: fold4w_wait_vm<0>(); break;
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);        // lgkmcnt(0): (first pass) the zero-row stores
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_barrier();               // group `it` is visible to every wave; every wave is done with the slot of group it - 1
        asm volatile("" ::: "memory");
        if (it + NBUF - 1 < NIT) issue(it + NBUF - 1, cur == 0 ? NBUF - 1 : cur - 1);
        const unsigned char *sl = smem + cur * SLOT;
#pragma unroll
        for (int kz = 0; kz < 3; ++kz) {
          const f16x8 a = *reinterpret_cast<const f16x8 *>(sl + aoff[kz]);
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const f16x8 b1 = *reinterpret_cast<const f16x8 *>(sl + b1off[j] + kz * (BN * 64));
            const f16x8 b2 = *reinterpret_cast<const f16x8 *>(sl + (b2off[j] == 0xffffffffu ? (unsigned)ZROW : b2off[j] + kz * (BN * 64)));
            acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b1, acc[j], 0, 0, 0);
            acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b2, acc[j], 0, 0, 0);
          }
        }
        cur = cur + 1 == NBUF ? 0 : cur + 1;
      }
      // M[xi] of this tile is complete (xi = 6 i + jj): fold it into the row accumulators
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float c = kFold4wAt[e][jj];
        if (c != 0.f) {
#pragma unroll
          for (int j = 0; j < 2; ++j) P[e][j] = c * acc[j] + P[e][j];
        }
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[j] = z4;
    }
    // the six jj of row i are in: out[a][e] += At[a][i] * P[e]
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const float c = kFold4wAt[a][i];
      if (c != 0.f) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int j = 0; j < 2; ++j) out[a][e][j] = c * P[e][j] + out[a][e][j];
      }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int j = 0; j < 2; ++j) P[e][j] = z4;
  }
  __syncthreads();                                  // the staging area of the epilogue overlaps the ring
  // ---- epilogue: out[a][e] is the (4 tx + a, 4 ty + e) output of the tile rows; v = act(M * mscale * scale + shift [+ res]) as wino_output_kernel.
  // A wave transposes its 16 x 32 block through 2 KB of LDS so that a lane owns 8 consecutive channels of a row (two 16-byte stores).
  float *stage = reinterpret_cast<float *>(smem + w * 2048);
  const float mscale = 1.0f / (fold4w_vscale(__uint_as_float(f.hdr_v[0])) * f.uscale[0]);
  const int rrow = lane >> 2, c8 = (lane & 3) * 8;
  const int nb = n0 + wc * 32 + c8;
  const bool nok = nb < p.Cout;                     // (Cout % 8 == 0 is checked by the launcher)
  f32x4 sc0 = {mscale, mscale, mscale, mscale}, sc1 = sc0, sf0 = z4, sf1 = z4;
  if (nok && f.scale) { sc0 = mscale * *reinterpret_cast<const f32x4 *>(f.scale + nb); sc1 = mscale * *reinterpret_cast<const f32x4 *>(f.scale + nb + 4); }
  if (nok && f.shift) { sf0 = *reinterpret_cast<const f32x4 *>(f.shift + nb); sf1 = *reinterpret_cast<const f32x4 *>(f.shift + nb + 4); }
  size_t base = 0;
  int xlim = 0, ylim = 0;
  {
    const int o = wr * 16 + rrow, m = m0 + o;
    if (o < BMO && m < p.M && nok) {
      const int z = m % Z, col = m / Z;
      const int ty = col % f.TY, t2 = col / f.TY;
      const int tx = t2 % f.TX, b = t2 / f.TX;
      base = ((((size_t)b * f.Xo + 4 * tx) * f.Yo + 4 * ty) * Z + z) * (size_t)f.Co + nb;
      xlim = f.Xo - 4 * tx;
      ylim = f.Yo - 4 * ty;
    }
  }
  const size_t ystep = (size_t)Z * f.Co, xstep = (size_t)f.Yo * ystep;
  float omax = 0.f;
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) stage[(4 * kq + r) * 32 + j * 16 + rl] = out[a][e][j][r];
      f32x4 v0 = *reinterpret_cast<const f32x4 *>(stage + rrow * 32 + c8);
      f32x4 v1 = *reinterpret_cast<const f32x4 *>(stage + rrow * 32 + c8 + 4);
      if (a < xlim && e < ylim) {
        const size_t o = base + a * xstep + e * ystep;
        v0 = v0 * sc0 + sf0;
        v1 = v1 * sc1 + sf1;
        f32x4 r0 = z4, r1 = z4;
        if (f.res_mode) { r0 = *reinterpret_cast<const f32x4 *>(f.res + o); r1 = *reinterpret_cast<const f32x4 *>(f.res + o + 4); }
        if (f.res_mode && !f.res_after_act) { v0 += r0; v1 += r1; }
        if (f.relu) {
#pragma unroll
          for (int q = 0; q < 4; ++q) { v0[q] = v0[q] > 0.f ? v0[q] : 0.f; v1[q] = v1[q] > 0.f ? v1[q] : 0.f; }
        }
        if (f.res_mode && f.res_after_act) { v0 += r0; v1 += r1; }
        v0 *= f.post_scale;
        v1 *= f.post_scale;
        *reinterpret_cast<f32x4 *>(f.out + o) = v0;
        *reinterpret_cast<f32x4 *>(f.out + o + 4) = v1;
#pragma unroll
        for (int q = 0; q < 4; ++q) omax = fmaxf(omax, fmaxf(fabsf(v0[q]), fabsf(v1[q])));
      }
    }
  if (f.pmax) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) omax = fmaxf(omax, __shfl_xor(omax, o));
    if (lane == 0) wmax[w] = omax;
    __syncthreads();
    if (tid == 0) {
      float m = wmax[0];
#pragma unroll
      for (int k = 1; k < 8; ++k) m = fmaxf(m, wmax[k]);
      f.pmax[blockIdx.x] = m;
    }
  }
}

// Same tile geometry as the one-wave form: ivx_conv_fold4_blocks(M, Cout) workgroups (= entries of the partial-maximum array).
int ivx_conv_launch_fold4w(const _Float16 *V, long long vs, const _Float16 *U, long long us, int M, int Cin_stored, int Cout, int Z, const IvxWinoFold &f,
                           hipStream_t st) {
  if (Cin_stored % 64 != 0 || Cout % 8 != 0) {
    ivx_set_error("ivx_conv_launch_fold4w: Cin %% 32 == 0 and Cout %% 8 == 0 only");
    return IVX_ERR_INVALID_ARG;
  }
  const long long in_bytes = 36LL * vs * 2, w_bytes = 36LL * us * 2;
  if (in_bytes >= (1LL << 31) || w_bytes >= (1LL << 31)) {
    ivx_set_error("ivx_conv_launch_fold4w: the 36 transformed planes must stay below 2 GiB together");
    return IVX_ERR_UNSUPPORTED;
  }
  Fold4wParams p;
  p.V = V; p.U = U; p.vs = vs; p.us = us; p.M = M; p.Cin = Cin_stored; p.Cout = Cout; p.K = 3 * Cin_stored; p.Z = Z;
  const long long Mt = (M + 61) / 62, Nt = (Cout + 63) / 64;
  p.q_total = (int)((Mt + 7) / 8);
  p.f = f;
  const dim3 grid((unsigned)(8LL * p.q_total * Nt));
  hipLaunchKernelGGL((conv_wino_fold4w_kernel<6>), grid, dim3(512), 0, st, p, (unsigned)in_bytes, (unsigned)w_bytes);
  return IVX_OK;
}
