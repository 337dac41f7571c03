// Shared helpers for the HIP side of libimvoxel_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/imvoxel.h"

void ivx_set_error(const char *fmt, ...);

#define IVX_REQUIRE(cond, ...)             \
  do {                                     \
    if (!(cond)) {                         \
      ivx_set_error(__VA_ARGS__);          \
      return IVX_ERR_INVALID_ARG;          \
    }                                      \
  } while (0)

#define IVX_CHECK_LAUNCH(what)                                                       \
  do {                                                                               \
    hipError_t e_ = hipGetLastError();                                               \
    if (e_ != hipSuccess) {                                                          \
      ivx_set_error("%s: HIP launch failed: %s", what, hipGetErrorString(e_));       \
      return IVX_ERR_HIP;                                                            \
    }                                                                                \
  } while (0)

static inline int64_t ivx_align_up(int64_t v, int64_t a) { return (v + a - 1) / a * a; }

// Epilogue description of the fused Winograd GEMM + output transform (conv_igemm.hip conv_wino_fold4_kernel; filled by winograd.hip):
// out[b, 4 tx + a, 4 ty + e, z, :] = act((At M A)[a][e] * mscale * scale + shift [+ res]) with M accumulated on chip.
struct IvxWinoFold {
  float *out;
  const float *res, *scale, *shift;
  const unsigned *hdr_v;     // device: bits of max |layer input| (the operand scale of V is derived from it as in winograd.hip)
  const float *uscale;       // device: the filter scale chosen by ivx_conv_winograd_weights
  float *pmax;               // per-workgroup max |out| (the next layer's operand scale), or NULL
  int B, TX, TY, Z, Xo, Yo, Co;
  int relu, res_mode, res_after_act;
  float post_scale;
};

#if defined(__HIPCC__)
// Power-of-two scale of an fp16 (hi, lo) pair tensor (IVX_F16_PAIR): s = 2^k with amax * s in [2^14, 2^15); 1 for 0 / non-finite amax.
__device__ __forceinline__ float ivx_pow2_scale(const float amax) {
  if (!(amax > 0.f) || !(amax < 3.0e38f)) return 1.0f;
  int e;
  (void)frexpf(amax, &e);                        // amax in [2^(e-1), 2^e)
  int k = 15 - e;
  k = k < -120 ? -120 : (k > 120 ? 120 : k);
  return ldexpf(1.0f, k);
}
// max |tensor| accumulated into IVX_AMAX_SLOTS device words (include/imvoxel.h, ivx_pair_io): max over the calling wave (every lane
// calls), then one atomic max on slot `salt` % 64 -- bits of non-negative floats order like the floats.  The slot is read first: the
// values only grow, so a (possibly stale) slot that already holds >= m makes the atomic unnecessary, and after the first few waves
// almost every wave skips it (same-address atomics serialise: 8192 of them cost 0.27 ms in round 3).
__device__ __forceinline__ void ivx_amax_commit(unsigned *slots, float m, int salt) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  if ((threadIdx.x & 63) == 0 && m > 0.f) {
    unsigned *s = slots + (salt & (IVX_AMAX_SLOTS - 1));
    if (__float_as_uint(m) > __hip_atomic_load(s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(s, __float_as_uint(m));
  }
}
// One atomic per WORKGROUP (round 5): every thread of the workgroup calls (uniformly); the waves leave their maxima in `red` (>= blockDim.x / 64
// floats of LDS), one barrier, wave 0 commits.  The 64 slots of a tensor lie in two cache lines, so atomics on different slots still queue at
// one memory channel: with one atomic per wave the thousands of waves that finish together behind freshly zeroed slots held up the tail of
// their launch by 10 - 20 us (profiles/r05_trunk_wg_timeline.md).
__device__ __forceinline__ void ivx_amax_commit_wg(unsigned *slots, float m, float *red, int salt) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = ((int)blockDim.x + 63) >> 6;
  if (lane == 0) red[wid] = m;
  __syncthreads();
  if (wid == 0) ivx_amax_commit(slots, lane < nw ? red[lane] : 0.f, salt);
}
// the maximum the slots hold (every lane of a full wave calls; all lanes get the result)
__device__ __forceinline__ float ivx_amax_read(const unsigned *slots) {
  float a = __uint_as_float(slots[threadIdx.x & (IVX_AMAX_SLOTS - 1)]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) a = fmaxf(a, __shfl_xor(a, o));
  return a;
}
#endif
