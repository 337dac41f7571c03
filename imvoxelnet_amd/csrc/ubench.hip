// Device ceilings measured on THIS box ("do not hard-code the peaks"): the issue rate of the MFMA forms the conv
// kernel uses and the streaming copy rate of HBM.  bench.py runs them once at start-up and reports every roofline fraction
// against both the data-sheet figure (MI355X_MICROARCH.md) and these measured ceilings.  Measurement only: nothing on the
// product path calls this file.
#include <hip/hip_runtime.h>

#include "../../include/imvoxel.h"
#include "ivx_common.h"

#define IVX_HIP_CHECK(call)                                              \
  do {                                                                   \
    hipError_t e_ = (call);                                              \
    if (e_ != hipSuccess) {                                              \
      ivx_set_error("%s: %s", #call, hipGetErrorString(e_));             \
      return IVX_ERR_HIP;                                                \
    }                                                                    \
  } while (0)

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// 4 waves per workgroup, 4 independent 32x32 accumulators per wave, operands varied so no two MFMAs are identical.
template <int BF16>
__global__ __launch_bounds__(256) void ubench_mfma_kernel(float *out, int iters) {
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float a[8], b[8];
  bf16x8 ah[4], bh[4];
  for (int i = 0; i < 8; ++i) { a[i] = threadIdx.x * 0.001f + i; b[i] = threadIdx.x * 0.002f - i; }
  for (int i = 0; i < 4; ++i)
    for (int r = 0; r < 8; ++r) { ah[i][r] = (__bf16)(a[(i + r) & 7] * 0.01f); bh[i][r] = (__bf16)(b[(i * 3 + r) & 7] * 0.01f); }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (BF16) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[(u + i) & 3], bh[(u * 3 + i) & 3], acc[i], 0, 0, 0);
        else acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(u + i) & 7], b[(u * 3 + i) & 7], acc[i], 0, 0, 0);
      }
    }
  }
  float s = 0;
  for (int i = 0; i < 4; ++i)
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// Streaming copies, three shapes (the best one is reported): one 16-byte item per lane over a huge grid; a grid-stride loop
// with U items in flight per lane; the same with non-temporal loads / stores.
__global__ __launch_bounds__(256) void ubench_copy_flat_kernel(const float4 *__restrict__ src, float4 *__restrict__ dst, long long n) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) dst[i] = src[i];
}

template <int U, bool NT>
__global__ __launch_bounds__(256) void ubench_copy_kernel(const f32x4 *__restrict__ src, f32x4 *__restrict__ dst, long long n) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i + (U - 1) * stride < n; i += U * stride) {
    f32x4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = NT ? __builtin_nontemporal_load(src + i + u * stride) : src[i + u * stride];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (NT) __builtin_nontemporal_store(v[u], dst + i + u * stride);
      else dst[i + u * stride] = v[u];
    }
  }
  for (; i < n; i += stride) dst[i] = src[i];
}

}  // namespace

// dtype IVX_F32: v_mfma_f32_32x32x2_f32, IVX_BF16: v_mfma_f32_32x32x16_bf16.  scratch: >= 2 MiB device memory.  Synchronises
// the stream.  *tflops = dense rate with every CU holding two workgroups of four waves.
extern "C" int ivx_ubench_mfma(int32_t dtype, void *scratch, int64_t scratch_bytes, double *tflops, ivx_stream_t stream) {
  IVX_REQUIRE(scratch && tflops && (dtype == IVX_F32 || dtype == IVX_BF16), "ivx_ubench_mfma: bad argument");
  const int nb = 512, iters = dtype == IVX_F32 ? 2000 : 8000;
  IVX_REQUIRE(scratch_bytes >= (int64_t)nb * 256 * 4, "ivx_ubench_mfma: scratch must hold %d bytes", nb * 256 * 4);
  hipStream_t st = (hipStream_t)stream;
  hipEvent_t e0, e1;
  IVX_HIP_CHECK(hipEventCreate(&e0));
  IVX_HIP_CHECK(hipEventCreate(&e1));
  double best = 0.0;
  for (int rep = 0; rep < 3; ++rep) {            // rep 0 warms the clocks up
    IVX_HIP_CHECK(hipEventRecord(e0, st));
    if (dtype == IVX_F32) hipLaunchKernelGGL(ubench_mfma_kernel<0>, dim3(nb), dim3(256), 0, st, (float *)scratch, iters);
    else hipLaunchKernelGGL(ubench_mfma_kernel<1>, dim3(nb), dim3(256), 0, st, (float *)scratch, iters);
    IVX_HIP_CHECK(hipEventRecord(e1, st));
    IVX_HIP_CHECK(hipEventSynchronize(e1));
    float ms = 0.f;
    IVX_HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double flop_per_inst = dtype == IVX_F32 ? 2.0 * 32 * 32 * 2 : 2.0 * 32 * 32 * 16;
    const double tf = (double)iters * 64 * flop_per_inst * 4.0 * nb / (ms * 1e-3) / 1e12;
    if (rep && tf > best) best = tf;
  }
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  *tflops = best;
  return IVX_OK;
}

// Streaming copy of `bytes` (a multiple of 16) from src to dst, both device buffers: *gbps = (read + written bytes) / time,
// best over five kernel shapes (flat, grid-stride x4 / x8, the same non-temporal), each run three times.  Size the buffers well past the 256 MiB Infinity Cache (bench.py uses 1 GiB each).
extern "C" int ivx_ubench_copy(const void *src, void *dst, int64_t bytes, double *gbps, ivx_stream_t stream) {
  IVX_REQUIRE(src && dst && gbps && bytes >= 16 && bytes % 16 == 0, "ivx_ubench_copy: bad argument");
  hipStream_t st = (hipStream_t)stream;
  hipEvent_t e0, e1;
  IVX_HIP_CHECK(hipEventCreate(&e0));
  IVX_HIP_CHECK(hipEventCreate(&e1));
  double best = 0.0;
  const long long n = (long long)(bytes / 16);
  for (int variant = 0; variant < 5; ++variant)
    for (int rep = 0; rep < 3; ++rep) {
      IVX_HIP_CHECK(hipEventRecord(e0, st));
      const float4 *s4 = (const float4 *)src;
      float4 *d4 = (float4 *)dst;
      switch (variant) {
        case 0: hipLaunchKernelGGL(ubench_copy_flat_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, s4, d4, n); break;
        case 1: hipLaunchKernelGGL((ubench_copy_kernel<4, false>), dim3(256 * 8), dim3(256), 0, st, (const f32x4 *)src, (f32x4 *)dst, n); break;
        case 2: hipLaunchKernelGGL((ubench_copy_kernel<8, false>), dim3(256 * 8), dim3(256), 0, st, (const f32x4 *)src, (f32x4 *)dst, n); break;
        case 3: hipLaunchKernelGGL((ubench_copy_kernel<4, true>), dim3(256 * 8), dim3(256), 0, st, (const f32x4 *)src, (f32x4 *)dst, n); break;
        default: hipLaunchKernelGGL((ubench_copy_kernel<8, true>), dim3(256 * 16), dim3(256), 0, st, (const f32x4 *)src, (f32x4 *)dst, n); break;
      }
      IVX_HIP_CHECK(hipEventRecord(e1, st));
      IVX_HIP_CHECK(hipEventSynchronize(e1));
      float ms = 0.f;
      IVX_HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
      const double g = 2.0 * (double)bytes / (ms * 1e-3) / 1e9;
      if (rep && g > best) best = g;
    }
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  *gbps = best;
  return IVX_OK;
}
