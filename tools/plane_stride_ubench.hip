// Microbenchmark for the Winograd transforms' memory pattern: one thread moves 8 bytes to / from each of 64 "planes".
//   layout 0 (as built): plane-major -- plane p of item t at  p * PS + t        (64 streams, PS = items per plane: 18.6 MB apart)
//   layout 1 (blocked):  [block][plane][items of the block] -- a workgroup's 64 accesses fall into ONE 2 MB block
// Modes: W = 64 stores per thread (the input transform's write side), R = 64 loads per thread + one store (the output transform's
// read side), C = linear copy of the same bytes.  Question: is the 4.7-5.2 TB/s of the transforms (copy: 6.1-6.3) the price of touching
// 64 distant regions per workgroup (address translation, DRAM pages), i.e. would a blocked V / M layout be faster?
// hipcc --offload-arch=gfx950 -O3 tools/plane_stride_ubench.hip -o /tmp/plane_stride_ubench && /tmp/plane_stride_ubench
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define HK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int LAYOUT, int BLK>
__device__ __forceinline__ size_t addr(size_t t, int p, size_t PS) {
  if (LAYOUT == 0) return (size_t)p * PS + t;
  return ((t / BLK) * 64 + p) * (size_t)BLK + (t % BLK);
}

template <int LAYOUT, int BLK>
__global__ __launch_bounds__(256) void kw(f32x2 *out, size_t PS, size_t n) {
  const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= n) return;
  const float b = (float)(t & 1023);
#pragma unroll
  for (int p = 0; p < 64; ++p) out[addr<LAYOUT, BLK>(t, p, PS)] = f32x2{b + p, b - p};
}

template <int LAYOUT, int BLK>
__global__ __launch_bounds__(256) void kr(const f32x2 *in, f32x2 *out, size_t PS, size_t n) {
  const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= n) return;
  f32x2 s = {0.f, 0.f};
#pragma unroll
  for (int p = 0; p < 64; ++p) s += in[addr<LAYOUT, BLK>(t, p, PS)];
  out[t] = s;
}

__global__ __launch_bounds__(256) void kc(const float4 *in, float4 *out, size_t n4) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) out[i] = in[i];
}

int main() {
  // the 64-channel KITTI neck layer: 72 576 tile-z rows x 32 two-channel items per plane
  const size_t n = 72576ull * 32, PS = n;
  const size_t bytes = 64 * n * 8;
  f32x2 *a, *b;
  HK(hipMalloc(&a, bytes + (1 << 24)));
  HK(hipMalloc(&b, bytes + (1 << 24)));
  HK(hipMemset(a, 0, bytes));
  HK(hipMemset(b, 0, bytes));
  hipEvent_t e0, e1;
  HK(hipEventCreate(&e0)); HK(hipEventCreate(&e1));
  const unsigned grid = (unsigned)((n + 255) / 256);
  auto time = [&](auto f, const char *name, double gb) {
    f(); (void)hipDeviceSynchronize();
    float best = 1e9f;
    for (int r = 0; r < 7; ++r) {
      (void)hipEventRecord(e0); f(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
      float ms; (void)hipEventElapsedTime(&ms, e0, e1);
      best = ms < best ? ms : best;
    }
    printf("%-46s %7.3f ms  %7.1f GB/s\n", name, best, gb / best * 1e3);
    return 0;
  };
  const double gb = bytes / 1e9;
  time([&] { hipLaunchKernelGGL((kw<0, 4096>), dim3(grid), dim3(256), 0, 0, a, PS, n); }, "W plane-major (as built)", gb);
  time([&] { hipLaunchKernelGGL((kw<1, 4096>), dim3(grid), dim3(256), 0, 0, a, PS, n); }, "W blocked, 4096 items (2 MB blocks)", gb);
  time([&] { hipLaunchKernelGGL((kw<1, 256>), dim3(grid), dim3(256), 0, 0, a, PS, n); }, "W blocked, 256 items (128 KB blocks)", gb);
  time([&] { hipLaunchKernelGGL((kr<0, 4096>), dim3(grid), dim3(256), 0, 0, a, b, PS, n); }, "R plane-major (as built)", gb + n * 8 / 1e9);
  time([&] { hipLaunchKernelGGL((kr<1, 4096>), dim3(grid), dim3(256), 0, 0, a, b, PS, n); }, "R blocked, 4096 items", gb + n * 8 / 1e9);
  time([&] { hipLaunchKernelGGL((kr<1, 256>), dim3(grid), dim3(256), 0, 0, a, b, PS, n); }, "R blocked, 256 items", gb + n * 8 / 1e9);
  time([&] { hipLaunchKernelGGL(kc, dim3(256 * 16), dim3(256), 0, 0, (const float4 *)a, (float4 *)b, bytes / 16); }, "linear copy of the same bytes (R + W)", 2 * gb);
  return 0;
}
