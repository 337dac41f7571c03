// Microbenchmark: issue rate of the f32 MFMA forms on gfx950 (cycles per instruction per wave, via s_memtime).
// hipcc --offload-arch=gfx950 -O3 tools/mfma_ubench.hip -o /tmp/mfma_ubench && /tmp/mfma_ubench
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC, int DISTINCT>
__global__ __launch_bounds__(256) void k32(float *out, long long *cyc, int iters) {
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float a[8], b[8];
  for (int i = 0; i < 8; ++i) { a[i] = threadIdx.x * 0.001f + i; b[i] = threadIdx.x * 0.002f - i; }
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
#pragma unroll
      for (int i = 0; i < NACC; ++i)
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[DISTINCT ? ((u + i) & 7) : 0], b[DISTINCT ? ((u * 3 + i) & 7) : 0], acc[i], 0, 0, 0);
    }
  }
  long long t1 = __builtin_readcyclecounter();
  float s = 0;
  for (int i = 0; i < NACC; ++i)
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int NACC>
__global__ __launch_bounds__(256) void k16(float *out, long long *cyc, int iters) {
  f32x4 acc[NACC];
  for (int i = 0; i < NACC; ++i)
    for (int r = 0; r < 4; ++r) acc[i][r] = 0.f;
  float a[8], b[8];
  for (int i = 0; i < 8; ++i) { a[i] = threadIdx.x * 0.001f + i; b[i] = threadIdx.x * 0.002f - i; }
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
#pragma unroll
      for (int i = 0; i < NACC; ++i)
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[(u + i) & 7], b[(u * 3 + i) & 7], acc[i], 0, 0, 0);
    }
  }
  long long t1 = __builtin_readcyclecounter();
  float s = 0;
  for (int i = 0; i < NACC; ++i)
    for (int r = 0; r < 4; ++r) s += acc[i][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <typename K>
void run(const char *name, K kern, int nacc, int blocks_per_cu, double flop_per_inst) {
  float *out; long long *cyc;
  int nb = 256 * blocks_per_cu, iters = 2000;
  hipMalloc(&out, nb * 256 * 4); hipMalloc(&cyc, nb * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(kern, dim3(nb), dim3(256), 0, 0, out, cyc, 10);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(kern, dim3(nb), dim3(256), 0, 0, out, cyc, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  double n_inst = (double)iters * 16 * nacc;
  double tf = n_inst * flop_per_inst * 4.0 /*waves per block*/ * nb / (ms * 1e-3) / 1e12;
  printf("%-28s nacc %2d blocks/CU %d: %7.2f counter-ticks/inst  %8.3f ms  %7.1f TFLOP/s\n", name, nacc, blocks_per_cu, (double)c / n_inst, ms, tf);
  hipFree(out); hipFree(cyc);
}

int main() {
  for (int bpc = 1; bpc <= 2; ++bpc) {
    run("32x32x2 distinct operands", k32<1, 1>, 1, bpc, 4096);
    run("32x32x2 distinct operands", k32<2, 1>, 2, bpc, 4096);
    run("32x32x2 distinct operands", k32<4, 1>, 4, bpc, 4096);
    run("32x32x2 same operands", k32<4, 0>, 4, bpc, 4096);
    run("16x16x4", k16<4>, 4, bpc, 2048);
    run("16x16x4", k16<8>, 8, bpc, 2048);
    run("16x16x4", k16<16>, 16, bpc, 2048);
  }
  return 0;
}
