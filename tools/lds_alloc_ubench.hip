// What does a workgroup's static LDS allocation cost at dispatch?  (round 5: conv tile 66 declaring 52 KB instead of 17 KB ran 25 us slower
// per launch with identical code.)  A kernel that touches one LDS word and does `work` dependent FMAs per thread, templated on the LDS bytes it
// declares; time per launch over grids of 32 .. 32768 workgroups of 256 threads, back-to-back launches.
//   hipcc --offload-arch=gfx950 -O3 tools/lds_alloc_ubench.hip -o gpurun_out/bin/lds_alloc_ubench && gpurun_out/bin/lds_alloc_ubench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

template <int KB>
__global__ __launch_bounds__(256) void k_lds(float *out, int work) {
  __shared__ float s[KB * 256];
  s[threadIdx.x] = (float)threadIdx.x;
  __syncthreads();
  float v = s[(threadIdx.x + 1) & 255];
  for (int i = 0; i < work; ++i) v = v * 1.0001f + 0.5f;
  if (v == 12345.678f) out[blockIdx.x] = v + s[(KB * 256 - 1) & (KB * 256 - 1)];
}

template <int KB>
static float run(float *out, int grid, int work, int iters) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(k_lds<KB>, dim3(grid), dim3(256), 0, 0, out, work);
  hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(k_lds<KB>, dim3(grid), dim3(256), 0, 0, out, work);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  hipEventDestroy(e0); hipEventDestroy(e1);
  return ms / iters * 1e3f;
}

int main() {
  float *out;
  hipMalloc(&out, 1 << 20);
  const int grids[] = {32, 128, 512, 2048, 8192, 32768};
  for (int work : {0, 4000}) {
    printf("## work = %d dependent FMAs per thread; us per launch (100 back-to-back launches)\n", work);
    printf("| LDS KB |");
    for (int g : grids) printf(" %d wg |", g);
    printf("\n|---|");
    for (size_t i = 0; i < sizeof(grids) / sizeof(int); ++i) printf("---|");
    printf("\n");
#define ROW(KB)                                                       \
    printf("| %d |", KB);                                             \
    for (int g : grids) printf(" %.1f |", run<KB>(out, g, work, 100)); \
    printf("\n");
    ROW(1) ROW(8) ROW(16) ROW(24) ROW(32) ROW(40) ROW(48) ROW(56) ROW(64) ROW(80) ROW(96) ROW(128) ROW(160)
  }
  hipFree(out);
  return 0;
}
