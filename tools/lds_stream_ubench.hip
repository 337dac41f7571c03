// Round 5: a 4-wave workgroup that streams `slabs` x 16 KB through a two-buffer LDS ring with LDS-DMA (the skeleton of conv_igemm_v4_kernel's
// K loop: DMA -> vmcnt(0) -> barrier -> ds_read_b128 -> a few FMAs), templated on the LDS bytes it DECLARES.  Does the declared size alone change
// the time of a 480-workgroup launch (the conv tile 66 ran 8.8 us declaring 32 KB and 34 us declaring 72 KB)?
//   hipcc --offload-arch=gfx950 -O3 tools/lds_stream_ubench.hip -o tools/bin/lds_stream_ubench
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void *lds_ptr_t;

template <int KB, int WPE>
__global__ __launch_bounds__(256, WPE) void k_stream(const float *src, unsigned bytes, float *out, int slabs, int rows_mod) {
  __shared__ __attribute__((aligned(16))) float lds[KB * 256];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)src, 0, bytes, 0x00020000);
  const int wid_u = __builtin_amdgcn_readfirstlane(wid);
  // a workgroup reads 16 KB per slab: 4 waves x 4 instructions x 1 KB; consecutive workgroups share rows (rows_mod distinct 16 KB blocks)
  const unsigned base = (unsigned)((blockIdx.x % rows_mod) * 16384u) + (unsigned)wid_u * 4096u + (unsigned)lane * 16u;
  auto load = [&](int s, int buf) {
    float *dst = lds + buf * 4096 + wid_u * 1024;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(dst + j * 256), 16, base + (unsigned)j * 1024u + (unsigned)s * (unsigned)rows_mod * 16384u, 0, 0, 0);
  };
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  load(0, 0);
  if (slabs > 1) load(1, 1);
  __builtin_amdgcn_s_waitcnt(0x0f70);
  __syncthreads();
  for (int s = 0; s < slabs; ++s) {
    const int cur = s & 1;
    const f32x4 a = *reinterpret_cast<const f32x4 *>(lds + cur * 4096 + ((tid * 4) & 4095));
    const f32x4 b = *reinterpret_cast<const f32x4 *>(lds + cur * 4096 + ((tid * 4 + 2048) & 4095));
    __builtin_amdgcn_s_waitcnt(0x0f70);
    __syncthreads();
    if (s + 2 < slabs) load(s + 2, cur);
    acc += a * b;
  }
  out[(size_t)blockIdx.x * 256 + tid] = acc[0] + acc[1] + acc[2] + acc[3];
}

template <int KB, int WPE>
static float run(const float *src, unsigned bytes, float *out, int grid, int slabs, int rows_mod, int iters) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((k_stream<KB, WPE>), dim3(grid), dim3(256), 0, 0, src, bytes, out, slabs, rows_mod);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0, 0);
  for (int i = 0; i < iters; ++i) hipLaunchKernelGGL((k_stream<KB, WPE>), dim3(grid), dim3(256), 0, 0, src, bytes, out, slabs, rows_mod);
  (void)hipEventRecord(e1, 0);
  (void)hipEventSynchronize(e1);
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, e0, e1);
  return ms / iters * 1e3f;
}

int main() {
  const size_t bytes = 256u << 20;
  float *src, *out;
  (void)hipMalloc(&src, bytes); (void)hipMemset(src, 0, bytes);
  (void)hipMalloc(&out, 32768 * 256 * 4);
  const int grids[] = {32, 120, 480, 1920, 7680};
  for (int slabs : {2, 8, 32}) {
    printf("## %d slabs of 16 KB per workgroup, 120 distinct row blocks; us per launch (50 back-to-back launches)\n| LDS KB (launch bound) |", slabs);
    for (int g : grids) printf(" %d wg |", g);
    printf("\n|---|---|---|---|---|---|\n");
#define ROW(KB, WPE)                                                                                         \
    printf("| %d (%d) |", KB, WPE);                                                                            \
    for (int g : grids) printf(" %.1f |", run<KB, WPE>(src, (unsigned)bytes, out, g, slabs, 120, 50));         \
    printf("\n");
    ROW(32, 1) ROW(40, 1) ROW(48, 1) ROW(56, 1) ROW(64, 1) ROW(72, 1) ROW(96, 1) ROW(128, 1) ROW(32, 4) ROW(72, 2)
  }
  return 0;
}
