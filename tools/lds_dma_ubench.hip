// Microbenchmark: does an in-flight LDS-DMA load (buffer_load_dwordx4 ... lds) delay an unrelated ds_read's
// s_waitcnt lgkmcnt(0)?  One wave per workgroup; the DMA source is a cold (never touched) 1 GiB buffer so the load misses
// L2 (~2 us); the ds_read targets LDS the DMA does not write.  Prints cycles (s_memtime) for:
//   A  ds_read + lgkmcnt(0) alone
//   B  DMA issued first, then ds_read + lgkmcnt(0) only      (if ~A: the DMA is not on the lgkm counter)
//   C  DMA issued, then vmcnt(0)                              (the DMA's own latency)
// Measured on MI355X (profiles/r01_lds_dma_ubench.log): A 168, B 330, C 528 cycles -- the DMA is not waited for by
// lgkmcnt, but issuing it (64 lanes, 64 pages) holds the wave ~160 cycles.
// hipcc --offload-arch=gfx950 -O3 tools/lds_dma_ubench.hip -o /tmp/lds_dma_ubench && /tmp/lds_dma_ubench
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void *lds_ptr_t;

__global__ __launch_bounds__(64) void k(const float *src, unsigned bytes, long long *out, float *sink, int mode) {
  __shared__ __attribute__((aligned(16))) float lds[4096];
  const int lane = threadIdx.x;
  for (int i = lane; i < 4096; i += 64) lds[i] = (float)i;
  __syncthreads();
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)src, 0, bytes, 0x00020000);
  const unsigned vo = ((unsigned)blockIdx.x * 1048576u + (unsigned)lane * 4096u) % (bytes - 4096u);   // a page per lane: all miss
  long long t0 = __builtin_readcyclecounter();
  if (mode >= 1) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(lds + 2048), 16, vo & ~15u, 0, 0, 0);
  f32x4 v = {0, 0, 0, 0};
  if (mode <= 1) {
    v = *reinterpret_cast<const f32x4 *>(lds + lane * 4);
    __builtin_amdgcn_s_waitcnt(0xc07f);    // lgkmcnt(0) only
  } else {
    __builtin_amdgcn_s_waitcnt(0x0f70);    // vmcnt(0) only
  }
  long long t1 = __builtin_readcyclecounter();
  __builtin_amdgcn_s_waitcnt(0x0070);      // drain everything before leaving
  sink[blockIdx.x * 64 + lane] = v[0] + v[3] + lds[2048 + lane];
  if (lane == 0) out[blockIdx.x] = t1 - t0;
}

int main() {
  const size_t bytes = 1ull << 30;
  float *src, *sink; long long *out;
  (void)hipMalloc(&src, bytes); (void)hipMemset(src, 0, bytes);
  const int nb = 64;
  (void)hipMalloc(&sink, nb * 64 * 4); (void)hipMalloc(&out, nb * 8);
  long long h[nb];
  const char *names[3] = {"A ds_read + lgkmcnt(0) alone          ", "B DMA in flight, ds_read + lgkmcnt(0) ", "C DMA + vmcnt(0)                      "};
  for (int mode = 0; mode < 3; ++mode) {
    for (int rep = 0; rep < 3; ++rep) {
      hipLaunchKernelGGL(k, dim3(nb), dim3(64), 0, 0, src, (unsigned)(bytes - 1), out, sink, mode);
      (void)hipDeviceSynchronize();
    }
    (void)hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
    double s = 0; long long mn = 1ll << 60, mx = 0;
    for (int i = 0; i < nb; ++i) { s += h[i]; if (h[i] < mn) mn = h[i]; if (h[i] > mx) mx = h[i]; }
    printf("%s cycles (s_memtime, 100 MHz-domain ticks may apply): mean %.0f  min %lld  max %lld\n", names[mode], s / nb, mn, mx);
  }
  return 0;
}
