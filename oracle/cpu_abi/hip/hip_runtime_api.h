/* TEST INFRASTRUCTURE (oracle/): a host-memory stand-in for the slice of the HIP runtime C API that csrc/model.cpp and
 * tests/c/e2e_small.c use, so that the SAME model-level C-ABI (include/imvoxel.h) can be served by the CPU restatement of
 * oracle/cpu_abi/cpu_ops.cpp (SURVEY 8d "same ABI served by the CPU restatement").  Found first on the include path of the
 * libimvoxel_cpu.so build only (oracle/cpu_abi/build.py); the product library is built against the real <hip/...> headers and
 * never sees this file.  "Device" memory is malloc'ed host memory, a stream is a no-op (every call completes before it returns),
 * events are wall-clock stamps, graphs are not available. */
#ifndef IVX_CPU_ABI_HIP_SHIM_H
#define IVX_CPU_ABI_HIP_SHIM_H
#include <stddef.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#ifdef __cplusplus
extern "C" {
#endif
typedef int hipError_t;
enum { hipSuccess = 0, hipErrorOutOfMemory = 2, hipErrorNotSupported = 801 };
typedef void *hipStream_t;
typedef struct ivx_cpu_event { double t_ms; } *hipEvent_t;
typedef void *hipGraph_t;
typedef void *hipGraphExec_t;
typedef enum { hipMemcpyHostToHost = 0, hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3, hipMemcpyDefault = 4 } hipMemcpyKind;
typedef enum { hipStreamCaptureModeGlobal = 0, hipStreamCaptureModeThreadLocal = 1, hipStreamCaptureModeRelaxed = 2 } hipStreamCaptureMode;

static inline const char *hipGetErrorString(hipError_t e) { return e == hipSuccess ? "no error" : (e == hipErrorOutOfMemory ? "out of memory" : "not supported on the CPU restatement"); }
/* hipMalloc returns 256-byte aligned memory; the entry points check workspace alignment */
static inline hipError_t hipMalloc(void **p, size_t n) { *p = aligned_alloc(256, (n + 255) / 256 * 256 + 256); return *p ? hipSuccess : hipErrorOutOfMemory; }
static inline hipError_t hipFree(void *p) { free(p); return hipSuccess; }
static inline hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind k) { (void)k; memcpy(d, s, n); return hipSuccess; }
static inline hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind k, hipStream_t st) { (void)k; (void)st; memcpy(d, s, n); return hipSuccess; }
static inline hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t st) { (void)st; memset(d, v, n); return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t st) { (void)st; return hipSuccess; }
static inline hipError_t hipDeviceSynchronize(void) { return hipSuccess; }
static inline hipError_t hipEventCreate(hipEvent_t *e) { *e = (hipEvent_t)malloc(sizeof(**e)); return *e ? hipSuccess : hipErrorOutOfMemory; }
static inline hipError_t hipEventDestroy(hipEvent_t e) { free(e); return hipSuccess; }
static inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t st) {
  struct timespec ts; (void)st;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  e->t_ms = ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
  return hipSuccess;
}
static inline hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b) { *ms = (float)(b->t_ms - a->t_ms); return hipSuccess; }
/* side streams of the step interpreter (csrc/model.cpp: the shortcut convs overlap conv1 / conv2): on the host everything runs in program order */
#define hipStreamNonBlocking 1
#define hipEventDisableTiming 2
static inline hipError_t hipStreamCreateWithFlags(hipStream_t *st, unsigned flags) { (void)flags; *st = (hipStream_t)1; return hipSuccess; }
static inline hipError_t hipStreamDestroy(hipStream_t st) { (void)st; return hipSuccess; }
static inline hipError_t hipStreamWaitEvent(hipStream_t st, hipEvent_t e, unsigned flags) { (void)st; (void)e; (void)flags; return hipSuccess; }
static inline hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned flags) { (void)flags; return hipEventCreate(e); }
static inline hipError_t hipStreamBeginCapture(hipStream_t st, hipStreamCaptureMode m) { (void)st; (void)m; return hipErrorNotSupported; }
static inline hipError_t hipStreamEndCapture(hipStream_t st, hipGraph_t *g) { (void)st; *g = NULL; return hipErrorNotSupported; }
static inline hipError_t hipGraphInstantiate(hipGraphExec_t *x, hipGraph_t g, void *a, void *b, size_t n) { (void)g; (void)a; (void)b; (void)n; *x = NULL; return hipErrorNotSupported; }
static inline hipError_t hipGraphLaunch(hipGraphExec_t x, hipStream_t st) { (void)x; (void)st; return hipErrorNotSupported; }
static inline hipError_t hipGraphDestroy(hipGraph_t g) { (void)g; return hipSuccess; }
static inline hipError_t hipGraphExecDestroy(hipGraphExec_t x) { (void)x; return hipSuccess; }
#ifdef __cplusplus
}
#endif
#endif
