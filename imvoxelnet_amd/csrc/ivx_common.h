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
