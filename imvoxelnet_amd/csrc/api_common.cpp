// Error reporting + version for libimvoxel_hip.so.
#include <stdarg.h>
#include <stdio.h>

#include "../../include/imvoxel.h"

static thread_local char g_err[512] = "";

void ivx_set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" int ivx_version(void) { return 200; /* 0.2.0: ivx_conv_desc gained res_scale, ivx_model_cfg the indoor-neck fields */ }
extern "C" const char *ivx_last_error(void) { return g_err; }
