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

extern "C" int ivx_version(void) { return 300; /* 0.3.0: ivx_model_cfg gained the head / DCNv2 / LayoutHead fields; ivx_model_detect, ivx_indoor_tail_* */ }
extern "C" const char *ivx_last_error(void) { return g_err; }
