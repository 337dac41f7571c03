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

extern "C" int ivx_version(void) { return 400; /* 0.4.0: ivx_pair_io / ivx_conv_fwd_pio (chained fp16-pair activations), ivx_model_cfg.trunk_operands; 0.3.1: ivx_conv_desc.wino_operands, IVX_BF16_PAIR / IVX_F16_PAIR, ivx_model_cfg.wino_operands (0.3.0: head / DCNv2 / LayoutHead fields, ivx_model_detect) */ }
extern "C" const char *ivx_last_error(void) { return g_err; }
