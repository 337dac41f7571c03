/* see hip_runtime_api.h in this directory (test infrastructure: CPU stand-in for the HIP runtime C API) */
#include "hip_runtime_api.h"
