// Error reporting and version of the C ABI (include/nerf_hip.h).
#include <stdarg.h>
#include <stdio.h>

#include "../../include/nerf_hip.h"

namespace nerf {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

}  // namespace nerf

extern "C" int nerf_hip_abi_version(void) { return NERF_HIP_ABI_VERSION; }
extern "C" const char* nerf_last_error(void) { return nerf::g_err; }
