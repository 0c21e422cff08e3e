// Error plumbing and ABI version of libwan_hip.so.
#include <stdarg.h>
#include <stdio.h>

#include <hip/hip_runtime.h>

#include "../../include/wan_hip.h"

static thread_local char g_err[512] = "";

void wan_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* wan_last_error(void) { return g_err; }
extern "C" int wan_abi_version(void) { return WAN_ABI_VERSION; }

// Launch planning (tile quantisation) wants the CU count; it is host arithmetic and must also work where no GPU is
// visible (the CPU-side ABI tests), hence the fallback.
int wan_cu_count() {
    static int ncu = 0;
    if (ncu == 0) {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) != hipSuccess ||
            hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
        ncu = v;
    }
    return ncu;
}
