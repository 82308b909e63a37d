// Library-level entry points: ABI version, per-thread error text, device facts.
#include <stdarg.h>
#include <string.h>
#include "zest_common.cuh"

static thread_local char g_err[512] = "";

void zest_set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" int zest_abi_version(void) { return ZEST_ABI_VERSION; }
extern "C" const char *zest_last_error(void) { return g_err; }

extern "C" int zest_device_info(int *cu_count, int *clock_khz, char *name, size_t name_len) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    hipDeviceProp_t p;
    if (e == hipSuccess) e = hipGetDeviceProperties(&p, dev);
    if (e != hipSuccess) {
        zest_set_error("zest_device_info: %s", hipGetErrorString(e));
        return (int)e;
    }
    if (cu_count) *cu_count = p.multiProcessorCount;
    if (clock_khz) *clock_khz = p.clockRate;
    if (name && name_len) {
        strncpy(name, p.gcnArchName, name_len - 1);
        name[name_len - 1] = 0;
    }
    return 0;
}
