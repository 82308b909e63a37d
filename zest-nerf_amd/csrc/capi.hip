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

// SURVEY.md 8(b)'s names for two entry points (include/zest_render.h)
extern "C" int zest_gather_encode_fwd(const float *ndc, const float *pts, const float *rays_dir, int R, int S, int has_time,
                                      float t, const float *vol_cl, int D, int Hv, int Wv, const float *imgs_cl, int V,
                                      int H, int W, const float *w2cs, const float *intrinsics, float *x, void *stream) {
    return zest_encode_fwd(ndc, pts, rays_dir, R, S, has_time, t, vol_cl, D, Hv, Wv, imgs_cl, V, H, W, w2cs, intrinsics, x,
                           stream);
}
extern "C" int zest_pack_weights(const zest_mlp_desc *desc, int precision, const float *const *params, void *packed,
                                 void *stream) {
    return zest_mlp_pack(desc, precision, params, packed, stream);
}
