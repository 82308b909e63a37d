// Fused inference renderer (placeholder entry point until the kernel lands this round).
#include "zest_common.cuh"

extern "C" int zest_render_fused_fwd(const float *, const float *, const float *, const float *, int,
                                     int, const zest_mlp_desc *, const void *, const zest_view_set *,
                                     const zest_mlp_desc *, const void *, const zest_view_set *, float,
                                     int, int, float *, void *) {
    zest_set_error("zest_render_fused_fwd: not implemented yet");
    return (int)hipErrorNotSupported;
}
