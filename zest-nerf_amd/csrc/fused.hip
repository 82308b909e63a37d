// zest_render_fused_fwd: argument checks and dispatch to the fused renderer variants
// (fused.cuh).  Before any launch the shapes the kernel indexes with are validated on the
// host: a bad pointer or size here must never reach the GPU.
#include <string.h>
#include "fused.cuh"
#include "mlp_plan.h"

namespace zest {

// Chain the block records of each ray: out += T * sums, T *= exit transmittance.
__global__ void fused_combine_kernel(const float *__restrict__ partials, int R, int bpr, int dyn,
                                     int white_bkgd, float *__restrict__ out) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    combine_ray([&](int b) { return partials + ((size_t)r * bpr + b) * kPartialFloats; }, bpr, dyn != 0,
                white_bkgd, out + (size_t)r * 16);
}

}  // namespace zest

namespace {

bool fill_net(const zest_mlp_desc *d, const void *packed, const zest_view_set *vs, int pts_ch, int precision,
              zest::FusedNet *n, int *nt_feat, int *units, const char **err) {
    memset(n, 0, sizeof(*n));
    if (!d || !packed) return *err = "descriptor and packed weights are required", false;
    if (d->in_ch_pts != pts_ch) return *err = "unexpected in_ch_pts for this slot", false;
    if (d->net_type != 0 && d->net_type != 2) return *err = "net_type must be 0 ('v0') or 2 ('v2')", false;
    zest::MlpPlan plan;
    if (!zest::build_plan(*d, precision, zest::ORDER_ACC, &plan, err, false)) return false;
    n->bias = (const float *)packed;
    n->tiles = (const uint4 *)((const char *)packed + plan.bias_bytes);
    n->head = d->head, n->v2 = d->net_type == 2;
    *nt_feat = d->use_feat ? plan.nt_feat : 0;
    *units = plan.n_tiles;
    if (((uintptr_t)packed & 15) != 0) return *err = "packed weights must be 16-byte aligned", false;
    if (vs) {
        n->w2cs = vs->w2cs, n->intr = vs->intrinsics;
        n->vol = (const float4 *)vs->vol_cl, n->imgs = (const float4 *)vs->imgs_cl;
        n->D = vs->D, n->Hv = vs->Hv, n->Wv = vs->Wv, n->V = vs->V, n->H = vs->H, n->W = vs->W;
    }
    if (d->use_feat) {
        if (!vs || !vs->vol_cl || !vs->imgs_cl || !vs->w2cs || !vs->intrinsics)
            return *err = "a net with features needs volume, images and cameras", false;
        if (vs->D < 1 || vs->Hv < 1 || vs->Wv < 1 || vs->H < 2 || vs->W < 2)
            return *err = "bad volume / image shape", false;
        if ((long long)vs->D * vs->Hv * vs->Wv >= (1LL << 30))
            return *err = "volume too large for the fused kernel's 32-bit voxel indices (2^30 voxels)", false;
        if (vs->V < 1 || vs->V > zest::kMaxViews || 8 + 4 * vs->V != d->in_ch_feat)
            return *err = "view count does not match in_ch_feat = 8 + 4V", false;
        if (((uintptr_t)vs->vol_cl | (uintptr_t)vs->imgs_cl) & 15)
            return *err = "volume / images must be 16-byte aligned", false;
    } else {
        n->vol = nullptr, n->imgs = nullptr;          // features unused: never dereferenced
    }
    return true;
}

}  // namespace

#ifdef ZEST_STAMPS
constexpr size_t kStampBytes = 1024 * 8 * 8 * 8;       // diagnostic builds: 8 u64 per wave, <= 8192 waves
#else
constexpr size_t kStampBytes = 0;
#endif

static int g_pass_shape = 0;       // 0: chosen per launch, 1: equal block shares (dense), 2: ray ranges

extern "C" int zest_render_fused_set_passes(int shape) {
    ZEST_CHECK_ARG(shape >= 0 && shape <= 2,
                   "zest_render_fused_set_passes: shape must be 0 (auto), 1 (dense: equal block shares) or 2 (ray ranges)");
    g_pass_shape = shape;
    return 0;
}

// Pass shape (a pass = the kFusedWaves blocks the waves of a workgroup run through the network together).
//   ray ranges (*ray_ranges = 1): workgroup w of n = min(cus, R) owns the rays [w R / n, (w+1) R / n) and walks their
//     blocks kFusedWaves at a time; a ray that continues into the workgroup's next pass carries its sums in LDS.
//     Every wave of every pass but a workgroup's last has a block, for any number of blocks per ray, and the rays
//     are finished in the kernel.  Taken unless it needs more rounds than the dense shape.
//   dense (0): every workgroup owns an equal share of ALL blocks (whole passes), rays notwithstanding; the block
//     records go to the workspace and a second tiny launch chains them.  For a few long rays on a big device.
// *n_wg: workgroups launched; *rounds: passes of the busiest workgroup.  Either may be NULL.
extern "C" int zest_render_fused_pass_shape(int R, int S, int precision, int cus, int *ray_ranges, int *n_wg,
                                            int *rounds) {
    ZEST_CHECK_ARG(R >= 0 && S >= 1 && zest::prec_is_engine(precision) && ray_ranges,
                   "zest_render_fused_pass_shape: bad argument");
    if (cus <= 0) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess ||
            hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
            cus = 256;
    }
    const int bs = 16 * zest::fused_cb(precision), bpr = (S + bs - 1) / bs;
    const int W = zest::kFusedWaves;
    // dense: P passes in all, ceil(P / G0) per workgroup, as many workgroups as that needs
    const int P = zest_div_up((long long)R * bpr, W), G0 = P < cus ? P : cus;
    const int dense_rounds = G0 ? zest_div_up(P, G0) : 0, dense_wg = G0 ? zest_div_up(P, dense_rounds) : 0;
    // ray ranges: n workgroups, the busiest has ceil(R / n) rays
    const int n = R < cus ? R : cus;
    const int range_rounds = n ? zest_div_up((long long)zest_div_up(R, n) * bpr, W) : 0;
    int take = range_rounds <= dense_rounds ? 1 : 0;
    if (g_pass_shape == 1) take = 0;
    if (g_pass_shape == 2) take = 1;
    *ray_ranges = take;
    if (n_wg) *n_wg = take ? n : dense_wg;
    if (rounds) *rounds = take ? range_rounds : dense_rounds;
    return 0;
}

extern "C" size_t zest_render_fused_workspace(int R, int S) {
    if (R <= 0 || S <= 0) return 16 + kStampBytes;
    // one record per block of a ray; the smallest block (split-fp16 operands) is 16 samples
    return (size_t)R * ((S + 15) / 16) * zest::kPartialFloats * sizeof(float) + kStampBytes;
}

extern "C" int zest_render_fused_fwd(const float *ndc, const float *pts, const float *z,
                                     const float *rays_dir, int R, int S,
                                     const zest_mlp_desc *desc_static, const void *packed_static,
                                     const zest_view_set *views_static,
                                     const zest_mlp_desc *desc_dynamic, const void *packed_dynamic,
                                     const zest_view_set *views_dynamic, float frame_idx,
                                     int precision, int white_bkgd, void *workspace, float *out,
                                     void *stream) {
    if (R == 0) return 0;                       // an empty batch is a no-op (its tensors have no storage)
    ZEST_CHECK_ARG(ndc && z && rays_dir && out && workspace,
                   "zest_render_fused_fwd: ndc, z, rays_dir, workspace, out required");
    ZEST_CHECK_ARG(((uintptr_t)workspace & 15) == 0, "zest_render_fused_fwd: workspace must be 16-byte aligned");
    ZEST_CHECK_ARG(R >= 0 && S >= 1, "zest_render_fused_fwd: bad shape R=%d S=%d", R, S);
    ZEST_CHECK_ARG(zest::prec_is_engine(precision),
                   "zest_render_fused_fwd: precision must be ZEST_PREC_BF16, _F16 or _F16X3 (the exact-product "
                   "fp32 kernel exists per operator only: zest_mlp_fwd)");
    ZEST_CHECK_ARG(((uintptr_t)out & 15) == 0, "zest_render_fused_fwd: out must be 16-byte aligned");
    zest::FusedArgs a;
    memset(&a, 0, sizeof(a));
    a.ndc = ndc, a.pts = pts, a.z = z, a.dir = rays_dir, a.R = R, a.S = S;
    a.frame_idx = frame_idx, a.white_bkgd = white_bkgd, a.out = out;
    const int bs = 16 * zest::fused_cb(precision);                    // samples per block
    a.bpr = (S + bs - 1) / bs, a.partials = (float *)workspace;
#ifdef ZEST_STAMPS
    a.stamps = (unsigned long long *)((char *)workspace + zest_render_fused_workspace(R, S) - kStampBytes);
#endif
    const char *err = nullptr;
    int nts = 0, ntd = 0, units_s = 0, units_d = 0;
    ZEST_CHECK_ARG(fill_net(desc_static, packed_static, views_static, 63, precision, &a.st, &nts, &units_s, &err),
                   "zest_render_fused_fwd: static net: %s", err);
    const bool dyn = desc_dynamic != nullptr;
    if (dyn) {
        ZEST_CHECK_ARG(fill_net(desc_dynamic, packed_dynamic, views_dynamic, 84, precision, &a.dy, &ntd, &units_d, &err),
                       "zest_render_fused_fwd: dynamic net: %s", err);
        ZEST_CHECK_ARG(desc_static->head == ZEST_HEAD_BLEND && desc_dynamic->head == ZEST_HEAD_DYNAMIC,
                       "zest_render_fused_fwd: blending needs a static net with the blend head and "
                       "a dynamic net with the scene-flow heads");
    }
    ZEST_CHECK_ARG(pts || (!a.st.vol && !a.dy.vol), "zest_render_fused_fwd: pts required with features");
    if (R == 0) return 0;
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
        cus = 256;
    int blocks = 0;                        // workgroups: at most one per CU (128 KiB ring)
    zest_render_fused_pass_shape(R, S, precision, cus, &a.ray_ranges, &blocks, nullptr);
    hipStream_t st = (hipStream_t)stream;
    // 'v2' static nets (additive modulation) have kernels of their own, single-net only
    const bool v2s = a.st.v2 != 0;
    ZEST_CHECK_ARG(!(v2s && (dyn || nts == 0)),
                   "zest_render_fused_fwd: a 'v2' net is rendered by the single-net kernels with features only");
    const int key = (v2s ? 1000 : 0) + (dyn ? 100 : 0) + nts * 10 + ntd;
    int rc = -1;
#define ZEST_CASE1(ptag, tag)                                                                    \
    ZEST_CHECK_ARG(zest::fused_units_##ptag##_##tag(0) == units_s && zest::fused_units_##ptag##_##tag(1) == units_d, \
                   "zest_render_fused_fwd: packed stream is %d+%d units, kernel expects %d+%d",  \
                   units_s, units_d, zest::fused_units_##ptag##_##tag(0), zest::fused_units_##ptag##_##tag(1)); \
    rc = zest::fused_launch_##ptag##_##tag(a, blocks, st);
#define ZEST_CASE(k, tag)                                                                        \
    case k:                                                                                      \
        if (precision == ZEST_PREC_BF16) { ZEST_CASE1(bf16, tag) }                               \
        else if (precision == ZEST_PREC_F16) { ZEST_CASE1(f16, tag) }                            \
        else { ZEST_CASE1(x3, tag) }                                                             \
        break;
    switch (key) {
        ZEST_CASE(0, s0)
        ZEST_CASE(20, s2)
        ZEST_CASE(40, s4)
        ZEST_CASE(100, s0d0)
        ZEST_CASE(120, s2d0)
        ZEST_CASE(140, s4d0)
        ZEST_CASE(122, s2d2)
        ZEST_CASE(142, s4d2)
        ZEST_CASE(1020, s2v)
        ZEST_CASE(1040, s4v)
        default:
            zest_set_error("zest_render_fused_fwd: no fused variant for %d static / %d dynamic feature "
                           "tiles%s", nts, ntd, dyn ? "" : " (static only)");
            return (int)hipErrorInvalidValue;
    }
#undef ZEST_CASE
#undef ZEST_CASE1
    if (rc != 0 || a.ray_ranges) return rc;
    hipLaunchKernelGGL(zest::fused_combine_kernel, dim3(zest_div_up(R, 128)), dim3(128), 0, st,
                       a.partials, R, a.bpr, dyn ? 1 : 0, white_bkgd, out);
    ZEST_RETURN_LAUNCH("zest_render_fused_fwd(combine)");
}
