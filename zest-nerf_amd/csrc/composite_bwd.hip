// Backward of the compositing kernels (training path).
// One wavefront per ray.  The forward quantities (alpha, transmittance T, weights w) are
// recomputed chunk by chunk exactly as composite.hip does; the gradient of the exclusive
// product T_i = prod_{j<i} f_j needs, per sample, the sum over all LATER samples of
// (dL/dw_k) w_k, i.e. a reverse scan, so the ray is walked a second time from its far end with
// the per-chunk entry transmittances saved by the first walk.
//   dL/dalpha_i = G_i T_i - (sum_{k>i} G_k w_k) / f_i,   G_i = dL/dw_i (direct + via the maps)
// Replaces the autograd of raw2outputs / raw2outputs_blending (reference renderer.py:115-219).
// Outputs with no consumer in the reference's losses (disp_map, alpha) carry no gradient.
#include "zest_common.cuh"

namespace {

constexpr int kWaves = 4, kMaxChunks = 32;          // S <= 2048

__device__ __forceinline__ float dist_of(const float *__restrict__ zr, const float *__restrict__ dr, int s, int S,
                                         float z, float dn) {
    if (dr) return dr[s];                       // the caller's own spacings
    return ((s + 1 < S) ? (zr[s + 1] - z) : 1e10f) * dn;
}

// inclusive/exclusive suffix sum over the 64 lanes (lane 63 is the "first" element)
__device__ __forceinline__ float suffix_excl_sum(float v, int lane, float *total) {
    float inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const float o = __shfl_down(inc, d, 64);
        if (lane + d < 64) inc += o;
    }
    *total = __shfl(inc, 0, 64);
    float ex = __shfl_down(inc, 1, 64);
    if (lane == 63) ex = 0.0f;
    return ex;
}

__device__ __forceinline__ float prefix_excl_prod(float f, int lane, float *total) {
    const float incl = seg_scan_mul<64>(f, lane);
    float ex = __shfl_up(incl, 1, 64);
    if (lane == 0) ex = 1.0f;
    *total = __shfl(incl, 63, 64);
    return ex;
}

__global__ __launch_bounds__(kWaves * 64) void composite_bwd_kernel(
    const float4 *__restrict__ raw, const float *__restrict__ z, const float *__restrict__ dir,
    const float *__restrict__ dists, const float *__restrict__ noise, float noise_std, int white_bkgd, int R, int S,
    const float *__restrict__ g_rgb, const float *__restrict__ g_depth, const float *__restrict__ g_acc,
    const float *__restrict__ g_w, float4 *__restrict__ g_raw) {
    __shared__ float carry_in[kWaves][kMaxChunks];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int r = blockIdx.x * kWaves + wv;
    if (r >= R) return;
    float dn = 0.0f;
    if (!dists) {
        const float dx = dir[3 * r], dy = dir[3 * r + 1], dzv = dir[3 * r + 2];
        dn = sqrtf(dx * dx + dy * dy + dzv * dzv);
    }
    const float *zr = z + (size_t)r * S, *dr = dists ? dists + (size_t)r * S : nullptr;
    const float gr = g_rgb ? g_rgb[3 * r] : 0.f, gg = g_rgb ? g_rgb[3 * r + 1] : 0.f,
                gb = g_rgb ? g_rgb[3 * r + 2] : 0.f;
    const float gd = g_depth ? g_depth[r] : 0.f;
    const float ga = (g_acc ? g_acc[r] : 0.f) - (white_bkgd ? gr + gg + gb : 0.f);
    const int nchunk = (S + 63) / 64;
    // walk 1: entry transmittance of every chunk
    float carry = 1.0f;
    for (int c = 0; c < nchunk; c++) {
        const int s = c * 64 + lane;
        float f = 1.0f;
        if (s < S) {
            float sig = raw[(size_t)r * S + s].w;
            if (noise) sig += noise[(size_t)r * S + s] * noise_std;
            const float zz = zr[s];
            const float alpha = 1.0f - expf(-fmaxf(sig, 0.f) * dist_of(zr, dr, s, S, zz, dn));
            f = 1.0f - alpha + 1e-10f;
        }
        if (lane == 0) carry_in[wv][c] = carry;
        float tot;
        prefix_excl_prod(f, lane, &tot);
        carry *= tot;
    }
    // walk 2: from the far end, carrying sum_{k in later chunks} G_k w_k
    float later = 0.0f;
    for (int c = nchunk - 1; c >= 0; c--) {
        const int s = c * 64 + lane;
        const bool on = s < S;
        float4 v = make_float4(0, 0, 0, 0);
        float zz = 0.f, sigp = 0.f, dist = 0.f;
        if (on) {
            v = raw[(size_t)r * S + s];
            sigp = v.w + (noise ? noise[(size_t)r * S + s] * noise_std : 0.f);
            zz = zr[s];
            dist = dist_of(zr, dr, s, S, zz, dn);
        }
        const float sig = fmaxf(sigp, 0.f);
        const float e = on ? expf(-sig * dist) : 1.0f;
        const float alpha = 1.0f - e, f = on ? 1.0f - alpha + 1e-10f : 1.0f;
        float tot;
        const float T = carry_in[wv][c] * prefix_excl_prod(f, lane, &tot);
        const float w = alpha * T;
        const float cr = zest_sigmoid(v.x), cg = zest_sigmoid(v.y), cb = zest_sigmoid(v.z);
        const float G = on ? (g_w ? g_w[(size_t)r * S + s] : 0.f) + gr * cr + gg * cg + gb * cb + gd * zz + ga
                           : 0.f;
        float chunk_sum;
        const float after = suffix_excl_sum(G * w, lane, &chunk_sum) + later;
        later += chunk_sum;
        if (on) {
            const float d_alpha = G * T - after / f;
            const float d_sig = sigp > 0.f ? d_alpha * dist * e : 0.f;
            g_raw[(size_t)r * S + s] = make_float4(gr * w * cr * (1.f - cr), gg * w * cg * (1.f - cg),
                                                   gb * w * cb * (1.f - cb), d_sig);
        }
    }
}

__global__ __launch_bounds__(kWaves * 64) void composite_blend_bwd_kernel(
    const float4 *__restrict__ raw_dy, const float4 *__restrict__ raw_st, const float *__restrict__ blend,
    const float *__restrict__ z, const float *__restrict__ dir, const float *__restrict__ dists,
    const float *__restrict__ noise,
    float noise_std, int R, int S, const float *__restrict__ g_rgb, const float *__restrict__ g_depth,
    const float *__restrict__ g_rgb_fg, const float *__restrict__ g_depth_fg,
    const float *__restrict__ g_wfg, const float *__restrict__ g_wd, float4 *__restrict__ g_raw_dy,
    float4 *__restrict__ g_raw_st, float *__restrict__ g_blend) {
    __shared__ float carry_in[kWaves][kMaxChunks], carry_fg[kWaves][kMaxChunks];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int r = blockIdx.x * kWaves + wv;
    if (r >= R) return;
    float dn = 0.0f;
    if (!dists) {
        const float dx = dir[3 * r], dy = dir[3 * r + 1], dzv = dir[3 * r + 2];
        dn = sqrtf(dx * dx + dy * dy + dzv * dzv);
    }
    const float *zr = z + (size_t)r * S, *dr = dists ? dists + (size_t)r * S : nullptr;
    float gm[3] = {0, 0, 0}, gf[3] = {0, 0, 0};
    if (g_rgb) gm[0] = g_rgb[3 * r], gm[1] = g_rgb[3 * r + 1], gm[2] = g_rgb[3 * r + 2];
    if (g_rgb_fg) gf[0] = g_rgb_fg[3 * r], gf[1] = g_rgb_fg[3 * r + 1], gf[2] = g_rgb_fg[3 * r + 2];
    const float gd = g_depth ? g_depth[r] : 0.f, gdf = g_depth_fg ? g_depth_fg[r] : 0.f;
    const int nchunk = (S + 63) / 64;

    auto alphas = [&](int s, float4 &vd, float4 &vs, float &b, float &zz, float &dist, float &ed, float &es,
                      float &spd, float &sps) {
        const size_t i = (size_t)r * S + s;
        vd = raw_dy[i], vs = raw_st[i], b = blend[i], zz = zr[s];
        const float n = noise ? noise[i] * noise_std : 0.f;
        dist = dist_of(zr, dr, s, S, zz, dn);
        spd = vd.w + n, sps = vs.w + n;
        ed = expf(-fmaxf(spd, 0.f) * dist), es = expf(-fmaxf(sps, 0.f) * dist);
    };
    float carry = 1.f, carryf = 1.f;
    for (int c = 0; c < nchunk; c++) {
        const int s = c * 64 + lane;
        float f = 1.f, ff = 1.f;
        if (s < S) {
            float4 vd, vs;
            float b, zz, dist, ed, es, spd, sps;
            alphas(s, vd, vs, b, zz, dist, ed, es, spd, sps);
            const float a_fg = 1.f - ed, a_d = a_fg * b, a_s = (1.f - es) * (1.f - b);
            f = (1.f - a_d) * (1.f - a_s) + 1e-10f, ff = 1.f - a_fg + 1e-10f;
        }
        if (lane == 0) carry_in[wv][c] = carry, carry_fg[wv][c] = carryf;
        float t1, t2;
        prefix_excl_prod(f, lane, &t1);
        prefix_excl_prod(ff, lane, &t2);
        carry *= t1, carryf *= t2;
    }
    float later = 0.f, later_fg = 0.f;
    for (int c = nchunk - 1; c >= 0; c--) {
        const int s = c * 64 + lane;
        const bool on = s < S;
        float4 vd = make_float4(0, 0, 0, 0), vs = vd;
        float b = 0.f, zz = 0.f, dist = 0.f, ed = 1.f, es = 1.f, spd = 0.f, sps = 0.f;
        if (on) alphas(s, vd, vs, b, zz, dist, ed, es, spd, sps);
        const float a_fg = 1.f - ed, a_sr = 1.f - es;
        const float a_d = a_fg * b, a_s = a_sr * (1.f - b);
        const float f = on ? (1.f - a_d) * (1.f - a_s) + 1e-10f : 1.f, ff = on ? 1.f - a_fg + 1e-10f : 1.f;
        float t1, t2;
        const float T = carry_in[wv][c] * prefix_excl_prod(f, lane, &t1);
        const float Tf = carry_fg[wv][c] * prefix_excl_prod(ff, lane, &t2);
        const float wd = T * a_d, ws = T * a_s, wf = a_fg * Tf;
        const float cd[3] = {zest_sigmoid(vd.x), zest_sigmoid(vd.y), zest_sigmoid(vd.z)};
        const float cs[3] = {zest_sigmoid(vs.x), zest_sigmoid(vs.y), zest_sigmoid(vs.z)};
        float Gd = 0.f, Gs = 0.f, Gf = 0.f;
        if (on) {
            const size_t i = (size_t)r * S + s;
            Gd = (g_wd ? g_wd[i] : 0.f) + gm[0] * cd[0] + gm[1] * cd[1] + gm[2] * cd[2] + gd * zz;
            Gs = gm[0] * cs[0] + gm[1] * cs[1] + gm[2] * cs[2] + gd * zz;
            Gf = (g_wfg ? g_wfg[i] : 0.f) + gf[0] * cd[0] + gf[1] * cd[1] + gf[2] * cd[2] + gdf * zz;
        }
        float cs1, cs2;
        const float after = suffix_excl_sum(Gd * wd + Gs * ws, lane, &cs1) + later;
        const float after_fg = suffix_excl_sum(Gf * wf, lane, &cs2) + later_fg;
        later += cs1, later_fg += cs2;
        if (on) {
            const size_t i = (size_t)r * S + s;
            const float dLdf = after / f;                       // through T of the later samples
            const float d_ad = Gd * T - dLdf * (1.f - a_s);
            const float d_as = Gs * T - dLdf * (1.f - a_d);
            const float d_afg = (Gf * Tf - after_fg / ff) + d_ad * b;
            const float d_asr = d_as * (1.f - b);
            g_blend[i] = d_ad * a_fg - d_as * a_sr;
            float4 od, os;
            od.x = (gm[0] * wd + gf[0] * wf) * cd[0] * (1.f - cd[0]);
            od.y = (gm[1] * wd + gf[1] * wf) * cd[1] * (1.f - cd[1]);
            od.z = (gm[2] * wd + gf[2] * wf) * cd[2] * (1.f - cd[2]);
            od.w = spd > 0.f ? d_afg * dist * ed : 0.f;
            os.x = gm[0] * ws * cs[0] * (1.f - cs[0]);
            os.y = gm[1] * ws * cs[1] * (1.f - cs[1]);
            os.z = gm[2] * ws * cs[2] * (1.f - cs[2]);
            os.w = sps > 0.f ? d_asr * dist * es : 0.f;
            g_raw_dy[i] = od, g_raw_st[i] = os;
        }
    }
}

}  // namespace

extern "C" int zest_composite_bwd(const float *raw, const float *z, const float *rays_dir, const float *dists,
                                  const float *noise, float noise_std, int white_bkgd, int R, int S,
                                  const float *g_rgb_map, const float *g_depth_map, const float *g_acc_map,
                                  const float *g_weights, float *g_raw, void *stream) {
    if (R == 0) return 0;                       // an empty batch is a no-op (its tensors have no storage)
    ZEST_CHECK_ARG(raw && z && (rays_dir || dists) && g_raw, "zest_composite_bwd: raw, z, rays_dir (or dists), g_raw required");
    ZEST_CHECK_ARG(R >= 0 && S >= 1 && S <= 64 * kMaxChunks, "zest_composite_bwd: bad shape R=%d S=%d", R, S);
    ZEST_CHECK_ARG((((uintptr_t)raw | (uintptr_t)g_raw) & 15) == 0, "zest_composite_bwd: 16-byte alignment");
    if (R == 0) return 0;
    hipLaunchKernelGGL(composite_bwd_kernel, dim3(zest_div_up(R, kWaves)), dim3(kWaves * 64), 0,
                       (hipStream_t)stream, (const float4 *)raw, z, rays_dir, dists, noise, noise_std, white_bkgd, R,
                       S, g_rgb_map, g_depth_map, g_acc_map, g_weights, (float4 *)g_raw);
    ZEST_RETURN_LAUNCH("zest_composite_bwd");
}

extern "C" int zest_composite_blend_bwd(const float *raw_dy, const float *raw_st, const float *blend,
                                        const float *z, const float *rays_dir, const float *dists,
                                        const float *noise,
                                        float noise_std, int R, int S, const float *g_rgb_map,
                                        const float *g_depth_map, const float *g_rgb_map_fg,
                                        const float *g_depth_map_fg, const float *g_weights_fg,
                                        const float *g_weights_dy, float *g_raw_dy, float *g_raw_st,
                                        float *g_blend, void *stream) {
    if (R == 0) return 0;                       // an empty batch is a no-op (its tensors have no storage)
    ZEST_CHECK_ARG(raw_dy && raw_st && blend && z && (rays_dir || dists) && g_raw_dy && g_raw_st && g_blend,
                   "zest_composite_blend_bwd: null argument");
    ZEST_CHECK_ARG(R >= 0 && S >= 1 && S <= 64 * kMaxChunks, "zest_composite_blend_bwd: bad shape");
    ZEST_CHECK_ARG((((uintptr_t)raw_dy | (uintptr_t)raw_st | (uintptr_t)g_raw_dy | (uintptr_t)g_raw_st) & 15) == 0,
                   "zest_composite_blend_bwd: 16-byte alignment");
    if (R == 0) return 0;
    hipLaunchKernelGGL(composite_blend_bwd_kernel, dim3(zest_div_up(R, kWaves)), dim3(kWaves * 64), 0,
                       (hipStream_t)stream, (const float4 *)raw_dy, (const float4 *)raw_st, blend, z, rays_dir, dists,
                       noise, noise_std, R, S, g_rgb_map, g_depth_map, g_rgb_map_fg, g_depth_map_fg,
                       g_weights_fg, g_weights_dy, (float4 *)g_raw_dy, (float4 *)g_raw_st, g_blend);
    ZEST_RETURN_LAUNCH("zest_composite_blend_bwd");
}
