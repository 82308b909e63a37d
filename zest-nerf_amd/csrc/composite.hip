// Alpha compositing along rays: one wavefront per ray, samples taken 64 at a time in lane
// order (coalesced float4 reads of the raw predictions), the exclusive transmittance product
// as a shuffle scan with a carry between 64-sample chunks, per-ray sums by butterfly
// reduction.  HBM-bound: 28 B read + up to 8 B written per sample.
//
// Replaces depth2dist, raw2alpha, raw2outputs, raw2outputs_blending and compute_2d_prob of
// the reference (renderer.py:22-32, 74-219).
#include "zest_common.cuh"

namespace {

constexpr int kWavesPerBlock = 4;

__device__ __forceinline__ float sample_dist(const float *__restrict__ zr, const float *__restrict__ dr, int s,
                                             int S, float z, float dnorm) {
    // distance to the next sample; the last one is "infinite" (1e10), both scaled by |dir|
    // (depth2dist, reference renderer.py:74-89) - or the caller's own spacings when it hands them in
    if (dr) return dr[s];
    const float dz = (s + 1 < S) ? (zr[s + 1] - z) : 1e10f;
    return dz * dnorm;
}

__global__ __launch_bounds__(kWavesPerBlock *ZEST_WAVE) void composite_kernel(
    const float4 *__restrict__ raw, const float *__restrict__ z, const float *__restrict__ dir,
    const float *__restrict__ dists, const float *__restrict__ noise, float noise_std, int white_bkgd, int R, int S,
    float *__restrict__ rgb_map, float *__restrict__ depth_map, float *__restrict__ acc_map,
    float *__restrict__ disp_map, float *__restrict__ weights, float *__restrict__ alpha_out) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    if (r >= R) return;                                   // wave-uniform
    float dnorm = 0.0f;
    if (!dists) {
        const float dx = dir[3 * r], dy = dir[3 * r + 1], dzv = dir[3 * r + 2];
        dnorm = sqrtf(dx * dx + dy * dy + dzv * dzv);
    }
    const float *zr = z + (size_t)r * S, *drow = dists ? dists + (size_t)r * S : nullptr;
    float carry = 1.0f, a_r = 0.f, a_g = 0.f, a_b = 0.f, a_d = 0.f, a_w = 0.f;
    for (int s0 = 0; s0 < S; s0 += 64) {
        const int s = s0 + lane;
        const bool on = s < S;
        float alpha = 0.f, zz = 0.f, cr = 0.f, cg = 0.f, cb = 0.f;
        if (on) {
            const float4 v = raw[(size_t)r * S + s];
            zz = zr[s];
            float sig = v.w;
            if (noise) sig += noise[(size_t)r * S + s] * noise_std;
            sig = fmaxf(sig, 0.f);
            alpha = 1.0f - expf(-sig * sample_dist(zr, drow, s, S, zz, dnorm));
            cr = zest_sigmoid(v.x), cg = zest_sigmoid(v.y), cb = zest_sigmoid(v.z);
        }
        const float f = on ? (1.0f - alpha + 1e-10f) : 1.0f;
        const float incl = seg_scan_mul<64>(f, lane);
        float excl = __shfl_up(incl, 1, 64);
        if (lane == 0) excl = 1.0f;
        const float w = alpha * (carry * excl);
        carry *= __shfl(incl, 63, 64);
        if (on) {
            if (weights) weights[(size_t)r * S + s] = w;
            if (alpha_out) alpha_out[(size_t)r * S + s] = alpha;
        }
        a_r += w * cr, a_g += w * cg, a_b += w * cb, a_d += w * zz, a_w += w;
    }
    a_r = wave_sum(a_r), a_g = wave_sum(a_g), a_b = wave_sum(a_b);
    a_d = wave_sum(a_d), a_w = wave_sum(a_w);
    if (lane == 0) {
        const float bg = white_bkgd ? (1.0f - a_w) : 0.0f;
        if (rgb_map) {
            rgb_map[3 * r] = a_r + bg, rgb_map[3 * r + 1] = a_g + bg, rgb_map[3 * r + 2] = a_b + bg;
        }
        if (depth_map) depth_map[r] = a_d;
        if (acc_map) acc_map[r] = a_w;
        if (disp_map) disp_map[r] = 1.0f / fmaxf(1e-10f, a_d / a_w);
    }
}

__global__ __launch_bounds__(kWavesPerBlock *ZEST_WAVE) void composite_blend_kernel(
    const float4 *__restrict__ raw_dy, const float4 *__restrict__ raw_st,
    const float *__restrict__ blend, const float *__restrict__ z, const float *__restrict__ dir,
    const float *__restrict__ dists, const float *__restrict__ noise, float noise_std, int R, int S,
    float *__restrict__ rgb_map,
    float *__restrict__ depth_map, float *__restrict__ rgb_map_fg, float *__restrict__ depth_map_fg,
    float *__restrict__ weights_fg, float *__restrict__ weights_dy, float *__restrict__ wdd_sum) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    if (r >= R) return;
    float dnorm = 0.0f;
    if (!dists) {
        const float dx = dir[3 * r], dy = dir[3 * r + 1], dzv = dir[3 * r + 2];
        dnorm = sqrtf(dx * dx + dy * dy + dzv * dzv);
    }
    const float *zr = z + (size_t)r * S, *drow = dists ? dists + (size_t)r * S : nullptr;
    float carry = 1.0f, carry_fg = 1.0f;
    float m_r = 0.f, m_g = 0.f, m_b = 0.f, m_d = 0.f, f_r = 0.f, f_g = 0.f, f_b = 0.f, f_d = 0.f,
          dd = 0.f;
    for (int s0 = 0; s0 < S; s0 += 64) {
        const int s = s0 + lane;
        const bool on = s < S;
        float a_fg = 0.f, a_d = 0.f, a_s = 0.f, zz = 0.f;
        float dr = 0.f, dg = 0.f, db = 0.f, sr = 0.f, sg = 0.f, sb = 0.f;
        if (on) {
            const size_t i = (size_t)r * S + s;
            const float4 vd = raw_dy[i], vs = raw_st[i];
            const float b = blend[i];
            zz = zr[s];
            const float n = noise ? noise[i] * noise_std : 0.0f;
            const float dist = sample_dist(zr, drow, s, S, zz, dnorm);
            a_fg = 1.0f - expf(-fmaxf(vd.w + n, 0.f) * dist);
            a_d = a_fg * b;
            a_s = (1.0f - expf(-fmaxf(vs.w + n, 0.f) * dist)) * (1.0f - b);
            dr = zest_sigmoid(vd.x), dg = zest_sigmoid(vd.y), db = zest_sigmoid(vd.z);
            sr = zest_sigmoid(vs.x), sg = zest_sigmoid(vs.y), sb = zest_sigmoid(vs.z);
        }
        const float f = on ? ((1.0f - a_d) * (1.0f - a_s) + 1e-10f) : 1.0f;
        const float g = on ? (1.0f - a_fg + 1e-10f) : 1.0f;
        const float incl = seg_scan_mul<64>(f, lane), incl_fg = seg_scan_mul<64>(g, lane);
        float excl = __shfl_up(incl, 1, 64), excl_fg = __shfl_up(incl_fg, 1, 64);
        if (lane == 0) excl = 1.0f, excl_fg = 1.0f;
        const float T = carry * excl;
        const float wd = T * a_d, ws = T * a_s, wf = a_fg * (carry_fg * excl_fg);
        carry *= __shfl(incl, 63, 64);
        carry_fg *= __shfl(incl_fg, 63, 64);
        if (on) {
            const size_t i = (size_t)r * S + s;
            if (weights_fg) weights_fg[i] = wf;
            if (weights_dy) weights_dy[i] = wd;
        }
        m_r += wd * dr + ws * sr, m_g += wd * dg + ws * sg, m_b += wd * db + ws * sb;
        m_d += (wd + ws) * zz;
        f_r += wf * dr, f_g += wf * dg, f_b += wf * db, f_d += wf * zz;
        dd += wd;
    }
    m_r = wave_sum(m_r), m_g = wave_sum(m_g), m_b = wave_sum(m_b), m_d = wave_sum(m_d);
    f_r = wave_sum(f_r), f_g = wave_sum(f_g), f_b = wave_sum(f_b), f_d = wave_sum(f_d);
    dd = wave_sum(dd);
    if (lane == 0) {
        if (rgb_map) rgb_map[3 * r] = m_r, rgb_map[3 * r + 1] = m_g, rgb_map[3 * r + 2] = m_b;
        if (depth_map) depth_map[r] = m_d;
        if (rgb_map_fg)
            rgb_map_fg[3 * r] = f_r, rgb_map_fg[3 * r + 1] = f_g, rgb_map_fg[3 * r + 2] = f_b;
        if (depth_map_fg) depth_map_fg[r] = f_d;
        if (wdd_sum) wdd_sum[r] = dd;
    }
}

__global__ __launch_bounds__(kWavesPerBlock *ZEST_WAVE) void weighted_complement_kernel(
    const float *__restrict__ w, const float *__restrict__ p, int R, int S, float *__restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    if (r >= R) return;
    float acc = 0.f;
    for (int s = lane; s < S; s += 64) acc += w[(size_t)r * S + s] * (1.0f - p[(size_t)r * S + s]);
    acc = wave_sum(acc);
    if (lane == 0) out[r] = acc;
}

}  // namespace

extern "C" int zest_composite_fwd(const float *raw, const float *z, const float *rays_dir, const float *dists,
                                  const float *noise, float noise_std, int white_bkgd, int R, int S,
                                  float *rgb_map, float *depth_map, float *acc_map, float *disp_map,
                                  float *weights, float *alpha, void *stream) {
    if (R == 0) return 0;                       // an empty batch is a no-op (its tensors have no storage)
    ZEST_CHECK_ARG(raw && z && (rays_dir || dists), "zest_composite_fwd: raw, z and rays_dir (or dists) are required");
    ZEST_CHECK_ARG(R >= 0 && S >= 1, "zest_composite_fwd: bad shape R=%d S=%d", R, S);
    ZEST_CHECK_ARG(((uintptr_t)raw & 15) == 0, "zest_composite_fwd: raw must be 16-byte aligned");
    if (R == 0) return 0;
    hipLaunchKernelGGL(composite_kernel, dim3(zest_div_up(R, kWavesPerBlock)),
                       dim3(kWavesPerBlock * ZEST_WAVE), 0, (hipStream_t)stream,
                       (const float4 *)raw, z, rays_dir, dists, noise, noise_std, white_bkgd, R, S,
                       rgb_map, depth_map, acc_map, disp_map, weights, alpha);
    ZEST_RETURN_LAUNCH("zest_composite_fwd");
}

extern "C" int zest_composite_blend_fwd(const float *raw_dy, const float *raw_st,
                                        const float *blend, const float *z, const float *rays_dir,
                                        const float *dists, const float *noise, float noise_std, int R, int S,
                                        float *rgb_map, float *depth_map, float *rgb_map_fg,
                                        float *depth_map_fg, float *weights_fg, float *weights_dy,
                                        float *weights_dd_sum, void *stream) {
    if (R == 0) return 0;                       // an empty batch is a no-op (its tensors have no storage)
    ZEST_CHECK_ARG(raw_dy && raw_st && blend && z && (rays_dir || dists),
                   "zest_composite_blend_fwd: raw_dy, raw_st, blend, z, rays_dir (or dists) are required");
    ZEST_CHECK_ARG(R >= 0 && S >= 1, "zest_composite_blend_fwd: bad shape R=%d S=%d", R, S);
    ZEST_CHECK_ARG((((uintptr_t)raw_dy | (uintptr_t)raw_st) & 15) == 0,
                   "zest_composite_blend_fwd: raw tensors must be 16-byte aligned");
    if (R == 0) return 0;
    hipLaunchKernelGGL(composite_blend_kernel, dim3(zest_div_up(R, kWavesPerBlock)),
                       dim3(kWavesPerBlock * ZEST_WAVE), 0, (hipStream_t)stream,
                       (const float4 *)raw_dy, (const float4 *)raw_st, blend, z, rays_dir, dists, noise,
                       noise_std, R, S, rgb_map, depth_map, rgb_map_fg, depth_map_fg, weights_fg,
                       weights_dy, weights_dd_sum);
    ZEST_RETURN_LAUNCH("zest_composite_blend_fwd");
}

extern "C" int zest_weighted_complement_sum(const float *w, const float *p, int R, int S,
                                            float *out, void *stream) {
    if (R == 0) return 0;                       // an empty batch is a no-op (its tensors have no storage)
    ZEST_CHECK_ARG(w && p && out, "zest_weighted_complement_sum: null pointer");
    ZEST_CHECK_ARG(R >= 0 && S >= 1, "zest_weighted_complement_sum: bad shape R=%d S=%d", R, S);
    if (R == 0) return 0;
    hipLaunchKernelGGL(weighted_complement_kernel, dim3(zest_div_up(R, kWavesPerBlock)),
                       dim3(kWavesPerBlock * ZEST_WAVE), 0, (hipStream_t)stream, w, p, R, S, out);
    ZEST_RETURN_LAUNCH("zest_weighted_complement_sum");
}
