// Ray sampling: the step right before the renderer.
// One thread per sample: depth candidate (uniform in [near, far], optionally jittered inside
// its stratum), world point o + z d, and its coordinates in the reference view's padded
// encoding volume.  Replaces the per-sample part of build_rays_base and get_ndc_coordinate
// (reference utils.py:232-288, 361-387); pixel selection stays on the host (it is R integers
// and defines the RNG call order).  HBM-bound: 4 B in (jitter), 28 B out per sample.
#include "zest_sample_ops.cuh"

namespace {

struct RayCams {                     // device pointers into the batch's camera tensors
    const float *k_tgt, *c2w_tgt;    // target view: intrinsics [3,3], camera-to-world [4,4]
    const float *w2c_ref, *k_ref;    // reference view (volume frame)
    const float *nf_tgt, *nf_ref;    // (near, far) of the two views
};

// torch.linspace(0, 1, S)[i] in fp32: forward from the start for the first half, backward
// from the end for the second (the library's symmetric evaluation)
__device__ __forceinline__ float linspace01(int i, int S) {
    if (S == 1) return 0.0f;
    const float step = 1.0f / (float)(S - 1);
    return i < S / 2 ? step * (float)i : 1.0f - step * (float)(S - 1 - i);
}

__device__ __forceinline__ void ndc_of(const float *w2c, const float *K, float px, float py, float pz,
                                       float inv_w, float inv_h, float near, float far, int pad,
                                       int lindisp, float out[3]) {
    float cx = px, cy = py, cz = pz;
    if (w2c) {
        cx = fmaf(pz, w2c[2], fmaf(py, w2c[1], px * w2c[0])) + w2c[3];
        cy = fmaf(pz, w2c[6], fmaf(py, w2c[5], px * w2c[4])) + w2c[7];
        cz = fmaf(pz, w2c[10], fmaf(py, w2c[9], px * w2c[8])) + w2c[11];
    }
    const float qx = fmaf(cz, K[2], fmaf(cy, K[1], cx * K[0]));
    const float qy = fmaf(cz, K[5], fmaf(cy, K[4], cx * K[3]));
    const float qz = fmaf(cz, K[8], fmaf(cy, K[7], cx * K[6]));
    float u = (qx / qz) / inv_w, v = (qy / qz) / inv_h;
    const float zn = lindisp ? (1.0f / qz - 1.0f / near) / (1.0f / far - 1.0f / near)
                             : (qz - near) / (far - near);
    if (pad > 0) {
        const float wf = (inv_w + 1.0f) / 4.0f, hf = (inv_h + 1.0f) / 4.0f;
        u = u * wf / (wf + pad * 2) + pad / (wf + pad * 2);
        v = v * hf / (hf + pad * 2) + pad / (hf + pad * 2);
    }
    out[0] = u, out[1] = v, out[2] = zn;
}

__global__ void build_rays_kernel(RayCams c, const float *__restrict__ xs, const float *__restrict__ ys,
                                  const float *__restrict__ t_rand, int R, int S, int pad, int W, int H,
                                  float *__restrict__ rays_dir, float *__restrict__ depth,
                                  float *__restrict__ pts, float *__restrict__ ndc) {
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= R * S) return;
    const int r = m / S, s = m % S;
    const float near_tgt = c.nf_tgt[0], far_tgt = c.nf_tgt[1], near_ref = c.nf_ref[0], far_ref = c.nf_ref[1];
    const float dx = (xs[r] - c.k_tgt[2]) / c.k_tgt[0], dy = (ys[r] - c.k_tgt[5]) / c.k_tgt[4];
    float d[3];
#pragma unroll
    for (int i = 0; i < 3; i++) d[i] = dx * c.c2w_tgt[4 * i] + dy * c.c2w_tgt[4 * i + 1] + c.c2w_tgt[4 * i + 2];
    if (s == 0) rays_dir[3 * r] = d[0], rays_dir[3 * r + 1] = d[1], rays_dir[3 * r + 2] = d[2];
    auto zc = [&](int i) {
        const float t = linspace01(i, S);
        return near_tgt * (1.0f - t) + far_tgt * t;
    };
    float z = zc(s);
    if (t_rand) {       // stratified: jitter inside [mid(s-1,s), mid(s,s+1)], clamped at the ends
        const float lower = s > 0 ? 0.5f * (z + zc(s - 1)) : z;
        const float upper = s + 1 < S ? 0.5f * (zc(s + 1) + z) : z;
        z = lower + (upper - lower) * t_rand[m];
    }
    depth[m] = z;
    const float px = c.c2w_tgt[3] + z * d[0], py = c.c2w_tgt[7] + z * d[1], pz = c.c2w_tgt[11] + z * d[2];
    pts[3 * m] = px, pts[3 * m + 1] = py, pts[3 * m + 2] = pz;
    float o[3];
    ndc_of(c.w2c_ref, c.k_ref, px, py, pz, (float)(W - 1), (float)(H - 1), near_ref, far_ref, pad, 0, o);
    ndc[3 * m] = o[0], ndc[3 * m + 1] = o[1], ndc[3 * m + 2] = o[2];
}

__global__ void ndc_kernel(const float *__restrict__ w2c, const float *__restrict__ K,
                           const float *__restrict__ pts, int M, float inv_w, float inv_h, float near,
                           float far, int pad, int lindisp, float *__restrict__ out) {
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= M) return;
    float o[3];
    ndc_of(w2c, K, pts[3 * m], pts[3 * m + 1], pts[3 * m + 2], inv_w, inv_h, near, far, pad, lindisp, o);
    out[3 * m] = o[0], out[3 * m + 1] = o[1], out[3 * m + 2] = o[2];
}

// Hierarchical (coarse -> fine) resampling, the `sample_pdf` BASELINE.json's north_star names.  The
// reference has no such function ("Read this first" 1): this is the canonical NeRF inverse-CDF
// sampler, a build extension with unpinned parity, tested by its properties and against a numpy
// inverse CDF.  One wave per ray: pdf = (w + 1e-5) / sum, its running sum by a shuffle scan into LDS
// (cdf[0] = 0), then every fine sample finds its bin by binary search in LDS and interpolates inside it.
constexpr int kMaxBins = 512;

__global__ __launch_bounds__(256) void sample_pdf_kernel(const float *__restrict__ bins, const float *__restrict__ weights,
                                                         const float *__restrict__ u_in, int R, int Nb, int Ns,
                                                         float *__restrict__ out) {
    __shared__ float cdf_s[4][kMaxBins + 1];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int r = blockIdx.x * 4 + wv;
    if (r >= R) return;                                    // wave-uniform
    float *cdf = cdf_s[wv];
    const float *w = weights + (size_t)r * Nb, *b = bins + (size_t)r * (Nb + 1);
    float total = 0.0f;
    for (int i = lane; i < Nb; i += 64) total += w[i] + 1e-5f;
    total = wave_sum(total);
    float carry = 0.0f;
    if (lane == 0) cdf[0] = 0.0f;
    for (int i0 = 0; i0 < Nb; i0 += 64) {
        const int i = i0 + lane;
        float v = i < Nb ? (w[i] + 1e-5f) / total : 0.0f;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const float o = __shfl_up(v, d, 64);
            if (lane >= d) v += o;
        }
        if (i < Nb) cdf[i + 1] = carry + v;
        carry += __shfl(v, 63, 64);
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);                    // lgkmcnt(0): the wave's own LDS writes are visible to its reads
    for (int j = lane; j < Ns; j += 64) {
        const float u = u_in ? u_in[(size_t)r * Ns + j] : (Ns > 1 ? (float)j / (float)(Ns - 1) : 0.5f);
        int lo = 0, hi = Nb + 1;                           // first index with cdf[idx] > u  (searchsorted, right = True)
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (cdf[mid] > u) hi = mid;
            else lo = mid + 1;
        }
        const int below = max(lo - 1, 0), above = min(lo, Nb);
        const float c0 = cdf[below], c1 = cdf[above];
        float den = c1 - c0;
        if (den < 1e-5f) den = 1.0f;
        const float t = (u - c0) / den;
        out[(size_t)r * Ns + j] = b[below] + t * (b[above] - b[below]);
    }
}

}  // namespace

extern "C" int zest_sample_pdf_fwd(const float *bins, const float *weights, const float *u, int R, int n_bins,
                                   int n_samples, float *samples, void *stream) {
    if (R == 0) return 0;                       // an empty batch is a no-op (its tensors have no storage)
    ZEST_CHECK_ARG(bins && weights && samples, "zest_sample_pdf_fwd: null argument");
    ZEST_CHECK_ARG(R >= 0 && n_bins >= 1 && n_bins <= kMaxBins && n_samples >= 1,
                   "zest_sample_pdf_fwd: bad shape R=%d bins=%d samples=%d (at most %d bins)", R, n_bins, n_samples, kMaxBins);
    if (R == 0) return 0;
    hipLaunchKernelGGL(sample_pdf_kernel, dim3(zest_div_up(R, 4)), dim3(256), 0, (hipStream_t)stream, bins, weights, u, R,
                       n_bins, n_samples, samples);
    ZEST_RETURN_LAUNCH("zest_sample_pdf_fwd");
}

extern "C" int zest_build_rays_fwd(const float *xs, const float *ys, const float *t_rand, int R, int S,
                                   const float *k_tgt, const float *c2w_tgt, const float *w2c_ref,
                                   const float *k_ref, const float *near_far_tgt,
                                   const float *near_far_ref, int pad, int W, int H, float *rays_dir,
                                   float *depth, float *pts, float *ndc, void *stream) {
    if (R == 0) return 0;                       // an empty batch is a no-op (its tensors have no storage)
    ZEST_CHECK_ARG(xs && ys && k_tgt && c2w_tgt && w2c_ref && k_ref && near_far_tgt && near_far_ref &&
                       rays_dir && depth && pts && ndc, "zest_build_rays_fwd: null argument");
    ZEST_CHECK_ARG(R >= 0 && S >= 1 && W >= 2 && H >= 2 && pad >= 0, "zest_build_rays_fwd: bad shape");
    if (R == 0) return 0;
    const RayCams c{k_tgt, c2w_tgt, w2c_ref, k_ref, near_far_tgt, near_far_ref};
    hipLaunchKernelGGL(build_rays_kernel, dim3(zest_div_up((long long)R * S, 256)), dim3(256), 0,
                       (hipStream_t)stream, c, xs, ys, t_rand, R, S, pad, W, H, rays_dir, depth, pts, ndc);
    ZEST_RETURN_LAUNCH("zest_build_rays_fwd");
}

extern "C" int zest_ndc_fwd(const float *pts, int M, const float *w2c, const float *k, float inv_w,
                            float inv_h, float near, float far, int pad, int lindisp, float *out,
                            void *stream) {
    if (M == 0) return 0;                       // an empty batch is a no-op (its tensors have no storage)
    ZEST_CHECK_ARG(pts && k && out, "zest_ndc_fwd: null argument");
    ZEST_CHECK_ARG(M >= 0 && pad >= 0, "zest_ndc_fwd: bad shape");
    if (M == 0) return 0;
    hipLaunchKernelGGL(ndc_kernel, dim3(zest_div_up(M, 256)), dim3(256), 0, (hipStream_t)stream, w2c, k, pts,
                       M, inv_w, inv_h, near, far, pad, lindisp, out);
    ZEST_RETURN_LAUNCH("zest_ndc_fwd");
}
