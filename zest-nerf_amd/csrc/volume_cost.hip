// Plane-sweep cost volume of the MVS volume builder.
//
// Replaces MVSNet.build_volume_cost (reference networks.py:1077-1140) and utils.homo_warp
// (utils.py:49-99).  The reference materialises, per source view, a [C,D,H,W] warped feature
// volume, its square, a warped image volume and a sampling grid, and reduces them with a dozen
// elementwise passes.  Here one thread owns one voxel (d, y, x) of the padded reference grid:
// it projects the voxel into every source view (homography at the plane's depth), takes the
// four bilinear taps of the 32-channel feature map (channels-last: 128 contiguous bytes per
// tap) and of the image, keeps the running sum / sum of squares / in-frame count in registers
// and writes the 3 V image channels, the 32 variance channels and the V masks once.  Bound: the
// HBM write of the output (41 x D x Hp x Wp floats); the source maps are a few MB and stay in L2.
#include "zest_common.cuh"
#include "../../include/zest_render.h"

namespace {

constexpr int kThreads = 256;
constexpr int kC = 32;               // feature channels of FeatureNet's top level

struct Tap4 {                        // bilinear taps, zero padding, align_corners (grid_sample)
    int off[4];                      // pixel offsets y * W + x (clamped when the weight is 0)
    float w[4];
};

// Source-view sampling position of reference pixel (xr, yr) on the plane at `depth`, following
// homo_warp's arithmetic: p = R [xr, yr, 1]^T + T / depth; (sx, sy) = p.xy / p.z; normalised to
// [-1, 1] (the in-frame mask is taken on the normalised value, strictly inside) and mapped
// back to pixels as grid_sample(align_corners=True) does.
__device__ __forceinline__ void project(const float *__restrict__ P, float xr, float yr, float depth,
                                        int H, int W, float &gx, float &gy) {
    const float X = fmaf(P[0], xr, fmaf(P[1], yr, P[2])) + P[3] / depth;
    const float Y = fmaf(P[4], xr, fmaf(P[5], yr, P[6])) + P[7] / depth;
    const float Z = fmaf(P[8], xr, fmaf(P[9], yr, P[10])) + P[11] / depth;
    gx = (X / Z) / ((float)(W - 1) / 2.0f) - 1.0f;
    gy = (Y / Z) / ((float)(H - 1) / 2.0f) - 1.0f;
}

__device__ __forceinline__ Tap4 taps(float gx, float gy, int H, int W) {
    Tap4 t;
    float px = (gx + 1.0f) * 0.5f * (float)(W - 1), py = (gy + 1.0f) * 0.5f * (float)(H - 1);
    // NaN / far-away positions (a plane behind the source camera): no contribution
    const bool finite = fabsf(px) < 1e8f && fabsf(py) < 1e8f;
    px = finite ? px : -4.0f, py = finite ? py : -4.0f;
    const float x0f = floorf(px), y0f = floorf(py);
    const float tx = px - x0f, ty = py - y0f;
    const int x0 = (int)x0f, y0 = (int)y0f;
#pragma unroll
    for (int c = 0; c < 4; c++) {
        const int dx = c & 1, dy = c >> 1, xi = x0 + dx, yi = y0 + dy;
        const bool ok = (unsigned)xi < (unsigned)W && (unsigned)yi < (unsigned)H;
        t.w[c] = ok ? (dx ? tx : 1.0f - tx) * (dy ? ty : 1.0f - ty) : 0.0f;
        t.off[c] = min(max(yi, 0), H - 1) * W + min(max(xi, 0), W - 1);
    }
    return t;
}

__global__ void nchw_to_nhwc_kernel(const float *__restrict__ in, int N, int C, long long npix,
                                    float *__restrict__ out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;      // over N * npix * C
    if (i >= (long long)N * npix * C) return;
    const int c = (int)(i % C);
    const long long p = (i / C) % npix, n = i / C / npix;
    out[i] = in[((size_t)n * C + c) * npix + p];
}

// feats_cl [V,H,W,32], imgs_cl [V,H,W,4] (rgb + pad), proj [V-1,3,4], depth [D]
// img_feat [3V + 32, D, Hp, Wp], in_masks [V, D, Hp, Wp]
__global__ __launch_bounds__(kThreads) void volume_cost_kernel(
    const float4 *__restrict__ feats, const float4 *__restrict__ imgs, const float *__restrict__ proj,
    const float *__restrict__ depth, int V, int D, int H, int W, int pad, float *__restrict__ img_feat,
    float *__restrict__ in_masks) {
    const int Hp = H + 2 * pad, Wp = W + 2 * pad;
    const long long nvox = (long long)D * Hp * Wp;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= nvox) return;
    const int x = (int)(idx % Wp), y = (int)((idx / Wp) % Hp), d = (int)(idx / ((long long)Wp * Hp));
    const int xr = x - pad, yr = y - pad;
    const bool inside = (unsigned)xr < (unsigned)W && (unsigned)yr < (unsigned)H;
    const float dep = depth[d];
    float sum[kC], sq[kC];
    {   // reference view: its own feature map, zero in the padding ring
        const float4 *f = feats + (size_t)(inside ? yr * W + xr : 0) * (kC / 4);
#pragma unroll
        for (int q = 0; q < kC / 4; q++) {
            const float4 v = inside ? f[q] : make_float4(0.f, 0.f, 0.f, 0.f);
            sum[4 * q] = v.x, sum[4 * q + 1] = v.y, sum[4 * q + 2] = v.z, sum[4 * q + 3] = v.w;
        }
#pragma unroll
        for (int c = 0; c < kC; c++) sq[c] = sum[c] * sum[c];
        // the reference leaves channels 0-2 of the padding ring uninitialised (torch.empty); 0 here
        const float4 c0 = inside ? imgs[(size_t)yr * W + xr] : make_float4(0.f, 0.f, 0.f, 0.f);
        img_feat[0 * nvox + idx] = c0.x, img_feat[1 * nvox + idx] = c0.y, img_feat[2 * nvox + idx] = c0.z;
        in_masks[idx] = 1.0f;
    }
    float count = 1.0f;
    for (int i = 1; i < V; i++) {
        float gx, gy;
        project(proj + 12 * (i - 1), (float)xr, (float)yr, dep, H, W, gx, gy);
        const float m = (gx > -1.0f && gx < 1.0f && gy > -1.0f && gy < 1.0f) ? 1.0f : 0.0f;
        in_masks[(size_t)i * nvox + idx] = m;
        count += m;
        const Tap4 t = taps(gx, gy, H, W);
        const float4 *f = feats + (size_t)i * H * W * (kC / 4);
#pragma unroll
        for (int q = 0; q < kC / 4; q++) {
            float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const float4 v = f[(size_t)t.off[c] * (kC / 4) + q];
                a.x = fmaf(t.w[c], v.x, a.x), a.y = fmaf(t.w[c], v.y, a.y);
                a.z = fmaf(t.w[c], v.z, a.z), a.w = fmaf(t.w[c], v.w, a.w);
            }
            sum[4 * q] += a.x, sum[4 * q + 1] += a.y, sum[4 * q + 2] += a.z, sum[4 * q + 3] += a.w;
            sq[4 * q] = fmaf(a.x, a.x, sq[4 * q]), sq[4 * q + 1] = fmaf(a.y, a.y, sq[4 * q + 1]);
            sq[4 * q + 2] = fmaf(a.z, a.z, sq[4 * q + 2]), sq[4 * q + 3] = fmaf(a.w, a.w, sq[4 * q + 3]);
        }
        const float4 *im = imgs + (size_t)i * H * W;
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int c = 0; c < 4; c++) {
            const float4 v = im[t.off[c]];
            a.x = fmaf(t.w[c], v.x, a.x), a.y = fmaf(t.w[c], v.y, a.y), a.z = fmaf(t.w[c], v.z, a.z);
        }
        img_feat[(size_t)(3 * i) * nvox + idx] = a.x;
        img_feat[(size_t)(3 * i + 1) * nvox + idx] = a.y;
        img_feat[(size_t)(3 * i + 2) * nvox + idx] = a.z;
    }
    const float inv = 1.0f / count;
#pragma unroll
    for (int c = 0; c < kC; c++) {
        const float mean = sum[c] * inv;
        img_feat[(size_t)(3 * V + c) * nvox + idx] = sq[c] * inv - mean * mean;
    }
}

// src [C,H,W]; grid_in (optional) [D,Hp,Wp,2] normalised positions to reuse; outputs
// warped [C,D,Hp,Wp] and (when computed here) grid_out [D,Hp,Wp,2]
__global__ __launch_bounds__(kThreads) void homo_warp_kernel(
    const float *__restrict__ src, const float *__restrict__ proj, const float *__restrict__ depth,
    const float *__restrict__ grid_in, int C, int D, int H, int W, int Hp, int Wp, int pad,
    float *__restrict__ warped, float *__restrict__ grid_out) {
    const long long nvox = (long long)D * Hp * Wp;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= nvox) return;
    float gx, gy;
    if (grid_in) {
        gx = grid_in[2 * idx], gy = grid_in[2 * idx + 1];
    } else {
        const int x = (int)(idx % Wp), y = (int)((idx / Wp) % Hp), d = (int)(idx / ((long long)Wp * Hp));
        project(proj, (float)(x - pad), (float)(y - pad), depth[d], H, W, gx, gy);
        grid_out[2 * idx] = gx, grid_out[2 * idx + 1] = gy;
    }
    const Tap4 t = taps(gx, gy, H, W);
    for (int c = 0; c < C; c++) {
        const float *s = src + (size_t)c * H * W;
        float a = 0.0f;
#pragma unroll
        for (int k = 0; k < 4; k++) a = fmaf(t.w[k], s[t.off[k]], a);
        warped[(size_t)c * nvox + idx] = a;
    }
}

inline bool aligned16(const void *p) { return ((uintptr_t)p & 15) == 0; }

}  // namespace

extern "C" int zest_nchw_to_nhwc(const float *in, int N, int C, int H, int W, float *out, void *stream) {
    ZEST_CHECK_ARG(in && out, "zest_nchw_to_nhwc: null pointer");
    ZEST_CHECK_ARG(N >= 1 && C >= 1 && H >= 1 && W >= 1, "zest_nchw_to_nhwc: bad shape");
    const long long n = (long long)N * C * H * W;
    hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3(zest_div_up(n, kThreads)), dim3(kThreads), 0,
                       (hipStream_t)stream, in, N, C, (long long)H * W, out);
    ZEST_RETURN_LAUNCH("zest_nchw_to_nhwc");
}

extern "C" int zest_volume_cost_fwd(const float *feats_cl, const float *imgs_cl, const float *proj,
                                    const float *depth, int V, int C, int D, int H, int W, int pad,
                                    float *img_feat, float *in_masks, void *stream) {
    ZEST_CHECK_ARG(feats_cl && imgs_cl && proj && depth && img_feat && in_masks && aligned16(feats_cl) &&
                       aligned16(imgs_cl), "zest_volume_cost_fwd: bad pointer");
    ZEST_CHECK_ARG(C == kC, "zest_volume_cost_fwd: %d feature channels (the FeatureNet top level has %d)", C, kC);
    ZEST_CHECK_ARG(V >= 2 && D >= 1 && H >= 2 && W >= 2 && pad >= 0, "zest_volume_cost_fwd: bad shape");
    const long long nvox = (long long)D * (H + 2 * pad) * (W + 2 * pad);
    hipLaunchKernelGGL(volume_cost_kernel, dim3(zest_div_up(nvox, kThreads)), dim3(kThreads), 0,
                       (hipStream_t)stream, (const float4 *)feats_cl, (const float4 *)imgs_cl, proj, depth,
                       V, D, H, W, pad, img_feat, in_masks);
    ZEST_RETURN_LAUNCH("zest_volume_cost_fwd");
}

extern "C" int zest_homo_warp_fwd(const float *src, const float *proj, const float *depth,
                                  const float *grid_in, int C, int D, int H, int W, int Hp, int Wp, int pad,
                                  float *warped, float *grid_out, void *stream) {
    ZEST_CHECK_ARG(src && warped, "zest_homo_warp_fwd: null pointer");
    ZEST_CHECK_ARG(grid_in || (proj && depth && grid_out),
                   "zest_homo_warp_fwd: either a grid or projection + depths + grid output are needed");
    ZEST_CHECK_ARG(C >= 1 && D >= 1 && H >= 2 && W >= 2 && Hp >= 1 && Wp >= 1 && pad >= 0,
                   "zest_homo_warp_fwd: bad shape");
    ZEST_CHECK_ARG(grid_in || (Hp == H + 2 * pad && Wp == W + 2 * pad),
                   "zest_homo_warp_fwd: padded grid %dx%d does not match %dx%d + 2*%d", Hp, Wp, H, W, pad);
    const long long nvox = (long long)D * Hp * Wp;
    hipLaunchKernelGGL(homo_warp_kernel, dim3(zest_div_up(nvox, kThreads)), dim3(kThreads), 0,
                       (hipStream_t)stream, src, proj, depth, grid_in, C, D, H, W, Hp, Wp, pad, warped, grid_out);
    ZEST_RETURN_LAUNCH("zest_homo_warp_fwd");
}
