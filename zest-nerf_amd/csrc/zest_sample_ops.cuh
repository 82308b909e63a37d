// Per-sample operators shared by the standalone kernels (encode.hip) and the fused renderer:
// trilinear lookup into the channels-last encoding volume, per-view projection + bilinear
// colour gather, view-direction feature.  Arithmetic follows the reference's fp32 op order
// where a different order could flip a discontinuous result (the strict in-frame mask).
#pragma once
#include "zest_common.cuh"

struct ZestCam {          // one source view: rows of w2c[:3,:4] and of K
    float r[3][4];
    float k[3][3];
};

// grid_sample's align_corners=True un-normalisation of a [0,1] coordinate that the caller
// mapped to [-1,1] first, in the reference's rounding order (utils.py:451, 487).
__device__ __forceinline__ float zest_unnorm(float c01, int size) {
    const float g = c01 * 2.0f - 1.0f;
    return (g + 1.0f) * 0.5f * (float)(size - 1);
}

// Voxel (z, y, x) of the kernels' copy of the encoding volume: channels-last and DEPTH-INNERMOST, [H][W][D][8].
// Consecutive samples of a ray advance about one depth plane each at slowly varying (x, y) (the volume is the
// reference view's frustum), so the taps that the 16 lanes of a lane group issue for one corner are one
// contiguous run of 32-byte voxels: 4-5 lines of 128 B instead of 16 lines 675 KB apart in a [D][H][W][8] copy.
__device__ __forceinline__ size_t zest_vox(int zi, int yi, int xi, int D, int W) {
    return ((size_t)yi * W + xi) * D + zi;
}
#ifdef ZEST_VOX_PLANES
#error "the plane-major [D][H][W][8] copy of rounds 1-2 is gone; profiles/r03_ab_volume_layout.txt holds the A/B"
#endif

// Trilinear, zero padding.  vol: [H,W,D,8] (zest_vox) as float4 pairs.  CH4 selects which
// half of the 8 channels (0: ch0-3, 1: ch4-7, 2: all eight -> out[0..7]).
template <int CH4>
__device__ __forceinline__ void zest_volume_trilerp(const float4 *__restrict__ vol, int D, int H,
                                                    int W, float nx, float ny, float nz,
                                                    float *out) {
    constexpr int NO = (CH4 == 2) ? 8 : 4;
#pragma unroll
    for (int i = 0; i < NO; i++) out[i] = 0.f;
    float fx = zest_unnorm(nx, W), fy = zest_unnorm(ny, H), fz = zest_unnorm(nz, D);
    // everything further than one voxel outside contributes nothing; clamp so the int
    // conversion is defined for any input (NaN falls to the low side)
    fx = fminf(fmaxf(fx, -2.0f), (float)W + 1.0f);
    fy = fminf(fmaxf(fy, -2.0f), (float)H + 1.0f);
    fz = fminf(fmaxf(fz, -2.0f), (float)D + 1.0f);
    const float x0f = floorf(fx), y0f = floorf(fy), z0f = floorf(fz);
    const float tx = fx - x0f, ty = fy - y0f, tz = fz - z0f;
    const int x0 = (int)x0f, y0 = (int)y0f, z0 = (int)z0f;
#pragma unroll
    for (int dz = 0; dz < 2; dz++) {
        const int zi = z0 + dz;
        const float wz = dz ? tz : 1.0f - tz;
#pragma unroll
        for (int dy = 0; dy < 2; dy++) {
            const int yi = y0 + dy;
            const float wy = dy ? ty : 1.0f - ty;
#pragma unroll
            for (int dx = 0; dx < 2; dx++) {
                const int xi = x0 + dx;
                const float w = (dx ? tx : 1.0f - tx) * wy * wz;
                const bool ok = (unsigned)xi < (unsigned)W && (unsigned)yi < (unsigned)H &&
                                (unsigned)zi < (unsigned)D;
                if (ok) {
                    const size_t vox = zest_vox(zi, yi, xi, D, W);
                    if (CH4 != 1) {
                        const float4 a = vol[2 * vox];
                        out[0] = fmaf(w, a.x, out[0]), out[1] = fmaf(w, a.y, out[1]);
                        out[2] = fmaf(w, a.z, out[2]), out[3] = fmaf(w, a.w, out[3]);
                    }
                    if (CH4 != 0) {
                        const float4 b = vol[2 * vox + 1];
                        float *o = out + (CH4 == 2 ? 4 : 0);
                        o[0] = fmaf(w, b.x, o[0]), o[1] = fmaf(w, b.y, o[1]);
                        o[2] = fmaf(w, b.z, o[2]), o[3] = fmaf(w, b.w, o[3]);
                    }
                }
            }
        }
    }
}

// One source view: world point -> pixel -> bilinear rgb (border clamp) + strict in-frame mask.
// img: [H,W,4] of this view.  out = (r, g, b, mask).
template <bool SAME_PIXEL = false>   // SAME_PIXEL: timing experiments only (all four taps read pixel 0)
__device__ __forceinline__ float4 zest_color_tap(const float4 *__restrict__ img, int H, int W,
                                                 const ZestCam &c, float px, float py, float pz) {
    // p_cam = R p + T, q = K p_cam (reference utils.py:262-268), fp32, left-to-right sums
    const float cx = fmaf(pz, c.r[0][2], fmaf(py, c.r[0][1], px * c.r[0][0])) + c.r[0][3];
    const float cy = fmaf(pz, c.r[1][2], fmaf(py, c.r[1][1], px * c.r[1][0])) + c.r[1][3];
    const float cz = fmaf(pz, c.r[2][2], fmaf(py, c.r[2][1], px * c.r[2][0])) + c.r[2][3];
    const float qx = fmaf(cz, c.k[0][2], fmaf(cy, c.k[0][1], cx * c.k[0][0]));
    const float qy = fmaf(cz, c.k[1][2], fmaf(cy, c.k[1][1], cx * c.k[1][0]));
    const float qz = fmaf(cz, c.k[2][2], fmaf(cy, c.k[2][1], cx * c.k[2][0]));
    const float u = (qx / qz) / (float)(W - 1), v = (qy / qz) / (float)(H - 1);
    const float gx = u * 2.0f - 1.0f, gy = v * 2.0f - 1.0f;
    const float mask = (gx > -1.0f && gx < 1.0f && gy > -1.0f && gy < 1.0f) ? 1.0f : 0.0f;
    float fx = (gx + 1.0f) * 0.5f * (float)(W - 1), fy = (gy + 1.0f) * 0.5f * (float)(H - 1);
    fx = fminf(fmaxf(fx, 0.0f), (float)(W - 1));          // padding_mode='border'
    fy = fminf(fmaxf(fy, 0.0f), (float)(H - 1));
    const float x0f = floorf(fx), y0f = floorf(fy);
    const float tx = fx - x0f, ty = fy - y0f;
    const int x0 = (int)x0f, y0 = (int)y0f;
    const int x1 = min(x0 + 1, W - 1), y1 = min(y0 + 1, H - 1);
    const size_t keep = SAME_PIXEL ? 0 : ~(size_t)0;
    const float4 a = img[((size_t)y0 * W + x0) & keep], b = img[((size_t)y0 * W + x1) & keep];
    const float4 d = img[((size_t)y1 * W + x0) & keep], e = img[((size_t)y1 * W + x1) & keep];
    const float w00 = (1.0f - tx) * (1.0f - ty), w10 = tx * (1.0f - ty);
    const float w01 = (1.0f - tx) * ty, w11 = tx * ty;
    float4 o;
    o.x = fmaf(w11, e.x, fmaf(w01, d.x, fmaf(w10, b.x, w00 * a.x)));
    o.y = fmaf(w11, e.y, fmaf(w01, d.y, fmaf(w10, b.y, w00 * a.y)));
    o.z = fmaf(w11, e.z, fmaf(w01, d.z, fmaf(w10, b.z, w00 * a.z)));
    o.w = mask;
    return o;
}

__device__ __forceinline__ ZestCam zest_load_cam(const float *__restrict__ w2cs,
                                                 const float *__restrict__ intr, int v) {
    ZestCam c;
#pragma unroll
    for (int i = 0; i < 3; i++) {
#pragma unroll
        for (int j = 0; j < 4; j++) c.r[i][j] = w2cs[16 * v + 4 * i + j];
#pragma unroll
        for (int j = 0; j < 3; j++) c.k[i][j] = intr[9 * v + 3 * i + j];
    }
    return c;
}

// unit direction rotated into the first view of the set (reference renderer.py:34-49, 256-260)
__device__ __forceinline__ void zest_view_dir(const float *__restrict__ dir3,
                                              const float *__restrict__ w2c0, float out[3]) {
    const float dx = dir3[0], dy = dir3[1], dz = dir3[2];
    const float nrm = sqrtf(dx * dx + dy * dy + dz * dz);
    const float ux = dx / nrm, uy = dy / nrm, uz = dz / nrm;
    if (w2c0) {
#pragma unroll
        for (int i = 0; i < 3; i++)
            out[i] = fmaf(uz, w2c0[4 * i + 2], fmaf(uy, w2c0[4 * i + 1], ux * w2c0[4 * i]));
    } else {
        out[0] = ux, out[1] = uy, out[2] = uz;
    }
}
