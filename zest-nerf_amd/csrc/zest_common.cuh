// Shared device/host helpers for the gfx950 ZeST rendering kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/zest_render.h"

#define ZEST_WAVE 64

// ---- host-side error plumbing -------------------------------------------------------
void zest_set_error(const char *fmt, ...);

#define ZEST_CHECK_ARG(cond, ...)                                   \
    do {                                                            \
        if (!(cond)) {                                              \
            zest_set_error(__VA_ARGS__);                            \
            return (int)hipErrorInvalidValue;                       \
        }                                                           \
    } while (0)

#define ZEST_RETURN_LAUNCH(name)                                            \
    do {                                                                    \
        hipError_t e_ = hipGetLastError();                                  \
        if (e_ != hipSuccess) {                                             \
            zest_set_error("%s: launch failed: %s", name, hipGetErrorString(e_)); \
            return (int)e_;                                                 \
        }                                                                   \
        return 0;                                                           \
    } while (0)

static inline int zest_div_up(long long a, long long b) { return (int)((a + b - 1) / b); }

// ---- wave-level primitives (64 lanes) ---------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}

// sum over the 32 lanes of this lane's half-wave
__device__ __forceinline__ float half_sum(float v) {
#pragma unroll
    for (int m = 16; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}

// inclusive product scan over `width` consecutive lanes (width = 32 or 64)
template <int WIDTH>
__device__ __forceinline__ float seg_scan_mul(float v, int lane_in_seg) {
#pragma unroll
    for (int d = 1; d < WIDTH; d <<= 1) {
        float o = __shfl_up(v, d, WIDTH);
        if (lane_in_seg >= d) v *= o;
    }
    return v;
}

// ---- DPP forms for a 32-lane half (plain VALU: no LDS round trip per step) --------------------
// dpp_ctrl: quad_perm[1,0,3,2]=0xB1, quad_perm[2,3,0,1]=0x4E, row_half_mirror=0x141,
// row_mirror=0x140, row_shr:n=0x110+n, row_bcast:15=0x142, wave_shr:1=0x138
template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ float dpp_f(float old, float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(v), CTRL,
                                                       ROW_MASK, 0xF, false));
}

// value of lane `l` as a wave-uniform float (the builtin is int-typed: bit-cast, do not convert)
template <int L>
__device__ __forceinline__ float readlane_f(float v) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), L));
}

// sum over lanes 0..31, returned wave-uniform (meaningful for the lower half)
__device__ __forceinline__ float lower_half_sum(float v) {
    v += dpp_f<0xB1>(0.f, v);
    v += dpp_f<0x4E>(0.f, v);
    v += dpp_f<0x141>(0.f, v);
    v += dpp_f<0x140>(0.f, v);                 // every lane of a 16-lane row holds the row sum
    return readlane_f<0>(v) + readlane_f<16>(v);
}

// product scan over lanes 0..31: returns the exclusive prefix product for this lane (col =
// lane & 31; correct for the lower half) and the product of all 32 factors in *total
__device__ __forceinline__ float lower_half_excl_prod(float f, int col, float *total) {
    float v = f;
    v *= dpp_f<0x111>(1.f, v);
    v *= dpp_f<0x112>(1.f, v);
    v *= dpp_f<0x114>(1.f, v);
    v *= dpp_f<0x118>(1.f, v);                 // inclusive within each 16-lane row
    v *= dpp_f<0x142, 0xA>(1.f, v);            // rows 1 and 3 take the last lane of the row before
    *total = readlane_f<31>(v);
    float ex = dpp_f<0x138>(1.f, v);           // shift the wave right by one lane
    return col == 0 ? 1.0f : ex;
}

// ---- scalar math ---------------------------------------------------------------------
__device__ __forceinline__ float zest_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }

// sin and cos of a moderate fp32 argument, <1 ulp: three-term Cody-Waite reduction by
// pi/2 with FMAs, near-minimax polynomials on [-pi/4, pi/4].  Arguments on this path are
// 2^k * coordinate with k <= 9; beyond |x| > 3e4 the reduction loses bits, so hand over to
// the library routine there (wave-uniformly rare).
__device__ __forceinline__ void zest_sincos(float x, float *s_out, float *c_out) {
    if (__builtin_expect(!(fabsf(x) < 30000.0f), 0)) {
        sincosf(x, s_out, c_out);
        return;
    }
    const float n = rintf(x * 0.63661974668502807617f);
    float r = fmaf(n, -1.57079637050628662109375f, x);     // pi/2 = hi + mid + lo
    r = fmaf(n, 4.371138828673792886547744e-08f, r);
    r = fmaf(n, 1.715124510005881872803934e-15f, r);
    const float r2 = r * r;
    // sin(r) = r + r^3 S(r^2), cos(r) = 1 - r^2/2 + r^4 C(r^2); Chebyshev fits on
    // [0,(pi/4)^2], max abs error of the whole routine 9.3e-8 for |x| <= 3e4 (checked
    // against fp64 on 2M points per decade; tests/test_hip_ops.py pins the encoder on the GPU).
    float sp = fmaf(r2, 2.7243888780503766611e-06f, -0.00019840040476992726326f);
    sp = fmaf(sp, r2, 0.0083333319053053855896f);
    sp = fmaf(sp, r2, -0.16666667163372039795f);
    const float sn = fmaf(r * r2, sp, r);
    float cp = fmaf(r2, 2.4613620553282089531e-05f, -0.0013888835674151778221f);
    cp = fmaf(cp, r2, 0.041666671633720397949f);
    const float cs = fmaf(r2 * r2, cp, fmaf(r2, -0.5f, 1.0f));
    const int q = (int)n;
    const float s1 = (q & 1) ? cs : sn;
    const float c1 = (q & 1) ? sn : cs;
    *s_out = (q & 2) ? -s1 : s1;
    *c_out = ((q + 1) & 2) ? -c1 : c1;
}
