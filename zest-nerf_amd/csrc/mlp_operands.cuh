// Operand assembly from rows of x [M, C_in] fp32 (the reference's prepare_pts layout): shared by the
// standalone MLP kernels (mlp_engine.hip) and the training kernels (mlp_train16.hip).
#pragma once
#include "mlp_engine.cuh"

namespace zest {

// Operand assembly.  The position -> input-column maps of the plan (mlp_plan.hip pe_map_acc,
// feat_map_acc) are affine in the lane group, so a lane needs one row pointer per operand and
// compile-time offsets - no table lookups, no per-element address arithmetic:
//   PE operand of C coordinates, L bands: element e of k-tile kt is m = 8 kt + e;
//     m < (L/2) C:  column C + 2C (2 (m / C) + (g >> 1)) + (g & 1) C + m % C
//                   = [C + 4C (m / C) + m % C] + [(g >> 1) 2C + (g & 1) C];   m = (L/2) C: column g (< C)
//   feature operand: quad q = 8 kt + 2 g + (e >> 2), channel c = e & 3: column feat_quad_col(q, V) + c (mlp_plan.h)
template <int EP, int C, int L, int NK>
__device__ __forceinline__ void load_pe_operand(const float *__restrict__ xrow, bool valid, int grp,
                                                OpArr<NK, ep_parts(EP)> &op) {
    const float *xg = xrow + (grp >> 1) * 2 * C + (grp & 1) * C;
    const float raw = (valid && grp < C) ? xrow[grp < C ? grp : 0] : 0.0f;
#pragma unroll
    for (int t = 0; t < NK; t++) {
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; e++) {
            const int m = 8 * t + e;
            if (m < (L / 2) * C)
                v[e] = valid ? xg[C + 4 * C * (m / C) + m % C] : 0.0f;
            else
                v[e] = m == (L / 2) * C ? raw : 0.0f;
        }
        store_tile<EP>(v, op, t);
    }
}

template <int EP, int NK>
__device__ __forceinline__ void load_feat_operand(const float *__restrict__ xf, int F, bool valid, int grp,
                                                  OpArr<NK, ep_parts(EP)> &op) {
    const int V = (F - 8) / 4;
#pragma unroll
    for (int t = 0; t < NK; t++) {
        // first columns of this lane's two quads
        const int ca = feat_quad_col(8 * t + 2 * grp, V), cb = feat_quad_col(8 * t + 2 * grp + 1, V);
        const bool va = valid && ca >= 0, vb = valid && cb >= 0;
        const float *pa = xf + (va ? ca : 0), *pb = xf + (vb ? cb : 0);
        float v[8];
#pragma unroll
        for (int c = 0; c < 4; c++) v[c] = va ? pa[c] : 0.0f, v[4 + c] = vb ? pb[c] : 0.0f;
        store_tile<EP>(v, op, t);
    }
}

// One k-tile of the same operands with the k-tile index at run time, as the eight fp32 values of the lane (the
// weight kernel of the training path gives each k-tile to a wave of its own and requests the values one block of
// samples ahead: they are rounded and packed only when they are stored): same columns as above, the divisions by
// C are a few VALU instructions against the latency of the row loads.
template <int C, int L>
__device__ __forceinline__ void load_pe_tile_raw(const float *__restrict__ xrow, bool valid, int grp, int t, float (&v)[8]) {
    const float *xg = xrow + (grp >> 1) * 2 * C + (grp & 1) * C;
#pragma unroll
    for (int e = 0; e < 8; e++) {
        const int m = 8 * t + e;
        const bool in_bands = m < (L / 2) * C;
        const int col = in_bands ? C + 4 * C * (m / C) + m % C : 0;
        const float band = (valid && in_bands) ? xg[col] : 0.0f;
        const float raw = (valid && m == (L / 2) * C && grp < C) ? xrow[grp < C ? grp : 0] : 0.0f;
        v[e] = in_bands ? band : raw;
    }
}

__device__ __forceinline__ void load_feat_tile_raw(const float *__restrict__ xf, int F, bool valid, int grp, int t, float (&v)[8]) {
    const int V = (F - 8) / 4;
    const int ca = feat_quad_col(8 * t + 2 * grp, V), cb = feat_quad_col(8 * t + 2 * grp + 1, V);
    const bool va = valid && ca >= 0, vb = valid && cb >= 0;
    const float *pa = xf + (va ? ca : 0), *pb = xf + (vb ? cb : 0);
#pragma unroll
    for (int c = 0; c < 4; c++) v[c] = va ? pa[c] : 0.0f, v[4 + c] = vb ? pb[c] : 0.0f;
}

}  // namespace zest
