// bf16 MFMA engine for the width-256 NeRF MLP: activations stay in registers.
//
// Orientation: Y^T = W X^T with v_mfma_f32_32x32x16_bf16.  A wave owns NB column blocks of
// 32 samples (sample = lane & 31; both lane halves own the same sample).  The A operand is a
// pre-packed 1 KiB weight tile (mlp_plan.h), the B operand 8 bf16 per lane = 8 "slots" of the
// wave's activation operand.  A finished 32x32 accumulator tile (16 fp32 per lane, output
// features on the register axis) is biased via its initial value, modulated, rectified,
// rounded to bf16 pairs and IS two B-operand tiles of the next op: no LDS round trip and no
// cross-lane traffic between layers (cdna_hip_programming.md section 3, "an accumulator tile
// as the next MFMA's operand"; the k permutation that implies is folded into the packed
// weights).  Row-blocks are the outer loop, so only one accumulator tile (plus one modulation
// tile) per column block is live next to the 64-register input and output operands.
//
// Per 32-sample column block, MOD on, xyz input: 1 376 MFMAs of 32 cycles; the arithmetic
// intensity against the weight stream is set by how many samples share a tile read
// (NB x waves per workgroup), see DESIGN.md.
#pragma once
#include <hip/hip_bf16.h>
#include "mlp_plan.h"
#include "zest_common.cuh"

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) short bf16x8;

namespace zest {

// Source of weight tiles: v1 reads them straight from global memory (L1/L2 resident).
struct GlobalTiles {
    const uint4 *base;     // wave-uniform: first tile of the stream
    int lane;
    __device__ __forceinline__ bf16x8 load(int tile) const {
        const uint4 v = base[(size_t)tile * 64 + lane];
        return *reinterpret_cast<const bf16x8 *>(&v);
    }
};

__device__ __forceinline__ unsigned pack_bf16(float lo, float hi) {
    __hip_bfloat162 v = __float22bfloat162_rn(make_float2(lo, hi));
    return *reinterpret_cast<unsigned *>(&v);
}

__device__ __forceinline__ f32x16 load_bias_block(const float *__restrict__ bias, int block, int half) {
    const float4 *b = reinterpret_cast<const float4 *>(bias + (size_t)block * 32 + half * 16);
    const float4 b0 = b[0], b1 = b[1], b2 = b[2], b3 = b[3];
    f32x16 r;
    r[0] = b0.x, r[1] = b0.y, r[2] = b0.z, r[3] = b0.w, r[4] = b1.x, r[5] = b1.y, r[6] = b1.z, r[7] = b1.w;
    r[8] = b2.x, r[9] = b2.y, r[10] = b2.z, r[11] = b2.w, r[12] = b3.x, r[13] = b3.y, r[14] = b3.z, r[15] = b3.w;
    return r;
}

// 16 activated fp32 values of a tile -> the two bf16 operand tiles they form
__device__ __forceinline__ void acc_to_operand(const f32x16 &v, bf16x8 &t0, bf16x8 &t1) {
    unsigned w[8];
#pragma unroll
    for (int j = 0; j < 8; j++) w[j] = pack_bf16(v[2 * j], v[2 * j + 1]);
    uint4 a = make_uint4(w[0], w[1], w[2], w[3]), b = make_uint4(w[4], w[5], w[6], w[7]);
    t0 = *reinterpret_cast<bf16x8 *>(&a);
    t1 = *reinterpret_cast<bf16x8 *>(&b);
}

template <int N>
struct OpArr {              // N operand tiles; N = 0 allowed
    bf16x8 t[N > 0 ? N : 1];
};

// One Linear: NJB row blocks over the operand [A (NTA tiles) | B (NTB tiles)].
//   MOD:  tile is modulated by the feature operand (NTF tiles) with the op's modulation tiles
//   MODE: 0 = produce operand tiles into `out` (2 per row block), 1 = keep the accumulator of
//         row block 0 in `keep` (head / rgb tiles)
template <int NB, int NJB, int NTA, int NTB, bool MOD, int NTF, bool RELU, int MODE, class Tiles>
__device__ __forceinline__ void engine_layer(const Tiles &tiles, int &tile, const float *__restrict__ bias,
                                             int bias_block, int half, bool v2,
                                             const OpArr<NTA> (&opa)[NB], const OpArr<NTB> (&opb)[NB],
                                             const OpArr<NTF> (&opf)[NB], OpArr<16> (&out)[NB],
                                             f32x16 (&keep)[NB]) {
#pragma unroll
    for (int jb = 0; jb < NJB; jb++) {
        f32x16 acc[NB], macc[NB];
        const f32x16 b0 = load_bias_block(bias, bias_block + jb, half);
#pragma unroll
        for (int nb = 0; nb < NB; nb++) acc[nb] = b0;
        if (MOD) {
            const f32x16 m0 = load_bias_block(bias, jb, half);
#pragma unroll
            for (int nb = 0; nb < NB; nb++) macc[nb] = m0;
#pragma unroll
            for (int k = 0; k < NTF; k++) {
                const bf16x8 a = tiles.load(tile++);
#pragma unroll
                for (int nb = 0; nb < NB; nb++)
                    macc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, opf[nb].t[k], macc[nb], 0, 0, 0);
            }
        }
#pragma unroll
        for (int k = 0; k < NTA; k++) {
            const bf16x8 a = tiles.load(tile++);
#pragma unroll
            for (int nb = 0; nb < NB; nb++)
                acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, opa[nb].t[k], acc[nb], 0, 0, 0);
        }
#pragma unroll
        for (int k = 0; k < NTB; k++) {
            const bf16x8 a = tiles.load(tile++);
#pragma unroll
            for (int nb = 0; nb < NB; nb++)
                acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, opb[nb].t[k], acc[nb], 0, 0, 0);
        }
#pragma unroll
        for (int nb = 0; nb < NB; nb++) {
            f32x16 v = acc[nb];
            if (MOD) {
#pragma unroll
                for (int i = 0; i < 16; i++) v[i] = v2 ? v[i] + macc[nb][i] : v[i] * macc[nb][i];
            }
            if (RELU) {
#pragma unroll
                for (int i = 0; i < 16; i++) v[i] = fmaxf(v[i], 0.0f);
            }
            if (MODE == 0)
                acc_to_operand(v, out[nb].t[2 * jb], out[nb].t[2 * jb + 1]);
            else if (jb == 0)
                keep[nb] = v;
        }
    }
}

// The whole network for NB column blocks.  pts/feat/views: encoder operands in plan slot
// order.  Results: head tile (row 0 alpha, rows 1.. extra heads) and rgb tile (rows 0-2),
// raw (no output activation).
template <int NB, int NT_PTS, bool MOD, int NT_FEAT, class Tiles>
__device__ __forceinline__ void engine_forward(const Tiles &tiles, const float *__restrict__ bias,
                                               int half, bool v2, const OpArr<NT_PTS> (&pts)[NB],
                                               const OpArr<NT_FEAT> (&feat)[NB],
                                               const OpArr<2> (&views)[NB], f32x16 (&head)[NB],
                                               f32x16 (&rgb)[NB]) {
    OpArr<16> hA[NB], hB[NB];
    OpArr<0> none[NB];
    f32x16 unused[NB];
    int tile = 0;
    int bb = MOD ? 8 : 0;
    engine_layer<NB, 8, NT_PTS, 0, MOD, NT_FEAT, true, 0>(tiles, tile, bias, bb, half, v2, pts, none, feat, hA, unused);
    bb += 8;
    engine_layer<NB, 8, 16, 0, MOD, NT_FEAT, true, 0>(tiles, tile, bias, bb, half, v2, hA, none, feat, hB, unused);
    bb += 8;
    engine_layer<NB, 8, 16, 0, MOD, NT_FEAT, true, 0>(tiles, tile, bias, bb, half, v2, hB, none, feat, hA, unused);
    bb += 8;
    engine_layer<NB, 8, 16, 0, MOD, NT_FEAT, true, 0>(tiles, tile, bias, bb, half, v2, hA, none, feat, hB, unused);
    bb += 8;
    engine_layer<NB, 8, 16, 0, MOD, NT_FEAT, true, 0>(tiles, tile, bias, bb, half, v2, hB, none, feat, hA, unused);
    bb += 8;
    engine_layer<NB, 8, NT_PTS, 16, MOD, NT_FEAT, true, 0>(tiles, tile, bias, bb, half, v2, pts, hA, feat, hB, unused);
    bb += 8;
    engine_layer<NB, 8, 16, 0, MOD, NT_FEAT, true, 0>(tiles, tile, bias, bb, half, v2, hB, none, feat, hA, unused);
    bb += 8;
    engine_layer<NB, 8, 16, 0, MOD, NT_FEAT, true, 0>(tiles, tile, bias, bb, half, v2, hA, none, feat, hB, unused);
    bb += 8;
    // trunk output in hB
    engine_layer<NB, 1, 16, 0, false, NT_FEAT, false, 1>(tiles, tile, bias, bb, half, v2, hB, none, feat, hA, head);
    bb += 1;
    engine_layer<NB, 8, 16, 0, false, NT_FEAT, false, 0>(tiles, tile, bias, bb, half, v2, hB, none, feat, hA, unused);
    bb += 8;
    engine_layer<NB, 4, 16, 2, false, NT_FEAT, true, 0>(tiles, tile, bias, bb, half, v2, hA, views, feat, hB, unused);
    bb += 4;
    // rgb: 128 hidden features = first 8 tiles of hB
    OpArr<8> h128[NB];
#pragma unroll
    for (int nb = 0; nb < NB; nb++)
#pragma unroll
        for (int k = 0; k < 8; k++) h128[nb].t[k] = hB[nb].t[k];
    engine_layer<NB, 1, 8, 0, false, NT_FEAT, false, 1>(tiles, tile, bias, bb, half, v2, h128, none, feat, hA, rgb);
}

// standalone launcher (mlp.hip -> zest_mlp_fwd)
int mlp_bf16_launch(const MlpPlan &p, const float *bias, const void *tiles, const float *x, int M,
                    float *out, hipStream_t stream);

}  // namespace zest
