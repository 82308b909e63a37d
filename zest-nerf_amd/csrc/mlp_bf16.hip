// Standalone bf16 MLP forward (zest_mlp_fwd, ZEST_PREC_BF16): x [M,C_in] fp32 in HBM ->
// operand registers -> engine -> out [M,C_out].  Backs MVSNeRF.forward in bf16 mode and the
// MFMA-utilisation measurement of the MLP alone.
#include "mlp_engine.cuh"

namespace zest {

struct SlotMaps {            // feature index per slot and lane half, -1 = zero pad
    short pts[48][2];
    short feat[32][2];
    short views[16][2];
};

template <int NT>
__device__ __forceinline__ void load_operand(const float *__restrict__ xrow, bool valid, int half,
                                             const short (*map)[2], OpArr<NT> &op) {
#pragma unroll
    for (int t = 0; t < NT; t++) {
        unsigned w[4];
#pragma unroll
        for (int jj = 0; jj < 4; jj++) {
            float v[2];
#pragma unroll
            for (int u = 0; u < 2; u++) {
                const int s = 8 * t + 2 * jj + u;
                const int idx = half ? map[s][1] : map[s][0];
                v[u] = (valid && idx >= 0) ? xrow[idx < 0 ? 0 : idx] : 0.0f;
            }
            w[jj] = pack_bf16(v[0], v[1]);
        }
        uint4 a = make_uint4(w[0], w[1], w[2], w[3]);
        op.t[t] = *reinterpret_cast<bf16x8 *>(&a);
    }
}

template <int NB, int NT_PTS, bool MOD, int NT_FEAT>
__global__ __launch_bounds__(256, NB == 1 ? 2 : 1) void mlp_bf16_kernel(
    SlotMaps maps, const uint4 *__restrict__ tiles, const float *__restrict__ x, int M, int P, int F, int C_in, int C_out, int head, int v2,
    float *__restrict__ out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = lane & 31, half = lane >> 5;
    const long long m_base = ((long long)blockIdx.x * 4 + wave) * (32 * NB);
    if (m_base >= M) return;                       // wave-uniform
    OpArr<NT_PTS> pts[NB];
    OpArr<NT_FEAT> feat[NB];
#pragma unroll
    for (int nb = 0; nb < NB; nb++) {
        const long long m = m_base + 32 * nb + col;
        const bool valid = m < M;
        const float *xrow = x + (size_t)(valid ? m : 0) * C_in;
        load_operand<NT_PTS>(xrow, valid, half, maps.pts, pts[nb]);
        if (MOD) load_operand<NT_FEAT>(xrow + P, valid, half, maps.feat, feat[nb]);
    }
    auto views_fn = [&](OpArr<2> (&views)[NB]) {
#pragma unroll
        for (int nb = 0; nb < NB; nb++) {
            const long long m = m_base + 32 * nb + col;
            const bool valid = m < M;
            load_operand<2>(x + (size_t)(valid ? m : 0) * C_in + P + F, valid, half, maps.views, views[nb]);
        }
    };
    f32x16 headt[NB], rgbt[NB];
    GlobalTiles gt{(gptr_u4)tiles, lane, half};
    int unit = 0;
    engine_forward<NB, NT_PTS, MOD, NT_FEAT>(gt, unit, v2 != 0, pts, feat, views_fn, headt, rgbt);
#pragma unroll
    for (int nb = 0; nb < NB; nb++) {
        const long long m = m_base + 32 * nb + col;
        if (m >= M) continue;
        float *o = out + (size_t)m * C_out;
        // tile row r sits in lane half (r>>2)&1, register (r&3) + 4*(r>>3)
        if (half == 0) {
            o[0] = v2 ? zest_sigmoid(rgbt[nb][0]) : rgbt[nb][0];
            o[1] = v2 ? zest_sigmoid(rgbt[nb][1]) : rgbt[nb][1];
            o[2] = v2 ? zest_sigmoid(rgbt[nb][2]) : rgbt[nb][2];
            o[3] = v2 ? fmaxf(headt[nb][0], 0.0f) : headt[nb][0];
            if (head == ZEST_HEAD_BLEND) o[4] = zest_sigmoid(headt[nb][1]);
            if (head == ZEST_HEAD_DYNAMIC) {
                o[4] = tanhf(headt[nb][1]), o[5] = tanhf(headt[nb][2]), o[6] = tanhf(headt[nb][3]);
                o[11] = zest_sigmoid(headt[nb][4]);            // row 8
            }
        } else if (head == ZEST_HEAD_DYNAMIC) {
            o[7] = tanhf(headt[nb][0]), o[8] = tanhf(headt[nb][1]), o[9] = tanhf(headt[nb][2]);  // rows 4-6
            o[10] = zest_sigmoid(headt[nb][3]);                // row 7
        }
    }
}

template <int NB, int NT_PTS, bool MOD, int NT_FEAT>
static int launch_one(const MlpPlan &p, const SlotMaps &maps, const void *tiles, const float *x, int M,
                      float *out, hipStream_t stream) {
    if (p.n_tiles != stream_units(NT_PTS, MOD ? NT_FEAT : 0)) {
        zest_set_error("zest_mlp_fwd(bf16): plan has %d stream units, kernel expects %d", p.n_tiles,
                       stream_units(NT_PTS, MOD ? NT_FEAT : 0));
        return (int)hipErrorInvalidValue;
    }
    const zest_mlp_desc &d = p.desc;
    const int F = d.use_feat ? d.in_ch_feat : 0;
    const int C_in = d.in_ch_pts + F + d.in_ch_views;
    const int C_out = d.head == ZEST_HEAD_NONE ? 4 : (d.head == ZEST_HEAD_BLEND ? 5 : 12);
    const int blocks = zest_div_up(M, 4 * 32 * NB);
    hipLaunchKernelGGL((mlp_bf16_kernel<NB, NT_PTS, MOD, NT_FEAT>), dim3(blocks), dim3(256), 0, stream,
                       maps, (const uint4 *)tiles, x, M, d.in_ch_pts, F, C_in, C_out, d.head,
                       d.net_type == 2 ? 1 : 0, out);
    ZEST_RETURN_LAUNCH("zest_mlp_fwd(bf16)");
}

int mlp_bf16_launch(const MlpPlan &p, const void *tiles, const float *x, int M, float *out,
                    hipStream_t stream) {
    SlotMaps maps;
    for (auto &r : maps.pts) r[0] = r[1] = -1;
    for (auto &r : maps.feat) r[0] = r[1] = -1;
    for (auto &r : maps.views) r[0] = r[1] = -1;
    for (int s = 0; s < p.ns_pts; s++) maps.pts[s][0] = p.map_pts[2 * s], maps.pts[s][1] = p.map_pts[2 * s + 1];
    for (int s = 0; s < p.ns_feat; s++) maps.feat[s][0] = p.map_feat[2 * s], maps.feat[s][1] = p.map_feat[2 * s + 1];
    for (int s = 0; s < p.ns_views; s++) maps.views[s][0] = p.map_views[2 * s], maps.views[s][1] = p.map_views[2 * s + 1];
    const bool mod = p.desc.use_feat != 0;
    const int key = p.nt_pts * 10 + (mod ? p.nt_feat : 0);
    switch (key) {
        case 40: return launch_one<1, 4, false, 0>(p, maps, tiles, x, M, out, stream);
        case 42: return launch_one<1, 4, true, 2>(p, maps, tiles, x, M, out, stream);
        case 43: return launch_one<1, 4, true, 3>(p, maps, tiles, x, M, out, stream);
        case 60: return launch_one<1, 6, false, 0>(p, maps, tiles, x, M, out, stream);
        case 62: return launch_one<1, 6, true, 2>(p, maps, tiles, x, M, out, stream);
        case 63: return launch_one<1, 6, true, 3>(p, maps, tiles, x, M, out, stream);
    }
    zest_set_error("zest_mlp_fwd(bf16): no kernel for %d point tiles / %d feature tiles "
                   "(supported: 3..8 source views)", p.nt_pts, mod ? p.nt_feat : 0);
    return (int)hipErrorInvalidValue;
}

}  // namespace zest
