// Fused inference renderer: per-sample work never leaves the chip.
//
// Rays are cut into blocks of 32 samples.  For its blocks a wave's lanes build the MLP
// operands in registers (sin/cos on the two lane halves of one sincos, volume channels and
// source views split between the halves), the bf16 engine (mlp_engine.cuh) runs the whole
// network on them with the weights streamed through an LDS ring shared by the workgroup, and
// the raw colour/density land back on the lanes, where a 32-lane shuffle scan composites the
// block.  Per block 80 B leave the chip; a second tiny kernel chains the blocks of each ray
// into the per-ray maps (64 B per ray).
//
// Replaces rendering(..., val=True) of the reference (renderer.py:579-626 with the early
// return at :444-445): prepare_pts / prepare_dynamic_pts, gen_pts_feats, run_network,
// raw2outputs and raw2outputs_blending.
#pragma once
#include "mlp_engine.cuh"
#include "zest_sample_ops.cuh"

namespace zest {

constexpr int kMaxViews = 16;
constexpr int kCamStride = 24;            // floats per camera in LDS: 12 (R|T rows) + 9 (K) + pad

struct FusedNet {                         // one MLP + its feature sources
    const float *bias;                    // packed: bias area
    const uint4 *tiles;                   // packed: tile stream
    const float4 *vol;                    // [D,Hv,Wv,8] or null
    const float4 *imgs;                   // [V,H,W,4] or null
    const float *w2cs, *intr;             // cameras or null
    int D, Hv, Wv, V, H, W;
    int head, v2;
};

struct FusedArgs {
    const float *ndc, *pts, *z, *dir;
    int R, S;
    FusedNet st, dy;
    float frame_idx;
    int white_bkgd;
    int bpr;                              // 32-sample blocks per ray = ceil(S / 32)
    float *partials;                      // [R*bpr, kPartialFloats] workspace
    float *out;                           // [R,16]
    unsigned long long *stamps;           // diagnostic builds (ZEST_STAMPS): 8 u64 per wave, else null
};

__device__ __forceinline__ bf16x8 pack_tile(const float (&v)[8]) {
    uint4 a = make_uint4(pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3]), pack_bf16(v[4], v[5]),
                         pack_bf16(v[6], v[7]));
    return *reinterpret_cast<bf16x8 *>(&a);
}

// Positional-encoding operand for C coordinates and L bands in plan slot order
// (mlp_plan.hip pe_map_acc): slot q < L*C holds sin (half 0) / cos (half 1) of
// 2^(q/C) * x[q%C]; then the raw coordinates two per slot; zero padding after that.
// The engine rounds operands to bf16 (8 significant bits), so the hardware sine is used:
// v_sin_f32 takes revolutions, cos is sin shifted by a quarter revolution, and the range
// reduction is one v_fract.  Its absolute error (~1e-6) plus the rounding of arg/(2 pi)
// (3e-5 rad at 2^9 x) is 1/50 of a bf16 ulp; the fp32 per-op path (encode.hip) keeps the
// accurate sincos.
template <int C, int L, int NT>
__device__ __forceinline__ void encode_pe_operand(const float (&x)[4], int half, OpArr<NT> &op) {
    static_assert(L * C + (C + 1) / 2 <= NT * 8, "slot layout");
    const float quarter = half ? 0.25f : 0.0f;
    float rev[C];
#pragma unroll
    for (int c = 0; c < C; c++) rev[c] = x[c] * 0.15915494309189535f;      // revolutions at band 0
#pragma unroll
    for (int t = 0; t < NT; t++) {
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; e++) {
            const int q = 8 * t + e;
            if (q < L * C) {
                const float r = fmaf(rev[q % C], (float)(1 << (q / C)), quarter);
                v[e] = __builtin_amdgcn_sinf(__builtin_amdgcn_fractf(r));
            } else if (q < L * C + (C + 1) / 2) {
                const int p = q - L * C;
                const float lo = x[2 * p], hi = (2 * p + 1 < C) ? x[2 * p + 1] : 0.0f;
                v[e] = half ? hi : lo;
            } else {
                v[e] = 0.0f;
            }
        }
        op.t[t] = pack_tile(v);
    }
}

// Feature operand (mlp_plan.hip feat_map_acc): slots 0-3 = volume channels 4*half..+3,
// slots 4+4p+c = channel c of source view 2p+half.
template <int NT>
__device__ __forceinline__ void encode_feat_operand(const FusedNet &n, const float *cams_lds,
                                                    const float (&ndc)[4], const float (&pw)[3],
                                                    int half, bool valid, OpArr<NT> &op) {
    float v[NT * 8];
#pragma unroll
    for (int i = 0; i < NT * 8; i++) v[i] = 0.0f;
    if (valid) {
        // Trilinear lookup of this half's four channels.  Branch-free: out-of-volume corners
        // read a clamped address with weight 0, so all eight 16-byte loads are in flight
        // together instead of one divergent branch (and one memory round trip) per corner.
        {
            float fx = zest_unnorm(ndc[0], n.Wv), fy = zest_unnorm(ndc[1], n.Hv), fz = zest_unnorm(ndc[2], n.D);
            fx = fminf(fmaxf(fx, -2.0f), (float)n.Wv + 1.0f);
            fy = fminf(fmaxf(fy, -2.0f), (float)n.Hv + 1.0f);
            fz = fminf(fmaxf(fz, -2.0f), (float)n.D + 1.0f);
            const float x0f = floorf(fx), y0f = floorf(fy), z0f = floorf(fz);
            const float tx = fx - x0f, ty = fy - y0f, tz = fz - z0f;
            const int x0 = (int)x0f, y0 = (int)y0f, z0 = (int)z0f;
            float4 tap[8];
            float wgt[8];
#pragma unroll
            for (int c = 0; c < 8; c++) {
                const int dx = c & 1, dy = (c >> 1) & 1, dz = c >> 2;
                const int xi = x0 + dx, yi = y0 + dy, zi = z0 + dz;
                const bool ok = (unsigned)xi < (unsigned)n.Wv && (unsigned)yi < (unsigned)n.Hv &&
                                (unsigned)zi < (unsigned)n.D;
                const int xc = min(max(xi, 0), n.Wv - 1), yc = min(max(yi, 0), n.Hv - 1),
                          zc = min(max(zi, 0), n.D - 1);
                wgt[c] = ok ? (dx ? tx : 1.0f - tx) * (dy ? ty : 1.0f - ty) * (dz ? tz : 1.0f - tz) : 0.0f;
                tap[c] = n.vol[2 * (((size_t)zc * n.Hv + yc) * n.Wv + xc) + half];
            }
#pragma unroll
            for (int c = 0; c < 8; c++) {
                v[0] = fmaf(wgt[c], tap[c].x, v[0]), v[1] = fmaf(wgt[c], tap[c].y, v[1]);
                v[2] = fmaf(wgt[c], tap[c].z, v[2]), v[3] = fmaf(wgt[c], tap[c].w, v[3]);
            }
        }
        // source views 2p+half: the view index beyond V is clamped and its result discarded
#pragma unroll
        for (int p = 0; p < (NT * 8 - 4) / 4; p++) {
            const int view = 2 * p + half;
            const int vc = view < n.V ? view : n.V - 1;
            ZestCam cam;
            const float *cl = cams_lds + vc * kCamStride;
#pragma unroll
            for (int i = 0; i < 3; i++) {
#pragma unroll
                for (int j = 0; j < 4; j++) cam.r[i][j] = cl[4 * i + j];
#pragma unroll
                for (int j = 0; j < 3; j++) cam.k[i][j] = cl[12 + 3 * i + j];
            }
            const float4 o = zest_color_tap(n.imgs + (size_t)vc * n.H * n.W, n.H, n.W, cam, pw[0], pw[1], pw[2]);
            const bool on = view < n.V;
            v[4 + 4 * p] = on ? o.x : 0.f, v[5 + 4 * p] = on ? o.y : 0.f;
            v[6 + 4 * p] = on ? o.z : 0.f, v[7 + 4 * p] = on ? o.w : 0.f;
        }
    }
#pragma unroll
    for (int t = 0; t < NT; t++) {
        float w8[8];
#pragma unroll
        for (int e = 0; e < 8; e++) w8[e] = v[8 * t + e];
        op.t[t] = pack_tile(w8);
    }
}

__device__ __forceinline__ void stage_cams(const FusedNet &n, float *lds) {
    if (!n.w2cs) return;
    const int nv = n.imgs ? n.V : 1;
    for (int i = threadIdx.x; i < nv * kCamStride; i += blockDim.x) {
        const int v = i / kCamStride, k = i % kCamStride;
        float val = 0.0f;
        if (k < 12) val = n.w2cs[16 * v + k];
        else if (k < 21 && n.intr) val = n.intr[9 * v + (k - 12)];
        lds[i] = val;
    }
}

#ifndef ZEST_FUSED_WAVES
#define ZEST_FUSED_WAVES 8         // waves per workgroup (8 = two per SIMD, 256 VGPRs each)
#define ZEST_FUSED_NB 1            // 32-sample column blocks per wave (tile reads feed NB MFMAs)
#endif
constexpr int kFusedWaves = ZEST_FUSED_WAVES, kFusedNB = ZEST_FUSED_NB;
constexpr int kPartialFloats = 20; // per-block record, see combine_kernel

struct BlockSamples {              // what a lane keeps about its sample across the two nets
    float x[4], pw[3], zz, dist;
    bool valid;
};

// Work unit: a block of 32 consecutive samples of one ray (rays are padded to bpr blocks).  A
// pass of the workgroup covers kFusedWaves * kFusedNB blocks; every wave runs the full network
// on its two blocks while all four waves share the weight stream through the LDS ring.  Each
// block is composited on its own with entry transmittance 1 and leaves a record (exit
// transmittance + weighted sums); combine_kernel chains the records of a ray.
template <int NT_FEAT_S, bool DYN, int NT_FEAT_D>
#ifndef ZEST_FUSED_WG_PER_CU
#define ZEST_FUSED_WG_PER_CU 1
#endif
__global__ __launch_bounds__(kFusedWaves * 64, kFusedWaves * ZEST_FUSED_WG_PER_CU / 4) void fused_blocks_kernel(FusedArgs a) {
    constexpr bool MOD_S = NT_FEAT_S > 0, MOD_D = NT_FEAT_D > 0;
    constexpr int NB = kFusedNB;
    constexpr int UNITS_S = stream_units(4, NT_FEAT_S), UNITS_D = DYN ? stream_units(6, NT_FEAT_D) : 0;
    using Ring = RingTiles<kFusedWaves, UNITS_S, UNITS_D>;
    // LDS: weight ring | cameras of both nets | per-lane (z, dist) of the pass's samples
    __shared__ __attribute__((aligned(16))) char lds[kRingUnits * 1024 + 2 * kMaxViews * kCamStride * 4 +
                                                     kFusedWaves * NB * 32 * 8];
    float *cams_s = reinterpret_cast<float *>(lds + kRingUnits * 1024), *cams_d = cams_s + kMaxViews * kCamStride;
    float2 *zd_lds = reinterpret_cast<float2 *>(cams_d + kMaxViews * kCamStride);
    stage_cams(a.st, cams_s);
    if (DYN) stage_cams(a.dy, cams_d);
    __syncthreads();

    const int lane = threadIdx.x & 63, col = lane & 31, half = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const Ring tiles{lds, (gptr_u4)a.st.tiles, (gptr_u4)a.dy.tiles, lane, half, wave,
                     (unsigned)(wave * Ring::kPieces * 64 + lane) * 16u,
                     (unsigned)(uintptr_t)(__attribute__((address_space(3))) char *)lds +
                         (unsigned)wave * Ring::kPieces * 1024u};
    tiles.prologue();

    const int n_blocks = a.R * a.bpr;
    const int n_pass = (n_blocks + kFusedWaves * NB - 1) / (kFusedWaves * NB);
    // A lane's sample is re-read from global memory (L1/L2 hits) wherever it is needed instead
    // of being held in registers across the network: the engine needs the registers more.
    auto fetch = [&](int pass, int nb, BlockSamples &b, int &gidx, const float *&dir) {
        const int g = (pass * kFusedWaves + wave) * NB + nb;
        gidx = g < n_blocks ? g : -1;
        const int r = g < n_blocks ? g / a.bpr : 0, s = (g % a.bpr) * 32 + col;
        b.valid = g < n_blocks && s < a.S;
        b.x[0] = b.x[1] = b.x[2] = 0.f, b.x[3] = a.frame_idx;
        b.pw[0] = b.pw[1] = b.pw[2] = 0.f, b.zz = 0.f, b.dist = 0.f;
        dir = a.dir + 3 * r;
        if (b.valid) {
            const size_t m = (size_t)r * a.S + s;
            const float *zr = a.z + (size_t)r * a.S;
            b.x[0] = a.ndc[3 * m], b.x[1] = a.ndc[3 * m + 1], b.x[2] = a.ndc[3 * m + 2];
            if (a.pts) b.pw[0] = a.pts[3 * m], b.pw[1] = a.pts[3 * m + 1], b.pw[2] = a.pts[3 * m + 2];
            b.zz = zr[s];
            const float dnorm = sqrtf(dir[0] * dir[0] + dir[1] * dir[1] + dir[2] * dir[2]);
            b.dist = ((s + 1 < a.S) ? (zr[s + 1] - b.zz) : 1e10f) * dnorm;
        }
    };
#ifdef ZEST_STAMPS
    unsigned long long st_enc = 0, st_eng = 0, st_comp = 0, st_n = 0;
    const unsigned long long st_t0 = __builtin_amdgcn_s_memtime(), st_r0 = __builtin_amdgcn_s_memrealtime();
#define ZEST_STAMP(var) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); var += t_ - st_last; st_last = t_; } while (0)
#else
#define ZEST_STAMP(var) do {} while (0)
#endif
    for (int pass = blockIdx.x; pass < n_pass; pass += gridDim.x) {
#ifdef ZEST_STAMPS
        unsigned long long st_last = __builtin_amdgcn_s_memtime();
        st_n++;
#endif
        int unit = 0;
        f32x16 head_s[NB], rgb_s[NB], head_d[NB], rgb_d[NB];
        // direction operand of net `n`, built when the engine reaches the view layer
        auto views_of = [&](const FusedNet &n, const float *cams) {
            return [&, cams](OpArr<2> (&views)[NB]) {
#pragma unroll
                for (int nb = 0; nb < NB; nb++) {
                    const int g = (pass * kFusedWaves + wave) * NB + nb;
                    const float *dir = a.dir + 3 * (g < n_blocks ? g / a.bpr : 0);
                    float dv[4] = {0.f, 0.f, 0.f, 0.f};
                    zest_view_dir(dir, n.w2cs ? cams : nullptr, dv);
                    encode_pe_operand<3, 4, 2>(dv, half, views[nb]);
                }
            };
        };
        {
            OpArr<4> pts_s[NB];
            OpArr<NT_FEAT_S> feat_s[NB];
#pragma unroll
            for (int nb = 0; nb < NB; nb++) {
                BlockSamples b;
                int g;
                const float *dir;
                fetch(pass, nb, b, g, dir);
                // compositing needs z and the sample spacing after the engine: park them in LDS so
                // that phase issues no global load (a vmcnt wait there would drain the weight DMA)
                if (half == 0) zd_lds[(wave * NB + nb) * 32 + col] = make_float2(b.zz, b.valid ? b.dist : -1.0f);
#ifdef ZEST_EXPERIMENT_NO_ENCODE        // timing experiment only
#pragma unroll
                for (int t = 0; t < 4; t++) pts_s[nb].t[t] = bf16x8{(short)lane, 1, 2, 3, 4, 5, 6, 7};
#else
                encode_pe_operand<3, 10, 4>(b.x, half, pts_s[nb]);
#endif
                if constexpr (MOD_S) encode_feat_operand<NT_FEAT_S>(a.st, cams_s, b.x, b.pw, half, b.valid, feat_s[nb]);
            }
            ZEST_STAMP(st_enc);
            engine_forward<NB, 4, MOD_S, NT_FEAT_S>(tiles, unit, a.st.v2 != 0, pts_s, feat_s,
                                                    views_of(a.st, cams_s), head_s, rgb_s);
            ZEST_STAMP(st_eng);
        }
        if (DYN) {
            OpArr<6> pts_d[NB];
            OpArr<NT_FEAT_D> feat_d[NB];
#pragma unroll
            for (int nb = 0; nb < NB; nb++) {
                BlockSamples b;
                int g;
                const float *dir;
                fetch(pass, nb, b, g, dir);
                encode_pe_operand<4, 10, 6>(b.x, half, pts_d[nb]);
                if constexpr (MOD_D) encode_feat_operand<NT_FEAT_D>(a.dy, cams_d, b.x, b.pw, half, b.valid, feat_d[nb]);
            }
            ZEST_STAMP(st_enc);
            engine_forward<NB, 6, MOD_D, NT_FEAT_D>(tiles, unit, false, pts_d, feat_d,
                                                    views_of(a.dy, cams_d), head_d, rgb_d);
            ZEST_STAMP(st_eng);
        }
        // ---- per-block compositing on lane half 0 (rgb tile rows 0-2, head tile rows 0,1)
#pragma unroll
        for (int nb = 0; nb < NB; nb++) {
            BlockSamples b;
            const int g_ = (pass * kFusedWaves + wave) * NB + nb;
            const int gidx = g_ < n_blocks ? g_ : -1;
            {
                const float2 zd = zd_lds[(wave * NB + nb) * 32 + col];
                b.zz = zd.x, b.dist = fmaxf(zd.y, 0.0f), b.valid = zd.y >= 0.0f;
            }
            float cr = rgb_s[nb][0], cg = rgb_s[nb][1], cb = rgb_s[nb][2], sg = head_s[nb][0];
            if (a.st.v2) {   // 'v2' nets activate inside the network; the compositor does it again
                cr = zest_sigmoid(cr), cg = zest_sigmoid(cg), cb = zest_sigmoid(cb), sg = fmaxf(sg, 0.f);
            }
            cr = zest_sigmoid(cr), cg = zest_sigmoid(cg), cb = zest_sigmoid(cb);
            const float al_s = b.valid ? 1.0f - expf(-fmaxf(sg, 0.f) * b.dist) : 0.0f;
            float rec[kPartialFloats];
#pragma unroll
            for (int i = 0; i < kPartialFloats; i++) rec[i] = 0.f;
            {
                float tot;
                const float w = al_s * lower_half_excl_prod(1.0f - al_s + 1e-10f, col, &tot);
                rec[0] = tot;
                rec[1] = lower_half_sum(w * cr), rec[2] = lower_half_sum(w * cg), rec[3] = lower_half_sum(w * cb);
                rec[4] = lower_half_sum(w * b.zz), rec[5] = lower_half_sum(w);
            }
            if (DYN) {
                const float blend = zest_sigmoid(head_s[nb][1]);
                const float er = zest_sigmoid(rgb_d[nb][0]), eg = zest_sigmoid(rgb_d[nb][1]),
                            eb = zest_sigmoid(rgb_d[nb][2]);
                const float a_fg = b.valid ? 1.0f - expf(-fmaxf(head_d[nb][0], 0.f) * b.dist) : 0.0f;
                const float a_d = a_fg * blend, a_st = al_s * (1.0f - blend);
                float tot_b, tot_f;
                const float Tb = lower_half_excl_prod((1.0f - a_d) * (1.0f - a_st) + 1e-10f, col, &tot_b);
                const float wf = a_fg * lower_half_excl_prod(1.0f - a_fg + 1e-10f, col, &tot_f);
                const float wd = Tb * a_d, ws = Tb * a_st;
                rec[6] = tot_b;
                rec[7] = lower_half_sum(wd * er + ws * cr), rec[8] = lower_half_sum(wd * eg + ws * cg);
                rec[9] = lower_half_sum(wd * eb + ws * cb), rec[10] = lower_half_sum((wd + ws) * b.zz);
                rec[11] = lower_half_sum(wd);
                rec[12] = tot_f;
                rec[13] = lower_half_sum(wf * er), rec[14] = lower_half_sum(wf * eg), rec[15] = lower_half_sum(wf * eb);
                rec[16] = lower_half_sum(wf * b.zz);
            }
            if (lane == 0 && gidx >= 0) {
                float4 *o = reinterpret_cast<float4 *>(a.partials + (size_t)gidx * kPartialFloats);
#pragma unroll
                for (int i = 0; i < (DYN ? 5 : 2); i++)
                    o[i] = make_float4(rec[4 * i], rec[4 * i + 1], rec[4 * i + 2], rec[4 * i + 3]);
            }
        }
        ZEST_STAMP(st_comp);
    }
    tiles.drain();
#ifdef ZEST_STAMPS
    if (a.stamps && lane == 0) {
        unsigned long long *o = a.stamps + (size_t)(blockIdx.x * kFusedWaves + wave) * 8;
        o[0] = __builtin_amdgcn_s_memtime() - st_t0, o[1] = st_enc, o[2] = st_eng, o[3] = st_comp;
        o[4] = tiles.t_wait, o[5] = tiles.t_issue, o[6] = __builtin_amdgcn_s_memrealtime() - st_r0, o[7] = st_n;
    }
#endif
}

// one translation unit per variant (fused_*.hip) so they compile in parallel
#define ZEST_FUSED_VARIANT(tag, NTS, DYN, NTD)                                                   \
    int fused_launch_##tag(const FusedArgs &a, int blocks, hipStream_t stream) {                 \
        hipLaunchKernelGGL((fused_blocks_kernel<NTS, DYN, NTD>), dim3(blocks),                   \
                           dim3(kFusedWaves * 64), 0, stream, a);                                \
        ZEST_RETURN_LAUNCH("zest_render_fused_fwd(" #tag ")");                                   \
    }                                                                                            \
    int fused_wg_per_cu_##tag() { return ZEST_FUSED_WG_PER_CU; }                                 \
    int fused_units_##tag(int which) {                                                           \
        return which == 0 ? stream_units(4, NTS) : (DYN ? stream_units(6, NTD) : 0);             \
    }

#define ZEST_FUSED_DECL(tag)                                                    \
    int fused_launch_##tag(const FusedArgs &a, int blocks, hipStream_t stream); \
    int fused_units_##tag(int which);                                           \
    int fused_wg_per_cu_##tag();
ZEST_FUSED_DECL(s0)
ZEST_FUSED_DECL(s2)
ZEST_FUSED_DECL(s3)
ZEST_FUSED_DECL(s0d0)
ZEST_FUSED_DECL(s2d0)
ZEST_FUSED_DECL(s3d0)
ZEST_FUSED_DECL(s2d2)
ZEST_FUSED_DECL(s3d2)

}  // namespace zest
