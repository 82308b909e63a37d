// Fused inference renderer: per-sample work never leaves the chip.
//
// Rays are cut into blocks of 32 samples = two MFMA column blocks of 16.  For its blocks a
// wave's lanes build the MLP operands in registers (lane l: sample column l & 15 of each
// column block, as group g = l >> 4: sin/cos and the band of a pair, volume channel quads and
// source views are split over the four groups), the bf16 engine (mlp_engine.cuh) runs the
// whole network on them with the weights streamed through an LDS ring shared by the workgroup,
// and the raw colour/density land on lanes 0-15 of each column block; one v_permlane16_swap
// per value lines the 32 samples up on lanes 0-31, where a shuffle scan composites the block.
// Each block is composited with entry transmittance 1 and leaves an 80-byte record; the records
// of a ray are chained into the per-ray maps (64 B per ray) right here: a workgroup owns a contiguous
// range of whole rays and walks their blocks 8 at a time, so the records of a pass sit in LDS and a ray
// that continues into the workgroup's next pass carries its running sums there (any S, any number of
// blocks per ray).  The other pass shape (taken for a few long rays on a big device, or forced with
// zest_render_fused_set_passes) splits ALL blocks evenly over the workgroups, rays notwithstanding: the
// records then go to an HBM workspace and a second tiny kernel (fused_combine_kernel) chains them.
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
    const float4 *vol;                    // [Hv,Wv,D,8] (zest_vox) or null
    const float4 *imgs;                   // [V,H,W,4] or null
    const float *w2cs, *intr;             // cameras or null
    int D, Hv, Wv, V, H, W;
    int head, v2;                         // v2: checked against the kernel's V2S by the host
};

struct FusedArgs {
    const float *ndc, *pts, *z, *dir;
    int R, S;
    FusedNet st, dy;
    float frame_idx;
    int white_bkgd;
    int bpr;                              // blocks per ray = ceil(S / block samples)
    float *partials;                      // [R*bpr, kPartialFloats] workspace
    float *out;                           // [R,16]
    unsigned long long *stamps;           // diagnostic builds (ZEST_STAMPS): 8 u64 per wave, else null
    int ray_ranges;                       // 1: workgroup w of gridDim.x owns the rays [w R / n, (w+1) R / n) and finishes
                                          //    them in the kernel (a ray spanning two passes is carried in LDS);
                                          // 0: it owns an equal share of ALL blocks, records go to `partials`
};

// Running composite of one ray over its block records (exit transmittance + weighted sums, entry
// transmittance 1 each): the blocks are chained in order, so every pass shape sums in the same order.
struct RayState {
    float Ts = 1.f, Tb = 1.f, Tf = 1.f;
    float s[5] = {0, 0, 0, 0, 0}, bl[5] = {0, 0, 0, 0, 0}, fg[4] = {0, 0, 0, 0};
    __device__ __forceinline__ void chain(const float *p, bool dyn) {
#pragma unroll
        for (int i = 0; i < 5; i++) s[i] += Ts * p[1 + i];
        Ts *= p[0];
        if (dyn) {
#pragma unroll
            for (int i = 0; i < 5; i++) bl[i] += Tb * p[7 + i];
            Tb *= p[6];
#pragma unroll
            for (int i = 0; i < 4; i++) fg[i] += Tf * p[13 + i];
            Tf *= p[12];
        }
    }
    // per-ray maps: column layout of include/zest_render.h
    __device__ __forceinline__ void emit(int white_bkgd, float *__restrict__ out_row) const {
        float4 *o = reinterpret_cast<float4 *>(out_row);
        const float bg = white_bkgd ? 1.0f - s[4] : 0.0f;
        o[0] = make_float4(s[0] + bg, s[1] + bg, s[2] + bg, s[3]);
        o[1] = make_float4(s[4], bl[0], bl[1], bl[2]);
        o[2] = make_float4(bl[3], fg[0], fg[1], fg[2]);
        // (the zero of the two reserved columns is made opaque here: as a plain constant hipcc materialises the
        // pair once at kernel entry, finds no register for it across the unrolled network and spills it - 1 MiB
        // of scratch write-back per launch for two zeros)
        float zero = 0.f;
        asm volatile("" : "+v"(zero));
        o[3] = make_float4(fg[3], bl[4], zero, zero);
    }
    // the open ray of a workgroup between two of its passes (kCarryFloats floats in LDS)
    __device__ __forceinline__ void save(float *c) const {
        c[0] = Ts, c[1] = Tb, c[2] = Tf;
#pragma unroll
        for (int i = 0; i < 5; i++) c[3 + i] = s[i], c[8 + i] = bl[i];
#pragma unroll
        for (int i = 0; i < 4; i++) c[13 + i] = fg[i];
    }
    __device__ __forceinline__ void load(const float *c) {
        Ts = c[0], Tb = c[1], Tf = c[2];
#pragma unroll
        for (int i = 0; i < 5; i++) s[i] = c[3 + i], bl[i] = c[8 + i];
#pragma unroll
        for (int i = 0; i < 4; i++) fg[i] = c[13 + i];
    }
};
constexpr int kCarryFloats = 20;

template <class Rec>
__device__ __forceinline__ void combine_ray(Rec rec, int bpr, bool dyn, int white_bkgd, float *__restrict__ out_row) {
    RayState st;
    for (int b = 0; b < bpr; b++) st.chain(rec(b), dyn);
    st.emit(white_bkgd, out_row);
}

// Positional-encoding operand for C coordinates and L (even) bands in plan position order
// (mlp_plan.hip pe_map_acc): element e of k-tile kt is m = 8 kt + e; m < (L/2) C is
// sin (g even) / cos (g odd) of 2^(2 (m / C) + (g >> 1)) x[m % C]; m = (L/2) C is the raw
// coordinate x[g]; zero after that.
// bf16 / fp16 operands (8 / 11 significant bits): the hardware sine is used: v_sin_f32 takes
// revolutions, cos is sin shifted by a quarter revolution, and the range reduction is one
// v_fract.  Its absolute error (~1e-6) plus the rounding of arg/(2 pi) (3e-5 rad at 2^9 x) is
// 1/50 of a bf16 ulp and 1/6 of an fp16 ulp.  The split-fp16 mode (fp32-class results) takes
// the accurate Cody-Waite sincos of the fp32 per-op path (zest_common.cuh, 9e-8 abs).
template <int EP, int C, int L, int NK>
__device__ __forceinline__ void encode_pe_operand(const float (&x)[4], int grp, OpArr<NK, ep_parts(EP)> &op) {
    static_assert((L / 2) * C + 1 <= NK * 8 && L % 2 == 0, "position layout");
    const float quarter = (grp & 1) ? 0.25f : 0.0f;
    const float gscale = (grp & 2) ? 2.0f * 0.15915494309189535f : 0.15915494309189535f;
    const float gband = (grp & 2) ? 2.0f : 1.0f;
    float rev[C];
#pragma unroll
    for (int c = 0; c < C; c++) rev[c] = x[c] * gscale;      // revolutions at the group's band of pair 0
    const float raw = grp == 0 ? x[0] : (grp == 1 ? x[1] : (grp == 2 ? x[2] : (C > 3 ? x[3] : 0.0f)));
#pragma unroll
    for (int t = 0; t < NK; t++) {
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; e++) {
            const int m = 8 * t + e;
            if (m < (L / 2) * C) {
                if constexpr (EP == ZEST_PREC_F16X3) {
                    float sn, cs;           // the argument 2^band x is exact, as in the reference
                    zest_sincos(x[m % C] * (gband * (float)(1 << (2 * (m / C)))), &sn, &cs);
                    v[e] = (grp & 1) ? cs : sn;
                } else {
                    const float r = fmaf(rev[m % C], (float)(1 << (2 * (m / C))), quarter);
                    v[e] = __builtin_amdgcn_sinf(__builtin_amdgcn_fractf(r));
                }
            } else if (m == (L / 2) * C) {
                v[e] = raw;
            } else {
                v[e] = 0.0f;
            }
        }
        store_tile<EP>(v, op, t);
    }
}

// One source view in the 16-bit operand modes: pixel position straight from the staged 3 x 4 matrix P = K [R | T]
// (stage_cams) and ONE hardware reciprocal - fx = (P p)_x / (P p)_z is the reference's
// ((q_x / q_z) / (W - 1) * 2 - 1 + 1) / 2 * (W - 1) (utils.py:262-269, 486-487) up to fp32 rounding, and its strict
// in-frame test -1 < 2 u - 1 < 1 is 0 < fx < W - 1.  The reference's own operation order (four IEEE divisions, the
// normalise / un-normalise round trip) matters where a rounding can flip the mask of a sample ON the frame border; it
// is kept by zest_color_tap for the fp32-class modes.  Here it would be 2/3 of the instructions of a tap whose result
// is rounded to 8 or 11 bits.
__device__ __forceinline__ float4 zest_color_tap_fast(const float4 *__restrict__ img, int H, int W,
                                                      const float *__restrict__ P, float px, float py, float pz) {
    const float4 r0 = *reinterpret_cast<const float4 *>(P), r1 = *reinterpret_cast<const float4 *>(P + 4),
                 r2 = *reinterpret_cast<const float4 *>(P + 8);
    const float qx = fmaf(pz, r0.z, fmaf(py, r0.y, fmaf(px, r0.x, r0.w)));
    const float qy = fmaf(pz, r1.z, fmaf(py, r1.y, fmaf(px, r1.x, r1.w)));
    const float qz = fmaf(pz, r2.z, fmaf(py, r2.y, fmaf(px, r2.x, r2.w)));
    const float rz = __builtin_amdgcn_rcpf(qz);
    float fx = qx * rz, fy = qy * rz;
    const float wm = (float)(W - 1), hm = (float)(H - 1);
    const float mask = (fx > 0.0f && fx < wm && fy > 0.0f && fy < hm) ? 1.0f : 0.0f;
    fx = fminf(fmaxf(fx, 0.0f), wm), fy = fminf(fmaxf(fy, 0.0f), hm);       // padding_mode='border'; NaN -> 0
    const float x0f = floorf(fx), y0f = floorf(fy);
    const float tx = fx - x0f, ty = fy - y0f;
    const int x0 = (int)x0f, y0 = (int)y0f;
    const int i00 = y0 * W + x0, dx = x0 + 1 < W ? 1 : 0, dy = y0 + 1 < H ? W : 0;
#ifdef ZEST_EXPERIMENT_TAP_SAME
    const float4 a = img[i00 & 0], b = img[(i00 + dx) & 0], d = img[(i00 + dy) & 0], e = img[(i00 + dy + dx) & 0];
#else
    const float4 a = img[i00], b = img[i00 + dx], d = img[i00 + dy], e = img[i00 + dy + dx];
#endif
    const float w00 = (1.0f - tx) * (1.0f - ty), w10 = tx * (1.0f - ty), w01 = (1.0f - tx) * ty, w11 = tx * ty;
    float4 o;
    o.x = fmaf(w11, e.x, fmaf(w01, d.x, fmaf(w10, b.x, w00 * a.x)));
    o.y = fmaf(w11, e.y, fmaf(w01, d.y, fmaf(w10, b.y, w00 * a.y)));
    o.z = fmaf(w11, e.z, fmaf(w01, d.z, fmaf(w10, b.z, w00 * a.z)));
    o.w = mask;
    return o;
}

// Feature operand (mlp_plan.h feat_quad_col): a lane's eight values of k-tile kt are the 4-channel quads
// q = 8 kt + 2 g and q + 1, filled in ROUNDS r = 2 kt + (q & 1) of one quad per lane group.  Rounds below
// rv = feat_volume_round(V) hold source views 4 r + g: one bilinear colour tap per lane, the same instructions for
// all 64 lanes; round rv holds the encoding volume's two channel quads at groups 0, 1 and up to two more views at
// groups 2, 3; later rounds are empty and cost nothing (all of this is wave-uniform: V is a kernel argument).
// 8 views: two rounds of colour taps and the trilinear lookup; 4 views (the dynamic net): one and the lookup.
template <int EP, int NK>
__device__ __forceinline__ void encode_feat_operand(const FusedNet &n, const float *cams_lds,
                                                    const float (&ndc)[4], const float (&pw)[3],
                                                    int grp, bool valid, OpArr<NK, ep_parts(EP)> &op) {
    float v[NK * 8];
#pragma unroll
    for (int i = 0; i < NK * 8; i++) v[i] = 0.0f;
    auto view_tap = [&](int view, float *dst) {       // view >= V: clamped address, result dropped
        const int vc = view < n.V ? view : n.V - 1;
        const float *cl = cams_lds + vc * kCamStride;
        float4 o;
        if constexpr (EP == ZEST_PREC_F16X3) {
            ZestCam cam;
#pragma unroll
            for (int i = 0; i < 3; i++) {
#pragma unroll
                for (int j = 0; j < 4; j++) cam.r[i][j] = cl[4 * i + j];
#pragma unroll
                for (int j = 0; j < 3; j++) cam.k[i][j] = cl[12 + 3 * i + j];
            }
#ifdef ZEST_EXPERIMENT_TAP_SAME        // timing experiment only: every tap reads pixel 0 of view 0 (arithmetic kept, memory trivial)
            o = zest_color_tap<true>(n.imgs, n.H, n.W, cam, pw[0], pw[1], pw[2]);
#else
            o = zest_color_tap(n.imgs + (size_t)vc * n.H * n.W, n.H, n.W, cam, pw[0], pw[1], pw[2]);
#endif
        } else {
            o = zest_color_tap_fast(n.imgs + vc * (n.H * n.W), n.H, n.W, cl, pw[0], pw[1], pw[2]);
        }
        const bool on = view < n.V;
        dst[0] = on ? o.x : 0.f, dst[1] = on ? o.y : 0.f, dst[2] = on ? o.z : 0.f, dst[3] = on ? o.w : 0.f;
    };
#ifdef ZEST_EXPERIMENT_NO_GATHER      // timing experiment only: features are zero
    valid = false;
#endif
#ifdef ZEST_EXPERIMENT_FAKE_FEAT      // timing experiment only: no lookups, but feature values with realistic bit patterns
    if (valid) {                      // (zero features let the chip hold a higher clock: not a clean knock-out)
#pragma unroll
        for (int i = 0; i < NK * 8; i++)
            v[i] = __builtin_amdgcn_fractf(ndc[0] * (13.37f + (float)i) + ndc[2] * 7.1f + (float)grp * 0.31f) - 0.25f;
    }
    valid = false;
#endif
    const int rv = feat_volume_round(n.V);
    if (valid) {
        float vv[4] = {0.f, 0.f, 0.f, 0.f};
        if (grp < 2) {
            // Trilinear lookup of this group's four channels.  Branch-free: out-of-volume corners
            // read a clamped address with weight 0, so all eight 16-byte loads are in flight
            // together instead of one divergent branch (and one memory round trip) per corner.
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
#ifdef ZEST_EXPERIMENT_TAP_SAME
                tap[c] = n.vol[(((yc * n.Wv + xc) * n.D + zc) & 0) + grp];
#else
                tap[c] = n.vol[2 * ((yc * n.Wv + xc) * n.D + zc) + grp];      // zest_vox, 32-bit (checked by the host)
#endif
            }
#pragma unroll
            for (int c = 0; c < 8; c++) {
                vv[0] = fmaf(wgt[c], tap[c].x, vv[0]), vv[1] = fmaf(wgt[c], tap[c].y, vv[1]);
                vv[2] = fmaf(wgt[c], tap[c].z, vv[2]), vv[3] = fmaf(wgt[c], tap[c].w, vv[3]);
            }
        }
#pragma unroll
        for (int r = 0; r < 2 * NK; r++) {
            float *dst = v + 8 * (r >> 1) + 4 * (r & 1);
            if (r < rv) {
                view_tap(4 * r + grp, dst);
            } else if (r == rv) {
                if (n.V > 4 * rv) view_tap(grp >= 2 ? 4 * rv + grp - 2 : n.V, dst);     // at most views 4 rv, 4 rv + 1
                if (grp < 2) dst[0] = vv[0], dst[1] = vv[1], dst[2] = vv[2], dst[3] = vv[3];
            }
        }
    }
#pragma unroll
    for (int t = 0; t < NK; t++) {
        float w8[8];
#pragma unroll
        for (int e = 0; e < 8; e++) w8[e] = v[8 * t + e];
        store_tile<EP>(w8, op, t);
    }
}

// Cameras of a net's source views -> LDS, kCamStride floats per view: the rows of w2c[:3, :4] and K (fp32-class mode:
// the reference's two-stage projection) or, PROJ, the product P = K [R | T] (12 floats, zest_color_tap_fast); the
// first view's rotation rows are read by zest_view_dir in both forms, so PROJ keeps them at floats 12 .. 23.
template <bool PROJ>
__device__ __forceinline__ void stage_cams(const FusedNet &n, float *lds) {
    if (!n.w2cs) return;
    const int nv = n.imgs ? n.V : 1;
    for (int i = threadIdx.x; i < nv * kCamStride; i += blockDim.x) {
        const int v = i / kCamStride, k = i % kCamStride;
        float val = 0.0f;
        if constexpr (PROJ) {
            if (k < 12 && n.intr) {
                const int row = k >> 2, col = k & 3;
                const float *K = n.intr + 9 * v + 3 * row, *Rt = n.w2cs + 16 * v + col;
                val = fmaf(K[2], Rt[8], fmaf(K[1], Rt[4], K[0] * Rt[0]));
            } else if (k >= 12) {
                val = n.w2cs[16 * v + (k - 12)];
            }
        } else {
            if (k < 12) val = n.w2cs[16 * v + k];
            else if (k < 21 && n.intr) val = n.intr[9 * v + (k - 12)];
        }
        lds[i] = val;
    }
}

#ifndef ZEST_FUSED_WAVES
#define ZEST_FUSED_WAVES 8         // waves per workgroup (8 = two per SIMD, 256 VGPRs each)
#endif
constexpr int kFusedWaves = ZEST_FUSED_WAVES;
constexpr int kPartialFloats = 20; // per-block record, see combine_ray
// column blocks of 16 samples a wave carries through the network: two, or one where every operand
// is a register pair (split fp16).  A block of a ray = 16 * CB consecutive samples.
constexpr int fused_cb(int EP) { return EP == ZEST_PREC_F16X3 ? 1 : 2; }

struct BlockSamples {              // what a lane keeps about its sample across the two nets
    float x[4], pw[3], zz, dist;
    bool valid;
};

// sum / exclusive product scan over the BS (16 or 32) lanes that hold a block's samples
template <int BS>
__device__ __forceinline__ float block_sum(float v) {
    if constexpr (BS == 32) return lower_half_sum(v);
    v += dpp_f<0xB1>(0.f, v);
    v += dpp_f<0x4E>(0.f, v);
    v += dpp_f<0x141>(0.f, v);
    v += dpp_f<0x140>(0.f, v);                 // every lane of a 16-lane row holds the row sum
    return readlane_f<0>(v);
}
template <int BS>
__device__ __forceinline__ float block_excl_prod(float f, int c, float *total) {
    if constexpr (BS == 32) return lower_half_excl_prod(f, c, total);
    float v = f;
    v *= dpp_f<0x111>(1.f, v);
    v *= dpp_f<0x112>(1.f, v);
    v *= dpp_f<0x114>(1.f, v);
    v *= dpp_f<0x118>(1.f, v);                 // inclusive within the 16-lane row
    *total = readlane_f<15>(v);
    const float ex = dpp_f<0x138>(1.f, v);     // shift the wave right by one lane
    return c == 0 ? 1.0f : ex;
}

// Work unit: a block of BS = 16 CB consecutive samples of one ray (rays are padded to bpr
// blocks).  A pass of the workgroup covers kFusedWaves blocks, one per wave: every wave runs the
// full network on its block while all waves share the weight stream through the LDS ring.  Each
// block is composited on its own with entry transmittance 1 and leaves a record (exit
// transmittance + weighted sums).  A workgroup owns a contiguous range of blocks and takes them kFusedWaves at a
// time (wave w of pass k: block 8 k + w of the range): whole rays (a.ray_ranges: the records of a ray are chained
// in the pass itself, a ray that continues into the next pass carries its sums in LDS), or an equal share of
// all blocks (the records go to HBM and fused_combine_kernel chains them).
// NT_FEAT_*: stream tiles of the feature operand per row block (2 x its k-tiles; 0 = no features)
#ifndef ZEST_FUSED_WG_PER_CU
#define ZEST_FUSED_WG_PER_CU 1
#endif
// Register-pressure knobs of the two-net kernels (measured: tools/kernel_resources.py):
#ifndef ZEST_REBUILD_PTS
#define ZEST_REBUILD_PTS 0         // 1: the point operand is rebuilt from LDS at the skip layer instead of kept in registers
#endif
#ifndef ZEST_GATHER_FENCE
#define ZEST_GATHER_FENCE 1        // 1: two-net kernels with two k-tiles of static features gather one column block at a time
#endif
#ifndef ZEST_EARLY_DYN_GATHER
#define ZEST_EARLY_DYN_GATHER 0    // 1: the dynamic net's gathers are issued at the start of the pass, operand parked in LDS
#endif
// V2S: the static net is a 'v2' net (additive modulation; only single-net kernels with features are built for it)
template <int EP, int NT_FEAT_S, bool DYN, int NT_FEAT_D, bool V2S = false>
__global__ __launch_bounds__(kFusedWaves * 64, kFusedWaves * ZEST_FUSED_WG_PER_CU / 4) void fused_blocks_kernel(FusedArgs a) {
    constexpr bool MOD_S = NT_FEAT_S > 0, MOD_D = NT_FEAT_D > 0;
    constexpr int NP = ep_parts(EP), CB = fused_cb(EP), BS = 16 * CB;
    if constexpr (EP == ZEST_PREC_F16) engine_fp16_overflow_clamp();
    constexpr int UNITS_S = stream_units(4, NT_FEAT_S, NP), UNITS_D = DYN ? stream_units(6, NT_FEAT_D, NP) : 0;
    using Ring = RingTiles<kFusedWaves, UNITS_S, UNITS_D>;
    // LDS: weight ring | cameras of both nets | per-lane (z, dist) of the pass's samples | ring flags |
    // block records | sample coordinates (the point operand is rebuilt from them at the skip layer) |
    // with a dynamic net: the static net's per-sample results and the dynamic feature operand, parked
    // while the other net runs (registers the engine needs: the two-net kernels spilled them before)
    constexpr int kFdBytes = (DYN && MOD_D && ZEST_EARLY_DYN_GATHER) ? kFusedWaves * CB * (NT_FEAT_D / 2) * NP * 1024 : 0;
    constexpr int kStBytes = DYN ? kFusedWaves * 32 * 24 : 0;
    // modulation cache (mlp_engine.cuh ZEST_MCACHE_JB): kMcJB row blocks of m per wave, [row block][column block][lane] x 16 B;
    // the two nets use it one after the other
    constexpr bool kMc = mcache_for(EP, MOD_S, NT_FEAT_S / 2) || (DYN && mcache_for(EP, MOD_D, NT_FEAT_D / 2));
    constexpr int kMcWaveBytes = kMc ? kMcJB * CB * 1024 : 0;
    __shared__ __attribute__((aligned(16))) char lds[kRingUnits * 1024 + 2 * kMaxViews * kCamStride * 4 +
                                                     kFusedWaves * 32 * 8 + 2 * kSlots * 4 +
                                                     kFusedWaves * kPartialFloats * 4 + kFusedWaves * 32 * 16 +
                                                     kFdBytes + kStBytes + 2 * kCarryFloats * 4 +
                                                     kFusedWaves * kMcWaveBytes];
    static_assert(sizeof(lds) <= 163840, "LDS budget of one workgroup per CU");
    static_assert((sizeof(lds) - kFusedWaves * kMcWaveBytes) % 16 == 0, "the modulation cache starts 16-byte aligned");
    float *cams_s = reinterpret_cast<float *>(lds + kRingUnits * 1024), *cams_d = cams_s + kMaxViews * kCamStride;
    float2 *zd_lds = reinterpret_cast<float2 *>(cams_d + kMaxViews * kCamStride);
    constexpr bool PROJ = EP != ZEST_PREC_F16X3;         // 16-bit operand modes: fast projection
    stage_cams<PROJ>(a.st, cams_s);
    if (DYN) stage_cams<PROJ>(a.dy, cams_d);
    int *ring_flags = reinterpret_cast<int *>(zd_lds + kFusedWaves * 32);
    float *rec_lds = reinterpret_cast<float *>(ring_flags + 2 * kSlots);      // [waves][kPartialFloats]
    float4 *x_lds = reinterpret_cast<float4 *>(rec_lds + kFusedWaves * kPartialFloats);    // [waves][32]
    uint4 *fd_lds = reinterpret_cast<uint4 *>(x_lds + kFusedWaves * 32);                   // [waves][CB][k-tile][part][64]
    float2 *st_lds = reinterpret_cast<float2 *>(reinterpret_cast<char *>(fd_lds) + kFdBytes);   // [waves][3][32]
    float *carry_lds = reinterpret_cast<float *>(reinterpret_cast<char *>(st_lds) + kStBytes);  // [2][kCarryFloats]
    char *mc_lds = lds + sizeof(lds) - kFusedWaves * kMcWaveBytes;                              // 16-byte aligned (sizes above)
#ifdef ZEST_RING_FLAGS
    Ring::init_flags(ring_flags);
#endif
    __syncthreads();

    const int lane0 = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const Ring tiles{lds, (gptr_u4)a.st.tiles, (gptr_u4)a.dy.tiles, lane0, lane0 >> 4, wave,
                     (unsigned)(wave * Ring::kPieces * 64 + lane0) * 16u,
                     (unsigned)(uintptr_t)(__attribute__((address_space(3))) char *)lds +
                         (unsigned)wave * Ring::kPieces * 1024u
#ifdef ZEST_RING_FLAGS
                     , (unsigned)(uintptr_t)(__attribute__((address_space(3))) char *)ring_flags
#endif
    };
    tiles.init_addr();
    if constexpr (kMc)
        tiles.init_mcache((unsigned)(uintptr_t)(__attribute__((address_space(3))) char *)mc_lds +
                          (unsigned)wave * kMcWaveBytes + (unsigned)lane0 * 16u);
    tiles.prologue();

    // The workgroup's blocks.  Workgroups are numbered so that those of one XCD (equal blockIdx % 8: one L2)
    // own neighbouring ranges: the rays whose gathers touch neighbouring voxels and pixels (whole-image loops
    // render contiguous pixel runs) meet in one L2 instead of being fetched into all eight.
    const bool ranges = a.ray_ranges != 0;
    const int n_wg = (int)gridDim.x;
#ifndef ZEST_NO_XCD_ORDER
    const int wg = n_wg % 8 == 0 ? (int)(blockIdx.x % 8) * (n_wg / 8) + (int)(blockIdx.x / 8) : (int)blockIdx.x;
#else
    const int wg = (int)blockIdx.x;
#endif
    int range_b0, range_nb;                 // first block, number of blocks
    if (ranges) {
        range_b0 = (int)((long long)wg * a.R / n_wg) * a.bpr;
        range_nb = (int)((long long)(wg + 1) * a.R / n_wg) * a.bpr - range_b0;
    } else {
        const int n_blocks = a.R * a.bpr;
        const int share = ((n_blocks + kFusedWaves - 1) / kFusedWaves + n_wg - 1) / n_wg * kFusedWaves;   // whole passes
        range_b0 = wg * share, range_nb = min(share, n_blocks - range_b0);
    }
    // this wave's block in pass `pass` of the workgroup (-1: none)
    auto block_of = [&](int pass) {
        const int lb = pass * kFusedWaves + wave;
        return __builtin_amdgcn_readfirstlane(lb < range_nb ? range_b0 + lb : -1);
    };
    // A lane's samples (column block cb: sample 16 cb + col of the block) are re-read from
    // global memory (L1/L2 hits) wherever they are needed instead of being held in registers
    // across the network: the engine needs the registers more.
    auto fetch = [&](int g, int cb, int col, BlockSamples &b) {
        const int r = g >= 0 ? g / a.bpr : 0, s = (g >= 0 ? g % a.bpr : 0) * BS + 16 * cb + col;
        b.valid = g >= 0 && s < a.S;
        b.x[0] = b.x[1] = b.x[2] = 0.f, b.x[3] = a.frame_idx;
        b.pw[0] = b.pw[1] = b.pw[2] = 0.f, b.zz = 0.f, b.dist = 0.f;
        const float *dir = a.dir + 3 * r;
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
    // a value of the block's samples: column block 0 on lanes 0-15 of `lo`, column block 1 on lanes
    // 0-15 of `hi` -> lanes 0-31 (v_permlane16_swap: row 1 of the first operand <-> row 0 of the second)
    auto join = [](const f32x4 (&t)[CB], int i) {
        if constexpr (CB == 1) {
            return t[0][i];
        } else {
            const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(t[0][i]), __float_as_uint(t[CB - 1][i]), false, false);
            return __uint_as_float(r[0]);
        }
    };
#ifdef ZEST_STAMPS
    unsigned long long st_enc = 0, st_eng = 0, st_comp = 0, st_n = 0;
    const unsigned long long st_t0 = __builtin_amdgcn_s_memtime(), st_r0 = __builtin_amdgcn_s_memrealtime();
#define ZEST_STAMP(var) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); var += t_ - st_last; st_last = t_; } while (0)
#else
#define ZEST_STAMP(var) do {} while (0)
#endif
    for (int k = 0; k * kFusedWaves < range_nb; k++) {
        const int pass = k;
#ifdef ZEST_STAMPS
        unsigned long long st_last = __builtin_amdgcn_s_memtime();
        st_n++;
#endif
        const int g = block_of(pass);
        // A wave without a block (the tail of the workgroup's range) only keeps the weight ring in step - its
        // DMA share and the rendezvous of every chunk - and leaves the matrix pipe of its SIMD to its partner
        // wave, which then runs the network about twice as fast.
        if (__builtin_expect(g < 0, 0)) {
            tiles.finish(0, UNITS_S + UNITS_D);
        } else {
        // The lane index is made opaque once per pass: every per-lane address below (samples, LDS
        // parking areas, record slots) is then recomputed in the pass - a handful of VALU instructions -
        // instead of being hoisted out of the loop as ~25 loop-invariant VGPRs that the two-net kernels
        // could only keep in scratch (11.5 MB of scratch write-back per launch in round 1).
        int lane = lane0;
        asm volatile("" : "+v"(lane));
        const int col = lane & 15, grp = lane >> 4, cbs = lane & (BS - 1);
        int unit = 0;
        f32x4 head_s[CB], rgb_s[CB], head_d[CB], rgb_d[CB];
        // direction operand of net `n`, built when the engine reaches the view layer
        auto views_of = [&](const FusedNet &n, const float *cams) {
            return [&, cams](OpArr<1, NP> (&views)[CB]) {
                const float *dir = a.dir + 3 * (g >= 0 ? g / a.bpr : 0);
                float dv[4] = {0.f, 0.f, 0.f, 0.f};
                zest_view_dir(dir, n.w2cs ? cams + (PROJ ? 12 : 0) : nullptr, dv);
                encode_pe_operand<EP, 3, 4, 1>(dv, grp, views[0]);
#pragma unroll
                for (int cb = 1; cb < CB; cb++) views[cb] = views[0];          // one ray per block
            };
        };
        // ---- the pass's samples: coordinates, depths and spacings go to LDS (the point operands and the
        // compositing read them from there: no global load - and no vmcnt wait that would drain the
        // weight DMA - once the networks run); the feature gathers of BOTH nets are issued here, while
        // nothing else needs the registers, and the dynamic net's finished operand is parked in LDS
        OpArr<NT_FEAT_S / 2, NP> feat_s[CB];
#pragma unroll
        for (int cb = 0; cb < CB; cb++) {
            BlockSamples b;
            fetch(g, cb, col, b);
            if (grp == 0) {
                zd_lds[wave * 32 + 16 * cb + col] = make_float2(b.zz, b.valid ? b.dist : -1.0f);
                x_lds[wave * 32 + 16 * cb + col] = make_float4(b.x[0], b.x[1], b.x[2], 0.0f);
            }
            if constexpr (MOD_S) encode_feat_operand<EP, NT_FEAT_S / 2>(a.st, cams_s, b.x, b.pw, grp, b.valid, feat_s[cb]);
#if ZEST_GATHER_FENCE
            // one column block's taps at a time: both in flight together (~190 registers of tap data with
            // 8 source views) push the two-net kernels, which hold more state, into scratch
            if constexpr (DYN && NT_FEAT_S >= 4) __builtin_amdgcn_sched_barrier(0);
#endif
            if constexpr (DYN && MOD_D && ZEST_EARLY_DYN_GATHER) {
                OpArr<NT_FEAT_D / 2, NP> fd;
                encode_feat_operand<EP, NT_FEAT_D / 2>(a.dy, cams_d, b.x, b.pw, grp, b.valid, fd);
#pragma unroll
                for (int t = 0; t < NT_FEAT_D / 2; t++)
#pragma unroll
                    for (int pt = 0; pt < NP; pt++)
                        fd_lds[(((wave * CB + cb) * (NT_FEAT_D / 2) + t) * NP + pt) * 64 + lane] =
                            __builtin_bit_cast(uint4, fd.t[pt][t]);
            }
        }
        // point operand of a net from the parked coordinates (C = 3: xyz; 4: xyz + frame index)
        auto pts_static = [&](OpArr<2, NP> (&o)[CB], int token) __attribute__((always_inline)) {
#pragma unroll
            for (int cb = 0; cb < CB; cb++) {
                int at = wave * 32 + 16 * cb + col;
                asm volatile("" : "+v"(at) : "v"(token));          // not before `token` exists (see engine_forward)
                const float4 xv = x_lds[at];
                const float x4[4] = {xv.x, xv.y, xv.z, 0.0f};
#ifdef ZEST_EXPERIMENT_NO_ENCODE        // timing experiment only
#pragma unroll
                for (int t = 0; t < 2; t++) o[cb].t[0][t] = bf16x8{(short)lane, 1, 2, 3, 4, 5, 6, 7};
#else
                encode_pe_operand<EP, 3, 10, 2>(x4, grp, o[cb]);
#endif
            }
        };
        auto pts_dynamic = [&](OpArr<3, NP> (&o)[CB], int token) __attribute__((always_inline)) {
#pragma unroll
            for (int cb = 0; cb < CB; cb++) {
                int at = wave * 32 + 16 * cb + col;
                asm volatile("" : "+v"(at) : "v"(token));
                const float4 xv = x_lds[at];
                const float x4[4] = {xv.x, xv.y, xv.z, a.frame_idx};
                encode_pe_operand<EP, 4, 10, 3>(x4, grp, o[cb]);
            }
        };
        ZEST_STAMP(st_enc);
        if constexpr (ZEST_REBUILD_PTS) {
            engine_forward<EP, CB, 4, MOD_S, NT_FEAT_S, V2S, true>(tiles, unit, pts_static, feat_s,
                                                        views_of(a.st, cams_s), head_s, rgb_s);
        } else {
            OpArr<2, NP> pts_keep[CB];
            pts_static(pts_keep, 0);
            auto pts_copy = [&](OpArr<2, NP> (&o)[CB], int) __attribute__((always_inline)) {
#pragma unroll
                for (int cb = 0; cb < CB; cb++) o[cb] = pts_keep[cb];
            };
            engine_forward<EP, CB, 4, MOD_S, NT_FEAT_S, V2S, true>(tiles, unit, pts_copy, feat_s,
                                                        views_of(a.st, cams_s), head_s, rgb_s);
        }
        ZEST_STAMP(st_eng);
        // ---- per-block compositing on lanes 0 .. BS-1 (sample = lane): rgb rows 0-2, head rows 0, 1
        // sit in elements 0-2 / 0, 1 of lane group 0 of each column block
        float rec[kPartialFloats];
#pragma unroll
        for (int i = 0; i < kPartialFloats; i++) rec[i] = 0.f;
        float zz, dist, cr, cg, cb, al_s;
        bool valid;
        {
            const float2 zd = zd_lds[wave * 32 + cbs];
            zz = zd.x, dist = fmaxf(zd.y, 0.0f), valid = zd.y >= 0.0f;
            float sg = join(head_s, 0);
            cr = join(rgb_s, 0), cg = join(rgb_s, 1), cb = join(rgb_s, 2);
            if (V2S) {       // 'v2' nets activate inside the network; the compositor does it again
                cr = zest_sigmoid(cr), cg = zest_sigmoid(cg), cb = zest_sigmoid(cb), sg = fmaxf(sg, 0.f);
            }
            cr = zest_sigmoid(cr), cg = zest_sigmoid(cg), cb = zest_sigmoid(cb);
            al_s = valid ? 1.0f - expf(-fmaxf(sg, 0.f) * dist) : 0.0f;
            float tot;
            const float w = al_s * block_excl_prod<BS>(1.0f - al_s + 1e-10f, cbs, &tot);
            rec[0] = tot;
            rec[1] = block_sum<BS>(w * cr), rec[2] = block_sum<BS>(w * cg), rec[3] = block_sum<BS>(w * cb);
            rec[4] = block_sum<BS>(w * zz), rec[5] = block_sum<BS>(w);
        }
        if (DYN) {
            // the static net's per-sample results wait in LDS while the dynamic net has the registers
            const float blend_s = zest_sigmoid(join(head_s, 1));
            if (lane < BS) {
                st_lds[(wave * 3 + 0) * 32 + lane] = make_float2(cr, cg);
                st_lds[(wave * 3 + 1) * 32 + lane] = make_float2(cb, al_s);
                st_lds[(wave * 3 + 2) * 32 + lane] = make_float2(blend_s, 0.0f);
            }
            OpArr<NT_FEAT_D / 2, NP> feat_d[CB];
            if constexpr (MOD_D && ZEST_EARLY_DYN_GATHER) {
#pragma unroll
                for (int cbi = 0; cbi < CB; cbi++)
#pragma unroll
                    for (int t = 0; t < NT_FEAT_D / 2; t++)
#pragma unroll
                        for (int pt = 0; pt < NP; pt++)
                            feat_d[cbi].t[pt][t] = __builtin_bit_cast(
                                bf16x8, fd_lds[(((wave * CB + cbi) * (NT_FEAT_D / 2) + t) * NP + pt) * 64 + lane]);
            } else if constexpr (MOD_D) {
                // the dynamic net's gathers, now that the static net's registers are free (issued together
                // with the static ones they push the gather phase into scratch)
#pragma unroll
                for (int cbi = 0; cbi < CB; cbi++) {
                    BlockSamples b;
                    fetch(g, cbi, col, b);
                    encode_feat_operand<EP, NT_FEAT_D / 2>(a.dy, cams_d, b.x, b.pw, grp, b.valid, feat_d[cbi]);
                }
            }
            ZEST_STAMP(st_comp);
            if constexpr (ZEST_REBUILD_PTS) {
                engine_forward<EP, CB, 6, MOD_D, NT_FEAT_D, false, true>(tiles, unit, pts_dynamic, feat_d,
                                                            views_of(a.dy, cams_d), head_d, rgb_d);
            } else {
                OpArr<3, NP> pts_keep[CB];
                pts_dynamic(pts_keep, 0);
                auto pts_copy = [&](OpArr<3, NP> (&o)[CB], int) __attribute__((always_inline)) {
#pragma unroll
                    for (int cb = 0; cb < CB; cb++) o[cb] = pts_keep[cb];
                };
                engine_forward<EP, CB, 6, MOD_D, NT_FEAT_D, false, true>(tiles, unit, pts_copy, feat_d,
                                                            views_of(a.dy, cams_d), head_d, rgb_d);
            }
            ZEST_STAMP(st_eng);
            const float2 s0 = st_lds[(wave * 3 + 0) * 32 + cbs], s1 = st_lds[(wave * 3 + 1) * 32 + cbs],
                         s2 = st_lds[(wave * 3 + 2) * 32 + cbs];
            cr = s0.x, cg = s0.y, cb = s1.x, al_s = s1.y;
            const float blend = s2.x;
            const float er = zest_sigmoid(join(rgb_d, 0)), eg = zest_sigmoid(join(rgb_d, 1)),
                        eb = zest_sigmoid(join(rgb_d, 2));
            const float sg_d = join(head_d, 0);
            const float a_fg = valid ? 1.0f - expf(-fmaxf(sg_d, 0.f) * dist) : 0.0f;
            const float a_d = a_fg * blend, a_st = al_s * (1.0f - blend);
            float tot_b, tot_f;
            const float Tb = block_excl_prod<BS>((1.0f - a_d) * (1.0f - a_st) + 1e-10f, cbs, &tot_b);
            const float wf = a_fg * block_excl_prod<BS>(1.0f - a_fg + 1e-10f, cbs, &tot_f);
            const float wd = Tb * a_d, ws = Tb * a_st;
            rec[6] = tot_b;
            rec[7] = block_sum<BS>(wd * er + ws * cr), rec[8] = block_sum<BS>(wd * eg + ws * cg);
            rec[9] = block_sum<BS>(wd * eb + ws * cb), rec[10] = block_sum<BS>((wd + ws) * zz);
            rec[11] = block_sum<BS>(wd);
            rec[12] = tot_f;
            rec[13] = block_sum<BS>(wf * er), rec[14] = block_sum<BS>(wf * eg), rec[15] = block_sum<BS>(wf * eb);
            rec[16] = block_sum<BS>(wf * zz);
        }
        if (lane == 0 && g >= 0) {
            // records go to HBM for combine_kernel, or stay in LDS when the pass holds whole rays
            float4 *o = ranges ? reinterpret_cast<float4 *>(rec_lds + wave * kPartialFloats)
                               : reinterpret_cast<float4 *>(a.partials + (size_t)g * kPartialFloats);
#pragma unroll
            for (int i = 0; i < (DYN ? 5 : 2); i++)
                o[i] = make_float4(rec[4 * i], rec[4 * i + 1], rec[4 * i + 2], rec[4 * i + 3]);
        }
        }   // g >= 0
#ifdef ZEST_EXPERIMENT_NO_FINISH       // timing experiment only: no rendezvous, no ray finishing (results are wrong)
        if (false) {
#else
        if (ranges) {
#endif
            // The pass holds consecutive blocks of the workgroup's rays.  After one rendezvous the wave with a
            // ray's first block OF THIS PASS chains that ray's records of this pass (80 B each, in LDS) - onto
            // the sums carried over from the previous pass if the ray began there (then it is wave 0) - and
            // either writes the ray's maps or, if the ray continues, leaves the sums for the next pass.  The
            // carry is double-buffered by pass parity: the wave that reads the old one and the wave that writes
            // the new one may differ.  Blocks are chained in ray order, as fused_combine_kernel does.  The next pass
            // cannot reach this point before every wave has passed the ring's chunk barriers, i.e. has left it.
            __syncthreads();
            const int rb = g >= 0 ? g % a.bpr : 0;                       // block within its ray
            if (lane0 == 0 && g >= 0 && (wave == 0 || rb == 0)) {
                RayState st;
                if (rb != 0) st.load(carry_lds + ((k + 1) & 1) * kCarryFloats);
                const int n = min(a.bpr - rb, kFusedWaves - wave);     // this ray's blocks in this pass
                for (int b = 0; b < n; b++) st.chain(rec_lds + (wave + b) * kPartialFloats, DYN);
                if (rb + n == a.bpr) st.emit(a.white_bkgd, a.out + (size_t)(g / a.bpr) * 16);
                else st.save(carry_lds + (k & 1) * kCarryFloats);
            }
        }
        ZEST_STAMP(st_comp);
        tiles.next_pass();
    }
    tiles.drain();
#ifdef ZEST_RING_FLAGS
    if (tiles.poisoned && lane0 == 0) {
        a.partials[0] = __builtin_nanf("");   // make a protocol failure visible
    }
#endif
#ifdef ZEST_STAMPS
    if (a.stamps && lane0 == 0) {
        unsigned long long *o = a.stamps + (size_t)(blockIdx.x * kFusedWaves + wave) * 8;
        o[0] = __builtin_amdgcn_s_memtime() - st_t0, o[1] = st_enc, o[2] = st_eng, o[3] = st_comp;
        o[4] = tiles.t_wait, o[5] = tiles.t_issue, o[6] = __builtin_amdgcn_s_memrealtime() - st_r0,
        o[7] = st_n | (tiles.t_vm << 16);
    }
#endif
}

// one translation unit per variant (fused_*.hip) so they compile in parallel
#define ZEST_FUSED_VARIANT(ptag, EP, tag, NTS, DYN, NTD, V2S) ZEST_FUSED_VARIANT_(ptag, EP, tag, NTS, DYN, NTD, V2S)
#define ZEST_FUSED_VARIANT_(ptag, EP, tag, NTS, DYN, NTD, V2S)                                   \
    int fused_launch_##ptag##_##tag(const FusedArgs &a, int blocks, hipStream_t stream) {        \
        hipLaunchKernelGGL((fused_blocks_kernel<EP, NTS, DYN, NTD, V2S>), dim3(blocks),          \
                           dim3(kFusedWaves * 64), 0, stream, a);                                \
        ZEST_RETURN_LAUNCH("zest_render_fused_fwd(" #ptag "_" #tag ")");                         \
    }                                                                                            \
    int fused_units_##ptag##_##tag(int which) {                                                  \
        return which == 0 ? stream_units(4, NTS, ep_parts(EP)) : (DYN ? stream_units(6, NTD, ep_parts(EP)) : 0); \
    }

#define ZEST_FUSED_DECL1(ptag, tag)                                                     \
    int fused_launch_##ptag##_##tag(const FusedArgs &a, int blocks, hipStream_t stream); \
    int fused_units_##ptag##_##tag(int which);
#define ZEST_FUSED_DECL(tag) ZEST_FUSED_DECL1(bf16, tag) ZEST_FUSED_DECL1(f16, tag) ZEST_FUSED_DECL1(x3, tag)
ZEST_FUSED_DECL(s0)
ZEST_FUSED_DECL(s2)
ZEST_FUSED_DECL(s4)
ZEST_FUSED_DECL(s0d0)
ZEST_FUSED_DECL(s2d0)
ZEST_FUSED_DECL(s4d0)
ZEST_FUSED_DECL(s2d2)
ZEST_FUSED_DECL(s4d2)
ZEST_FUSED_DECL(s2v)
ZEST_FUSED_DECL(s4v)

}  // namespace zest
