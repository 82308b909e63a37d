// bf16 training path of the width-256 MLP on the register engine (consumer:
// reference train.py:587-760 -> backward through Renderer.forward, networks.py:150-221).
//
// Forward = the inference engine kernel with an activation stash (mlp_engine.hip, TRAIN).  Backward =
// three kernels, none of which materialises a [samples x 256] fp32 tensor:
//   data kernel     the backward walk on the SAME register engine with the transposed weight stream
//                   (mlp_plan.hip build_bwd_plan): a wave carries the gradient of its 32 samples through
//                   rgb -> view layer -> feature_linear | heads -> trunk 7 .. 0 in registers; every
//                   transposed GEMM's epilogue applies the ReLU mask (one bit per element, stashed by the
//                   forward) and the modulation m = pts_bias(feats) (recomputed per row block with the
//                   same 2-4 MFMA pairs as the forward) and leaves d(pre-activation) as the next
//                   operand; those tiles also go to the gradient stash; the point-encoding rows of
//                   layers 5 and 0 leave as float atomics into g_x.
//   modulation      d m = sum_l d pre_l * h_l / m^2 per sample from the two stashes, d features = Wm^T d m.
//   weight kernel   d W_op = sum over samples of d pre_op (x) input_op, d b_op = sum d pre_op: stash tiles
//                   hold samples on the lane axis, the contraction runs over samples, so the tiles are
//                   staged in LDS and read back with the transposing LDS read (ds_read_b64_tr_b16) as
//                   MFMA operands with k = sample; a wave keeps a 32 x 256 slice of d W in registers over
//                   all the blocks of its workgroup and adds it to the fp32 gradient once.
#include <string.h>
#include <map>
#include <mutex>
#include <tuple>
#include <vector>
#include "mlp_operands.cuh"
#include "mlp_train16.h"

namespace zest {
size_t bwd_stream_bytes(const zest_mlp_desc &d);
int bwd_stream_units_of(const zest_mlp_desc &d);
int pack_bwd_stream(const zest_mlp_desc &d, const float *const *params, void *packed, hipStream_t stream);
size_t train16_dw_partial_bytes(int cus);                       // mlp_train16_dw.hip
int train16_dw_launch(const DwJob *jobs, int n_jobs, int n_wg, const uint4 *stash_tiles, const uint4 *grad, int M,
                      float4 *partial, float *const *g_params, hipStream_t st);
int mlp_engine_train_launch(const MlpPlan &p, const void *tiles, const float *x, int M, float *out, void *stash_tiles,
                            void *stash_masks, hipStream_t stream);
}  // namespace zest

namespace {

using namespace zest;
constexpr int EP = ZEST_PREC_BF16;
constexpr int kWaves = 8;

constexpr int kPtsSideFloat4 = 2 * 3 * 2 * 2 * 64;      // point-gradient side buffer per block: [layer 5 | 0][row block < 3][CB][row tile][lane]

struct Sizes {
    long long blocks;
    size_t tile_bytes, mask_bytes, grad_bytes, side_bytes;
};
constexpr size_t kPtrTableBytes = 256;          // reserved (rounds 1-2 kept a device copy of the gradient pointers here)
Sizes sizes_of(int M) {
    Sizes s;
    s.blocks = train_blocks(M);
    s.tile_bytes = (size_t)s.blocks * kStashTiles * 2 * 1024;
    s.mask_bytes = (size_t)s.blocks * kStashMasks * 2 * 64 * 8;
    s.grad_bytes = (size_t)s.blocks * kGradTiles * 2 * 1024;
    s.side_bytes = (size_t)s.blocks * kPtsSideFloat4 * 16;
    return s;
}

// a stash tile, read once by the kernel: streaming load (finishing kernel 323 -> 283 us, weight kernel 371 -> 353 us
// against plain loads, -DZEST_STASH_CACHED)
__device__ __forceinline__ uint4 stash_load(const uint4 *p) {
#ifndef ZEST_STASH_CACHED
    return __builtin_bit_cast(uint4, __builtin_nontemporal_load(reinterpret_cast<const v4u *>(p)));
#else
    return *p;
#endif
}

__device__ __forceinline__ void unpack8(const uint4 q, float (&v)[8]) {
    const unsigned w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
    for (int i = 0; i < 4; i++) v[2 * i] = __uint_as_float(w[i] << 16), v[2 * i + 1] = __uint_as_float(w[i] & 0xFFFF0000u);
}
__device__ __forceinline__ uint4 pack8(const float (&v)[8]) {
    return make_uint4(pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3]), pack_bf16(v[4], v[5]), pack_bf16(v[6], v[7]));
}

// ------------------------------------------------------------------------------ data kernel
// One transposed Linear: NJB row blocks of 32 INPUT indices over the operand [A (NKA k-tiles) | B (NKB)] of
// output-feature positions.  MOD: the row block also computes m (header modulation bias + NKF feature
// k-tiles).  MODE 0: epi.tile(jb, cb, v, m) finishes the 8 values (mask, modulation), which become k-tile jb
// of `out`;  MODE 2: epi.rows(jb, cb, v) takes the raw sums (point-encoding gradients).
template <int CB, int NJB, int NKA, int NKB, bool MOD, int NKF, int MODE, class Tiles, class Epi>
__device__ __forceinline__ void bwd_layer(const Tiles &tiles, int &unit, const OpArr<NKA> (&opa)[CB],
                                          const OpArr<NKB> (&opb)[CB], const OpArr<NKF> (&opf)[CB],
                                          OpArr<8> (&out)[CB], const Epi &epi) {
    constexpr int NM = MOD ? 2 * NKF : 0, T = NM + 2 * (NKA + NKB);
    struct Pre {
        f32x4 mbias[2];
        bf16x8 win[kPrefetch];
    };
    auto preload = [&](Pre &p, int u0) {
        tiles.touch(u0);                                         // the header unit: walked exactly once
#pragma unroll
        for (int rt = 0; rt < 2; rt++)
            if (MOD) p.mbias[rt] = tiles.load_bias(u0, 1, rt);
#pragma unroll
        for (int k = 0; k < kPrefetch; k++)
            if (k < T) p.win[k] = tiles.load(u0 + 1 + k);
    };
    Pre pre;
    preload(pre, unit);
#pragma unroll
    for (int jb = 0; jb < NJB; jb++) {
        const int u0 = unit;
        f32x4 acc[2][CB], macc[2][CB];
        bf16x8 win[kPrefetch];
#pragma unroll
        for (int k = 0; k < kPrefetch; k++) win[k] = pre.win[k];
#pragma unroll
        for (int rt = 0; rt < 2; rt++)
#pragma unroll
            for (int cb = 0; cb < CB; cb++) {
                acc[rt][cb] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (MOD) macc[rt][cb] = pre.mbias[rt];
            }
#pragma unroll
        for (int k = 0; k < T; k++) {
            const bf16x8 a = win[k % kPrefetch];
            const int rt = k % 2, kt = (k < NM ? k : k - NM) / 2;
#pragma unroll
            for (int cb = 0; cb < CB; cb++) {
                if (k < NM)
                    macc[rt][cb] = mfma16<EP>(a, opf[cb].t[0][k < NM ? kt : 0], macc[rt][cb]);
                else if (kt < NKA)
                    acc[rt][cb] = mfma16<EP>(a, opa[cb].t[0][(k >= NM && kt < NKA) ? kt : 0], acc[rt][cb]);
                else
                    acc[rt][cb] = mfma16<EP>(a, opb[cb].t[0][(k >= NM && kt >= NKA) ? kt - NKA : 0], acc[rt][cb]);
            }
            if (k + kPrefetch < T) win[k % kPrefetch] = tiles.load(u0 + 1 + k + kPrefetch);
        }
        unit = u0 + 1 + T;
        if (jb + 1 < NJB) preload(pre, unit);
#pragma unroll
        for (int cb = 0; cb < CB; cb++) {
            float v[8], m[8];
#pragma unroll
            for (int i = 0; i < 8; i++) v[i] = acc[i >> 2][cb][i & 3], m[i] = MOD ? macc[i >> 2][cb][i & 3] : 1.0f;
            if (MODE == 0) {
                epi.tile(jb, cb, v, m);
                store_tile<EP>(v, out[cb], jb);
            } else {
                epi.rows(jb, cb, v);
            }
        }
    }
}

// epilogue of a transposed GEMM whose result is d h of a masked (and modulated) layer
template <int CB>
struct MaskEpi {
    uint4 *grad;            // gradient stash tiles of this block: [kGradTiles][CB][64]
    int tile0, lane;
    unsigned lo[CB], hi[CB];
    bool masked;
    __device__ __forceinline__ void tile(int jb, int cb, float (&v)[8], const float (&m)[8]) const {
        const unsigned bits = masked ? ((jb < 4 ? lo[cb] >> (8 * jb) : hi[cb] >> (8 * (jb - 4))) & 0xFFu) : 0xFFu;
#pragma unroll
        for (int i = 0; i < 8; i++) v[i] = ((bits >> i) & 1u) ? v[i] * m[i] : 0.0f;
        // streaming store (the gradient stash is read back by the finishing and weight kernels only: data kernel 296 -> 248 us)
#ifndef ZEST_STASH_CACHED              // (defined: plain stores, for A/B timing)
        __builtin_nontemporal_store(__builtin_bit_cast(v4u, pack8(v)), reinterpret_cast<v4u *>(&grad[((tile0 + jb) * CB + cb) * 64 + lane]));
#else
        grad[((tile0 + jb) * CB + cb) * 64 + lane] = pack8(v);
#endif
    }
    __device__ __forceinline__ void rows(int, int, const float (&)[8]) const {}
};

// point-encoding rows (layers 5 and 0): the raw sums of a row block go to a side buffer exactly as the lanes
// hold them (two float4 per lane: rows 4g..4g+3 of row tiles 0 / 1; 1 KiB per wave store, coalesced); the
// finishing kernel adds the two layers' contributions and scatters them into the columns of g_x.  (Float
// atomics straight into g_x - every lane another row - ran at 1/17 of the atomic rate and dominated the kernel.)
template <int CB>
struct PtsEpi {
    float4 *side;           // this block's side buffer
    int which, lane;        // 0: layer 5, 1: layer 0
    __device__ __forceinline__ void tile(int, int, float (&)[8], const float (&)[8]) const {}
    __device__ __forceinline__ void rows(int jb, int cb, const float (&v)[8]) const {
        float4 *o = side + ((((which * 3 + jb) * CB + cb) * 2) * 64 + lane);
        o[0] = make_float4(v[0], v[1], v[2], v[3]);
        o[64] = make_float4(v[4], v[5], v[6], v[7]);
    }
};

template <int NT_PTS, bool MOD, int NT_FEAT>
__global__ __launch_bounds__(kWaves * 64, kWaves / 4) void train16_data_kernel(
    const uint4 *__restrict__ stream, const float *__restrict__ x, const float *__restrict__ out,
    const float *__restrict__ g_out, int M, int P, int F, int C_in, int C_out, int head,
    const uint2 *__restrict__ masks, uint4 *__restrict__ grad, float4 *__restrict__ pts_side) {
    constexpr int CB = 2, KP = NT_PTS / 2, KF = NT_FEAT / 2;
    constexpr int UNITS = bwd_stream_units(NT_PTS, MOD ? NT_FEAT : 0);
    using Ring = RingTiles<kWaves, UNITS, 0>;
    __shared__ __attribute__((aligned(16))) char lds[kRingUnits * 1024 + 2 * kSlots * 4];
    const int lane = threadIdx.x & 63, col = lane & 15, grp = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const Ring tiles{lds, (gptr_u4)stream, (gptr_u4)stream, lane, grp, wave,
                     (unsigned)(wave * Ring::kPieces * 64 + lane) * 16u,
                     (unsigned)(uintptr_t)(__attribute__((address_space(3))) char *)lds + (unsigned)wave * Ring::kPieces * 1024u};
    tiles.init_addr();
    tiles.prologue();
    const long long n_blocks = ((long long)M + 31) / 32, n_pass = (n_blocks + kWaves - 1) / kWaves;
    for (long long pass = blockIdx.x; pass < n_pass; pass += gridDim.x) {
        const long long block = pass * kWaves + wave;
        const long long m_base = block * 32;
        uint4 *gblk = grad + (size_t)block * kGradTiles * CB * 64;
        const uint2 *mblk = masks + (size_t)block * kStashMasks * CB * 64;
        OpArr<1> d_rgb[CB], d_head[CB];
        OpArr<KF> feat[CB];
        OpArr<0> none[CB];
        // ---- gradients of the raw network outputs from g_out (activation derivatives through `out`)
#pragma unroll
        for (int cb = 0; cb < CB; cb++) {
            const long long m = m_base + 16 * cb + col;
            const bool valid = m < M;
            const float *go = g_out + (size_t)(valid ? m : 0) * C_out, *o = out + (size_t)(valid ? m : 0) * C_out;
            float r[8] = {0, 0, 0, 0, 0, 0, 0, 0}, h[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            if (valid) {
                if (grp == 0) {
                    r[0] = go[0], r[1] = go[1], r[2] = go[2];                   // 'v0': rgb and alpha are raw
                    h[0] = go[3];
                    if (head == ZEST_HEAD_BLEND) h[1] = go[4] * o[4] * (1.0f - o[4]);
                    if (head == ZEST_HEAD_DYNAMIC)
                        for (int i = 1; i < 4; i++) h[i] = go[3 + i] * (1.0f - o[3 + i] * o[3 + i]);     // tanh rows 1-3
                } else if (head == ZEST_HEAD_DYNAMIC && grp == 1) {
                    for (int i = 0; i < 3; i++) h[i] = go[7 + i] * (1.0f - o[7 + i] * o[7 + i]);         // tanh rows 4-6
                    h[3] = go[10] * o[10] * (1.0f - o[10]);                                              // sigmoid row 7
                } else if (head == ZEST_HEAD_DYNAMIC && grp == 2) {
                    h[0] = go[11] * o[11] * (1.0f - o[11]);                                              // sigmoid row 8
                }
            }
            store_tile<EP>(r, d_rgb[cb], 0);
            store_tile<EP>(h, d_head[cb], 0);
            gblk[(76 * CB + cb) * 64 + lane] = __builtin_bit_cast(uint4, d_head[cb].t[0][0]);
            gblk[(77 * CB + cb) * 64 + lane] = __builtin_bit_cast(uint4, d_rgb[cb].t[0][0]);
            if (MOD) load_feat_operand<EP, KF>(x + (size_t)(valid ? m : 0) * C_in + P, F, valid, grp, feat[cb]);
        }
        auto mask_epi = [&](int mask_id, int tile0) {
            MaskEpi<CB> e;
            e.grad = gblk, e.tile0 = tile0, e.lane = lane, e.masked = mask_id >= 0;
#pragma unroll
            for (int cb = 0; cb < CB; cb++) {
                const uint2 w = mask_id >= 0 ? mblk[(mask_id * CB + cb) * 64 + lane] : make_uint2(0, 0);
                e.lo[cb] = w.x, e.hi[cb] = w.y;
            }
            return e;
        };
        int unit = 0;
        OpArr<8> ga[CB], gb[CB];
        // 1. rgb^T -> d view-layer output, masked by the view layer's ReLU (4 k-tiles)
        bwd_layer<CB, 4, 1, 0, false, KF, 0>(tiles, unit, d_rgb, none, feat, ga, mask_epi(8, 72));
        OpArr<4> g_hv[CB];
#pragma unroll
        for (int cb = 0; cb < CB; cb++)
#pragma unroll
            for (int k = 0; k < 4; k++) g_hv[cb].t[0][k] = ga[cb].t[0][k];
        // 2. view layer^T -> d feature_linear output (no activation)
        bwd_layer<CB, 8, 4, 0, false, KF, 0>(tiles, unit, g_hv, none, feat, gb, mask_epi(-1, 64));
        // 3. feature_linear^T | heads^T -> d h7, masked and modulated as layer 7 -> d pre_7
        bwd_layer<CB, 8, 8, 1, MOD, KF, 0>(tiles, unit, gb, d_head, feat, ga, mask_epi(7, 56));
        // 4. trunk layers 7 .. 1 (layer 5: the point rows first)
        PtsEpi<CB> pe5{pts_side + (size_t)block * kPtsSideFloat4, 0, lane}, pe0{pts_side + (size_t)block * kPtsSideFloat4, 1, lane};
        bwd_layer<CB, 8, 8, 0, MOD, KF, 0>(tiles, unit, ga, none, feat, gb, mask_epi(6, 48));     // layer 7 -> d pre_6
        bwd_layer<CB, 8, 8, 0, MOD, KF, 0>(tiles, unit, gb, none, feat, ga, mask_epi(5, 40));     // layer 6 -> d pre_5
        bwd_layer<CB, KP, 8, 0, false, KF, 2>(tiles, unit, ga, none, feat, gb, pe5);               // layer 5: points
        bwd_layer<CB, 8, 8, 0, MOD, KF, 0>(tiles, unit, ga, none, feat, gb, mask_epi(4, 32));     // layer 5 -> d pre_4
        bwd_layer<CB, 8, 8, 0, MOD, KF, 0>(tiles, unit, gb, none, feat, ga, mask_epi(3, 24));     // layer 4 -> d pre_3
        bwd_layer<CB, 8, 8, 0, MOD, KF, 0>(tiles, unit, ga, none, feat, gb, mask_epi(2, 16));     // layer 3 -> d pre_2
        bwd_layer<CB, 8, 8, 0, MOD, KF, 0>(tiles, unit, gb, none, feat, ga, mask_epi(1, 8));      // layer 2 -> d pre_1
        bwd_layer<CB, 8, 8, 0, MOD, KF, 0>(tiles, unit, ga, none, feat, gb, mask_epi(0, 0));      // layer 1 -> d pre_0
        bwd_layer<CB, KP, 8, 0, false, KF, 2>(tiles, unit, gb, none, feat, ga, pe0);               // layer 0: points
        tiles.finish(unit, UNITS);
        tiles.next_pass();
    }
    tiles.drain();
}

// Input column of accumulator row (row tile rt, lane group grp, element r) of row block jb of a transposed encoder
// operand = the plan's position map (mlp_plan.hip pe_map_acc / feat_map_acc) at position 32 jb + 16 rt + 4 grp + r,
// in closed form: everything but the lane group is a compile-time constant after unrolling.  (Looked up in the
// device copies of the maps, every element was a 2-byte global load with a store waiting on it: 48 + 16 exposed
// round trips per block were most of the finishing kernel's time.)
template <int C, int L>
__device__ __forceinline__ int pe_col_of(int jb, int rt, int r, int grp) {
    const int g = 2 * rt + (grp >> 1), mm = 8 * jb + 4 * (grp & 1) + r;
    if (mm < (L / 2) * C) return C + 2 * C * (2 * (mm / C) + (g >> 1)) + ((g & 1) ? C : 0) + mm % C;
    return (mm == (L / 2) * C && g < C) ? g : -1;
}
__device__ __forceinline__ int feat_col_of(int jb, int rt, int r, int grp, int V) {
    // accumulator row 16 rt + 4 grp + r of row block jb = operand position 32 jb + 16 rt + 4 grp + r = quad
    // 8 jb + 4 rt + grp, channel r (mlp_plan.h feat_quad_col)
    const int col = feat_quad_col(8 * jb + 4 * rt + grp, V);
    return col >= 0 ? col + r : -1;
}

// ------------------------------------------------------------------------------ finishing kernel
// Per block of 32 samples, one wave, everything straight from global memory / L2 (no ring):
//   * point-encoding gradient: the two side-buffer contributions (layers 5 and 0) are added and scattered
//     into the point columns of g_x through the position map;
//   * modulation (nets with features): m is recomputed with the forward stream's modulation tiles (MFMA),
//     d m = sum over the 8 trunk layers of d pre_l * h_l / m^2 from the two stashes (d pre_l = d h_l mask m and
//     h_l = pre_l m, so the product is d h_l mask pre_l) becomes gradient tiles 78 .. 85 for the weight kernel,
//     and d features = Wm^T d m (MFMA with the backward stream's tail) goes to the feature columns of g_x.
template <int NT_PTS, int NT_FEAT>
#ifdef ZEST_FIN_WAVES                  // occupancy experiment: cap the registers for this many waves per SIMD
__attribute__((amdgpu_waves_per_eu(ZEST_FIN_WAVES, ZEST_FIN_WAVES)))
#endif
__global__ __launch_bounds__(256) void train16_finish_kernel(
    const float *__restrict__ x, int M, int P, int F, int C_in, const uint4 *__restrict__ bwd_tail,
    const uint4 *__restrict__ stash, uint4 *__restrict__ grad,
    const float4 *__restrict__ pts_side, const short *__restrict__ map_pts, const short *__restrict__ map_feat,
    float *__restrict__ g_x) {
    constexpr int CB = 2, KP = NT_PTS / 2, KF = NT_FEAT / 2;
    const int lane = threadIdx.x & 63, col = lane & 15, grp = lane >> 4;
    const long long block = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (block * 32 >= M) return;                                   // wave-uniform
    // The point and feature columns of the block's 32 rows of g_x are assembled in LDS (row = sample) and written
    // out as contiguous 256-byte runs: straight from the accumulators every lane writes another row, 64 cache lines
    // per store instruction (12 % of the kernel).
    constexpr int kGxStride = 113;                                 // floats per sample row in LDS (>= 84 + 24, odd)
    __shared__ float gx_lds[4][32 * kGxStride];
    float *gx_t = gx_lds[threadIdx.x >> 6];
    float *gx_row[CB];
#pragma unroll
    for (int cb = 0; cb < CB; cb++) gx_row[cb] = gx_t + (16 * cb + col) * kGxStride;
    // ---- point columns
    const float4 *side = pts_side + (size_t)block * kPtsSideFloat4;
#pragma unroll
    for (int jb = 0; jb < KP; jb++)
#pragma unroll
        for (int cb = 0; cb < CB; cb++)
#pragma unroll
            for (int rt = 0; rt < 2; rt++) {
                const float4 a = side[(((0 * 3 + jb) * CB + cb) * 2 + rt) * 64 + lane];
                const float4 b = side[(((1 * 3 + jb) * CB + cb) * 2 + rt) * 64 + lane];
                const float v[4] = {a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w};
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int c = pe_col_of<NT_PTS == 4 ? 3 : 4, 10>(jb, rt, r, grp);
#ifdef ZEST_FIN_EXP_NO_GX               // timing experiment only
                    if (c >= 0 && v[r] == 123456.0f) gx_row[cb][c] = v[r];
#else
                    if (c >= 0) gx_row[cb][c] = v[r];
#endif
                }
            }
    if constexpr (KF > 0) {
        const uint4 *sblk = stash + (size_t)block * kStashTiles * CB * 64;
        uint4 *gblk = grad + (size_t)block * kGradTiles * CB * 64;
        OpArr<KF> feat[CB];           // the feature operand as the forward pass built it (stash tiles kStashFeat ..)
#pragma unroll
        for (int cb = 0; cb < CB; cb++)
#pragma unroll
            for (int k = 0; k < KF; k++)
                feat[cb].t[0][k] = __builtin_bit_cast(bf16x8, sblk[((kStashFeat + k) * CB + cb) * 64 + lane]);
        OpArr<8> dm[CB];
#pragma unroll
        for (int jb = 0; jb < 8; jb++) {
            // m of the row block: header (modulation bias block at byte 128) + NT_FEAT modulation tiles (tail, 2nd part)
            const uint4 *u = bwd_tail + (size_t)(KF * 17 + jb * (1 + NT_FEAT)) * 64;
            f32x4 macc[2][CB];
#pragma unroll
            for (int rt = 0; rt < 2; rt++) {
                const float4 b4 = reinterpret_cast<const float4 *>(u)[8 + 4 * rt + grp];     // floats 32 + 16 rt + 4 g ..
#pragma unroll
                for (int cb = 0; cb < CB; cb++) macc[rt][cb] = f32x4{b4.x, b4.y, b4.z, b4.w};
            }
#pragma unroll
            for (int k = 0; k < NT_FEAT; k++) {
                const bf16x8 a = __builtin_bit_cast(bf16x8, u[(1 + k) * 64 + lane]);
#pragma unroll
                for (int cb = 0; cb < CB; cb++) macc[k % 2][cb] = mfma16<EP>(a, feat[cb].t[0][k / 2], macc[k % 2][cb]);
            }
#pragma unroll
            for (int cb = 0; cb < CB; cb++) {
                float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
                for (int l = 0; l < 8; l++) {
                    float dp[8], hv[8];
                    unpack8(stash_load(&gblk[((8 * l + jb) * CB + cb) * 64 + lane]), dp);
                    unpack8(stash_load(&sblk[((8 * l + jb) * CB + cb) * 64 + lane]), hv);
#pragma unroll
                    for (int e = 0; e < 8; e++) acc[e] = fmaf(dp[e], hv[e], acc[e]);
                }
#pragma unroll
                for (int e = 0; e < 8; e++) {
                    const float mv = macc[e >> 2][cb][e & 3];
                    acc[e] = acc[e] != 0.0f ? acc[e] / (mv * mv) : 0.0f;
                }
                store_tile<EP>(acc, dm[cb], jb);
                gblk[((78 + jb) * CB + cb) * 64 + lane] = __builtin_bit_cast(uint4, dm[cb].t[0][jb]);
            }
        }
        // d features = Wm^T d m: row blocks = feature positions
#pragma unroll
        for (int jb = 0; jb < KF; jb++) {
            const uint4 *u = bwd_tail + (size_t)(jb * 17) * 64;
            f32x4 acc[2][CB];
#pragma unroll
            for (int rt = 0; rt < 2; rt++)
#pragma unroll
                for (int cb = 0; cb < CB; cb++) acc[rt][cb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int k = 0; k < 16; k++) {
                const bf16x8 a = __builtin_bit_cast(bf16x8, u[(1 + k) * 64 + lane]);
#pragma unroll
                for (int cb = 0; cb < CB; cb++) acc[k % 2][cb] = mfma16<EP>(a, dm[cb].t[0][k / 2], acc[k % 2][cb]);
            }
#pragma unroll
            for (int cb = 0; cb < CB; cb++) {
#pragma unroll
                for (int rt = 0; rt < 2; rt++)
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const int c = feat_col_of(jb, rt, r, grp, (F - 8) / 4);
#ifdef ZEST_FIN_EXP_NO_GX               // timing experiment only
                        if (c >= 0 && acc[rt][cb][r] == 123456.0f) gx_row[cb][P + c] = acc[rt][cb][r];
#else
                        if (c >= 0) gx_row[cb][P + c] = acc[rt][cb][r];
#endif
                    }
            }
        }
    }
    // ---- rows out (the LDS operations of a wave execute in order: no barrier between its writes and reads)
    const int ncol = P + (KF > 0 ? F : 0);
    for (int r = 0; r < 32; r++) {
        const long long m = block * 32 + r;
        if (m >= M) break;                                         // wave-uniform
        for (int c = lane; c < ncol; c += 64) g_x[(size_t)m * C_in + c] = gx_t[r * kGxStride + c];
    }
}

// ------------------------------------------------------------------------------ host side
struct TrainTables {                 // per MLP shape, device resident
    DwJob *jobs = nullptr;
    int n_jobs = 0, n_wg = 0;        // weight kernel: jobs and the workgroups they share (one per CU)
    short *map_pts = nullptr, *map_feat = nullptr;
    MlpPlan fwd;                     // forward plan (stream size check, operand tile counts)
};
std::mutex g_mu;
std::map<std::tuple<int, int, int, int, int, int, int, int, int, int>, TrainTables *> g_tables;    // per shape AND device

TrainTables *tables_for(const zest_mlp_desc &d) {
    // the kernels are unrolled for the shipped shape (8 x 256, skip after layer 4): refused here for every entry
    // point, whatever an earlier call has cached
    const char *err = nullptr;
    zest::MlpShape sh;
    if (!zest::mlp_shape(d, &sh, &err) || !sh.is_default) {
        zest_set_error("zest_mlp_train16: %s", err ? err : "the bf16 training kernels cover depth 8 / width 256 / skips [4] only");
        return nullptr;
    }
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    std::lock_guard<std::mutex> lk(g_mu);
    auto key = std::make_tuple(d.in_ch_pts, d.use_feat ? d.in_ch_feat : 0, d.in_ch_views, d.use_feat, d.net_type, d.head,
                               d.depth, d.width, d.skip_mask, dev);      // the tables live in the memory of `dev`
    auto it = g_tables.find(key);
    if (it != g_tables.end()) return it->second;
    TrainTables *t = new TrainTables();
    std::vector<DwJob> jobs;
    if (d.net_type != 0 || !build_plan(d, ZEST_PREC_BF16, ORDER_ACC, &t->fwd, &err, true) ||
        build_dw_jobs(d, &jobs, &err) < 0) {
        zest_set_error("zest_mlp_train16: shape not supported: %s", err ? err : "the bf16 training path covers 'v0' nets");
        delete t;
        return nullptr;
    }
    t->n_jobs = (int)jobs.size();
    {   // Workgroups per job: an even split of the CUs.  The time of a job is set by its number of blocks per
        // workgroup, hardly by the tiles it streams per block (an iteration costs about the same whether it stages
        // 5 tiles or 16), so a split in proportion to the tiles leaves the light jobs (rgb: 5 tiles, 7 workgroups)
        // running 1.6x longer than the even one (measured: 990 us against 600).
        int cus = 256;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 256;
        const int nj = t->n_jobs;
        if (cus < nj) cus = nj;
        std::vector<int> wgs(nj);
        for (int j = 0; j < nj; j++) wgs[j] = cus / nj + (j < cus % nj ? 1 : 0);
        for (int j = 0, w0 = 0; j < nj; j++) jobs[j].wg0 = w0, jobs[j].n_wg = wgs[j], w0 += wgs[j];
        t->n_wg = cus;
    }
    std::vector<short> mp(96, -1), mf(64, -1);
    for (size_t i = 0; i < t->fwd.map_pts.size() && i < 96; i++) mp[i] = t->fwd.map_pts[i];
    for (size_t i = 0; i < t->fwd.map_feat.size() && i < 64; i++) mf[i] = t->fwd.map_feat[i];
    hipError_t e = hipMalloc(&t->jobs, jobs.size() * sizeof(DwJob));
    if (e == hipSuccess) e = hipMemcpy(t->jobs, jobs.data(), jobs.size() * sizeof(DwJob), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMalloc(&t->map_pts, 96 * sizeof(short));
    if (e == hipSuccess) e = hipMemcpy(t->map_pts, mp.data(), 96 * sizeof(short), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMalloc(&t->map_feat, 64 * sizeof(short));
    if (e == hipSuccess) e = hipMemcpy(t->map_feat, mf.data(), 64 * sizeof(short), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        zest_set_error("zest_mlp_train16: uploading tables: %s", hipGetErrorString(e));
        delete t;
        return nullptr;
    }
    g_tables[key] = t;
    return t;
}

int cu_count() {
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
        cus = 256;
    return cus;
}
// partial sums of the weight kernel: one slice per workgroup (one workgroup per CU, at least one per job: 16 jobs at most)
size_t dw_partial_bytes() { return zest::train16_dw_partial_bytes(cu_count()); }

}  // namespace

// 0 + error text for a shape the bf16 training kernels do not cover
static bool train16_shape_ok(const zest_mlp_desc *desc) {
    const char *err = nullptr;
    zest::MlpShape sh;
    if (desc && zest::mlp_shape(*desc, &sh, &err) && sh.is_default && desc->net_type == 0) return true;
    zest_set_error("zest_mlp_train16: %s", !desc ? "null descriptor" : err ? err :
                   "the bf16 training kernels cover 'v0' nets of depth 8 / width 256 / skips [4] only");
    return false;
}

extern "C" size_t zest_mlp_train16_stash_bytes(const zest_mlp_desc *desc, int M) {
    if (!train16_shape_ok(desc) || M <= 0) return 0;
    const Sizes s = sizes_of(M);
    return s.tile_bytes + s.mask_bytes;
}
extern "C" size_t zest_mlp_train16_work_bytes(const zest_mlp_desc *desc, int M) {
    if (!train16_shape_ok(desc) || M <= 0) return 0;
    const Sizes s = sizes_of(M);
    return s.grad_bytes + s.side_bytes + kPtrTableBytes + dw_partial_bytes();
}
extern "C" size_t zest_mlp_train16_packed_bytes(const zest_mlp_desc *desc) { return train16_shape_ok(desc) ? zest::bwd_stream_bytes(*desc) : 0; }

extern "C" int zest_mlp_train16_pack(const zest_mlp_desc *desc, const float *const *params, void *packed, void *stream) {
    ZEST_CHECK_ARG(desc && params && packed && ((uintptr_t)packed & 15) == 0, "zest_mlp_train16_pack: bad argument");
    if (!train16_shape_ok(desc)) return (int)hipErrorInvalidValue;
    return zest::pack_bwd_stream(*desc, params, packed, (hipStream_t)stream);
}

extern "C" int zest_mlp_train16_fwd(const zest_mlp_desc *desc, const void *packed_fwd, const float *x, int M, void *stash,
                                    float *out, void *stream) {
    if (M == 0) return 0;                       // an empty batch is a no-op (its tensors have no storage)
    ZEST_CHECK_ARG(desc && packed_fwd && x && stash && out && M > 0, "zest_mlp_train16_fwd: bad argument");
    ZEST_CHECK_ARG(((uintptr_t)stash & 15) == 0, "zest_mlp_train16_fwd: stash must be 16-byte aligned");
    TrainTables *t = tables_for(*desc);
    if (!t) return (int)hipErrorInvalidValue;
    const Sizes s = sizes_of(M);
    const void *tiles = (const char *)packed_fwd + t->fwd.bias_bytes;
    return zest::mlp_engine_train_launch(t->fwd, tiles, x, M, out, stash, (char *)stash + s.tile_bytes, (hipStream_t)stream);
}

extern "C" int zest_mlp_train16_bwd(const zest_mlp_desc *desc, const void *packed_bwd, const float *const *params,
                                    const float *x, int M, const void *stash, const float *out, const float *g_out,
                                    void *work, float *g_x, float *const *g_params, int stages, void *stream) {
    if (M == 0) return 0;                       // an empty batch is a no-op (its tensors have no storage)
    ZEST_CHECK_ARG(desc && packed_bwd && params && x && stash && out && g_out && work && g_x && g_params && M > 0,
                   "zest_mlp_train16_bwd: bad argument");
    ZEST_CHECK_ARG((((uintptr_t)work | (uintptr_t)stash | (uintptr_t)packed_bwd) & 15) == 0, "zest_mlp_train16_bwd: 16-byte alignment");
    TrainTables *t = tables_for(*desc);
    if (!t) return (int)hipErrorInvalidValue;
    const Sizes s = sizes_of(M);
    hipStream_t st = (hipStream_t)stream;
    const zest_mlp_desc &d = *desc;
    const bool mod = d.use_feat != 0;
    const int F = mod ? d.in_ch_feat : 0, P = d.in_ch_pts, C_in = P + F + d.in_ch_views;
    const int C_out = d.head == ZEST_HEAD_NONE ? 4 : (d.head == ZEST_HEAD_BLEND ? 5 : 12);
    const uint4 *stash_tiles = (const uint4 *)stash;
    const uint2 *masks = (const uint2 *)((const char *)stash + s.tile_bytes);
    uint4 *grad = (uint4 *)work;
    float4 *pts_side = (float4 *)((char *)work + s.grad_bytes);
    const int cus = cu_count();
    if (stages & 1) {
        const int units = zest::bwd_stream_units_of(d);
        const long long n_pass = (((long long)M + 31) / 32 + kWaves - 1) / kWaves;
        const int blocks = (int)(n_pass < cus ? n_pass : cus);
#define ZEST_DATA(NTP, MODF, NTF)                                                                              \
    do {                                                                                                       \
        ZEST_CHECK_ARG(units == bwd_stream_units(NTP, MODF ? NTF : 0), "zest_mlp_train16_bwd: stream of %d units, kernel expects %d", \
                       units, bwd_stream_units(NTP, MODF ? NTF : 0));                                          \
        hipLaunchKernelGGL((train16_data_kernel<NTP, MODF, NTF>), dim3(blocks), dim3(kWaves * 64), 0, st,      \
                           (const uint4 *)packed_bwd, x, out, g_out, M, P, F, C_in, C_out, d.head, masks, grad, pts_side); \
    } while (0)
        const int key = t->fwd.nt_pts * 10 + (mod ? t->fwd.nt_feat : 0);
        switch (key) {
            case 40: ZEST_DATA(4, false, 0); break;
            case 42: ZEST_DATA(4, true, 2); break;
            case 44: ZEST_DATA(4, true, 4); break;
            case 60: ZEST_DATA(6, false, 0); break;
            case 62: ZEST_DATA(6, true, 2); break;
            case 64: ZEST_DATA(6, true, 4); break;
            default:
                zest_set_error("zest_mlp_train16_bwd: no data kernel for %d point units / %d feature units", t->fwd.nt_pts,
                               mod ? t->fwd.nt_feat : 0);
                return (int)hipErrorInvalidValue;
        }
#undef ZEST_DATA
    }
    if (stages & 2) {
        const long long n_blocks = ((long long)M + 31) / 32;
        const uint4 *tail = (const uint4 *)packed_bwd + (size_t)zest::bwd_stream_units_of(d) * 64;
#define ZEST_FIN(NTP, NTF)                                                                                       \
    hipLaunchKernelGGL((train16_finish_kernel<NTP, NTF>), dim3(zest_div_up(n_blocks, 4)), dim3(256), 0, st, x, M, P, F, \
                       C_in, tail, stash_tiles, grad, (const float4 *)pts_side, t->map_pts, t->map_feat, g_x)
        const int key = t->fwd.nt_pts * 10 + (mod ? t->fwd.nt_feat : 0);
        switch (key) {
            case 40: ZEST_FIN(4, 0); break;
            case 42: ZEST_FIN(4, 2); break;
            case 44: ZEST_FIN(4, 4); break;
            case 60: ZEST_FIN(6, 0); break;
            case 62: ZEST_FIN(6, 2); break;
            case 64: ZEST_FIN(6, 4); break;
            default: break;
        }
#undef ZEST_FIN
    }
    if (stages & 4) {
        float4 *partial = (float4 *)((char *)work + s.grad_bytes + s.side_bytes + kPtrTableBytes);
        zest::train16_dw_launch(t->jobs, t->n_jobs, t->n_wg, stash_tiles, (const uint4 *)grad, M, partial, g_params, st);
    }
    ZEST_RETURN_LAUNCH("zest_mlp_train16_bwd");
}
