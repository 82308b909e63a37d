// Standalone MLP forward on the register engine (zest_mlp_fwd, ZEST_PREC_BF16 / _F16 / _F16X3):
// x [M,C_in] fp32 in HBM -> operand registers -> engine -> out [M,C_out].  Backs MVSNeRF.forward
// in the 16-bit modes, the fp32-class per-op path (split fp16) and the MFMA-utilisation
// measurement of the MLP alone.  Same structure as the fused renderer: 8 waves
// per workgroup share the weight stream through the LDS ring (LDS-DMA), each wave carries 32
// rows through the network in registers; only the operand source (rows of x instead of the
// in-kernel encoders) and the sink (raw network outputs instead of compositing) differ.
#include "mlp_engine.cuh"
#include "mlp_train16.h"
#include "mlp_operands.cuh"

namespace zest {

// A wave runs CB column blocks of 16 rows (lane: column l & 15, group l >> 4): two, or one where
// every operand is a register pair (split fp16).
constexpr int kMlpWaves = 8;
constexpr int mlp_cb(int EP) { return EP == ZEST_PREC_F16X3 ? 1 : 2; }

// Training forward: every layer's output operand tiles go to the activation stash as they are
// produced (1 KiB per tile and column block, written by the wave that holds it: fully coalesced), plus
// one bit per element "was active" for the layers with a ReLU.  Layout (mlp_train16.h):
//   tiles  [block][kStashTiles][CB][64 lanes] x 16 B      tile = 8 * layer + k-tile (layers 0-7),
//                                                          64 + k-tile feature_linear, 72 + k-tile view layer
//   masks  [block][kStashMasks][CB][64 lanes] x 8 B       bit 8 * k-tile + element; 0-7 trunk, 8 view layer
__device__ __forceinline__ void stash_store(uint4 *p, uint4 q) {
#ifndef ZEST_STASH_CACHED              // (defined: plain stores, for A/B timing)
    __builtin_nontemporal_store(__builtin_bit_cast(v4u, q), reinterpret_cast<v4u *>(p));
#else
    *p = q;
#endif
}

template <int CB>
struct StashSink {
    uint4 *tiles;
    uint2 *masks;
    long long block;
    int lane;
    mutable unsigned lo[CB], hi[CB];
    __device__ __forceinline__ void operator()(int id, int jb, int cb, const float (&v)[8]) const {
        const int t = id < 8 ? 8 * id + jb : (id == 8 ? 64 + jb : 72 + jb);
        uint4 q = make_uint4(pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3]), pack_bf16(v[4], v[5]), pack_bf16(v[6], v[7]));
        // streaming stores: the stash (0.7 GB per pass) is read back by later kernels only and must not push the
        // weight stream and the rows of x out of L2 (forward 285 -> 245 us per 131k samples)
        stash_store(&tiles[((block * kStashTiles + t) * CB + cb) * 64 + lane], q);
        if (id == 8) return;                                     // feature_linear has no activation
        unsigned bits = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) bits |= (v[i] > 0.0f ? 1u : 0u) << i;
        if (jb == 0) lo[cb] = 0, hi[cb] = 0;
        if (jb < 4) lo[cb] |= bits << (8 * jb);
        else hi[cb] |= bits << (8 * (jb - 4));
        const int last = id == 9 ? 3 : 7, mid = id == 9 ? 8 : id;
        if (jb == last) masks[((block * kStashMasks + mid) * CB + cb) * 64 + lane] = make_uint2(lo[cb], hi[cb]);
    }
};

template <int EP, int NT_PTS, bool MOD, int NT_FEAT, bool TRAIN = false, bool V2 = false>
__global__ __launch_bounds__(kMlpWaves * 64, kMlpWaves / 4) void mlp_engine_kernel(
    const uint4 *__restrict__ tiles_g, const float *__restrict__ x, int M, int P, int F, int C_in, int C_out, int head,
    int act_out, float *__restrict__ out, uint4 *__restrict__ stash_tiles = nullptr, uint2 *__restrict__ stash_masks = nullptr) {
    constexpr int NP = ep_parts(EP), CB = mlp_cb(EP), UNITS = stream_units(NT_PTS, MOD ? NT_FEAT : 0, NP);
    using Ring = RingTiles<kMlpWaves, UNITS, 0>;
    if constexpr (EP == ZEST_PREC_F16) engine_fp16_overflow_clamp();
    __shared__ __attribute__((aligned(16))) char lds[kRingUnits * 1024 + 2 * kSlots * 4];
#ifdef ZEST_RING_FLAGS
    int *ring_flags = reinterpret_cast<int *>(lds + kRingUnits * 1024);
    Ring::init_flags(ring_flags);
    __syncthreads();
#endif
    const int lane = threadIdx.x & 63, col = lane & 15, grp = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const Ring tiles{lds, (gptr_u4)tiles_g, (gptr_u4)tiles_g, lane, grp, wave,
                     (unsigned)(wave * Ring::kPieces * 64 + lane) * 16u,
                     (unsigned)(uintptr_t)(__attribute__((address_space(3))) char *)lds +
                         (unsigned)wave * Ring::kPieces * 1024u
#ifdef ZEST_RING_FLAGS
                     , (unsigned)(uintptr_t)(__attribute__((address_space(3))) char *)ring_flags
#endif
    };
    tiles.init_addr();
    tiles.prologue();
    const int n_blocks = (M + 16 * CB - 1) / (16 * CB), n_pass = (n_blocks + kMlpWaves - 1) / kMlpWaves;
    for (int pass = blockIdx.x; pass < n_pass; pass += gridDim.x) {
        // every wave walks the stream in step, also one whose rows lie past M (its loads are
        // clamped and its stores masked)
        const long long m_base = ((long long)pass * kMlpWaves + wave) * (16 * CB);
        OpArr<NT_PTS / 2, NP> pts[CB];
        OpArr<NT_FEAT / 2, NP> feat[CB];
#pragma unroll
        for (int cb = 0; cb < CB; cb++) {
            const long long m = m_base + 16 * cb + col;
            const bool valid = m < M;
#ifdef ZEST_EXP_NO_XGATHER             // timing experiment only: every lane reads row 0 (one cache line set, L1 hits)
            const float *xrow = x;
#else
            const float *xrow = x + (size_t)(valid ? m : 0) * C_in;
#endif
            load_pe_operand<EP, NT_PTS == 4 ? 3 : 4, 10, NT_PTS / 2>(xrow, valid, grp, pts[cb]);
            if (MOD) load_feat_operand<EP, NT_FEAT / 2>(xrow + P, F, valid, grp, feat[cb]);
            if constexpr (TRAIN) {          // the encoder's operands go to the stash too: inputs of the weight kernel
                uint4 *st = stash_tiles + (((long long)pass * kMlpWaves + wave) * kStashTiles * CB + cb) * 64 + lane;
#pragma unroll
                for (int k = 0; k < NT_PTS / 2; k++) stash_store(&st[(size_t)(kStashPts + k) * CB * 64], __builtin_bit_cast(uint4, pts[cb].t[0][k]));
                if (MOD) {
#pragma unroll
                    for (int k = 0; k < NT_FEAT / 2; k++) stash_store(&st[(size_t)(kStashFeat + k) * CB * 64], __builtin_bit_cast(uint4, feat[cb].t[0][k]));
                }
            }
        }
        auto views_fn = [&](OpArr<1, NP> (&views)[CB]) {
#pragma unroll
            for (int cb = 0; cb < CB; cb++) {
                const long long m = m_base + 16 * cb + col;
                const bool valid = m < M;
                load_pe_operand<EP, 3, 4, 1>(x + (size_t)(valid ? m : 0) * C_in + P + F, valid, grp, views[cb]);
                if constexpr (TRAIN)
                    stash_store(&stash_tiles[((((long long)pass * kMlpWaves + wave) * kStashTiles + kStashViews) * CB + cb) * 64 + lane],
                                __builtin_bit_cast(uint4, views[cb].t[0][0]));
            }
        };
        f32x4 headt[CB], rgbt[CB];
        int unit = 0;
        // the rows of x come from global memory: the point operand is built once and kept (a reload in
        // the middle of the engine would drain the weight DMA behind its vmcnt wait)
        auto pts_fn = [&](OpArr<NT_PTS / 2, NP> (&o)[CB], int) {
#pragma unroll
            for (int cb = 0; cb < CB; cb++) o[cb] = pts[cb];
        };
        if constexpr (TRAIN) {
            const StashSink<CB> sink{stash_tiles, stash_masks, (long long)pass * kMlpWaves + wave, lane, {}, {}};
            engine_forward<EP, CB, NT_PTS, MOD, NT_FEAT, V2>(tiles, unit, pts_fn, feat, views_fn, headt, rgbt, sink);
        } else {
            engine_forward<EP, CB, NT_PTS, MOD, NT_FEAT, V2>(tiles, unit, pts_fn, feat, views_fn, headt, rgbt);
        }
#pragma unroll
        for (int cb = 0; cb < CB; cb++) {
            const long long m = m_base + 16 * cb + col;
            if (m >= M) continue;
            float *o = out + (size_t)m * C_out;
            // tile row r sits in lane group r >> 2, element r & 3
            if (grp == 0) {
                o[0] = act_out ? zest_sigmoid(rgbt[cb][0]) : rgbt[cb][0];
                o[1] = act_out ? zest_sigmoid(rgbt[cb][1]) : rgbt[cb][1];
                o[2] = act_out ? zest_sigmoid(rgbt[cb][2]) : rgbt[cb][2];
                o[3] = act_out ? fmaxf(headt[cb][0], 0.0f) : headt[cb][0];
                if (head == ZEST_HEAD_BLEND) o[4] = zest_sigmoid(headt[cb][1]);
                if (head == ZEST_HEAD_DYNAMIC)
                    o[4] = tanhf(headt[cb][1]), o[5] = tanhf(headt[cb][2]), o[6] = tanhf(headt[cb][3]);   // rows 1-3
            } else if (head == ZEST_HEAD_DYNAMIC && grp == 1) {
                o[7] = tanhf(headt[cb][0]), o[8] = tanhf(headt[cb][1]), o[9] = tanhf(headt[cb][2]);      // rows 4-6
                o[10] = zest_sigmoid(headt[cb][3]);                                                      // row 7
            } else if (head == ZEST_HEAD_DYNAMIC && grp == 2) {
                o[11] = zest_sigmoid(headt[cb][0]);                                                      // row 8
            }
        }
        tiles.next_pass();
    }
    tiles.drain();
}

template <int EP, int NT_PTS, bool MOD, int NT_FEAT, bool TRAIN = false, bool V2 = false>
static int launch_one(const MlpPlan &p, const void *tiles, const float *x, int M,
                      float *out, hipStream_t stream, void *stash_tiles = nullptr, void *stash_masks = nullptr) {
    constexpr int units = stream_units(NT_PTS, MOD ? NT_FEAT : 0, ep_parts(EP));
    if (p.n_tiles != units) {
        zest_set_error("zest_mlp_fwd(engine): plan has %d stream units, kernel expects %d", p.n_tiles, units);
        return (int)hipErrorInvalidValue;
    }
    const zest_mlp_desc &d = p.desc;
    const int F = d.use_feat ? d.in_ch_feat : 0;
    const int C_in = d.in_ch_pts + F + d.in_ch_views;
    const int C_out = d.head == ZEST_HEAD_NONE ? 4 : (d.head == ZEST_HEAD_BLEND ? 5 : 12);
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
        cus = 256;
    const int n_pass = zest_div_up(zest_div_up(M, 16 * mlp_cb(EP)), kMlpWaves);
    const int blocks = n_pass < cus ? n_pass : cus;             // one workgroup per CU (128 KiB ring)
    if ((d.net_type >= 2) != V2) {
        zest_set_error("zest_mlp_fwd(engine): kernel / net_type mismatch");
        return (int)hipErrorInvalidValue;
    }
    hipLaunchKernelGGL((mlp_engine_kernel<EP, NT_PTS, MOD, NT_FEAT, TRAIN, V2>), dim3(blocks), dim3(kMlpWaves * 64), 0, stream,
                       (const uint4 *)tiles, x, M, d.in_ch_pts, F, C_in, C_out, d.head,
                       d.net_type == 2 ? 1 : 0, out, (uint4 *)stash_tiles, (uint2 *)stash_masks);
    ZEST_RETURN_LAUNCH(TRAIN ? "zest_mlp_train16_fwd" : "zest_mlp_fwd(engine)");
}

template <int EP>
static int launch_prec(const MlpPlan &p, const void *tiles, const float *x, int M, float *out, hipStream_t stream) {
    const bool mod = p.desc.use_feat != 0;
    const int key = p.nt_pts * 10 + (mod ? p.nt_feat : 0);
    if (p.desc.net_type >= 2) {       // additive modulation ('v2' trunk): always with features, static nets (63 point channels)
        switch (key) {
            case 42: return launch_one<EP, 4, true, 2, false, true>(p, tiles, x, M, out, stream);
            case 44: return launch_one<EP, 4, true, 4, false, true>(p, tiles, x, M, out, stream);
            case 62: return launch_one<EP, 6, true, 2, false, true>(p, tiles, x, M, out, stream);
            case 64: return launch_one<EP, 6, true, 4, false, true>(p, tiles, x, M, out, stream);
        }
        zest_set_error("zest_mlp_fwd(engine): no 'v2' kernel for %d point units / %d feature units per row block",
                       p.nt_pts, mod ? p.nt_feat : 0);
        return (int)hipErrorInvalidValue;
    }
    switch (key) {
        case 40: return launch_one<EP, 4, false, 0>(p, tiles, x, M, out, stream);
        case 42: return launch_one<EP, 4, true, 2>(p, tiles, x, M, out, stream);
        case 44: return launch_one<EP, 4, true, 4>(p, tiles, x, M, out, stream);
        case 60: return launch_one<EP, 6, false, 0>(p, tiles, x, M, out, stream);
        case 62: return launch_one<EP, 6, true, 2>(p, tiles, x, M, out, stream);
        case 64: return launch_one<EP, 6, true, 4>(p, tiles, x, M, out, stream);
    }
    zest_set_error("zest_mlp_fwd(engine): no kernel for %d point units / %d feature units per row "
                   "block (supported: 1..14 source views)", p.nt_pts, mod ? p.nt_feat : 0);
    return (int)hipErrorInvalidValue;
}

// bf16 training forward: the same kernel with the activation stash (mlp_train16.hip)
int mlp_engine_train_launch(const MlpPlan &p, const void *tiles, const float *x, int M, float *out, void *stash_tiles,
                            void *stash_masks, hipStream_t stream) {
    constexpr int EP = ZEST_PREC_BF16;
    const bool mod = p.desc.use_feat != 0;
    const int key = p.nt_pts * 10 + (mod ? p.nt_feat : 0);
    switch (key) {
        case 40: return launch_one<EP, 4, false, 0, true>(p, tiles, x, M, out, stream, stash_tiles, stash_masks);
        case 42: return launch_one<EP, 4, true, 2, true>(p, tiles, x, M, out, stream, stash_tiles, stash_masks);
        case 44: return launch_one<EP, 4, true, 4, true>(p, tiles, x, M, out, stream, stash_tiles, stash_masks);
        case 60: return launch_one<EP, 6, false, 0, true>(p, tiles, x, M, out, stream, stash_tiles, stash_masks);
        case 62: return launch_one<EP, 6, true, 2, true>(p, tiles, x, M, out, stream, stash_tiles, stash_masks);
        case 64: return launch_one<EP, 6, true, 4, true>(p, tiles, x, M, out, stream, stash_tiles, stash_masks);
    }
    zest_set_error("zest_mlp_train16_fwd: no kernel for %d point units / %d feature units per row block", p.nt_pts,
                   mod ? p.nt_feat : 0);
    return (int)hipErrorInvalidValue;
}

int mlp_engine_launch(const MlpPlan &p, const void *tiles, const float *x, int M, float *out,
                      hipStream_t stream) {
    switch (p.precision) {
        case ZEST_PREC_BF16: return launch_prec<ZEST_PREC_BF16>(p, tiles, x, M, out, stream);
        case ZEST_PREC_F16: return launch_prec<ZEST_PREC_F16>(p, tiles, x, M, out, stream);
        case ZEST_PREC_F16X3: return launch_prec<ZEST_PREC_F16X3>(p, tiles, x, M, out, stream);
    }
    zest_set_error("zest_mlp_fwd(engine): precision %d is not an engine operand type", p.precision);
    return (int)hipErrorInvalidValue;
}

}  // namespace zest
