// Standalone MLP forward on the register engine (zest_mlp_fwd, ZEST_PREC_BF16 / _F16 / _F16X3):
// x [M,C_in] fp32 in HBM -> operand registers -> engine -> out [M,C_out].  Backs MVSNeRF.forward
// in the 16-bit modes, the fp32-class per-op path (split fp16) and the MFMA-utilisation
// measurement of the MLP alone.  Same structure as the fused renderer: 8 waves
// per workgroup share the weight stream through the LDS ring (LDS-DMA), each wave carries 32
// rows through the network in registers; only the operand source (rows of x instead of the
// in-kernel encoders) and the sink (raw network outputs instead of compositing) differ.
#include "mlp_engine.cuh"

namespace zest {

// Operand assembly.  The position -> input-column maps of the plan (mlp_plan.hip pe_map_acc,
// feat_map_acc) are affine in the lane group, so a lane needs one row pointer per operand and
// compile-time offsets - no table lookups, no per-element address arithmetic:
//   PE operand of C coordinates, L bands: element e of k-tile kt is m = 8 kt + e;
//     m < (L/2) C:  column C + 2C (2 (m / C) + (g >> 1)) + (g & 1) C + m % C
//                   = [C + 4C (m / C) + m % C] + [(g >> 1) 2C + (g & 1) C];   m = (L/2) C: column g (< C)
//   feature operand: quad q = 8 kt + 2 g + (e >> 2), channel c = e & 3:
//     q = 0, 2: volume columns c, 4 + c;  q = 1, 3: columns 8 + c, 12 + c;  q >= 4: column 4 q + c
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
#pragma unroll
    for (int t = 0; t < NK; t++) {
        // first columns of this lane's two quads
        int ca, cb;
        if (t == 0) {
            ca = grp == 0 ? 0 : (grp == 1 ? 4 : 8 * grp);              // quads 0, 2, 4, 6
            cb = grp == 0 ? 8 : (grp == 1 ? 12 : 8 * grp + 4);         // quads 1, 3, 5, 7
        } else {
            ca = 32 * t + 8 * grp, cb = ca + 4;
        }
        const bool va = valid && ca + 4 <= F, vb = valid && cb + 4 <= F;
        const float *pa = xf + (va ? ca : 0), *pb = xf + (vb ? cb : 0);
        float v[8];
#pragma unroll
        for (int c = 0; c < 4; c++) v[c] = va ? pa[c] : 0.0f, v[4 + c] = vb ? pb[c] : 0.0f;
        store_tile<EP>(v, op, t);
    }
}

// A wave runs CB column blocks of 16 rows (lane: column l & 15, group l >> 4): two, or one where
// every operand is a register pair (split fp16).
constexpr int kMlpWaves = 8;
constexpr int mlp_cb(int EP) { return EP == ZEST_PREC_F16X3 ? 1 : 2; }

template <int EP, int NT_PTS, bool MOD, int NT_FEAT>
__global__ __launch_bounds__(kMlpWaves * 64, kMlpWaves / 4) void mlp_engine_kernel(
    const uint4 *__restrict__ tiles_g, const float *__restrict__ x, int M, int P, int F, int C_in, int C_out, int head, int v2,
    int act_out, float *__restrict__ out) {
    constexpr int NP = ep_parts(EP), CB = mlp_cb(EP), UNITS = stream_units(NT_PTS, MOD ? NT_FEAT : 0, NP);
    using Ring = RingTiles<kMlpWaves, UNITS, 0>;
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
            const float *xrow = x + (size_t)(valid ? m : 0) * C_in;
            load_pe_operand<EP, NT_PTS == 4 ? 3 : 4, 10, NT_PTS / 2>(xrow, valid, grp, pts[cb]);
            if (MOD) load_feat_operand<EP, NT_FEAT / 2>(xrow + P, F, valid, grp, feat[cb]);
        }
        auto views_fn = [&](OpArr<1, NP> (&views)[CB]) {
#pragma unroll
            for (int cb = 0; cb < CB; cb++) {
                const long long m = m_base + 16 * cb + col;
                const bool valid = m < M;
                load_pe_operand<EP, 3, 4, 1>(x + (size_t)(valid ? m : 0) * C_in + P + F, valid, grp, views[cb]);
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
        engine_forward<EP, CB, NT_PTS, MOD, NT_FEAT>(tiles, unit, v2 != 0, pts_fn, feat, views_fn, headt, rgbt);
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

template <int EP, int NT_PTS, bool MOD, int NT_FEAT>
static int launch_one(const MlpPlan &p, const void *tiles, const float *x, int M,
                      float *out, hipStream_t stream) {
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
    hipLaunchKernelGGL((mlp_engine_kernel<EP, NT_PTS, MOD, NT_FEAT>), dim3(blocks), dim3(kMlpWaves * 64), 0, stream,
                       (const uint4 *)tiles, x, M, d.in_ch_pts, F, C_in, C_out, d.head,
                       d.net_type >= 2 ? 1 : 0, d.net_type == 2 ? 1 : 0, out);
    ZEST_RETURN_LAUNCH("zest_mlp_fwd(engine)");
}

template <int EP>
static int launch_prec(const MlpPlan &p, const void *tiles, const float *x, int M, float *out, hipStream_t stream) {
    const bool mod = p.desc.use_feat != 0;
    const int key = p.nt_pts * 10 + (mod ? p.nt_feat : 0);
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
