// MLP entry points: weight packing, the fp32 parity kernel, dispatch to the bf16 engine.
//
// fp32 kernel (ZEST_PREC_F32): one workgroup of 4 waves owns 32 samples; activations live
// in LDS as [feature][sample], each wave produces row-blocks of 32 output features with
// v_mfma_f32_32x32x2_f32 (exact fp32 products, fp32 accumulate: bitwise an fmaf chain), so
// results differ from the reference's fp32 GEMM only by summation order.  MFMA-bound at the
// fp32 matrix rate (157 TFLOP/s peak); used for parity tests and fp32 training forward.
#include <string.h>
#include <map>
#include <mutex>
#include <tuple>
#include "mlp_plan.h"
#include "zest_common.cuh"
#include "mlp_engine.cuh"

namespace {

using zest::MlpPlan;

struct DevPlan {
    MlpPlan plan;
    uint32_t *tile_src = nullptr, *bias_src = nullptr, *hdr_src = nullptr;   // device gather tables
    uint8_t *unit_part = nullptr;                                            // device: hi / lo flag per unit
};

std::mutex g_mu;
std::map<std::tuple<int, int, int, int, int, int, int, int, int, int, int, int>, DevPlan *> g_plans;   // per shape AND device

constexpr int kOrderBwd = 99;       // key of the training path's backward-data stream (bf16, build_bwd_plan)

// Plans are built once per (shape, precision, order, device) and kept for the life of the process: a plan's gather
// tables live in the memory of the device that was current when they were uploaded, so the current device is
// part of the key (include/zest_render.h, "Devices").
DevPlan *get_plan(const zest_mlp_desc &d, int precision, int order, bool need_tables) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    std::lock_guard<std::mutex> lk(g_mu);
    auto key = std::make_tuple(d.in_ch_pts, d.use_feat ? d.in_ch_feat : 0, d.in_ch_views,
                               d.use_feat, d.net_type, d.head, precision, order, d.depth, d.width, d.skip_mask, dev);
    auto it = g_plans.find(key);
    DevPlan *dp = it == g_plans.end() ? nullptr : it->second;
    if (!dp) {
        dp = new DevPlan();
        const char *err = nullptr;
        const bool ok = order == kOrderBwd ? zest::build_bwd_plan(d, &dp->plan, &err)
                                           : zest::build_plan(d, precision, order, &dp->plan, &err);
        if (!ok) {
            zest_set_error("MLP shape not supported: %s", err);
            delete dp;
            return nullptr;
        }
        g_plans[key] = dp;
    }
    if (need_tables && !dp->tile_src) {
        const size_t nt = dp->plan.tile_src.size() * 4, nb = dp->plan.bias_src.size() * 4;
        const size_t nh = dp->plan.hdr_src.size() * 4, np = dp->plan.unit_part.size();
        hipError_t e = hipMalloc(&dp->tile_src, nt);
        if (e == hipSuccess) e = hipMalloc(&dp->unit_part, np ? np : 4);
        if (e == hipSuccess && np) e = hipMemcpy(dp->unit_part, dp->plan.unit_part.data(), np, hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMalloc(&dp->bias_src, nb ? nb : 4);
        if (e == hipSuccess) e = hipMalloc(&dp->hdr_src, nh ? nh : 4);
        if (e == hipSuccess) e = hipMemcpy(dp->tile_src, dp->plan.tile_src.data(), nt, hipMemcpyHostToDevice);
        if (e == hipSuccess && nb) e = hipMemcpy(dp->bias_src, dp->plan.bias_src.data(), nb, hipMemcpyHostToDevice);
        if (e == hipSuccess && nh) e = hipMemcpy(dp->hdr_src, dp->plan.hdr_src.data(), nh, hipMemcpyHostToDevice);
        if (e != hipSuccess) {
            zest_set_error("zest_mlp_pack: uploading gather tables: %s", hipGetErrorString(e));
            dp->tile_src = nullptr;
            return nullptr;
        }
    }
    return dp;
}

inline int order_for(int precision) {
    return zest::prec_is_engine(precision) ? zest::ORDER_ACC : zest::ORDER_NATURAL;
}

struct ParamTable {
    const float *p[2 * ZEST_P_COUNT];
};

// PREC: element type of the packed tiles.  ZEST_PREC_F16X3: a unit flagged 1 in unit_part holds
// the scaled remainders fp16((w - fp16(w)) * 2^11) of the unit in front of it.
template <int PREC>
__global__ void pack_kernel(ParamTable pt, const uint32_t *__restrict__ tile_src, size_t n_w,
                            const uint32_t *__restrict__ bias_src, size_t n_b,
                            const uint8_t *__restrict__ unit_part,
                            float *__restrict__ bias_out, void *__restrict__ w_out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_b) {
        const uint32_t s = bias_src[i];
        bias_out[i] = s == 0xFFFFFFFFu ? 0.0f : pt.p[2 * (s >> 24) + 1][s & 0xFFFFFF];
    }
    if (i < n_w) {
        const uint32_t s = tile_src[i];
        const float v = s == 0xFFFFFFFFu ? 0.0f : pt.p[2 * (s >> 24)][s & 0xFFFFFF];
        if (PREC == ZEST_PREC_BF16) {
            ((__hip_bfloat16 *)w_out)[i] = __float2bfloat16(v);
        } else if (PREC == ZEST_PREC_F16) {
            ((_Float16 *)w_out)[i] = (_Float16)v;
        } else if (PREC == ZEST_PREC_F16X3) {
            const _Float16 hi = (_Float16)v;
            ((_Float16 *)w_out)[i] = unit_part[i / 512] ? (_Float16)((v - (float)hi) * 2048.0f) : hi;
        } else {
            ((float *)w_out)[i] = v;
        }
    }
}

// header units of the bf16 stream: 64 fp32 (bias block, modulation bias block) at the head
__global__ void pack_headers_kernel(ParamTable pt, const uint32_t *__restrict__ hdr_src, size_t n,
                                    char *__restrict__ w_out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t s = hdr_src[i];
    if (s == 0xFFFFFFFFu) return;
    const size_t unit = i / 64, j = i % 64;
    ((float *)(w_out + unit * 1024))[j] = pt.p[2 * (s >> 24) + 1][s & 0xFFFFFF];
}

// ------------------------------------------------------------------ fp32 parity kernel
struct F32Op {
    int njb, nseg, kind0, nt0, kind1, nt1, mod, relu, tile_base, tiles_per_jb, bias_block;
    int src_h, dst;       // LDS buffer of the H operand / destination (0 = hA, 1 = hB, 2 = global)
};
struct F32Prog {
    F32Op op[zest::kNumOps];
    int n_ops, W;              // depth + 4 ops; trunk width
    int P, F, Vw, C_in, C_out, nt_feat, net_v2, act_out, head;     // net_v2: additive modulation; act_out: sigmoid(rgb), relu(alpha)
    int rows_pts, rows_feat;   // padded row counts of the input buffers
};

constexpr int kF32Samples = 32;

__global__ __launch_bounds__(256, 1) void mlp_f32_kernel(F32Prog pr, const float *__restrict__ bias,
                                                         const float4 *__restrict__ tiles,
                                                         const float *__restrict__ x, int M,
                                                         float *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *b_pts = smem;                                   // [rows_pts][32]
    float *b_feat = b_pts + pr.rows_pts * 32;              // [rows_feat][32]
    float *b_views = b_feat + pr.rows_feat * 32;           // [32][32]
    float *b_h[2] = {b_views + 32 * 32, b_views + 32 * 32 + pr.W * 32};
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int col = lane & 31, half = lane >> 5;
    const int n_in_rows = pr.rows_pts + pr.rows_feat + 32;

    for (int m0 = blockIdx.x * kF32Samples; m0 < M; m0 += gridDim.x * kF32Samples) {
        __syncthreads();
        for (int i = tid; i < n_in_rows * 32; i += 256) smem[i] = 0.0f;    // pads must be zero
        __syncthreads();
        const int nvalid = min(kF32Samples, M - m0);
        for (int i = tid; i < nvalid * pr.C_in; i += 256) {
            const int s = i / pr.C_in, c = i % pr.C_in;
            const float v = x[(size_t)m0 * pr.C_in + i];
            float *dst = c < pr.P ? b_pts + c * 32
                                  : (c < pr.P + pr.F ? b_feat + (c - pr.P) * 32
                                                     : b_views + (c - pr.P - pr.F) * 32);
            dst[s] = v;
        }
        __syncthreads();
#pragma unroll 1
        for (int o = 0; o < pr.n_ops; o++) {
            const F32Op op = pr.op[o];
            for (int jb = wave; jb < op.njb; jb += 4) {
                const float4 *tp = tiles + ((size_t)op.tile_base + (size_t)jb * op.tiles_per_jb) * 64 + lane;
                f32x16 acc, macc;
                {
                    const float *bb = bias + (size_t)(op.bias_block + jb) * 32 + half * 16;
#pragma unroll
                    for (int i = 0; i < 16; i++) acc[i] = bb[i];
                }
                if (op.mod) {
                    const float *mb = bias + (size_t)jb * 32 + half * 16;
#pragma unroll
                    for (int i = 0; i < 16; i++) macc[i] = mb[i];
                    for (int t = 0; t < pr.nt_feat; t++, tp += 64) {
                        const float4 a = *tp;
                        const float *bp = b_feat + (t * 8 + 4 * half) * 32 + col;
                        macc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, bp[0], macc, 0, 0, 0);
                        macc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, bp[32], macc, 0, 0, 0);
                        macc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, bp[64], macc, 0, 0, 0);
                        macc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, bp[96], macc, 0, 0, 0);
                    }
                }
                for (int sg = 0; sg < op.nseg; sg++) {
                    const int kind = sg ? op.kind1 : op.kind0, nt = sg ? op.nt1 : op.nt0;
                    const float *src = kind == zest::SEG_PTS ? b_pts
                                       : kind == zest::SEG_VIEWS ? b_views : b_h[op.src_h];
                    for (int t = 0; t < nt; t++, tp += 64) {
                        const float4 a = *tp;
                        const float *bp = src + (t * 8 + 4 * half) * 32 + col;
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, bp[0], acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, bp[32], acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, bp[64], acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, bp[96], acc, 0, 0, 0);
                    }
                }
#pragma unroll
                for (int i = 0; i < 16; i++) {
                    const int row = 32 * jb + (i & 3) + 8 * (i >> 2) + 4 * half;
                    float v = acc[i];
                    if (op.mod) v = pr.net_v2 ? v + macc[i] : v * macc[i];
                    if (op.relu) v = fmaxf(v, 0.0f);
                    if (op.dst < 2) {
                        b_h[op.dst][row * 32 + col] = v;
                    } else if (col < nvalid) {
                        float *orow = out + (size_t)(m0 + col) * pr.C_out;
                        if (o == pr.n_ops - 1) {             // rgb tile
                            if (row < 3) orow[row] = pr.act_out ? zest_sigmoid(v) : v;
                        } else if (row == 0) {               // head tile: alpha
                            orow[3] = pr.act_out ? fmaxf(v, 0.0f) : v;
                        } else if (3 + row < pr.C_out) {     // blend weight | scene flow, prob
                            const bool is_sf = pr.head == ZEST_HEAD_DYNAMIC && row <= 6;
                            orow[3 + row] = is_sf ? tanhf(v) : zest_sigmoid(v);
                        }
                    }
                }
            }
            __syncthreads();
        }
    }
}

F32Prog make_f32_prog(const MlpPlan &p) {
    F32Prog pr = {};
    // trunk ping-pong: layer l lands in buffer l & 1, so the trunk output sits in t = (D-1) & 1 (depth 8: B);
    // the head tile reads t; feature t -> 1-t; view layer 1-t -> t; rgb reads t
    const int D = p.shape.D, t = (D - 1) & 1;
    pr.n_ops = p.n_ops, pr.W = p.shape.W;
    for (int o = 0; o < p.n_ops; o++) {
        const zest::OpPlan &s = p.op[o];
        F32Op &d = pr.op[o];
        const int src_h = o < D ? (o + 1) & 1 : (o == D + 2 ? 1 - t : t);
        const int dst = o < D ? o & 1 : (o == D + 1 ? 1 - t : (o == D + 2 ? t : 2));
        d.njb = s.njb, d.nseg = s.nseg, d.kind0 = s.seg[0].kind, d.nt0 = s.seg[0].ntiles;
        d.kind1 = s.seg[1].kind, d.nt1 = s.seg[1].ntiles, d.mod = s.mod, d.relu = s.relu;
        d.tile_base = s.tile_base, d.tiles_per_jb = s.tiles_per_jb, d.bias_block = s.bias_block;
        d.src_h = src_h, d.dst = dst;
    }
    pr.P = p.desc.in_ch_pts, pr.F = p.desc.use_feat ? p.desc.in_ch_feat : 0, pr.Vw = p.desc.in_ch_views;
    pr.C_in = pr.P + pr.F + pr.Vw;
    pr.C_out = p.desc.head == ZEST_HEAD_NONE ? 4 : (p.desc.head == ZEST_HEAD_BLEND ? 5 : 12);
    pr.nt_feat = p.nt_feat, pr.net_v2 = p.desc.net_type >= 2, pr.act_out = p.desc.net_type == 2, pr.head = p.desc.head;
    pr.rows_pts = p.nt_pts * 8, pr.rows_feat = p.nt_feat * 8;
    return pr;
}

}  // namespace

// the training path's transposed stream: same packer, other plan (mlp_train16.hip calls these)
namespace zest {
size_t bwd_stream_bytes(const zest_mlp_desc &d) {
    DevPlan *dp = get_plan(d, ZEST_PREC_BF16, kOrderBwd, false);
    return dp ? dp->plan.bytes : 0;
}
int bwd_stream_units_of(const zest_mlp_desc &d) {          // the ring-streamed part (the modulation^T tail follows it)
    DevPlan *dp = get_plan(d, ZEST_PREC_BF16, kOrderBwd, false);
    return dp ? dp->plan.tail_unit0 : -1;
}
int pack_bwd_stream(const zest_mlp_desc &d, const float *const *params, void *packed, hipStream_t stream) {
    DevPlan *dp = get_plan(d, ZEST_PREC_BF16, kOrderBwd, true);
    if (!dp) return (int)hipErrorInvalidValue;
    const MlpPlan &p = dp->plan;
    ParamTable pt;
    for (int i = 0; i < 2 * ZEST_P_COUNT; i++) pt.p[i] = params[i];
    const size_t n_w = p.tile_src.size(), n_h = p.hdr_src.size();
    hipLaunchKernelGGL(pack_kernel<ZEST_PREC_BF16>, dim3(zest_div_up(n_w, 256)), dim3(256), 0, stream, pt, dp->tile_src,
                       n_w, dp->bias_src, (size_t)0, dp->unit_part, (float *)packed, packed);
    hipLaunchKernelGGL(pack_headers_kernel, dim3(zest_div_up(n_h, 256)), dim3(256), 0, stream, pt, dp->hdr_src, n_h,
                       (char *)packed);
    ZEST_RETURN_LAUNCH("zest_mlp_train16_pack");
}
}  // namespace zest

extern "C" size_t zest_mlp_packed_bytes(const zest_mlp_desc *desc, int precision) {
    if (!desc) return 0;
    DevPlan *dp = get_plan(*desc, precision, order_for(precision), false);
    return dp ? dp->plan.bytes : 0;
}

extern "C" int zest_mlp_pack(const zest_mlp_desc *desc, int precision, const float *const *params,
                             void *packed, void *stream) {
    ZEST_CHECK_ARG(desc && params && packed, "zest_mlp_pack: null argument");
    DevPlan *dp = get_plan(*desc, precision, order_for(precision), true);
    if (!dp) return (int)hipErrorInvalidValue;
    const MlpPlan &p = dp->plan;
    ParamTable pt;
    for (int i = 0; i < 2 * ZEST_P_COUNT; i++) pt.p[i] = params[i];
    // every parameter the gather tables reference must be present
    bool need[ZEST_P_COUNT] = {};
    for (int i = 0; i < p.shape.D; i++) need[i] = true;
    need[ZEST_P_PTS_BIAS] = desc->use_feat != 0;
    need[ZEST_P_VIEWS] = need[ZEST_P_FEATURE] = need[ZEST_P_ALPHA] = need[ZEST_P_RGB] = true;
    need[ZEST_P_HEAD0] = desc->head != ZEST_HEAD_NONE;
    need[ZEST_P_HEAD1] = desc->head == ZEST_HEAD_DYNAMIC;
    for (int i = 0; i < ZEST_P_COUNT; i++)
        ZEST_CHECK_ARG(!need[i] || (params[2 * i] && params[2 * i + 1]),
                       "zest_mlp_pack: parameter slot %d (weight and bias) is required", i);
    ZEST_CHECK_ARG(((uintptr_t)packed & 15) == 0, "zest_mlp_pack: packed must be 16-byte aligned");
    const size_t n_w = p.tile_src.size(), n_b = p.bias_src.size();
    const size_t n = n_w > n_b ? n_w : n_b;
    void *w_out = (char *)packed + p.bias_bytes;
#define ZEST_PACK(PREC)                                                                            \
    hipLaunchKernelGGL(pack_kernel<PREC>, dim3(zest_div_up(n, 256)), dim3(256), 0, (hipStream_t)stream, \
                       pt, dp->tile_src, n_w, dp->bias_src, n_b, dp->unit_part, (float *)packed, w_out)
    if (zest::prec_is_engine(precision)) {
        if (precision == ZEST_PREC_BF16) ZEST_PACK(ZEST_PREC_BF16);
        else if (precision == ZEST_PREC_F16) ZEST_PACK(ZEST_PREC_F16);
        else ZEST_PACK(ZEST_PREC_F16X3);
        const size_t n_h = p.hdr_src.size();
        if (n_h)     // same stream, after the tile pass that zero-filled the header units
            hipLaunchKernelGGL(pack_headers_kernel, dim3(zest_div_up(n_h, 256)), dim3(256), 0,
                               (hipStream_t)stream, pt, dp->hdr_src, n_h, (char *)w_out);
    } else {
        ZEST_PACK(ZEST_PREC_F32);
    }
#undef ZEST_PACK
    ZEST_RETURN_LAUNCH("zest_mlp_pack");
}

extern "C" int zest_mlp_fwd(const zest_mlp_desc *desc, int precision, const void *packed,
                            const float *x, int M, float *out, void *stream) {
    if (M == 0) return 0;                       // an empty batch is a no-op (its tensors have no storage)
    ZEST_CHECK_ARG(desc && packed && x && out, "zest_mlp_fwd: null argument");
    ZEST_CHECK_ARG(M >= 0, "zest_mlp_fwd: M=%d", M);
    DevPlan *dp = get_plan(*desc, precision, order_for(precision), false);
    if (!dp) return (int)hipErrorInvalidValue;
    if (M == 0) return 0;
    const MlpPlan &p = dp->plan;
    const float *bias = (const float *)packed;
    const void *tiles = (const char *)packed + p.bias_bytes;
    if (precision == ZEST_PREC_F32) {
        const F32Prog pr = make_f32_prog(p);
        const size_t lds = (size_t)(pr.rows_pts + pr.rows_feat + 32 + 2 * pr.W) * 32 * sizeof(float);
        const int blocks = zest_div_up(M, kF32Samples);
        hipLaunchKernelGGL(mlp_f32_kernel, dim3(blocks < 4096 ? blocks : 4096), dim3(256), lds,
                           (hipStream_t)stream, pr, bias, (const float4 *)tiles, x, M, out);
        ZEST_RETURN_LAUNCH("zest_mlp_fwd(f32)");
    }
    return zest::mlp_engine_launch(p, tiles, x, M, out, (hipStream_t)stream);
}
