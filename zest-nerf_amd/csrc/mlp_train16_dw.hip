// Weight-gradient kernel of the bf16 training path (mlp_train16.hip has the rest: data and finishing kernels, host
// side) in a translation unit of its own: it is compiled WITH the SLP vectorizer, the others without - the packed
// f32 instructions SLP creates cost the data / finishing / forward kernels 2 - 5 % beside their MFMAs, while this
// kernel runs twice as long without it (profiles/r03_ab_train_noslp.txt).
#include "mlp_operands.cuh"
#include "mlp_train16.h"

namespace {

using namespace zest;
constexpr int EP = ZEST_PREC_BF16;
constexpr int kWaves = 8;

// a stash tile, read once by the kernel: streaming load (weight kernel 371 -> 353 us against plain loads,
// -DZEST_STASH_CACHED)
__device__ __forceinline__ uint4 stash_load(const uint4 *p) {
#ifndef ZEST_STASH_CACHED
    return __builtin_bit_cast(uint4, __builtin_nontemporal_load(reinterpret_cast<const v4u *>(p)));
#else
    return *p;
#endif
}

// ------------------------------------------------------------------------------ weight kernel
constexpr int kDwSlots = 34;                 // accumulators of a wave of the weight kernel: 2 row tiles x 16 column tiles + 2 bias
constexpr int kImgStride = 528;              // bytes per sample row of an LDS image: 256 positions x 2 B + 16 (bank spread)
constexpr int kImgBytes = 32 * kImgStride;

typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

// operand with k = the 32 samples of a block: rows/cols = positions p0 .. p0+15 of the image
__device__ __forceinline__ bf16x8 tr_operand(unsigned img_addr, int p0, int lane) {
    const int g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
    const unsigned a = img_addr + (unsigned)(8 * g + q) * kImgStride + (unsigned)(p0 + 4 * p) * 2u;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)(uintptr_t)a);
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)(uintptr_t)(a + 4 * kImgStride));
    return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}

__global__ __launch_bounds__(kWaves * 64, kWaves / 4) void train16_dw_kernel(
    const DwJob *__restrict__ jobs, int n_jobs, const uint4 *__restrict__ stash, const uint4 *__restrict__ grad,
    int M, float4 *__restrict__ partial) {
    constexpr int CB = 2;
    // An iteration of the main loop covers NB blocks of 32 samples: its cost is mostly fixed (stage -> rendezvous ->
    // transposing reads -> MFMA chain, about 2 us with one block), so two blocks per iteration halve it per sample.
#ifndef ZEST_DW_BLOCKS
#define ZEST_DW_BLOCKS 2
#endif
    constexpr int NB = ZEST_DW_BLOCKS;
    __shared__ __attribute__((aligned(16))) char img[2][2][NB * kImgBytes];   // [buffer][0: out (gradient), 1: in][sample row][...]
    static_assert(sizeof(img) <= 160 * 1024, "LDS");
    // jobs own runs of workgroups in proportion to the tiles they stream per block (tables_for)
    int ji = 0;
    for (int j = 1; j < n_jobs; j++)
        if ((int)blockIdx.x >= jobs[j].wg0) ji = j;
    const DwJob &job = jobs[ji];
    const int part = (int)blockIdx.x - job.wg0, wg_per_job = job.n_wg;
    const int lane = threadIdx.x & 63, col = lane & 15, grp = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long long n_blocks = ((long long)M + 31) / 32;
    const long long per = (n_blocks + wg_per_job - 1) / wg_per_job;
    const long long b0 = part * per, b1 = min(n_blocks, b0 + per);
    // (a workgroup without blocks - tiny batches - still writes its slice of the partial sums: zeros)
    // The job's scalars, read once into registers the compiler cannot re-derive from memory: left as `job.x` it
    // re-loads them with s_load inside the main loop (cheaper than an SGPR to it) - and every such load ends in an
    // s_waitcnt lgkmcnt(0) that also drains the LDS reads in flight: the loop ran at the pace of scalar-cache
    // round trips, whatever else it did (each of: the staging, the MFMAs, the rendezvous could be removed without
    // changing the kernel's 590 us).
    const int n_in = __builtin_amdgcn_readfirstlane(job.n_in_tiles), n_out = __builtin_amdgcn_readfirstlane(job.n_out_tiles);
    const int want_bias = __builtin_amdgcn_readfirstlane(job.want_bias);
    const int out_tile0 = __builtin_amdgcn_readfirstlane(job.out_tile0), in_tile0 = __builtin_amdgcn_readfirstlane(job.in_tile0);
    f32x4 acc[2][16], accb[2];
#pragma unroll
    for (int rt = 0; rt < 2; rt++) {
        accb[rt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ct = 0; ct < 16; ct++) acc[rt][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const bf16x8 ones = {0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80};
    const unsigned img0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char *)&img[0][0][0];
    // items = (tile, column block) of a block: a wave takes items wave, wave + 8, ... (at most 4); the tiles of
    // block b + 1 are requested before block b is computed, so their HBM latency hides behind the MFMAs
    const int items = (n_out + n_in) * CB;
    auto fetch_item = [&](long long b, int it) {
#ifdef ZEST_DW_EXP_NO_LOADS            // timing experiment only: every block reads the first one again (L2 hits)
        b = b0;
#endif
        const int t = it / CB, cb = it % CB;
        const bool is_out = t < n_out;
        const int kt = is_out ? t : t - n_out;
        return stash_load(is_out ? &grad[(((size_t)b * kGradTiles + out_tile0 + kt) * CB + cb) * 64 + lane]
                                 : &stash[(((size_t)b * kStashTiles + in_tile0 + kt) * CB + cb) * 64 + lane]);
    };
    // tiles of blocks b .. b + kDwAhead - 1: requested kDwAhead blocks ahead of their use (HBM latency under the
    // load of 255 workgroups streaming is several microseconds; a block is ~2 us of work)
#ifndef ZEST_DW_AHEAD
#define ZEST_DW_AHEAD 1        // iterations ahead: with two blocks per iteration the same two blocks as before
#endif
    constexpr int kDwAhead = ZEST_DW_AHEAD;          // iterations (of NB blocks) ahead
    uint4 pre[kDwAhead][NB][4];
    // a block past the end of this workgroup's range contributes nothing: its gradient tiles are zero
    auto fetch_or_zero = [&](long long b, int it) { return b < b1 ? fetch_item(b, it) : uint4{0u, 0u, 0u, 0u}; };
#pragma unroll
    for (int d = 0; d < kDwAhead; d++)
#pragma unroll
        for (int sb = 0; sb < NB; sb++)
#pragma unroll
            for (int i = 0; i < 4; i++)
                if (wave + 8 * i < items) pre[d][sb][i] = fetch_or_zero(b0 + d * NB + sb, wave + 8 * i);
    int buf = 0;
    for (long long b = b0; b < b1; b += NB, buf ^= 1) {
        char *im_out = img[buf][0], *im_in = img[buf][1];
        // ---- stage the blocks' tiles: image row = sample, 16 B at position 8 g of the k-tile
#pragma unroll
        for (int sb = 0; sb < NB; sb++) {
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int it = wave + 8 * i;
                if (it >= items) continue;
                const int t = it / CB, cb = it % CB;
                const bool is_out = t < n_out;
                const int kt = is_out ? t : t - n_out;
#ifdef ZEST_DW_EXP_NO_STAGE            // timing experiment only: the tiles never enter LDS (their loads become dead code)
                if (b == b0)
#endif
                *reinterpret_cast<uint4 *>((is_out ? im_out : im_in) + (32 * sb + 16 * cb + col) * kImgStride + kt * 64 + grp * 16) =
                    pre[0][sb][i];
            }
        }
#ifndef ZEST_DW_EXP_NO_SYNC             // (defined: timing experiment only, results are wrong)
        __syncthreads();
#endif
#pragma unroll
        for (int sb = 0; sb < NB; sb++) {
#pragma unroll
            for (int i = 0; i < 4; i++) {
#pragma unroll
                for (int d = 0; d + 1 < kDwAhead; d++) pre[d][sb][i] = pre[d + 1][sb][i];
                if (wave + 8 * i < items) pre[kDwAhead - 1][sb][i] = fetch_or_zero(b + (long long)kDwAhead * NB + sb, wave + 8 * i);
            }
        }
#ifdef ZEST_DW_EXP_NO_MFMA             // timing experiment only
        if (b == b0 && wave < n_out) {
#else
        if (wave < n_out) {
#endif
#pragma unroll
            for (int sb = 0; sb < NB; sb++) {
                const unsigned a_out = img0 + (unsigned)(buf * 2) * (NB * kImgBytes) + (unsigned)sb * kImgBytes;
                const unsigned a_in = a_out + NB * kImgBytes;
                bf16x8 A[2];
#pragma unroll
                for (int rt = 0; rt < 2; rt++) A[rt] = tr_operand(a_out, 32 * wave + 16 * rt, lane);
                // All 16 column tiles, whatever the job's number of input k-tiles: straight-line code, so the
                // transposing reads run ahead of the MFMAs (with a test of ct against n_in in front of each pair of
                // reads hipcc waited for every read right behind it: 16 exposed LDS latencies per block, the kernel's
                // bottleneck).  Tiles past 2 n_in multiply whatever the image holds there into accumulators that are
                // never written out.
                // The reads are pipelined by hand, in source order, kAheadB column tiles ahead of the MFMAs that use them.
                constexpr int kAheadB = 3;
                bf16x8 Bq[kAheadB];
#pragma unroll
                for (int ct = 0; ct < kAheadB; ct++) Bq[ct] = tr_operand(a_in, 16 * ct, lane);
#pragma unroll
                for (int ct = 0; ct < 16; ct++) {
                    const bf16x8 B = Bq[ct % kAheadB];
                    if (ct + kAheadB < 16) Bq[ct % kAheadB] = tr_operand(a_in, 16 * (ct + kAheadB), lane);
#pragma unroll
                    for (int rt = 0; rt < 2; rt++) acc[rt][ct] = mfma16<EP>(A[rt], B, acc[rt][ct]);
                    __builtin_amdgcn_sched_barrier(0);      // keep the read-ahead distance (hipcc otherwise sinks each read
                }                                           // to just in front of its use to save four registers)
                if (want_bias) {
#pragma unroll
                    for (int rt = 0; rt < 2; rt++) accb[rt] = mfma16<EP>(A[rt], ones, accb[rt]);
                }
            }
        }
    }
    if (wave >= n_out) return;
    // ---- this workgroup's slice of the sums goes to its own 272 KB of the partial buffer, accumulator by
    // accumulator (1 KiB per wave-instruction); train16_dw_reduce_kernel adds the slices of a job and scatters
    // the totals into the fp32 gradients.  (Float atomics from all 256 workgroups straight into the gradients -
    // 16.8 M of them - cost 88 of this kernel's 425 us.)
    float4 *mine = partial + ((size_t)blockIdx.x * kWaves + wave) * kDwSlots * 64 + lane;
#ifdef ZEST_DW_EXP_NO_ATOMICS          // timing experiment only: results are wrong
    if (acc[0][0][0] != 123456.0f) return;
#endif
#pragma unroll
    for (int rt = 0; rt < 2; rt++) {
#pragma unroll
        for (int ct = 0; ct < 16; ct++)
            mine[(rt * 16 + ct) * 64] = make_float4(acc[rt][ct][0], acc[rt][ct][1], acc[rt][ct][2], acc[rt][ct][3]);
        mine[(32 + rt) * 64] = make_float4(accb[rt][0], accb[rt][1], accb[rt][2], accb[rt][3]);
    }
}

// One thread per (job, wave, accumulator, lane): the sum over the job's workgroups of that accumulator's four values,
// added to the fp32 gradients through the job's position maps.  Every gradient element belongs to exactly one
// accumulator element of one job (the two jobs of the skip layer and of the view layer own different columns), so
// plain read-modify-write is enough - and the result does not depend on the order workgroups finish in.
struct GradPtrs {                    // the caller's gradient tensors, passed BY VALUE as a kernel argument (208 bytes)
    float *p[2 * ZEST_P_COUNT];
};
__global__ __launch_bounds__(256) void train16_dw_reduce_kernel(const DwJob *__restrict__ jobs, int n_jobs,
                                                                const float4 *__restrict__ partial, const GradPtrs g) {
    float *const *g_params = g.p;
    const int per_job = kWaves * kDwSlots * 64;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_jobs * per_job) return;
    const DwJob &job = jobs[i / per_job];
    const int r_ = i % per_job, wave = r_ / (kDwSlots * 64), slot = r_ / 64 % kDwSlots, lane = r_ % 64;
    const int col = lane & 15, grp = lane >> 4;
    const bool is_bias = slot >= 32;
    const int rt = is_bias ? slot - 32 : slot / 16, ct = slot % 16;
    if (wave >= job.n_out_tiles || (is_bias ? !job.want_bias || col != 0 : ct >= 2 * job.n_in_tiles)) return;
    const int ci = is_bias ? 0 : job.in_col[16 * ct + col];
    if (ci < 0) return;
    float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int w = 0; w < job.n_wg; w++) {
        const float4 v = partial[((size_t)(job.wg0 + w) * kWaves + wave) * kDwSlots * 64 + slot * 64 + lane];
        sum.x += v.x, sum.y += v.y, sum.z += v.z, sum.w += v.w;
    }
    const float vals[4] = {sum.x, sum.y, sum.z, sum.w};
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int po = 32 * wave + 16 * rt + 4 * grp + r;
        const int ps = job.out_slot[po], row = job.out_row[po];
        if (ps < 0) continue;
        if (is_bias) g_params[2 * ps + 1][row] += vals[r];
        else g_params[2 * ps][(size_t)row * job.ld + job.col0 + ci] += vals[r];
    }
}

}  // namespace

namespace zest {

// partial sums of the weight kernel: one slice per workgroup (one workgroup per CU, at least one per job: 16 jobs at most)
size_t train16_dw_partial_bytes(int cus) { return (size_t)(cus < 16 ? 16 : cus) * kWaves * kDwSlots * 1024; }

// d W of every op from the two stashes: the weight kernel (n_wg workgroups, one per CU) and the reduce kernel that adds a
// job's slices and scatters them into the caller's gradient tensors (g_params: 2 * ZEST_P_COUNT device pointers, host array)
int train16_dw_launch(const DwJob *jobs, int n_jobs, int n_wg, const uint4 *stash_tiles, const uint4 *grad, int M,
                      float4 *partial, float *const *g_params, hipStream_t st) {
    // the gradient pointers travel in the kernel's argument block (an upload from the caller's pageable
    // table was a staging copy per backward call)
    GradPtrs gp;
    for (int i = 0; i < 2 * ZEST_P_COUNT; i++) gp.p[i] = g_params[i];
    // one resident workgroup per CU (128 accumulator registers per lane): more workgroups would only run in a
    // second round and add their 272 KB of partial sums each
    hipLaunchKernelGGL(train16_dw_kernel, dim3(n_wg), dim3(kWaves * 64), 0, st, jobs, n_jobs, stash_tiles, grad, M, partial);
    hipLaunchKernelGGL(train16_dw_reduce_kernel, dim3(zest_div_up(n_jobs * kWaves * kDwSlots * 64, 256)), dim3(256), 0, st,
                       jobs, n_jobs, (const float4 *)partial, gp);
    return 0;
}

}  // namespace zest
