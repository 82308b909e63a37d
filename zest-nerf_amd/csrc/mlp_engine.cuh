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
// The weight stream (units of 1 KiB in consumption order, mlp_plan.h) comes from a `Tiles`
// source: GlobalTiles reads units straight from global memory (standalone MLP kernel);
// RingTiles (fused renderer) reads them from an LDS ring that the workgroup's waves keep
// filled with LDS-DMA (`global_load_lds`), several chunks ahead of the MFMAs.
#pragma once
#include <hip/hip_bf16.h>
#include "mlp_plan.h"
#include "zest_common.cuh"

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) float v4f;
typedef __attribute__((ext_vector_type(4))) unsigned v4u;
typedef const __attribute__((address_space(1))) v4u *gptr_u4;        // global memory, 16-byte units

namespace zest {

// units in the stream of one net (mirror of build_plan for ORDER_ACC, checked on the host)
constexpr int stream_units_raw(int nt_pts, int nt_feat) {
    const int mod = nt_feat;                       // modulation tiles per row block (0 = off)
    return 8 * (1 + mod + nt_pts) + 6 * 8 * (1 + mod + 16) + 8 * (1 + mod + nt_pts + 16)   // trunk
           + (1 + 16) + 8 * (1 + 16) + 4 * (1 + 16 + 2) + (1 + 8);                         // heads
}
constexpr int stream_units(int nt_pts, int nt_feat) {
    return (stream_units_raw(nt_pts, nt_feat) + kStreamAlign - 1) / kStreamAlign * kStreamAlign;
}

__device__ __forceinline__ f32x16 f32x16_from(const v4f b0, const v4f b1, const v4f b2, const v4f b3) {
    f32x16 r;
    r[0] = b0.x, r[1] = b0.y, r[2] = b0.z, r[3] = b0.w, r[4] = b1.x, r[5] = b1.y, r[6] = b1.z, r[7] = b1.w;
    r[8] = b2.x, r[9] = b2.y, r[10] = b2.z, r[11] = b2.w, r[12] = b3.x, r[13] = b3.y, r[14] = b3.z, r[15] = b3.w;
    return r;
}

// ---- weight source 1: global memory --------------------------------------------------------
struct GlobalTiles {
    gptr_u4 base;          // wave-uniform: first unit of the stream
    int lane, half;
    __device__ __forceinline__ bf16x8 load(int unit) const {
        const v4u v = base[(size_t)unit * 64 + lane];
        return *reinterpret_cast<const bf16x8 *>(&v);
    }
    // bias block `which` (0 = op bias, 1 = modulation bias) of header unit `unit`
    __device__ __forceinline__ f32x16 load_bias(int unit, int which) const {
        const __attribute__((address_space(1))) v4f *b =
            (const __attribute__((address_space(1))) v4f *)(base + (size_t)unit * 64) + which * 8 + half * 4;
        return f32x16_from(b[0], b[1], b[2], b[3]);
    }
    __device__ __forceinline__ void finish(int, int) const {}
};

// ---- weight source 2: LDS ring fed by LDS-DMA -----------------------------------------------
// The ring holds kRingUnits KiB = kSlots chunks of kChunk units.  All NW waves of the
// workgroup walk the stream in step.  On entering chunk c every wave waits until its own DMA
// pieces of chunk c have landed (counted vmcnt: the younger kAhead-1 chunks stay in flight),
// meets the others at a barrier - after which chunk c is complete and chunk c-1 is no longer
// read by anyone - and issues its share of chunk c+kAhead into the slot that frees.  The
// stream length is a multiple of the ring size, so unit u always sits at ring offset
// u % kRingUnits and passes follow each other without draining the ring.
#ifndef ZEST_CHUNK
#define ZEST_CHUNK 16      // units per chunk (one rendezvous per chunk)
#define ZEST_SLOTS 8       // chunks in the ring
#define ZEST_AHEAD 4       // chunks of DMA kept in flight
#endif
constexpr int kChunk = ZEST_CHUNK, kSlots = ZEST_SLOTS, kRingUnits = kChunk * kSlots, kAhead = ZEST_AHEAD;
static_assert(kStreamAlign % kRingUnits == 0, "stream padding must be a multiple of the ring size");
static_assert(kSlots >= kAhead + 1, "a slot is refilled while the previous chunk may still be read");

template <int NW, int UNITS_A, int UNITS_B>
struct RingTiles {
    static constexpr int kPieces = kChunk / NW;             // DMA pieces per wave per chunk
    static constexpr int kChunksA = UNITS_A / kChunk, kChunks = (UNITS_A + UNITS_B) / kChunk;
    static_assert(kChunk % NW == 0 && UNITS_A % kRingUnits == 0 && UNITS_B % kRingUnits == 0, "");
    char *ring;            // LDS, kRingUnits KiB, 16-byte aligned
    gptr_u4 src_a, src_b;  // the two nets' streams (src_b unused when UNITS_B == 0)
    int lane, half, wave;  // wave: provably uniform (readfirstlane)
    unsigned voff;         // (wave * kPieces * 64 + lane) * 16: this lane's byte offset in a chunk
    unsigned lds_wave_base; // LDS byte address of the ring + wave * kPieces * 1024
#ifdef ZEST_STAMPS         // diagnostic build: cycles spent in enter_chunk (DMA wait + barrier + issue)
    mutable unsigned long long t_wait = 0, t_issue = 0;
#endif

    // LDS-DMA of this wave's pieces of a chunk.  Issued through inline asm so that only the
    // counted waits in enter_chunk govern it: hipcc would otherwise drain vmcnt(0) in front of
    // every ds_read it cannot prove disjoint from the DMA destination
    // (cdna_hip_programming.md 5.7).  M0 = this wave's LDS destination; the instruction offset
    // advances both the global source and the LDS destination, so one M0 write serves all of
    // the wave's pieces.  M0 is not restored: hipcc keeps nothing live in it across an asm
    // statement that declares it clobbered.  Addressing is scalar base + 32-bit lane offset,
    // and the chunk's byte offset is made opaque so the 64-bit adds stay here instead of being
    // hoisted out of the pass loop for all chunks at once (which spills SGPRs to VGPR lanes).
    __device__ __forceinline__ void issue(int chunk) const {      // chunk: compile-time after unrolling
        const int c = chunk % kChunks;
        unsigned off = (unsigned)((c < kChunksA ? c : c - kChunksA) * kChunk * 1024);
        asm volatile("" : "+s"(off));
        const __attribute__((address_space(1))) char *sb =
            (const __attribute__((address_space(1))) char *)(c < kChunksA ? src_a : src_b) + off;
        const unsigned dst = lds_wave_base + (c % kSlots) * kChunk * 1024;
        static_assert(kPieces == 1 || kPieces == 2 || kPieces == 4, "pieces per wave");
        if (kPieces == 1)
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                         :: "v"(voff), "s"(sb), "s"(dst) : "memory", "m0");
        else if (kPieces == 2)
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1\n\t"
                         "global_load_lds_dwordx4 %0, %1 offset:1024"
                         :: "v"(voff), "s"(sb), "s"(dst) : "memory", "m0");
        else
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1\n\t"
                         "global_load_lds_dwordx4 %0, %1 offset:1024\n\t"
                         "global_load_lds_dwordx4 %0, %1 offset:2048\n\t"
                         "global_load_lds_dwordx4 %0, %1 offset:3072"
                         :: "v"(voff), "s"(sb), "s"(dst) : "memory", "m0");
    }
    __device__ __forceinline__ void prologue() const {
#pragma unroll
        for (int c = 0; c < kAhead; c++) issue(c);
    }
    __device__ __forceinline__ void enter_chunk(int chunk) const {
#ifdef ZEST_STAMPS
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#endif
        // all but the youngest (kAhead-1)*kPieces of this wave's DMA pieces have landed
        asm volatile("s_waitcnt vmcnt(%0)" ::"i"((kAhead - 1) * kPieces) : "memory");
#ifndef ZEST_EXPERIMENT_NO_BARRIER      // timing experiment only: results are wrong without it
        __builtin_amdgcn_s_barrier();
#endif
        asm volatile("" ::: "memory");
#ifdef ZEST_STAMPS
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
#endif
#ifndef ZEST_EXPERIMENT_NO_DMA          // timing experiment only
        issue(chunk + kAhead);
#endif
#ifdef ZEST_STAMPS
        t_wait += t1 - t0, t_issue += __builtin_amdgcn_s_memtime() - t1;
#endif
    }
    __device__ __forceinline__ void touch(int unit) const {
        if (unit % kChunk == 0) enter_chunk(unit / kChunk);
    }
    __device__ __forceinline__ bf16x8 load(int unit) const {
        touch(unit);
#ifdef ZEST_EXPERIMENT_NO_LDSREAD       // timing experiment only: one read per chunk
        const v4u v = *reinterpret_cast<const v4u *>(ring + (unit / kChunk * kChunk % kRingUnits) * 1024 + lane * 16);
#else
        const v4u v = *reinterpret_cast<const v4u *>(ring + (unit % kRingUnits) * 1024 + lane * 16);
#endif
        return *reinterpret_cast<const bf16x8 *>(&v);
    }
    __device__ __forceinline__ f32x16 load_bias(int unit, int which) const {
        if (which == 0) touch(unit);          // the modulation block (which = 1) is read right after
        const v4f *b = reinterpret_cast<const v4f *>(ring + (unit % kRingUnits) * 1024 + which * 128 + half * 64);
        return f32x16_from(b[0], b[1], b[2], b[3]);
    }
    // walk the padding [unit, end) of a net's stream so every chunk is entered exactly once
    __device__ __forceinline__ void finish(int unit, int end) const {
#pragma unroll
        for (int u = (unit + kChunk - 1) / kChunk * kChunk; u < end; u += kChunk) enter_chunk(u / kChunk);
    }
    __device__ __forceinline__ void drain() const { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
};

// two fp32 -> packed bf16 pair (round to nearest even): one v_cvt_pk_bf16_f32.  Written as a
// vector conversion the compiler understands - an inline-asm form would read MFMA results
// without the wait states hipcc inserts for its own instructions (cdna guide 5.7 item 2) -
// and not as __float22bfloat162_rn, which costs 2 cvt + shift + or on MFMA tile elements.
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
__device__ __forceinline__ unsigned pack_bf16(float lo, float hi) {
    const f32x2_t f = {lo, hi};
    const bf16x2_t r = __builtin_convertvector(f, bf16x2_t);
    return *reinterpret_cast<const unsigned *>(&r);
}

// relu as one v_med3_f32: max(v, 0) for every finite activation (fmaxf costs a second,
// canonicalising v_max in front; med3 against a finite bound is not folded back into it)
__device__ __forceinline__ float relu1(float v) { return __builtin_amdgcn_fmed3f(v, 0.0f, 3.0e38f); }

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

#ifndef ZEST_PREFETCH
#define ZEST_PREFETCH 3        // weight tiles kept in flight ahead of the MFMA that consumes them
#endif
constexpr int kPrefetch = ZEST_PREFETCH;

// What is fetched ahead for one row block: its bias initialisers and its first tiles.
template <bool MOD>
struct RowBlockPre {
    f32x16 bias, mbias;
    bf16x8 win[kPrefetch];
};

// One Linear: NJB row blocks over the operand [A (NTA tiles) | B (NTB tiles)].
//   MOD:  tile is modulated by the feature operand (NTF tiles) with the op's modulation tiles
//   MODE: 0 = produce operand tiles into `out` (2 per row block), 1 = keep the accumulator of
//         row block 0 in `keep` (head / rgb tiles)
// The LDS reads are software-pipelined by hand, in source order, because hipcc leaves them
// where they are written in a block this large: every tile is requested kPrefetch tiles before
// the MFMA that uses it, and the NEXT row block's header and first tiles are requested before
// the CURRENT epilogue, whose ~40 VALU instructions cover their latency.
template <int NB, int NJB, int NTA, int NTB, bool MOD, int NTF, bool RELU, int MODE, class Tiles>
__device__ __forceinline__ void engine_layer(const Tiles &tiles, int &unit, bool v2,
                                             const OpArr<NTA> (&opa)[NB], const OpArr<NTB> (&opb)[NB],
                                             const OpArr<NTF> (&opf)[NB], OpArr<16> (&out)[NB],
                                             f32x16 (&keep)[NB]) {
    constexpr int NM = MOD ? NTF : 0, T = NM + NTA + NTB;       // tiles per row block
    static_assert(T >= 1, "empty layer");
    auto preload = [&](RowBlockPre<MOD> &p, int u0) {           // u0: the row block's header unit
        p.bias = tiles.load_bias(u0, 0);
        if (MOD) p.mbias = tiles.load_bias(u0, 1);
#pragma unroll
        for (int k = 0; k < kPrefetch; k++)
            if (k < T) p.win[k] = tiles.load(u0 + 1 + k);
    };
    RowBlockPre<MOD> pre;
    preload(pre, unit);
#pragma unroll
    for (int jb = 0; jb < NJB; jb++) {
        const int u0 = unit;
        f32x16 acc[NB], macc[NB];
        bf16x8 win[kPrefetch];
#pragma unroll
        for (int k = 0; k < kPrefetch; k++) win[k] = pre.win[k];
#pragma unroll
        for (int nb = 0; nb < NB; nb++) {
            acc[nb] = pre.bias;
            if (MOD) macc[nb] = pre.mbias;
        }
#pragma unroll
        for (int k = 0; k < T; k++) {
            const bf16x8 a = win[k % kPrefetch];
#pragma unroll
            for (int nb = 0; nb < NB; nb++) {
                if (k < NM)
                    macc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, opf[nb].t[k < NM ? k : 0], macc[nb], 0, 0, 0);
                else if (k < NM + NTA)
                    acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, opa[nb].t[(k >= NM && k < NM + NTA) ? k - NM : 0], acc[nb], 0, 0, 0);
                else
                    acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, opb[nb].t[k >= NM + NTA ? k - NM - NTA : 0], acc[nb], 0, 0, 0);
            }
            if (k + kPrefetch < T) win[k % kPrefetch] = tiles.load(u0 + 1 + k + kPrefetch);
        }
        unit = u0 + 1 + T;
        if (jb + 1 < NJB) preload(pre, unit);                   // in flight during the epilogue
#pragma unroll
        for (int nb = 0; nb < NB; nb++) {
            f32x16 v = acc[nb];
            if (MOD) {
#pragma unroll
                for (int i = 0; i < 16; i++) v[i] = v2 ? v[i] + macc[nb][i] : v[i] * macc[nb][i];
            }
            if (RELU) {
#pragma unroll
                for (int i = 0; i < 16; i++) v[i] = relu1(v[i]);
            }
            if (MODE == 0)
                acc_to_operand(v, out[nb].t[2 * jb], out[nb].t[2 * jb + 1]);
            else if (jb == 0)
                keep[nb] = v;
        }
    }
}

// The whole network for NB column blocks, reading the stream from unit `unit` on (advanced to
// the end of the net's padded stream).  pts/feat: encoder operands in plan slot order;
// `views_fn(views)` builds the direction operand when it is first needed (op 10) so that it does
// not occupy registers through the trunk.  Results: head tile (row 0 alpha, rows 1.. extra
// heads) and rgb tile (rows 0-2), raw.
template <int NB, int NT_PTS, bool MOD, int NT_FEAT, class Tiles, class ViewsFn>
__device__ __forceinline__ void engine_forward(const Tiles &tiles, int &unit, bool v2,
                                               const OpArr<NT_PTS> (&pts)[NB],
                                               const OpArr<NT_FEAT> (&feat)[NB], ViewsFn views_fn,
                                               f32x16 (&head)[NB], f32x16 (&rgb)[NB]) {
    OpArr<16> hA[NB], hB[NB];
    OpArr<0> none[NB];
    f32x16 unused[NB];
    const int unit0 = unit;
    engine_layer<NB, 8, NT_PTS, 0, MOD, NT_FEAT, true, 0>(tiles, unit, v2, pts, none, feat, hA, unused);
    engine_layer<NB, 8, 16, 0, MOD, NT_FEAT, true, 0>(tiles, unit, v2, hA, none, feat, hB, unused);
    engine_layer<NB, 8, 16, 0, MOD, NT_FEAT, true, 0>(tiles, unit, v2, hB, none, feat, hA, unused);
    engine_layer<NB, 8, 16, 0, MOD, NT_FEAT, true, 0>(tiles, unit, v2, hA, none, feat, hB, unused);
    engine_layer<NB, 8, 16, 0, MOD, NT_FEAT, true, 0>(tiles, unit, v2, hB, none, feat, hA, unused);
    engine_layer<NB, 8, NT_PTS, 16, MOD, NT_FEAT, true, 0>(tiles, unit, v2, pts, hA, feat, hB, unused);
    engine_layer<NB, 8, 16, 0, MOD, NT_FEAT, true, 0>(tiles, unit, v2, hB, none, feat, hA, unused);
    engine_layer<NB, 8, 16, 0, MOD, NT_FEAT, true, 0>(tiles, unit, v2, hA, none, feat, hB, unused);
    // trunk output in hB
    engine_layer<NB, 1, 16, 0, false, NT_FEAT, false, 1>(tiles, unit, v2, hB, none, feat, hA, head);
    engine_layer<NB, 8, 16, 0, false, NT_FEAT, false, 0>(tiles, unit, v2, hB, none, feat, hA, unused);
    OpArr<2> views[NB];
    views_fn(views);
    engine_layer<NB, 4, 16, 2, false, NT_FEAT, true, 0>(tiles, unit, v2, hA, views, feat, hB, unused);
    // rgb: 128 hidden features = first 8 tiles of hB
    OpArr<8> h128[NB];
#pragma unroll
    for (int nb = 0; nb < NB; nb++)
#pragma unroll
        for (int k = 0; k < 8; k++) h128[nb].t[k] = hB[nb].t[k];
    engine_layer<NB, 1, 8, 0, false, NT_FEAT, false, 1>(tiles, unit, v2, h128, none, feat, hA, rgb);
    tiles.finish(unit, unit0 + stream_units(NT_PTS, MOD ? NT_FEAT : 0));
    unit = unit0 + stream_units(NT_PTS, MOD ? NT_FEAT : 0);
}

// standalone launcher (mlp.hip -> zest_mlp_fwd)
int mlp_bf16_launch(const MlpPlan &p, const void *tiles, const float *x, int M, float *out,
                    hipStream_t stream);

}  // namespace zest
