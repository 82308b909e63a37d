// MFMA engine for the width-256 NeRF MLP (bf16 / fp16 / split-fp16 operands): activations stay in registers.
//
// Orientation: Y^T = W X^T with v_mfma_f32_16x16x32_bf16 (under random data the chip holds a higher clock
// on this shape: at identical operand traffic it delivered 5-15 % more FLOP/s here than 32x32x16, cf.
// MI355X_MICROARCH.md "Shape"; its price is issue room - two plain VALU slots per MFMA - so the
// epilogues count their instructions).  A wave owns NB blocks of 32 samples =
// 2 NB column blocks of 16: lane l serves sample column l & 15 of every column block, as group
// g = l >> 4 of four.  The A operand is a pre-packed 1 KiB weight tile (16 output rows x 32
// k-positions, mlp_plan.h) and feeds one MFMA per column block; the B operand is one k-tile of
// a column block's activations, 8 bf16 per lane (positions 8g .. 8g+7).  Two finished 16x16
// accumulator tiles of a row block (4 fp32 per lane each: rows 4g .. 4g+3), biased via their
// initial value, modulated, rectified and rounded to bf16 pairs, ARE one k-tile of the next
// op's B operand: no LDS round trip and no cross-lane traffic between layers
// (cdna_hip_programming.md section 3, "an accumulator tile as the next MFMA's operand"; the k
// permutation that implies is folded into the packed weights).  Row blocks are the outer loop,
// so only four small accumulators (plus four for the modulation) per 32 samples are live next
// to the 64-register input and output operands.
//
// The weight stream (units of 1 KiB in consumption order, mlp_plan.h) comes from a `Tiles`
// source: RingTiles (fused renderer and standalone MLP kernel) reads them from an LDS ring that
// the workgroup's waves keep filled with LDS-DMA (`global_load_lds`), several chunks ahead of
// the MFMAs.  (Reading every unit from global memory in every wave, the first version of the
// standalone kernel, is 2.3x slower.)
#pragma once
#include <hip/hip_bf16.h>
#include "mlp_plan.h"
#include "zest_common.cuh"

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) float v4f;
typedef __attribute__((ext_vector_type(4))) unsigned v4u;
typedef const __attribute__((address_space(1))) v4u *gptr_u4;        // global memory, 16-byte units
typedef __attribute__((address_space(3))) int lds_int;                // an int in LDS

namespace zest {

// units in the stream of one net (mirror of build_plan for ORDER_ACC, checked on the host)
// (parts: stream units per weight tile, 2 for the split fp16 pair)
constexpr int stream_units_raw(int nt_pts, int nt_feat, int parts = 1) {
    const int mod = nt_feat * parts, P = nt_pts * parts, H = 16 * parts;   // mod: modulation units per row block (0 = off)
    return 8 * (1 + mod + P) + 6 * 8 * (1 + mod + H) + 8 * (1 + mod + P + H)   // trunk
           + (1 + H) + 8 * (1 + H) + 4 * (1 + H + 2 * parts) + (1 + H / 2);     // heads
}
constexpr int stream_units(int nt_pts, int nt_feat, int parts = 1) {
    return (stream_units_raw(nt_pts, nt_feat, parts) + kStreamAlign - 1) / kStreamAlign * kStreamAlign;
}

// ---- weight source: LDS ring fed by LDS-DMA -----------------------------------------------
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

#ifndef ZEST_DEFER_DMA
#define ZEST_DEFER_DMA 0   // 1: weight DMA issued in the row-block epilogues instead of at the rendezvous (see flush())
#endif
#ifndef ZEST_ISSUERS
#define ZEST_ISSUERS 8     // waves of the workgroup that issue the weight DMA (8: every wave its share; 4: waves 0-3,
#endif                     // one per SIMD, take all of it and their SIMD partners 4-7 none - measured in DESIGN.md section 4)
template <int NW, int UNITS_A, int UNITS_B>
struct RingTiles {
    static constexpr int kIssuers = ZEST_ISSUERS < NW ? ZEST_ISSUERS : NW;
    static constexpr int kPieces = kChunk / kIssuers;       // DMA pieces per issuing wave per chunk
    static constexpr int kChunksA = UNITS_A / kChunk, kChunks = (UNITS_A + UNITS_B) / kChunk;
    static_assert(kChunk % kIssuers == 0 && UNITS_A % kRingUnits == 0 && UNITS_B % kRingUnits == 0, "");
    char *ring;            // LDS, kRingUnits KiB, 16-byte aligned
    gptr_u4 src_a, src_b;  // the two nets' streams (src_b unused when UNITS_B == 0)
    int lane, grp, wave;   // grp = lane >> 4; wave: provably uniform (readfirstlane)
    unsigned voff;         // (wave * kPieces * 64 + lane) * 16: this lane's byte offset in a chunk
    unsigned lds_wave_base; // LDS byte address of the ring + wave * kPieces * 1024
#ifdef ZEST_RING_FLAGS     // rendezvous-free ring: per-slot counters in LDS instead of a barrier per chunk
    unsigned flags;        // LDS byte address of 2 * kSlots ints, see off_landed / off_done
    mutable int gen_base = 0;      // ring generations completed by earlier passes of this workgroup
    mutable int poisoned = 0;      // a bounded wait ran out (never in a correct run): results are invalid
#endif
#ifdef ZEST_STAMPS         // diagnostic build: cycles spent in enter_chunk (DMA wait + barrier + issue)
    mutable unsigned long long t_wait = 0, t_issue = 0, t_vm = 0;   // t_vm: the vmcnt part of t_wait
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
        if (kIssuers < NW && wave >= kIssuers) return;             // wave-uniform: a scalar branch
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
#ifdef ZEST_RING_FLAGS
    // Two monotonic counters per slot, each bumped once per wave and use of the slot:
    //   landed[s]: this wave's DMA pieces of the chunk now in slot s are in LDS (published one
    //              chunk early, when the wave enters the chunk before it)
    //   done[s]:   this wave has issued its last read of the chunk in slot s
    // A wave may read chunk q once landed[q % kSlots] == NW * (generation(q) + 1), and may
    // refill slot t once done[t] == NW * generation(new occupant).  landed[s] and the done
    // counter the same enter_chunk needs (slot s + kAhead) sit next to each other: one 8-byte
    // read.  No wave ever waits for the others to ARRIVE anywhere: the two waves of a SIMD
    // drift apart and one's MFMAs cover the other's waits, DMA issue and epilogues.
    static_assert(kSlots >= kAhead + 2, "flag ring: a slot is refilled while laggards read its neighbours");
    static constexpr int off_landed(int slot) { return 8 * (slot % kSlots); }
    static constexpr int off_done(int slot) { return 8 * ((slot % kSlots - kAhead + kSlots) % kSlots) + 4; }
    static __device__ void init_flags(int *f) {             // before the workgroup's first barrier
        if (threadIdx.x < 2 * kSlots) f[threadIdx.x] = 0;
        __syncthreads();
        // the very first enter_chunk "finishes" a chunk that never was: pre-charge its counter
        if (threadIdx.x == 0) f[off_done(kSlots - 1) / 4] = -NW;
    }
    // The flag traffic is inline asm on purpose: a C++ wait loop in every enter_chunk puts ~80
    // loops into the unrolled network, after which hipcc no longer keeps the operand arrays in
    // registers; an asm block is straight-line code to it.  Lane 0 does the adds (EXEC = 1).
    // No VGPR is first written inside an asm block: hipcc inserts no MFMA->VALU wait states for
    // asm (cdna guide 5.7), and a scratch register it hands out may be the destination of an MFMA
    // still in flight (seen: the adds after the last rgb MFMA added an activation instead of 1).
    // Constants come in as inputs and the read targets are initialised in C++.
    template <int OFF_A, int OFF_B>
    __device__ __forceinline__ void bump2() const {
        unsigned long long save;
        asm volatile("s_mov_b64 %0, exec\n\ts_mov_b64 exec, 1\n\t"
                     "ds_add_u32 %2, %1 offset:%3\n\tds_add_u32 %2, %1 offset:%4\n\ts_mov_b64 exec, %0"
                     : "=&s"(save) : "v"(1u), "v"(flags), "i"(OFF_A), "i"(OFF_B) : "memory");
    }
    __device__ __forceinline__ void prologue() const {
#pragma unroll
        for (int c = 0; c < kAhead; c++) issue(c);
        asm volatile("s_waitcnt vmcnt(%0)" ::"i"((kAhead - 1) * kPieces) : "memory");
        unsigned long long save;
        asm volatile("s_mov_b64 %0, exec\n\ts_mov_b64 exec, 1\n\t"
                     "ds_add_u32 %2, %1 offset:%3\n\ts_mov_b64 exec, %0"
                     : "=&s"(save) : "v"(1u), "v"(flags), "i"(off_landed(0)) : "memory");
    }
    __device__ __forceinline__ void next_pass() const { gen_base += kChunks / kSlots; }
    // The two counters a chunk entry checks are read kPreRead units ahead of it (pre_read), with
    // the tile reads around them hiding the LDS latency; the check itself is then scalar work
    // on registers.  Only if the early values are short of the targets does the wave fall into
    // the (bounded) re-read loop.
    static constexpr int kPreRead = 5;
    typedef int v2i __attribute__((ext_vector_type(2)));
    mutable v2i flag_pre = {0, 0};
    __device__ __forceinline__ void pre_read(int chunk) const {
        flag_pre = *(const volatile __attribute__((address_space(3))) v2i *)(flags + off_landed(chunk));
    }
    template <int OFF>
    __device__ __forceinline__ void await(int need_l, int need_d) const {
        unsigned va = (unsigned)flag_pre.x, vb = (unsigned)flag_pre.y;
        int sa, sb, tries = 1 << 14;       // bounded: a protocol bug must not hang the GPU
        asm volatile("0:\n\t"
                     "v_readfirstlane_b32 %2, %0\n\t"
                     "v_readfirstlane_b32 %3, %1\n\t"
                     "s_nop 1\n\t"
                     "s_cmp_ge_i32 %2, %6\n\t"
                     "s_cselect_b32 %2, 1, 0\n\t"
                     "s_cmp_ge_i32 %3, %7\n\t"
                     "s_cselect_b32 %3, 1, 0\n\t"
                     "s_and_b32 %2, %2, %3\n\t"
                     "s_cbranch_scc1 1f\n\t"
                     "s_sub_u32 %4, %4, 1\n\t"
                     "s_cmp_eq_u32 %4, 0\n\t"
                     "s_cbranch_scc1 1f\n\t"
                     "s_sleep 1\n\t"
                     "ds_read_b32 %0, %5 offset:%8\n\t"
                     "ds_read_b32 %1, %5 offset:%9\n\t"
                     "s_waitcnt lgkmcnt(0)\n\t"
                     "s_branch 0b\n\t"
                     "1:"
                     : "+v"(va), "+v"(vb), "=&s"(sa), "=&s"(sb), "+s"(tries)
                     : "v"(flags), "s"(need_l), "s"(need_d), "i"(OFF), "i"(OFF + 4)
                     : "memory", "scc");
        if (tries == 0) poisoned = 1;
    }
    __device__ __forceinline__ void enter_chunk(int chunk) const {
#ifdef ZEST_STAMPS
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#endif
        // own pieces of the NEXT chunk have landed (chunks chunk+2 .. chunk+kAhead-1 stay in flight);
        // all reads of the previous chunk are issued (LDS operations of a wave execute in order)
        asm volatile("s_waitcnt vmcnt(%0)" ::"i"((kAhead - 2) * kPieces) : "memory");
#ifdef ZEST_STAMPS
        t_vm += __builtin_amdgcn_s_memtime() - t0;
#endif
        // (a switch, because the asm offsets must be literal constants; `chunk` is one only after
        // unrolling - build with -mllvm -pragma-unroll-threshold raised, see build_hip.py)
        const int need_l = NW * (gen_base + chunk / kSlots + 1), need_d = NW * (gen_base + (chunk + kAhead) / kSlots);
        if (chunk % kChunks == 0) pre_read(chunk);           // first chunk of a pass: nothing ran ahead of it
        switch (chunk % kSlots) {
#define ZEST_CASE(S) case S: bump2<off_landed(S + 1), off_done(S + kSlots - 1)>(); await<off_landed(S)>(need_l, need_d); break;
            ZEST_CASE(0) ZEST_CASE(1) ZEST_CASE(2) ZEST_CASE(3) ZEST_CASE(4) ZEST_CASE(5) ZEST_CASE(6) ZEST_CASE(7)
            ZEST_CASE(8) ZEST_CASE(9) ZEST_CASE(10) ZEST_CASE(11) ZEST_CASE(12) ZEST_CASE(13) ZEST_CASE(14) ZEST_CASE(15)
#undef ZEST_CASE
        }
        static_assert(kSlots <= 16, "extend the slot switch");
#ifdef ZEST_STAMPS
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
#endif
        issue(chunk + kAhead);
#ifdef ZEST_STAMPS
        t_wait += t1 - t0, t_issue += __builtin_amdgcn_s_memtime() - t1;
#endif
    }
#else
    __device__ __forceinline__ void prologue() const {
#pragma unroll
        for (int c = 0; c < kAhead; c++) issue(c);
    }
    __device__ __forceinline__ void next_pass() const {}
    __device__ __forceinline__ void enter_chunk(int chunk) const {
#ifdef ZEST_STAMPS
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#endif
        flush();
        // all but the youngest (kAhead-1)*kPieces of this wave's DMA pieces have landed
        asm volatile("s_waitcnt vmcnt(%0)" ::"i"((kAhead - 1) * kPieces) : "memory");
#ifdef ZEST_STAMPS
        t_vm += __builtin_amdgcn_s_memtime() - t0;
#endif
#ifndef ZEST_EXPERIMENT_NO_BARRIER      // timing experiment only: results are wrong without it
        __builtin_amdgcn_s_barrier();
#endif
        asm volatile("" ::: "memory");
#ifdef ZEST_STAMPS
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
#endif
#ifndef ZEST_EXPERIMENT_NO_DMA          // timing experiment only
#if ZEST_DEFER_DMA
        pending = chunk + kAhead;               // issued by the next flush(): a layer's epilogue
#else
        issue(chunk + kAhead);
#endif
#endif
#ifdef ZEST_STAMPS
        t_wait += t1 - t0, t_issue += __builtin_amdgcn_s_memtime() - t1;
#endif
    }
#endif
    // ZEST_DEFER_DMA: the refill of the slot a rendezvous frees is not issued at the rendezvous - in the middle
    // of a run of MFMAs and tile reads, where an LDS-DMA piece holds the wave's issue for 100+ cycles
    // (MI355X_MICROARCH.md, row 'LDS-DMA piece issue cost') - but at the next row block's epilogue, among plain
    // VALU instructions.  `pending` is a compile-time constant at every use once the network is unrolled (it is -1
    // again at the end of a net's stream, so also around the pass loop).  The counted vmcnt of enter_chunk is
    // unchanged: a refill still pending at the next rendezvous is issued there, in front of the wait.
    mutable int pending = -1;
    __device__ __forceinline__ void flush() const {
#if ZEST_DEFER_DMA
#ifdef ZEST_STAMPS
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
#endif
        if (pending >= 0) issue(pending);
        pending = -1;
#ifdef ZEST_STAMPS
        t_issue += __builtin_amdgcn_s_memtime() - t1;
#endif
#endif
    }
    __device__ __forceinline__ void touch(int unit) const {
        if (unit % kChunk == 0) enter_chunk(unit / kChunk);
#ifdef ZEST_RING_FLAGS
        if (unit % kChunk == kChunk - kPreRead) pre_read(unit / kChunk + 1);
#endif
    }
    // LDS read addresses: exactly two base registers per access pattern (lower / upper 64 KiB of
    // the ring, the reach of the 16-bit DS offset field), made opaque once.  Left to itself hipcc
    // materialises every `const + lane * 16` beyond the first 64 KiB as a loop-invariant VGPR of
    // its own: ~60 registers of addresses that crowd out operands and end in scratch.
    mutable unsigned rd_lo = 0, rd_hi = 0, rb_lo = 0, rb_hi = 0;
    // modulation cache (ZEST_MCACHE_JB, engine_layer): LDS byte address of this lane's 16 bytes in the wave's region
    mutable unsigned mc = 0;
    __device__ __forceinline__ void init_mcache(unsigned lds_addr) const {
        mc = lds_addr;
        asm volatile("" : "+v"(mc));
    }
    __device__ __forceinline__ void init_addr() const {
        const unsigned base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char *)ring;
        rd_lo = base + lane * 16, rd_hi = rd_lo + 65536, rb_lo = base + grp * 16, rb_hi = rb_lo + 65536;
        asm volatile("" : "+v"(rd_lo), "+v"(rd_hi), "+v"(rb_lo), "+v"(rb_hi));
    }
    static_assert(kRingUnits <= 128, "two 64 KiB windows");
    template <class T>
    static __device__ __forceinline__ const __attribute__((address_space(3))) T *lds_at(unsigned addr) {
        return (const __attribute__((address_space(3))) T *)(uintptr_t)addr;
    }
    __device__ __forceinline__ bf16x8 load(int unit) const {
        touch(unit);
#ifdef ZEST_EXPERIMENT_NO_LDSREAD       // timing experiment only: one read per chunk
        const int r = unit / kChunk * kChunk % kRingUnits;
#else
        const int r = unit % kRingUnits;
#endif
        const v4u v = *lds_at<v4u>((r < 64 ? rd_lo : rd_hi) + (r % 64) * 1024);
        return *reinterpret_cast<const bf16x8 *>(&v);
    }
    __device__ __forceinline__ f32x4 load_bias(int unit, int which, int rt) const {
        if (which == 0 && rt == 0) touch(unit);     // the other blocks of the header are read right after
        const int r = unit % kRingUnits;
        return *lds_at<v4f>((r < 64 ? rb_lo : rb_hi) + (r % 64) * 1024 + which * 128 + rt * 64);
    }
    // walk the padding [unit, end) of a net's stream so every chunk is entered exactly once
    __device__ __forceinline__ void finish(int unit, int end) const {
#pragma unroll
        for (int u = (unit + kChunk - 1) / kChunk * kChunk; u < end; u += kChunk) {
#ifdef ZEST_RING_FLAGS
            pre_read(u / kChunk);
#endif
            enter_chunk(u / kChunk);
        }
        flush();
    }
    __device__ __forceinline__ void drain() const { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
};

// ---- operand types ------------------------------------------------------------------------
// EP (engine precision) is one of ZEST_PREC_BF16 / _F16 / _F16X3 (include/zest_render.h).  A tile
// of 8 operand values per lane is 4 VGPRs of 2-byte elements (`bf16x8` is used as the raw
// container for both element types).  ZEST_PREC_F16X3 carries every operand as the pair
// hi = fp16(v), lo = fp16((v - hi) * 2^11): together 22 significant bits.  The lo part is scaled
// so that it stays a normal fp16 number wherever hi is one, and the products that contain it
// are summed in an accumulator of their own that enters the result scaled by 2^-11:
//     sum_k w_k x_k  ~=  sum hi_w hi_x  +  2^-11 (sum hi_w lo_x + sum lo_w hi_x)
// (the dropped lo*lo term is 2^-22 relative).  Three MFMAs per product at the fp16 rate instead
// of one fp32 MFMA at 1/16 of that rate.
constexpr int ep_parts(int EP) { return EP == ZEST_PREC_F16X3 ? 2 : 1; }
constexpr float kLoScale = 2048.0f, kLoUnscale = 1.0f / 2048.0f;

typedef __attribute__((ext_vector_type(2))) float f32x2_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2_t;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;

// two fp32 -> one register of two 2-byte elements, round to nearest even: one v_cvt_pk_bf16_f32 /
// v_cvt_pk_f16_f32.  Written as a vector conversion the compiler understands - an inline-asm form
// would read MFMA results without the wait states hipcc inserts for its own instructions (cdna
// guide 5.7 item 2) - and not as __float22bfloat162_rn, which costs 2 cvt + shift + or.
template <int EP>
__device__ __forceinline__ unsigned pack_pair(float lo, float hi) {
    const f32x2_t f = {lo, hi};
    if constexpr (EP == ZEST_PREC_BF16) {
        const bf16x2_t r = __builtin_convertvector(f, bf16x2_t);
        return __builtin_bit_cast(unsigned, r);
    } else {
        const f16x2_t r = __builtin_convertvector(f, f16x2_t);
        return __builtin_bit_cast(unsigned, r);
    }
}
__device__ __forceinline__ unsigned pack_bf16(float lo, float hi) { return pack_pair<ZEST_PREC_BF16>(lo, hi); }

// the (hi, lo) registers of two values (ZEST_PREC_F16X3)
__device__ __forceinline__ void split_pair(float a, float b, unsigned &hi, unsigned &lo) {
    const f32x2_t f = {a, b};
    const f16x2_t h = __builtin_convertvector(f, f16x2_t);
    const f32x2_t back = __builtin_convertvector(h, f32x2_t);
    const f32x2_t rem = {(a - back[0]) * kLoScale, (b - back[1]) * kLoScale};     // a - hi is exact in fp32
    const f16x2_t l = __builtin_convertvector(rem, f16x2_t);
    hi = __builtin_bit_cast(unsigned, h), lo = __builtin_bit_cast(unsigned, l);
}

template <int EP>
__device__ __forceinline__ f32x4 mfma16(bf16x8 a, bf16x8 b, f32x4 c) {
    if constexpr (EP == ZEST_PREC_BF16)
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b),
                                                      c, 0, 0, 0);
}

// relu as one v_med3_f32: max(v, 0) for every finite activation (fmaxf costs a second,
// canonicalising v_max in front; med3 against a finite bound is not folded back into it).
// The fp16 modes take the largest fp16 number as the bound: an activation beyond it saturates
// instead of turning into infinity (and NaN one layer later).
template <int EP>
__device__ __forceinline__ float relu1(float v) {
    return __builtin_amdgcn_fmed3f(v, 0.0f, EP == ZEST_PREC_BF16 ? 3.0e38f : 65504.0f);
}

template <int N, int NP = 1>
struct OpArr {              // N k-tiles of one column block's operand in NP parts (hi [, lo]); N = 0 allowed
    bf16x8 t[NP][N > 0 ? N : 1];
};

// eight fp32 values of one lane -> k-tile `k` of an operand
template <int EP, int N>
__device__ __forceinline__ void store_tile(const float (&v)[8], OpArr<N, ep_parts(EP)> &op, int k) {
    if constexpr (EP == ZEST_PREC_F16X3) {
        uint4 h, l;
        split_pair(v[0], v[1], h.x, l.x), split_pair(v[2], v[3], h.y, l.y);
        split_pair(v[4], v[5], h.z, l.z), split_pair(v[6], v[7], h.w, l.w);
        op.t[0][k] = __builtin_bit_cast(bf16x8, h), op.t[1][k] = __builtin_bit_cast(bf16x8, l);
    } else {
        const uint4 q = make_uint4(pack_pair<EP>(v[0], v[1]), pack_pair<EP>(v[2], v[3]), pack_pair<EP>(v[4], v[5]),
                                   pack_pair<EP>(v[6], v[7]));
        op.t[0][k] = __builtin_bit_cast(bf16x8, q);
    }
}

// ReLU of a 16-bit operand tile AFTER rounding, on the packed pairs: a negative bf16 is a negative int16, so
// max(x, 0) per 16-bit half is one v_pk_max_i16 for two values where relu in fp32 costs one v_med3_f32 each.
// Same bits as relu-then-round (rounding keeps the sign; -0 becomes +0 either way).  The engine's issue port
// is the contended resource (MI355X_MICROARCH.md, row 'vector-instruction ISSUE cost': a 16x16x32 MFMA leaves
// room for two plain VALU instructions): this takes 8 of the ~33 VALU instructions of a 256-wide row block away.
#ifndef ZEST_PACKED_RELU
#define ZEST_PACKED_RELU 1
#endif
typedef __attribute__((ext_vector_type(2))) short s16x2_t;
__device__ __forceinline__ unsigned relu_bf16x2(unsigned u) {
    const s16x2_t r = __builtin_elementwise_max(__builtin_bit_cast(s16x2_t, u), s16x2_t{0, 0});
    return __builtin_bit_cast(unsigned, r);
}
template <int EP, int N>
__device__ __forceinline__ void store_tile_relu16(const float (&v)[8], OpArr<N, 1> &op, int k) {
    const uint4 q = make_uint4(relu_bf16x2(pack_pair<EP>(v[0], v[1])), relu_bf16x2(pack_pair<EP>(v[2], v[3])),
                               relu_bf16x2(pack_pair<EP>(v[4], v[5])), relu_bf16x2(pack_pair<EP>(v[6], v[7])));
    op.t[0][k] = __builtin_bit_cast(bf16x8, q);
}
// fp16 operands take the same packed ReLU (a negative fp16 is a negative int16 as well).  What the fp32 ReLU did on
// top for them - saturate an activation beyond 65504 instead of letting it round to infinity (and NaN one layer
// later) - is done by the hardware: with MODE.FP16_OVFL set an fp16 result that overflows is clamped to +-65504.
// The kernels that run the engine on fp16 operands set the bit once at their start (engine_fp16_overflow_clamp).
__device__ __forceinline__ void engine_fp16_overflow_clamp() {
    __builtin_amdgcn_s_setreg((0 << 11) | (23 << 6) | 1, 1);        // hwreg(HW_REG_MODE, offset 23 = FP16_OVFL, 1 bit) = 1
}

#ifndef ZEST_PREFETCH
#define ZEST_PREFETCH 3        // weight tiles kept in flight ahead of the MFMAs that consume them
#endif
constexpr int kPrefetch = ZEST_PREFETCH;

// Modulation cache.  m = pts_bias(features) is the same in all eight trunk layers; the engine recomputes it per
// row block and layer with 2 NKF extra MFMA pairs because 256 values per sample have no room in registers.  A kernel
// built with ZEST_MCACHE_JB = n > 0 (the fused feature kernels whose ring is 64 KiB: fused.cuh) keeps the first n
// row blocks of m in LDS instead, rounded to the operand type: layer 0 computes them as before and stores them (one
// ds_write_b128 per column block), layers 1-7 skip those row blocks' modulation tiles - their stream units are
// walked (touch) but neither read nor multiplied - and read m back (one ds_read_b128 per column block, unpacked
// with two integer ops per pair).  With two feature k-tiles that removes 8 of the 40 MFMAs and 4 of the 20 tile
// reads of a cached row block; with one k-tile the unpacking costs what the 4 MFMAs did, so only nets with
// NKF >= 2 take it.  MC: 0 = off, 1 = compute + store (layer 0), 2 = load (layers 1-7).
#ifndef ZEST_MCACHE_JB
#define ZEST_MCACHE_JB 0
#endif
constexpr int kMcJB = ZEST_MCACHE_JB;
constexpr bool mcache_for(int EP, bool mod, int nkf) { return kMcJB > 0 && mod && nkf >= 2 && EP != ZEST_PREC_F16X3; }

template <int EP>
__device__ __forceinline__ void unpack_pairs(const v4u q, float (&v)[8]) {
#pragma unroll
    for (int i = 0; i < 4; i++) {
        if constexpr (EP == ZEST_PREC_BF16) {
            v[2 * i] = __uint_as_float(q[i] << 16), v[2 * i + 1] = __uint_as_float(q[i] & 0xFFFF0000u);
        } else {
            const f32x2_t f = __builtin_convertvector(__builtin_bit_cast(f16x2_t, q[i]), f32x2_t);
            v[2 * i] = f[0], v[2 * i + 1] = f[1];
        }
    }
}

// What is fetched ahead for one row block: its bias initialisers and its first tiles.
template <int NP>
struct RowBlockPre {
    f32x4 bias[2], mbias[2];      // per row tile
    bf16x8 win[kPrefetch][NP];
};

// One Linear: NJB row blocks (32 outputs = row tiles 0, 1) over the operand
// [A (NKA k-tiles) | B (NKB k-tiles)] of each of the CB column blocks.
//   MOD:  the row block is modulated by the feature operand (NKF k-tiles) with the op's
//         modulation tiles
//   MODE: 0 = produce k-tile jb of `out`, 1 = keep row tile 0 of row block 0 in `keep`
//         (head / rgb: lane group g holds rows 4g .. 4g+3)
// Unit order inside a row block: header, then per k-tile the row tiles 0, 1 (modulation k-tiles
// first; a split tile is the unit pair hi, lo): independent accumulators take turns.  The LDS
// reads are software-pipelined by hand, in source order: every tile is requested kPrefetch tiles
// before the MFMAs that use it, and the NEXT row block's header and first tiles are requested
// before the CURRENT epilogue, whose VALU instructions cover their latency.
// A sink sees every finished operand tile of the layers that have one (training: the activation stash):
//   sink(id, jb, cb, v)   id: 0-7 trunk layers, 8 feature_linear, 9 view layer; v: the 8 fp32 values
struct NoSink {
    __device__ __forceinline__ void operator()(int, int, int, const float (&)[8]) const {}
};

// V2: additive modulation (Renderer_linear, networks.py:294) instead of the multiplicative one.  A template
// parameter on purpose: as a launch-time flag, `v2 ? v + m : v * m` per element makes hipcc compute both and select -
// three vector instructions per value where one is needed, 2 048 of the ~5 500 a feature pass issued beside its
// 2 880 MFMAs, on the port the engine is short of (MI355X_MICROARCH.md, 'vector-instruction ISSUE cost') - and a
// scalar branch per row block cuts the unrolled network into basic blocks the register allocator answers with
// ~200 spills.
template <int EP, int CB, int NJB, int NKA, int NKB, bool MOD, int NKF, bool RELU, int MODE, bool V2, int MC, class Tiles,
          class Sink = NoSink>
__device__ __forceinline__ void engine_layer(const Tiles &tiles, int &unit,
                                             const OpArr<NKA, ep_parts(EP)> (&opa)[CB],
                                             const OpArr<NKB, ep_parts(EP)> (&opb)[CB],
                                             const OpArr<NKF, ep_parts(EP)> (&opf)[CB],
                                             OpArr<8, ep_parts(EP)> (&out)[CB], f32x4 (&keep)[CB],
                                             int sink_id = 0, const Sink &sink = Sink()) {
    constexpr int NP = ep_parts(EP);
    constexpr bool X3 = EP == ZEST_PREC_F16X3;
    constexpr int NM = MOD ? 2 * NKF : 0, T = NM + 2 * (NKA + NKB);     // tiles per row block
    static_assert(T >= 1, "empty layer");
    // a sink wants the activated fp32 values: only the plain inference layer rounds first
    constexpr bool kPackedRelu = ZEST_PACKED_RELU && RELU && MODE == 0 && EP != ZEST_PREC_F16X3 && __is_same(Sink, NoSink);
    static_assert(MC == 0 || (MOD && MODE == 0 && NJB == 8), "the modulation cache serves modulated trunk layers");
    // first tile a row block executes: its modulation tiles are skipped where m comes from the cache
    auto k0_of = [](int jb) { return (MC == 2 && jb < kMcJB) ? NM : 0; };
    auto preload = [&](RowBlockPre<NP> &p, int u0, int k0) {    // u0: the row block's header unit
#pragma unroll
        for (int rt = 0; rt < 2; rt++) {
            p.bias[rt] = tiles.load_bias(u0, 0, rt);
            if (MOD && k0 == 0) p.mbias[rt] = tiles.load_bias(u0, 1, rt);
        }
#pragma unroll
        for (int k = 0; k < k0; k++)                            // skipped units: the ring's chunks are still entered in order
#pragma unroll
            for (int pt = 0; pt < NP; pt++) tiles.touch(u0 + 1 + NP * k + pt);
#pragma unroll
        for (int k = 0; k < kPrefetch; k++)
            if (k0 + k < T) {
#pragma unroll
                for (int pt = 0; pt < NP; pt++) p.win[k][pt] = tiles.load(u0 + 1 + NP * (k0 + k) + pt);
            }
    };
    RowBlockPre<NP> pre;
    preload(pre, unit, k0_of(0));
#pragma unroll
    for (int jb = 0; jb < NJB; jb++) {
        const int u0 = unit, k0 = k0_of(jb);
        const bool mc_rd = MC == 2 && jb < kMcJB, mc_wr = MC == 1 && jb < kMcJB;
        // acc: the hi*hi products (all products of the one-part types); corr: hi*lo + lo*hi, scaled 2^11
        f32x4 acc[2][CB], macc[2][CB], corr[2][CB], mcorr[2][CB];
        bf16x8 win[kPrefetch][NP];
#pragma unroll
        for (int k = 0; k < kPrefetch; k++)
#pragma unroll
            for (int pt = 0; pt < NP; pt++) win[k][pt] = pre.win[k][pt];
#pragma unroll
        for (int rt = 0; rt < 2; rt++)
#pragma unroll
            for (int cb = 0; cb < CB; cb++) {
                acc[rt][cb] = pre.bias[rt];
                if (MOD && !mc_rd) macc[rt][cb] = pre.mbias[rt];
                if (X3) corr[rt][cb] = f32x4{0.f, 0.f, 0.f, 0.f}, mcorr[rt][cb] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
        for (int k = k0; k < T; k++) {
            bf16x8 a[NP];
#pragma unroll
            for (int pt = 0; pt < NP; pt++) a[pt] = win[(k - k0) % kPrefetch][pt];
            const int rt = k % 2, kt = (k < NM ? k : k - NM) / 2;          // compile-time after unrolling
#pragma unroll
            for (int cb = 0; cb < CB; cb++) {
                // the operand k-tile this weight tile multiplies, part pt (every index is a
                // compile-time constant after unrolling; the clamps keep dead branches in range)
                auto opnd = [&](int pt) -> bf16x8 {
                    if (k < NM) return opf[cb].t[pt][k < NM ? kt : 0];
                    if (kt < NKA) return opa[cb].t[pt][(k >= NM && kt < NKA) ? kt : 0];
                    return opb[cb].t[pt][(k >= NM && kt >= NKA) ? kt - NKA : 0];
                };
                if (k < NM) {
                    macc[rt][cb] = mfma16<EP>(a[0], opnd(0), macc[rt][cb]);
                    if constexpr (X3) {
                        mcorr[rt][cb] = mfma16<EP>(a[0], opnd(NP - 1), mcorr[rt][cb]);
                        mcorr[rt][cb] = mfma16<EP>(a[NP - 1], opnd(0), mcorr[rt][cb]);
                    }
                } else {
                    acc[rt][cb] = mfma16<EP>(a[0], opnd(0), acc[rt][cb]);
                    if constexpr (X3) {
                        corr[rt][cb] = mfma16<EP>(a[0], opnd(NP - 1), corr[rt][cb]);
                        corr[rt][cb] = mfma16<EP>(a[NP - 1], opnd(0), corr[rt][cb]);
                    }
                }
            }
            if (k + kPrefetch < T) {
#pragma unroll
                for (int pt = 0; pt < NP; pt++) win[(k - k0) % kPrefetch][pt] = tiles.load(u0 + 1 + NP * (k + kPrefetch) + pt);
            }
#ifdef ZEST_SCHED_PIN
            __builtin_amdgcn_sched_barrier(0);      // keep the hand-made read-ahead distance
#endif
        }
        v4u mq[CB];
        if (mc_rd) {                                             // in flight while the last MFMAs drain
#pragma unroll
            for (int cb = 0; cb < CB; cb++) mq[cb] = *Tiles::template lds_at<v4u>(tiles.mc + (jb * CB + cb) * 1024);
        }
        unit = u0 + 1 + NP * T;
        if (jb + 1 < NJB) preload(pre, unit, k0_of(jb + 1));                   // in flight during the epilogue
        tiles.flush();
#pragma unroll
        for (int cb = 0; cb < CB; cb++) {
            float v[8];
#pragma unroll
            for (int i = 0; i < 8; i++) {
                v[i] = acc[i >> 2][cb][i & 3];
                if (X3) v[i] = fmaf(corr[i >> 2][cb][i & 3], kLoUnscale, v[i]);
            }
            if (MOD) {
                float mv[8];
                if (mc_rd) {
                    unpack_pairs<EP>(mq[cb], mv);
                } else {
#pragma unroll
                    for (int i = 0; i < 8; i++) {
                        mv[i] = macc[i >> 2][cb][i & 3];
                        if (X3) mv[i] = fmaf(mcorr[i >> 2][cb][i & 3], kLoUnscale, mv[i]);
                    }
                }
                if (mc_wr) {
                    if constexpr (!X3) {
                        const uint4 q = make_uint4(pack_pair<EP>(mv[0], mv[1]), pack_pair<EP>(mv[2], mv[3]),
                                                   pack_pair<EP>(mv[4], mv[5]), pack_pair<EP>(mv[6], mv[7]));
                        *(__attribute__((address_space(3))) v4u *)(uintptr_t)(tiles.mc + (jb * CB + cb) * 1024) =
                            __builtin_bit_cast(v4u, q);
                    }
                }
#pragma unroll
                for (int i = 0; i < 8; i++) v[i] = V2 ? v[i] + mv[i] : v[i] * mv[i];
            }
            if (RELU && !kPackedRelu) {
#pragma unroll
                for (int i = 0; i < 8; i++) v[i] = relu1<EP>(v[i]);
            }
            if (MODE == 0) {
                if constexpr (kPackedRelu) store_tile_relu16<EP>(v, out[cb], jb);
                else store_tile<EP>(v, out[cb], jb);
                sink(sink_id, jb, cb, v);
            } else if (jb == 0) {
                keep[cb] = f32x4{v[0], v[1], v[2], v[3]};
            }
        }
    }
}

// The whole network for CB column blocks of 16 samples, reading the stream from unit `unit` on
// (advanced to the end of the net's padded stream).  pts/feat: encoder operands in plan position
// order, NU_* = their logical stream tiles per row block = 2 x k-tiles.  `pts_fn(pts)` builds the
// point operand; it is called twice, `pts_fn(pts, token)`, for layer 0 and for the skip layer 5, so
// the operand does not occupy registers through layers 1-4 (a caller that prefers to keep it hands
// out copies);
// `views_fn(views)` builds the direction operand when it is first needed (op 10).  Results per column block: head (lane group g: rows 4g .. 4g+3 of the head
// tile; row 0 alpha, rows 1.. extra heads) and rgb (group 0: rows 0-2), raw.
template <int EP, int CB, int NU_PTS, bool MOD, int NU_FEAT, bool V2, bool MCACHE = false, class Tiles, class PtsFn, class ViewsFn,
          class Sink = NoSink>
__device__ __forceinline__ void engine_forward(const Tiles &tiles, int &unit, PtsFn pts_fn,
                                               const OpArr<NU_FEAT / 2, ep_parts(EP)> (&feat)[CB], ViewsFn views_fn,
                                               f32x4 (&head)[CB], f32x4 (&rgb)[CB], const Sink &sink = Sink()) {
    constexpr int KP = NU_PTS / 2, KF = NU_FEAT / 2, NP = ep_parts(EP);
    static_assert(NU_PTS % 2 == 0 && NU_FEAT % 2 == 0, "units per row block come in row-tile pairs");
    constexpr int MC0 = (MCACHE && mcache_for(EP, MOD, KF)) ? 1 : 0, MCL = MC0 ? 2 : 0;   // layer 0 stores m, 1-7 load it
    OpArr<8, NP> hA[CB], hB[CB];
    OpArr<0, NP> none[CB];
    f32x4 unused[CB];
    const int unit0 = unit;
    {
        OpArr<KP, NP> pts[CB];
        pts_fn(pts, 0);
        engine_layer<EP, CB, 8, KP, 0, MOD, KF, true, 0, V2, MC0>(tiles, unit, pts, none, feat, hA, unused, 0, sink);
    }
    engine_layer<EP, CB, 8, 8, 0, MOD, KF, true, 0, V2, MCL>(tiles, unit, hA, none, feat, hB, unused, 1, sink);
    engine_layer<EP, CB, 8, 8, 0, MOD, KF, true, 0, V2, MCL>(tiles, unit, hB, none, feat, hA, unused, 2, sink);
    engine_layer<EP, CB, 8, 8, 0, MOD, KF, true, 0, V2, MCL>(tiles, unit, hA, none, feat, hB, unused, 3, sink);
    engine_layer<EP, CB, 8, 8, 0, MOD, KF, true, 0, V2, MCL>(tiles, unit, hB, none, feat, hA, unused, 4, sink);
    {
        // `token` is a value layer 4 has just produced: a builder that ties its address arithmetic to it
        // cannot be scheduled ahead of layers 1-4 (where its registers would be live all along)
        OpArr<KP, NP> pts[CB];
        pts_fn(pts, (int)hA[CB - 1].t[0][7][0]);
        engine_layer<EP, CB, 8, KP, 8, MOD, KF, true, 0, V2, MCL>(tiles, unit, pts, hA, feat, hB, unused, 5, sink);
    }
    engine_layer<EP, CB, 8, 8, 0, MOD, KF, true, 0, V2, MCL>(tiles, unit, hB, none, feat, hA, unused, 6, sink);
    engine_layer<EP, CB, 8, 8, 0, MOD, KF, true, 0, V2, MCL>(tiles, unit, hA, none, feat, hB, unused, 7, sink);
    // trunk output in hB
    engine_layer<EP, CB, 1, 8, 0, false, KF, false, 1, V2, 0>(tiles, unit, hB, none, feat, hA, head);
    engine_layer<EP, CB, 8, 8, 0, false, KF, false, 0, V2, 0>(tiles, unit, hB, none, feat, hA, unused, 8, sink);
    OpArr<1, NP> views[CB];
    views_fn(views);
    engine_layer<EP, CB, 4, 8, 1, false, KF, true, 0, V2, 0>(tiles, unit, hA, views, feat, hB, unused, 9, sink);
    // rgb: 128 hidden features = first 4 k-tiles of hB
    OpArr<4, NP> h128[CB];
#pragma unroll
    for (int cb = 0; cb < CB; cb++)
#pragma unroll
        for (int pt = 0; pt < NP; pt++)
#pragma unroll
            for (int k = 0; k < 4; k++) h128[cb].t[pt][k] = hB[cb].t[pt][k];
    engine_layer<EP, CB, 1, 4, 0, false, KF, false, 1, V2, 0>(tiles, unit, h128, none, feat, hA, rgb);
    tiles.finish(unit, unit0 + stream_units(NU_PTS, MOD ? NU_FEAT : 0, NP));
    unit = unit0 + stream_units(NU_PTS, MOD ? NU_FEAT : 0, NP);
}

// standalone launcher (mlp.hip -> zest_mlp_fwd)
int mlp_engine_launch(const MlpPlan &p, const void *tiles, const float *x, int M, float *out,
                    hipStream_t stream);

}  // namespace zest
