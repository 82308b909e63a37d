// 3-D regularisation net of the MVS volume builder: CostRegNet, reference networks.py:1003-1059 -
// seven 3x3x3 convolutions (three of stride 2), three stride-2 transposed convolutions, a batch norm + leaky
// ReLU(0.01) after each (InPlaceABN, networks.py:938-960), three skip additions - on the plane-sweep cost volume,
// without autograd (whole-image evaluation; training goes through the library convolutions, which have a backward).
//
// Layout: every tensor is channels-last fp32 [D][H][W][C]: the K dimension of a convolution's contraction - the 3 C
// values of a voxel's x-window in one (dz, dy) row - is then contiguous in memory, 8 channels of a pixel are one
// 32-byte load, and no im2col buffer exists.  A convolution writes its RAW output and the per-channel sum / sum of
// squares of it (batch statistics); the norm + activation of a layer is applied by its CONSUMER when it loads the
// tensor (scale and shift per channel from zest_costreg_bn: two floats per channel), so an activation tensor is
// written once and read once per consumer.  (The library path runs, per layer: convolution, statistics pass,
// normalisation pass, activation pass, layout transposes around the convolution - 23 ms of device time for the two
// builders of an NSFF image, profiles/r03_builder_library_kernels.txt.)
//
// conv3d_mfma_kernel: v_mfma_f32_16x16x32_bf16, A = packed weights (16 output channels x 32 k), B = activations
// (32 k x 16 voxels consecutive in x): lane (n = l & 15, g = l >> 4) supplies the 8 channels of octet 4 c + g of voxel
// n's window and receives output channels 4 g .. 4 g + 3 of voxel n (one 16-byte store).  PASSES = 1: operands
// rounded to bf16 (the --precision 16 path; the library runs these layers under bf16 autocast there).  PASSES = 3:
// every operand is the pair hi = bf16(v), lo = bf16(v - hi) and a product is hi hi + hi lo + lo hi - 16 significant
// bits, bf16's exponent range (raw cost-volume variances are unbounded).  Bound: the loads of the operands through
// the vector L1 (a window is re-read by its neighbours), not HBM and not the MFMA pipe; see DESIGN.md.
// deconv3d_mfma_kernel: the transposed convolutions as eight small convolutions (one per output parity class) over
// the same staged input rows; see there.
#include "mlp_engine.cuh"

namespace {

using namespace zest;
constexpr int BF = ZEST_PREC_BF16;
constexpr float kSlope = 0.01f;           // InPlaceABN's leaky ReLU

__device__ __forceinline__ float leaky(float v) { return fmaxf(v, kSlope * v); }

// Per-channel sums of a workgroup -> its row of the partial-sum table [kStatRows + 1][2][COUT] (doubles); the first
// value of the extra row is the number of rows in use (the kernel's grid).
constexpr int kStatRows = 1024;           // = the largest grid of these kernels
template <int NT, int COUT>
__device__ __forceinline__ void stats_to_row(const float (&ssum)[NT][4], const float (&ssq)[NT][4], float (&red)[4][2 * NT * 16],
                                             double *stats, int lane, int wave) {
    const int n = lane & 15, g = lane >> 4;
#pragma unroll
    for (int nt = 0; nt < NT; nt++)
#pragma unroll
        for (int i = 0; i < 4; i++) {
            float s = ssum[nt][i], q = ssq[nt][i];
#pragma unroll
            for (int m = 8; m >= 1; m >>= 1) s += __shfl_xor(s, m, 64), q += __shfl_xor(q, m, 64);
            if (n == 0) red[wave][nt * 16 + 4 * g + i] = s, red[wave][NT * 16 + nt * 16 + 4 * g + i] = q;
        }
    __syncthreads();
    if (threadIdx.x < 2 * NT * 16) {
        const int which = threadIdx.x / (NT * 16), c = threadIdx.x % (NT * 16);
        if (c < COUT)
            stats[((size_t)blockIdx.x * 2 + which) * COUT + c] =
                (double)red[0][threadIdx.x] + (double)red[1][threadIdx.x] + (double)red[2][threadIdx.x] + (double)red[3][threadIdx.x];
        if (blockIdx.x == 0 && threadIdx.x == 0) stats[(size_t)kStatRows * 2 * COUT] = (double)gridDim.x;   // rows in use
    }
}

struct ConvArgs {
    const float *in, *pre;                // [Di,Hi,Wi,CIN]; [2,CIN] scale / shift of the producer's norm (PRE)
    const uint4 *w;                       // packed A operands (zest_networks.CostRegNet._pack_conv)
    float *out;                           // [Do,Ho,Wo,COUT] raw
    double *stats;                        // [kStatRows,2,COUT] sum, sum of squares per workgroup
    int Di, Hi, Wi, Do, Ho, Wo, n_xb, n_yg, n_tiles;
};

// eight values -> operand registers (hi [, lo])
template <int NP>
__device__ __forceinline__ void to_operand(const float (&v)[8], bf16x8 (&op)[NP]) {
    uint4 h;
    h.x = pack_pair<BF>(v[0], v[1]), h.y = pack_pair<BF>(v[2], v[3]), h.z = pack_pair<BF>(v[4], v[5]), h.w = pack_pair<BF>(v[6], v[7]);
    op[0] = __builtin_bit_cast(bf16x8, h);
    if constexpr (NP == 2) {
        float r[8];
        const unsigned hw[4] = {h.x, h.y, h.z, h.w};
#pragma unroll
        for (int e = 0; e < 8; e++) {
            const unsigned bits = (e & 1) ? (hw[e >> 1] & 0xFFFF0000u) : (hw[e >> 1] << 16);
            r[e] = v[e] - __uint_as_float(bits);              // exact
        }
        uint4 l;
        l.x = pack_pair<BF>(r[0], r[1]), l.y = pack_pair<BF>(r[2], r[3]), l.z = pack_pair<BF>(r[4], r[5]), l.w = pack_pair<BF>(r[6], r[7]);
        op[1] = __builtin_bit_cast(bf16x8, l);
    }
}

// K: taps per axis in y and x (3 or 5, padding K / 2); KD: taps in z (3, padding 1 - or 1: the slices are the images of
// a batch, a 2-D convolution: FeatureNet, reference networks.py:962-1001; z is then not strided)
template <int CIN, int COUT, int STRIDE, int PASSES, bool PRE, int RT, int K = 3, int KD = 3>
__global__ __launch_bounds__(256, 2) void conv3d_mfma_kernel(const ConvArgs a) {
    constexpr int OPT = CIN / 8;                   // channel octets per pixel
    constexpr int CPR = (K * OPT + 3) / 4;         // k-chunks (of 4 octets) per (dz, dy) row of the window
    constexpr int PAD = K / 2, SZ = KD == 1 ? 1 : STRIDE, PZ = KD / 2;
    constexpr int NT = (COUT + 15) / 16, NP = PASSES == 3 ? 2 : 1;
    constexpr int IY = STRIDE * (RT - 1) + K;      // input rows under RT output rows
    constexpr bool WREG = K == 3 && K * CPR * NT * NP * 4 <= 128 && RT > 1;       // weights of a z-tap in registers
    static_assert(CIN % 8 == 0 && (!PRE || OPT == 1 || OPT == 2 || OPT == 4 || OPT == 8), "channel counts");
    const int lane = threadIdx.x & 63, n = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    __shared__ float red[4][2 * NT * 16];
    float sc[8], sh[8];                            // norm constants of this lane's octet of the strip (64 % OPT == 0)
    if constexpr (PRE) {
        const int q = lane % OPT;
#pragma unroll
        for (int e = 0; e < 8; e++) sc[e] = a.pre[8 * q + e], sh[e] = a.pre[CIN + 8 * q + e];
    }
    float ssum[NT][4], ssq[NT][4];
#pragma unroll
    for (int nt = 0; nt < NT; nt++)
#pragma unroll
        for (int i = 0; i < 4; i++) ssum[nt][i] = ssq[nt][i] = 0.0f;

    // Strip staging.  The 3 C-wide windows of 16 neighbouring voxels overlap: the strip of input row (zi, yi) under
    // them - SP pixels, contiguous in memory - is loaded ONCE per wave (lane i takes octet i: 2 KiB contiguous per
    // round), normalised, rounded and split once, and left in a wave-private piece of LDS; the MFMA operand of
    // (voxel n, octet 4 c + g) is then one 16-byte LDS read per part.  (Read per window from global memory every octet
    // was fetched, normalised and rounded three times - and the loads were 32-byte pieces at a stride of C floats.)
    // No barrier: the strip belongs to one wave, and a wave's LDS instructions execute in order.
    constexpr int SP = STRIDE * 15 + K, NO = SP * OPT, NR = (NO + 63) / 64;
    constexpr int PSTR = OPT + 1;                   // octets per pixel in LDS: one of padding (spreads the banks)
    constexpr int PART_BYTES = SP * PSTR * 16;
    __shared__ __attribute__((aligned(16))) char strips[4][NP * PART_BYTES];
    char *const strip = strips[wave];
    // Rows in flight: the strip staged at step k is requested at step k - 1, or k - 2 (DEEP) where the registers allow
    // it without losing the second wave per SIMD: the bf16 kernels (first layer 483 -> 421 us); the split-bf16 ones
    // hold twice the weights and ran slower with it (498 -> 649 us at one wave per SIMD).
    constexpr bool DEEP = NP == 1;
    float4 pf[NR][2], pf2[DEEP ? NR : 1][2];
    unsigned pf_ok = 0, pf2_ok = 0;
    auto fetch_row = [&](int k, int z, int y0, int x0, auto &dst, unsigned &dst_ok) {   // row k of the tile's 3 IY
        const int zi = SZ * z + k / IY - PZ, yi = STRIDE * y0 - PAD + k % IY;
        const bool row_ok = k < KD * IY && (unsigned)zi < (unsigned)a.Di && (unsigned)yi < (unsigned)a.Hi;
        const float *base = a.in + (((size_t)(row_ok ? zi : 0) * a.Hi + (row_ok ? yi : 0)) * a.Wi) * CIN;
        const int xs = STRIDE * x0 - PAD;
        dst_ok = 0;
#pragma unroll
        for (int r = 0; r < NR; r++) {
            const int i = lane + 64 * r, xi = xs + i / OPT;
            const bool ok = row_ok && i < NO && (unsigned)xi < (unsigned)a.Wi;
            const float4 *src = reinterpret_cast<const float4 *>(ok ? base + ((ptrdiff_t)xs * CIN + 8 * i) : a.in);
            dst[r][0] = src[0], dst[r][1] = src[1];          // unconditional (a select of the ADDRESS): 16-byte loads, no branches
            dst_ok |= ok ? (1u << r) : 0u;
        }
    };
    auto stage = [&]() {                            // pf -> operand halves in LDS
#pragma unroll
        for (int r = 0; r < NR; r++) {
            const int i = lane + 64 * r, sp = i / OPT, q = i - sp * OPT;
            if (i >= NO) continue;
            float v[8] = {pf[r][0].x, pf[r][0].y, pf[r][0].z, pf[r][0].w, pf[r][1].x, pf[r][1].y, pf[r][1].z, pf[r][1].w};
            const bool ok = (pf_ok >> r) & 1;
#pragma unroll
            for (int e = 0; e < 8; e++) v[e] = ok ? (PRE ? leaky(fmaf(v[e], sc[e], sh[e])) : v[e]) : 0.0f;
            bf16x8 op[NP];
            to_operand<NP>(v, op);
            char *dst = strip + (sp * PSTR + q) * 16;
            *reinterpret_cast<bf16x8 *>(dst) = op[0];
            if constexpr (NP == 2) *reinterpret_cast<bf16x8 *>(dst + PART_BYTES) = op[NP - 1];
        }
    };

    for (int t = (int)blockIdx.x * 4 + wave; t < a.n_tiles; t += (int)gridDim.x * 4) {
        // z runs fastest: the four waves of a workgroup (and the workgroups next to it) work on neighbouring slices of
        // one (y, x) position at the same time, so the three uses of an input row by the output slices z - 1, z, z + 1
        // meet in L1 / L2: FETCH_SIZE of the first layer 1.25 -> 1.03 GB (0.52 GB of input, the y and x halos of a 4 x 16
        // tile are 1.7x), of the 16 -> 16 layer and the last transposed one -60 %; the time moves by 2 %: these kernels
        // are not HBM-bound.  (Numbering the workgroups XCD-major on top of it - blockIdx % 8 first - was WORSE: fetch
        // +12 %, time +8 %.)
        const int z = t % a.Do, r_ = t / a.Do, xb = r_ % a.n_xb, yg = r_ / a.n_xb;
        const int x0 = xb * 16, y0 = yg * RT;
        f32x4 acc[RT][NT];
#pragma unroll
        for (int ry = 0; ry < RT; ry++)
#pragma unroll
            for (int nt = 0; nt < NT; nt++) acc[ry][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
        fetch_row(0, z, y0, x0, pf, pf_ok);
        if constexpr (DEEP) fetch_row(1, z, y0, x0, pf2, pf2_ok);
        for (int dz = 0; dz < KD; dz++) {
            const int zi = SZ * z + dz - PZ;
            const bool z_ok = (unsigned)zi < (unsigned)a.Di;
            // the weights of this z-tap stay in registers over the tile's input rows when they fit (the three (dy)
            // operands of a chunk are used again by every input row: read per use they were 2/3 of the kernel's L1 traffic)
            bf16x8 wreg[WREG ? K : 1][WREG ? CPR : 1][WREG ? NT : 1][NP];
            if constexpr (WREG) {
                if (z_ok) {
#pragma unroll
                    for (int dy = 0; dy < K; dy++)
#pragma unroll
                        for (int c = 0; c < CPR; c++)
#pragma unroll
                            for (int nt = 0; nt < NT; nt++)
#pragma unroll
                                for (int pt = 0; pt < NP; pt++)
                                    wreg[dy][c][nt][pt] = __builtin_bit_cast(
                                        bf16x8, a.w[((size_t)(((dz * K + dy) * CPR + c) * NT + nt) * NP + pt) * 64 + lane]);
                }
            }
#pragma unroll
            for (int iy = 0; iy < IY; iy++) {
                const int yi = STRIDE * y0 - PAD + iy;
                const bool row_ok = z_ok && (unsigned)yi < (unsigned)a.Hi;
                if (row_ok) stage();
                __builtin_amdgcn_wave_barrier();
                if constexpr (DEEP) {
#pragma unroll
                    for (int r = 0; r < NR; r++) pf[r][0] = pf2[r][0], pf[r][1] = pf2[r][1];
                    pf_ok = pf2_ok;
                    fetch_row(dz * IY + iy + 2, z, y0, x0, pf2, pf2_ok);
                } else {
                    fetch_row(dz * IY + iy + 1, z, y0, x0, pf, pf_ok);
                }
                if (!row_ok) continue;
                bf16x8 act[CPR][NP];
#pragma unroll
                for (int c = 0; c < CPR; c++) {
                    const int o = 4 * c + g, p = o / OPT, q = o - p * OPT;
                    const bool okc = o < K * OPT;
                    const char *src = strip + ((okc ? STRIDE * n + p : 0) * PSTR + (okc ? q : 0)) * 16;
#pragma unroll
                    for (int pt = 0; pt < NP; pt++) {
                        bf16x8 v = *reinterpret_cast<const bf16x8 *>(src + pt * PART_BYTES);
                        if (4 * c + 3 >= K * OPT) {                   // a chunk with padding octets (zero weights: keep stale LDS bits out)
                            const bf16x8 z8 = {0, 0, 0, 0, 0, 0, 0, 0};
                            v = okc ? v : z8;
                        }
                        act[c][pt] = v;
                    }
                }
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int ry = 0; ry < RT; ry++) {
                    const int dy = iy - STRIDE * ry;                  // compile-time after unrolling
                    if (dy < 0 || dy >= K) continue;
                    const uint4 *wrow = a.w + (size_t)((dz * K + dy) * CPR) * NT * NP * 64 + lane;
#pragma unroll
                    for (int c = 0; c < CPR; c++)
#pragma unroll
                        for (int nt = 0; nt < NT; nt++) {
                            const uint4 *wp = wrow + (size_t)(c * NT + nt) * NP * 64;
                            const bf16x8 wh = WREG ? wreg[WREG ? dy : 0][WREG ? c : 0][WREG ? nt : 0][0] : __builtin_bit_cast(bf16x8, wp[0]);
                            acc[ry][nt] = mfma16<BF>(wh, act[c][0], acc[ry][nt]);
                            if constexpr (NP == 2) {
                                const bf16x8 wl = WREG ? wreg[WREG ? dy : 0][WREG ? c : 0][WREG ? nt : 0][NP - 1] : __builtin_bit_cast(bf16x8, wp[64]);
                                acc[ry][nt] = mfma16<BF>(wh, act[c][1], acc[ry][nt]);
                                acc[ry][nt] = mfma16<BF>(wl, act[c][0], acc[ry][nt]);
                            }
                        }
                }
            }
        }
        const int x = x0 + n;
#pragma unroll
        for (int ry = 0; ry < RT; ry++) {
            const int y = y0 + ry;
            if (y >= a.Ho || x >= a.Wo) continue;
            float *dst = a.out + (((size_t)z * a.Ho + y) * a.Wo + x) * COUT;
#pragma unroll
            for (int nt = 0; nt < NT; nt++) {
                if (nt * 16 + 4 * g >= COUT) continue;
                const f32x4 v = acc[ry][nt];
                *reinterpret_cast<float4 *>(dst + nt * 16 + 4 * g) = make_float4(v[0], v[1], v[2], v[3]);
#pragma unroll
                for (int i = 0; i < 4; i++) ssum[nt][i] += v[i], ssq[nt][i] = fmaf(v[i], v[i], ssq[nt][i]);
            }
        }
    }
    // statistics, in a fixed order (the image must not depend on which wave finished first): the 16 voxel lanes of a
    // group (butterfly), the workgroup's waves in order, and this workgroup's own row of the partial sums - which
    // zest_costreg_bn adds up in order
    stats_to_row<NT, COUT>(ssum, ssq, red, a.stats, lane, wave);
}

// ---------------------------------------------------------------------------- transposed convolution
// ConvTranspose3d(k 3, stride 2, padding 1, output_padding 1): out[o] = sum_k in[i] w[k] over o = 2 i - 1 + k.  Per
// dimension an even output o = 2 i has the single tap k = 1 (input i), an odd one o = 2 i + 1 the taps k = 2 (input i)
// and k = 0 (input i + 1): the eight parity classes (pz, py, px) of the output lattice are eight small convolutions
// over the SAME 2 x 2 x 2 input neighbourhood.  A tile is 16 consecutive voxels of the input lattice (one (z, y) row):
// the four input rows (z + oz, y + oy) are staged as strips of 17 pixels (as above; the input is
// act(norm(in0)) [+ act(norm(in1))] - the U-Net's skip addition happens here, on load), every class takes its taps from
// them - 27 (tap, class) products per tile, A = the class's packed weights of that tap row, B = the window of 1
// (px = 0) or 2 (px = 1) pixels - and writes its 16 voxels of the output (stride 2 in x).
struct DeconvArgs {
    const float *in0, *pre0, *in1, *pre1;   // [Di,Hi,Wi,CIN] raw, [2,CIN] each
    const uint4 *w;                         // [class 8][oz 2][oy 2][chunk][16-row tile][hi, lo][lane] (CostRegNet._pack_deconv)
    float *out;                             // [2Di,2Hi,2Wi,COUT] raw
    double *stats;
    int Di, Hi, Wi, n_xb, n_tiles;
};

template <int CIN, int COUT, bool IN2, int PASSES>
__global__ __launch_bounds__(256) void deconv3d_mfma_kernel(const DeconvArgs a) {
    constexpr int OPT = CIN / 8, NT = (COUT + 15) / 16, NP = PASSES == 3 ? 2 : 1;
    constexpr int CH1 = (OPT + 3) / 4, CH2 = (2 * OPT + 3) / 4;    // chunks that hold pixel 0 / pixels 0 and 1 of a window
    constexpr int SP = 17, NO = SP * OPT, NR = (NO + 63) / 64, PSTR = OPT + 1, PART_BYTES = SP * PSTR * 16;
    static_assert(OPT == 2 || OPT == 4 || OPT == 8, "channel counts");
    const int lane = threadIdx.x & 63, n = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    __shared__ float red[4][2 * NT * 16];
    __shared__ __attribute__((aligned(16))) char strips[4][NP * PART_BYTES];
    char *const strip = strips[wave];
    float sc0[8], sh0[8], sc1[8], sh1[8];
    {
        const int q = lane % OPT;
#pragma unroll
        for (int e = 0; e < 8; e++) {
            sc0[e] = a.pre0[8 * q + e], sh0[e] = a.pre0[CIN + 8 * q + e];
            if constexpr (IN2) sc1[e] = a.pre1[8 * q + e], sh1[e] = a.pre1[CIN + 8 * q + e];
        }
    }
    float ssum[NT][4], ssq[NT][4];
#pragma unroll
    for (int nt = 0; nt < NT; nt++)
#pragma unroll
        for (int i = 0; i < 4; i++) ssum[nt][i] = ssq[nt][i] = 0.0f;
    float4 pf[IN2 ? 2 : 1][NR][2];
    unsigned pf_ok = 0;
    auto fetch = [&](int zi, int yi, int x0) {
        const bool row_ok = zi < a.Di && yi < a.Hi;
        const size_t row = (((size_t)(row_ok ? zi : 0) * a.Hi + (row_ok ? yi : 0)) * a.Wi + x0) * CIN;
        pf_ok = 0;
#pragma unroll
        for (int r = 0; r < NR; r++) {
            const int i = lane + 64 * r;
            const bool ok = row_ok && i < NO && x0 + i / OPT < a.Wi;
            const size_t off = ok ? row + 8 * i : 0;
            pf[0][r][0] = *reinterpret_cast<const float4 *>(a.in0 + off), pf[0][r][1] = *reinterpret_cast<const float4 *>(a.in0 + off + 4);
            if constexpr (IN2)
                pf[IN2 ? 1 : 0][r][0] = *reinterpret_cast<const float4 *>(a.in1 + off),
                pf[IN2 ? 1 : 0][r][1] = *reinterpret_cast<const float4 *>(a.in1 + off + 4);
            pf_ok |= ok ? (1u << r) : 0u;
        }
    };
    auto stage = [&]() {
#pragma unroll
        for (int r = 0; r < NR; r++) {
            const int i = lane + 64 * r, sp = i / OPT, q = i - sp * OPT;
            if (i >= NO) continue;
            const bool ok = (pf_ok >> r) & 1;
            const float4 *u = pf[0][r];
            float v[8] = {u[0].x, u[0].y, u[0].z, u[0].w, u[1].x, u[1].y, u[1].z, u[1].w};
#pragma unroll
            for (int e = 0; e < 8; e++) v[e] = leaky(fmaf(v[e], sc0[e], sh0[e]));
            if constexpr (IN2) {
                const float4 *t = pf[IN2 ? 1 : 0][r];
                const float s[8] = {t[0].x, t[0].y, t[0].z, t[0].w, t[1].x, t[1].y, t[1].z, t[1].w};
#pragma unroll
                for (int e = 0; e < 8; e++) v[e] += leaky(fmaf(s[e], sc1[e], sh1[e]));
            }
#pragma unroll
            for (int e = 0; e < 8; e++) v[e] = ok ? v[e] : 0.0f;
            bf16x8 op[NP];
            to_operand<NP>(v, op);
            char *dst = strip + (sp * PSTR + q) * 16;
            *reinterpret_cast<bf16x8 *>(dst) = op[0];
            if constexpr (NP == 2) *reinterpret_cast<bf16x8 *>(dst + PART_BYTES) = op[NP - 1];
        }
    };

    for (int t = (int)blockIdx.x * 4 + wave; t < a.n_tiles; t += (int)gridDim.x * 4) {
        const int zi = t % a.Di, r_ = t / a.Di, xb = r_ % a.n_xb, yi = r_ / a.n_xb;      // z fastest, as above
        const int x0 = xb * 16;
        f32x4 acc[8][NT];
#pragma unroll
        for (int cls = 0; cls < 8; cls++)
#pragma unroll
            for (int nt = 0; nt < NT; nt++) acc[cls][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
        fetch(zi, yi, x0);
#pragma unroll 1
        for (int step = 0; step < 4; step++) {
            const int oz = step >> 1, oy = step & 1;
            const bool row_ok = zi + oz < a.Di && yi + oy < a.Hi;
            if (row_ok) stage();
            __builtin_amdgcn_wave_barrier();
            if (step < 3) fetch(zi + ((step + 1) >> 1), yi + ((step + 1) & 1), x0);
            if (!row_ok) continue;
            bf16x8 act[CH2][NP];
#pragma unroll
            for (int c = 0; c < CH2; c++) {
                const int o = 4 * c + g, p = o / OPT, q = o - p * OPT;
#pragma unroll
                for (int pt = 0; pt < NP; pt++)
                    act[c][pt] = *reinterpret_cast<const bf16x8 *>(strip + ((n + p) * PSTR + q) * 16 + pt * PART_BYTES);
            }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int cls = 0; cls < 8; cls++) {
                const int pz = cls >> 2, py = (cls >> 1) & 1, px = cls & 1;
                if ((oz && !pz) || (oy && !py)) continue;           // this class has no tap in this input row
                const uint4 *wrow = a.w + (size_t)(((cls * 2 + oz) * 2 + oy) * CH2) * NT * NP * 64 + lane;
#pragma unroll
                for (int c = 0; c < (px ? CH2 : CH1); c++)
#pragma unroll
                    for (int nt = 0; nt < NT; nt++) {
                        const uint4 *wp = wrow + (size_t)(c * NT + nt) * NP * 64;
                        const bf16x8 wh = __builtin_bit_cast(bf16x8, wp[0]);
                        acc[cls][nt] = mfma16<BF>(wh, act[c][0], acc[cls][nt]);
                        if constexpr (NP == 2) {
                            const bf16x8 wl = __builtin_bit_cast(bf16x8, wp[64]);
                            acc[cls][nt] = mfma16<BF>(wh, act[c][1], acc[cls][nt]);
                            acc[cls][nt] = mfma16<BF>(wl, act[c][0], acc[cls][nt]);
                        }
                    }
                // the weight loads of one class at a time (hoisted over all classes they take 500 registers at 64 -> 32)
                if constexpr (CIN * NT * NP >= 128) __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (x0 + n < a.Wi) {
#pragma unroll
            for (int cls = 0; cls < 8; cls++) {
                const int pz = cls >> 2, py = (cls >> 1) & 1, px = cls & 1;
                float *dst = a.out + ((((size_t)(2 * zi + pz)) * (2 * a.Hi) + (2 * yi + py)) * (2 * a.Wi) + (2 * (x0 + n) + px)) * COUT;
#pragma unroll
                for (int nt = 0; nt < NT; nt++) {
                    if (nt * 16 + 4 * g >= COUT) continue;
                    const f32x4 v = acc[cls][nt];
                    *reinterpret_cast<float4 *>(dst + nt * 16 + 4 * g) = make_float4(v[0], v[1], v[2], v[3]);
#pragma unroll
                    for (int i = 0; i < 4; i++) ssum[nt][i] += v[i], ssq[nt][i] = fmaf(v[i], v[i], ssq[nt][i]);
                }
            }
        }
    }
    stats_to_row<NT, COUT>(ssum, ssq, red, a.stats, lane, wave);
}

// ------------------------------------------------------------------------------------- batch norm constants
// scale = gamma / sqrt(var + eps), shift = beta - mean scale of one layer, from the batch statistics of its raw
// output (training-mode norm: the reference validates with batch statistics, networks.py:629, 644; the running
// estimates are updated as nn.BatchNorm does: momentum, unbiased variance, the step counter) or from the running ones.
__global__ __launch_bounds__(1024) void bn_constants_kernel(const double *__restrict__ stats, int C, double count, const float *__restrict__ gamma,
                                    const float *__restrict__ beta, float eps, int batch_stats, float *running_mean,
                                    float *running_var, float momentum, long long *steps, float *__restrict__ pre, float *__restrict__ moments) {
    __shared__ double part[1024];
    __shared__ double total[128];
    const int V = 2 * C, G = 1024 / V;           // V values (sum, sum of squares per channel), G row groups (C <= 64)
    if (batch_stats) {
        const int rows = (int)stats[(size_t)kStatRows * V];
        const int v = threadIdx.x % V, grp = threadIdx.x / V;
        double s = 0.0;
        if (grp < G) {
#pragma unroll 8
            for (int r = grp; r < rows; r += G) s += stats[(size_t)r * V + v];           // fixed order
        }
        part[threadIdx.x] = s;
        __syncthreads();
        if (threadIdx.x < V) {
            double t = 0.0;
            for (int k = 0; k < G; k++) t += part[k * V + threadIdx.x];
            total[threadIdx.x] = t;
        }
        __syncthreads();
    }
    const int c = threadIdx.x;
    if (c >= C) return;
    float mean, var;
    if (batch_stats) {
        const double m = total[c] / count;
        double v = total[C + c] / count - m * m;
        v = v > 0.0 ? v : 0.0;
        mean = (float)m, var = (float)v;
        if (running_mean) {
            running_mean[c] = (1.0f - momentum) * running_mean[c] + momentum * mean;
            running_var[c] = (1.0f - momentum) * running_var[c] + momentum * (float)(v * (count / (count > 1.0 ? count - 1.0 : 1.0)));
            if (c == 0 && steps) *steps += 1;
        }
    } else {
        mean = running_mean[c], var = running_var[c];
    }
    const float invstd = 1.0f / sqrtf(var + eps), scale = (gamma ? gamma[c] : 1.0f) * invstd;
    pre[c] = scale, pre[C + c] = (beta ? beta[c] : 0.0f) - mean * scale;
    if (moments) moments[c] = mean, moments[C + c] = invstd;         // what a batch-norm backward needs (save_mean, save_invstd)
}

// ------------------------------------------------------------------------------------- norm backward (training)
// Backward of act(norm(r)) with batch statistics on the [M, C] view of channels-last tensors (zest_autograd.CostRegFn):
//   g_y = g_a * (y > 0 ? 1 : 0.01),  xhat = (r - mean) invstd,
//   g_r = gamma invstd (g_y - sum(g_y) / M - xhat sum(g_y xhat) / M),  g_gamma = sum(g_y xhat),  g_beta = sum(g_y).
// Pass 1 (reduce): a thread walks float4 quads of its own channel quad (the grid stride is a multiple of C / 4), the
// lanes of a quad are added by butterflies, the waves in order, one table row per workgroup (kStatRows, as the forward
// statistics).  Pass 2 (totals): the rows in a fixed order -> [2, C] floats.  Pass 3 (apply): elementwise.
__device__ __forceinline__ void bn_bwd_terms(const float4 r, const float4 g, const float *sc, const float *sh, const float *mu,
                                             const float *is, float (&gy)[4], float (&xh)[4]) {
    const float rv[4] = {r.x, r.y, r.z, r.w}, gv[4] = {g.x, g.y, g.z, g.w};
#pragma unroll
    for (int j = 0; j < 4; j++) {
        gy[j] = gv[j] * (fmaf(rv[j], sc[j], sh[j]) > 0.0f ? 1.0f : kSlope);
        xh[j] = (rv[j] - mu[j]) * is[j];
    }
}

__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const float4 *__restrict__ raw, const float4 *__restrict__ g_act,
                                                            const float *__restrict__ pre, const float *__restrict__ moments,
                                                            int C, long long n_quads, double *__restrict__ stats) {
    const int Q = C / 4, q = threadIdx.x % Q;            // Q = 2, 4, 8, 16 divides 256 and the grid stride
    float sc[4], sh[4], mu[4], is[4];
#pragma unroll
    for (int j = 0; j < 4; j++)
        sc[j] = pre[4 * q + j], sh[j] = pre[C + 4 * q + j], mu[j] = moments[4 * q + j], is[j] = moments[C + 4 * q + j];
    float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n_quads; i += (long long)gridDim.x * 256) {
        float gy[4], xh[4];
        bn_bwd_terms(raw[i], g_act[i], sc, sh, mu, is, gy, xh);
#pragma unroll
        for (int j = 0; j < 4; j++) s1[j] += gy[j], s2[j] = fmaf(gy[j], xh[j], s2[j]);
    }
    __shared__ float red[4][2][64];                      // [wave][s1 | s2][channel]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        float a = s1[j], b = s2[j];
        for (int m = 32; m >= Q; m >>= 1) a += __shfl_xor(a, m, 64), b += __shfl_xor(b, m, 64);
        if (lane < Q) red[wave][0][4 * lane + j] = a, red[wave][1][4 * lane + j] = b;
    }
    __syncthreads();
    if (threadIdx.x < 2 * C) {
        const int which = threadIdx.x / C, c = threadIdx.x % C;
        stats[((size_t)blockIdx.x * 2 + which) * C + c] =
            (double)red[0][which][c] + (double)red[1][which][c] + (double)red[2][which][c] + (double)red[3][which][c];
        if (blockIdx.x == 0 && threadIdx.x == 0) stats[(size_t)kStatRows * 2 * C] = (double)gridDim.x;
    }
}

__global__ __launch_bounds__(1024) void bn_bwd_totals_kernel(const double *__restrict__ stats, int C, float *__restrict__ totals) {
    __shared__ double part[1024];
    const int V = 2 * C, G = 1024 / V, rows = (int)stats[(size_t)kStatRows * V];
    const int v = threadIdx.x % V, grp = threadIdx.x / V;
    double s = 0.0;
    if (grp < G) {
#pragma unroll 8
        for (int r = grp; r < rows; r += G) s += stats[(size_t)r * V + v];
    }
    part[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x < V) {
        double t = 0.0;
        for (int k = 0; k < G; k++) t += part[k * V + threadIdx.x];
        totals[threadIdx.x] = (float)t;                  // [0, C): sum g_y = g_beta; [C, 2C): sum g_y xhat = g_gamma
    }
}

__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float4 *__restrict__ raw, const float4 *__restrict__ g_act,
                                                           const float *__restrict__ pre, const float *__restrict__ moments,
                                                           const float *__restrict__ gamma, const float *__restrict__ totals,
                                                           int C, long long n_quads, float inv_m, float4 *__restrict__ g_raw) {
    const int Q = C / 4, q = threadIdx.x % Q;
    float sc[4], sh[4], mu[4], is[4], k0[4], k1[4], k2[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int c = 4 * q + j;
        sc[j] = pre[c], sh[j] = pre[C + c], mu[j] = moments[c], is[j] = moments[C + c];
        k0[j] = (gamma ? gamma[c] : 1.0f) * is[j], k1[j] = totals[c] * inv_m, k2[j] = totals[C + c] * inv_m;
    }
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n_quads; i += (long long)gridDim.x * 256) {
        float gy[4], xh[4], o[4];
        bn_bwd_terms(raw[i], g_act[i], sc, sh, mu, is, gy, xh);
#pragma unroll
        for (int j = 0; j < 4; j++) o[j] = k0[j] * (gy[j] - k1[j] - xh[j] * k2[j]);
        g_raw[i] = make_float4(o[0], o[1], o[2], o[3]);
    }
}

// encoding volume = act(norm(a)) + act(norm(b)) (the last skip addition), [D,H,W,8] -> the reference's [8,D,H,W]
__global__ __launch_bounds__(256) void costreg_out_kernel(const float4 *__restrict__ ra, const float *__restrict__ pa,
                                                          const float4 *__restrict__ rb, const float *__restrict__ pb,
                                                          long long nvox, float *__restrict__ out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nvox) return;
    const float4 a0 = ra[2 * i], a1 = ra[2 * i + 1], b0 = rb[2 * i], b1 = rb[2 * i + 1];
    const float va[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w}, vb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
    for (int c = 0; c < 8; c++)
        out[(size_t)c * nvox + i] = leaky(fmaf(va[c], pa[c], pa[8 + c])) + leaky(fmaf(vb[c], pb[c], pb[8 + c]));
}

template <int CIN, int COUT, int STRIDE, bool PRE, int K = 3, int KD = 3>
int launch_conv(const ConvArgs &a0, int passes, hipStream_t st) {
    ConvArgs a = a0;
    const long long rows = (long long)a.Do * a.Ho * ((a.Wo + 15) / 16);
    // small levels: one row per wave so that the tiles still cover the chip
    const bool small = rows < 8192;
    // rows per wave: 4 (1 on the small levels).  (8 for the first layer - fewer strips per output row, 10 / 8 instead
    // of 6 / 4 - does not fit two waves per SIMD: 500 - 800 spilled registers.)
    constexpr int RTL = 4;
    const int RT = small ? 1 : RTL;
    a.n_xb = (a.Wo + 15) / 16, a.n_yg = (a.Ho + RT - 1) / RT, a.n_tiles = a.Do * a.n_yg * a.n_xb;
    const int grid = a.n_tiles < 4096 ? (a.n_tiles + 3) / 4 : 1024;
#define ZEST_CONV(P, R) hipLaunchKernelGGL((conv3d_mfma_kernel<CIN, COUT, STRIDE, P, PRE, R, K, KD>), dim3(grid), dim3(256), 0, st, a)
    if (passes == 1) { if (small) ZEST_CONV(1, 1); else ZEST_CONV(1, RTL); }
    else { if (small) ZEST_CONV(3, 1); else ZEST_CONV(3, RTL); }
#undef ZEST_CONV
    return 0;
}

}  // namespace

extern "C" int zest_costreg_stat_rows(void) { return kStatRows + 1; }

static size_t conv_packed_bytes(int cin, int cout, int passes, int k, int kd) {
    const int opt = cin / 8, cpr = (k * opt + 3) / 4, nt = (cout + 15) / 16;
    return (size_t)kd * k * cpr * nt * (passes == 3 ? 2 : 1) * 1024;
}
extern "C" size_t zest_costreg_packed_bytes(int cin, int cout, int passes) { return conv_packed_bytes(cin, cout, passes, 3, 3); }
extern "C" size_t zest_conv2d_packed_bytes(int cin, int cout, int k, int passes) { return conv_packed_bytes(cin, cout, passes, k, 1); }

static int conv_check(const char *who, const float *in, const void *w_packed, float *out, double *stats, int stride, int passes,
                      int Di, int Hi, int Wi, int cin) {
    ZEST_CHECK_ARG(in && w_packed && out && stats, "%s: null pointer", who);
    ZEST_CHECK_ARG(((uintptr_t)in | (uintptr_t)out | (uintptr_t)w_packed) % 16 == 0, "%s: pointers must be 16-byte aligned", who);
    ZEST_CHECK_ARG((stride == 1 || stride == 2) && (passes == 1 || passes == 3), "%s: stride %d, passes %d", who, stride, passes);
    ZEST_CHECK_ARG(Di >= 1 && Hi >= 1 && Wi >= 1 && (long long)Di * Hi * Wi * cin < (1ll << 31), "%s: bad shape", who);
    return 0;
}

extern "C" int zest_costreg_conv_fwd(const float *in, const float *pre, const void *w_packed, int cin, int cout, int stride,
                                     int passes, int Di, int Hi, int Wi, float *out, double *stats, void *stream) {
    if (int e = conv_check("zest_costreg_conv_fwd", in, w_packed, out, stats, stride, passes, Di, Hi, Wi, cin)) return e;
    ConvArgs a{};
    a.in = in, a.pre = pre, a.w = (const uint4 *)w_packed, a.out = out, a.stats = stats;
    a.Di = Di, a.Hi = Hi, a.Wi = Wi;
    a.Do = (Di - 1) / stride + 1, a.Ho = (Hi - 1) / stride + 1, a.Wo = (Wi - 1) / stride + 1;
    const hipStream_t st = (hipStream_t)stream;
    const int key = cin * 1000 + cout * 10 + stride;
    ZEST_CHECK_ARG(pre || key == 48081, "zest_costreg_conv_fwd: only the first layer (48 -> 8) reads an un-normalised input");
    switch (key) {          // the layers of CostRegNet (reference networks.py:1007-1013)
    case 48081: launch_conv<48, 8, 1, false>(a, passes, st); break;       // conv0 (41 channels, padded)
    case 8162: launch_conv<8, 16, 2, true>(a, passes, st); break;         // conv1
    case 16161: launch_conv<16, 16, 1, true>(a, passes, st); break;       // conv2
    case 16322: launch_conv<16, 32, 2, true>(a, passes, st); break;       // conv3
    case 32321: launch_conv<32, 32, 1, true>(a, passes, st); break;       // conv4
    case 32642: launch_conv<32, 64, 2, true>(a, passes, st); break;       // conv5
    case 64641: launch_conv<64, 64, 1, true>(a, passes, st); break;       // conv6
    default:
        zest_set_error("zest_costreg_conv_fwd: no kernel for %d -> %d channels at stride %d", cin, cout, stride);
        return (int)hipErrorInvalidValue;
    }
    ZEST_RETURN_LAUNCH("zest_costreg_conv_fwd");
}

// 2-D convolution of a batch of N images [N,Hi,Wi,cin] (channels-last) with the same machinery: Conv2d(cin -> cout, k,
// stride, padding k / 2, no bias) on act(norm(in)) (pre NULL: on `in` itself): the layers of FeatureNet
extern "C" int zest_conv2d_fwd(const float *in, const float *pre, const void *w_packed, int cin, int cout, int k, int stride,
                               int passes, int N, int Hi, int Wi, float *out, double *stats, void *stream) {
    if (int e = conv_check("zest_conv2d_fwd", in, w_packed, out, stats, stride, passes, N, Hi, Wi, cin)) return e;
    ConvArgs a{};
    a.in = in, a.pre = pre, a.w = (const uint4 *)w_packed, a.out = out, a.stats = stats;
    a.Di = N, a.Hi = Hi, a.Wi = Wi;
    a.Do = N, a.Ho = (Hi - 1) / stride + 1, a.Wo = (Wi - 1) / stride + 1;
    const hipStream_t st = (hipStream_t)stream;
    const int key = ((cin * 100 + cout) * 10 + k) * 10 + stride;
    ZEST_CHECK_ARG(pre || key == 80831, "zest_conv2d_fwd: only the first layer (8 -> 8, k 3) reads an un-normalised input");
    if (!pre) launch_conv<8, 8, 1, false, 3, 1>(a, passes, st);           // conv0.0 (3 channels, padded)
    else switch (key) {     // the layers of FeatureNet (reference networks.py:967-979)
    case 80831: launch_conv<8, 8, 1, true, 3, 1>(a, passes, st); break;       // conv0.1
    case 81652: launch_conv<8, 16, 2, true, 5, 1>(a, passes, st); break;      // conv1.0
    case 161631: launch_conv<16, 16, 1, true, 3, 1>(a, passes, st); break;    // conv1.1, conv1.2
    case 163252: launch_conv<16, 32, 2, true, 5, 1>(a, passes, st); break;    // conv2.0
    case 323231: launch_conv<32, 32, 1, true, 3, 1>(a, passes, st); break;    // conv2.1, conv2.2
    default:
        zest_set_error("zest_conv2d_fwd: no kernel for %d -> %d channels, k %d, stride %d", cin, cout, k, stride);
        return (int)hipErrorInvalidValue;
    }
    ZEST_RETURN_LAUNCH("zest_conv2d_fwd");
}

extern "C" size_t zest_costreg_deconv_packed_bytes(int cin, int cout, int passes) {
    const int opt = cin / 8, ch2 = (2 * opt + 3) / 4, nt = (cout + 15) / 16;
    return (size_t)8 * 2 * 2 * ch2 * nt * (passes == 3 ? 2 : 1) * 1024;
}

extern "C" int zest_costreg_deconv_fwd(const float *in0, const float *pre0, const float *in1, const float *pre1,
                                       const void *w_packed, int cin, int cout, int passes, int Di, int Hi, int Wi,
                                       float *out, double *stats, void *stream) {
    ZEST_CHECK_ARG(in0 && pre0 && w_packed && out && stats && (!in1 || pre1), "zest_costreg_deconv_fwd: null pointer");
    ZEST_CHECK_ARG(((uintptr_t)in0 | (uintptr_t)in1 | (uintptr_t)out | (uintptr_t)w_packed) % 16 == 0,
                   "zest_costreg_deconv_fwd: pointers must be 16-byte aligned");
    ZEST_CHECK_ARG(passes == 1 || passes == 3, "zest_costreg_deconv_fwd: passes %d", passes);
    ZEST_CHECK_ARG(Di >= 1 && Hi >= 1 && Wi >= 1 && (long long)Di * Hi * Wi * 8 * cout < (1ll << 31), "zest_costreg_deconv_fwd: bad shape");
    DeconvArgs a{in0, pre0, in1, pre1, (const uint4 *)w_packed, out, stats, Di, Hi, Wi, 0, 0};
    a.n_xb = (Wi + 15) / 16, a.n_tiles = Di * Hi * a.n_xb;
    const dim3 grid(a.n_tiles < 4096 ? (a.n_tiles + 3) / 4 : 1024), block(256);
    const hipStream_t st = (hipStream_t)stream;
    const int key = cin * 100 + cout;
#define ZEST_DECONV_P(CI, CO, TWO)                                                                           \
    if (passes == 1) hipLaunchKernelGGL((deconv3d_mfma_kernel<CI, CO, TWO, 1>), grid, block, 0, st, a);     \
    else hipLaunchKernelGGL((deconv3d_mfma_kernel<CI, CO, TWO, 3>), grid, block, 0, st, a)
#define ZEST_DECONV(CI, CO)                                \
    if (in1) { ZEST_DECONV_P(CI, CO, true); }              \
    else { ZEST_DECONV_P(CI, CO, false); }
    if (key == 6432) { ZEST_DECONV(64, 32) }             // conv7
    else if (key == 3216) { ZEST_DECONV(32, 16) }        // conv9
    else if (key == 1608) { ZEST_DECONV(16, 8) }         // conv11
    else {
        zest_set_error("zest_costreg_deconv_fwd: no kernel for %d -> %d channels", cin, cout);
        return (int)hipErrorInvalidValue;
    }
#undef ZEST_DECONV
#undef ZEST_DECONV_P
    ZEST_RETURN_LAUNCH("zest_costreg_deconv_fwd");
}

extern "C" int zest_costreg_bn(const double *stats, int C, long long count, const float *gamma, const float *beta, float eps,
                               int batch_stats, float *running_mean, float *running_var, float momentum,
                               long long *steps, float *pre, float *moments, void *stream) {
    ZEST_CHECK_ARG(pre && C >= 1 && C <= 64 && count >= 1, "zest_costreg_bn: bad argument (at most 64 channels)");
    ZEST_CHECK_ARG(batch_stats ? stats != nullptr : (running_mean && running_var), "zest_costreg_bn: statistics missing");
    hipLaunchKernelGGL(bn_constants_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, stats, C, (double)count, gamma, beta,
                       eps, batch_stats, running_mean, running_var, momentum, steps, pre, moments);
    ZEST_RETURN_LAUNCH("zest_costreg_bn");
}

extern "C" int zest_costreg_bn_bwd(const float *raw, const float *g_act, const float *pre, const float *moments,
                                   const float *gamma, int C, long long M, double *stats, float *totals, float *g_raw,
                                   void *stream) {
    ZEST_CHECK_ARG(raw && g_act && pre && moments && stats && totals && g_raw, "zest_costreg_bn_bwd: null pointer");
    ZEST_CHECK_ARG((C == 8 || C == 16 || C == 32 || C == 64) && M >= 1, "zest_costreg_bn_bwd: %d channels, %lld voxels", C, M);
    ZEST_CHECK_ARG(((uintptr_t)raw | (uintptr_t)g_act | (uintptr_t)g_raw) % 16 == 0, "zest_costreg_bn_bwd: pointers must be 16-byte aligned");
    const long long nq = M * (C / 4);
    const int grid = nq < 1024ll * 256 ? zest_div_up(nq, 256) : 1024;          // <= kStatRows; 256 % (C / 4) == 0
    const hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3(grid), dim3(256), 0, st, (const float4 *)raw, (const float4 *)g_act, pre, moments, C, nq, stats);
    hipLaunchKernelGGL(bn_bwd_totals_kernel, dim3(1), dim3(1024), 0, st, (const double *)stats, C, totals);
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(grid), dim3(256), 0, st, (const float4 *)raw, (const float4 *)g_act, pre, moments, gamma,
                       (const float *)totals, C, nq, 1.0f / (float)M, (float4 *)g_raw);
    ZEST_RETURN_LAUNCH("zest_costreg_bn_bwd");
}

extern "C" int zest_costreg_out(const float *raw_a, const float *pre_a, const float *raw_b, const float *pre_b, int D,
                                int H, int W, float *volume, void *stream) {
    ZEST_CHECK_ARG(raw_a && pre_a && raw_b && pre_b && volume, "zest_costreg_out: null pointer");
    const long long nvox = (long long)D * H * W;
    ZEST_CHECK_ARG(nvox >= 1 && nvox < (1ll << 31), "zest_costreg_out: bad shape");
    hipLaunchKernelGGL(costreg_out_kernel, dim3(zest_div_up(nvox, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const float4 *)raw_a, pre_a, (const float4 *)raw_b, pre_b, nvox, volume);
    ZEST_RETURN_LAUNCH("zest_costreg_out");
}
