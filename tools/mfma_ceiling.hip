// What this chip sustains on the instruction mix of the fused renderer's network, without the renderer:
// 256 workgroups x 8 waves (2 per SIMD, as the fused kernel) issuing v_mfma_f32_16x16x32_bf16 into four
// independent accumulators per wave, in three mixes
//   mfma       registers only
//   mfma+lds   one ds_read_b128 weight tile per two MFMAs (the engine's A-operand traffic: 1 KiB per wave per CB=2 pair)
//   mfma+lds+epi  additionally, per 32 MFMAs, the layer epilogue's VALU work on 16 accumulator registers
//   ...+barrier   additionally one s_barrier per 32 MFMAs (the weight ring's rendezvous)
//   "half a block behind": waves 4-7 (the SIMD partners of waves 0-3) run 16 MFMAs behind, so one wave's epilogue
//                 meets the other's MFMA run instead of the other's epilogue
// for launches of ~0.1 ms, ~1 ms and ~10 ms (board power / clock management has time constants), reporting
// TFLOP/s, the fraction of the 2.5 PFLOP/s dense peak and the in-kernel clock (s_memtime cycles per 100 MHz
// s_memrealtime tick).  The number to read the renderer's roofline fraction against.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_ceiling tools/mfma_ceiling.hip && /tmp/mfma_ceiling
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) unsigned v4u;

#define CHECK(x)                                                                         \
    do {                                                                                 \
        hipError_t e_ = (x);                                                             \
        if (e_ != hipSuccess) {                                                          \
            std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                 \
            std::exit(1);                                                                \
        }                                                                                \
    } while (0)

template <int MIX, int STAG>
__global__ __launch_bounds__(512, 1) void mix_kernel(int iters, int zero_data, float *out, unsigned long long *clk) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];     // 128 KiB "ring"
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // RANDOM operands (the clock the chip holds under an MFMA load depends on the data: MI355X_MICROARCH.md, DVFS
    // give-back; zero or trivial operands overstate it).  Weights ~ U(-a, a) with a chosen so that a 256-term dot
    // product followed by ReLU keeps the activations O(1) from block to block; `zero_data` != 0 fills zeros instead.
    unsigned rng = 0x9E3779B9u * (blockIdx.x * 512 + threadIdx.x + 1);
    auto next = [&]() { rng ^= rng << 13, rng ^= rng >> 17, rng ^= rng << 5; return (float)(rng >> 8) * (1.0f / 8388608.0f) - 1.0f; };
    for (int i = threadIdx.x; i < 65536; i += 512) ((__bf16 *)smem)[i] = (__bf16)(zero_data ? 0.0f : 0.153f * next());
    __syncthreads();
    bf16x8 b0, b1;
    for (int e = 0; e < 8; e++) b0[e] = (__bf16)(zero_data ? 0.0f : next()), b1[e] = (__bf16)(zero_data ? 0.0f : next());
    f32x4 acc[4] = {};
    const unsigned long long t0 = __builtin_readcyclecounter(), r0 = wall_clock64();
    const v4u *tiles = (const v4u *)smem + lane;
    // Weight tiles are read four MFMA pairs ahead into a ring of eight register slots (counted lgkmcnt waits,
    // as the engine's hand-pipelined tile reads); address = LDS byte offset of this lane's 16 bytes of a tile.
    v4u w[8];
    float keep = 0.f;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)smem + lane * 16;
#define TILE_READ(slot, t) asm volatile("ds_read_b128 %0, %1" : "=v"(w[slot]) : "v"(lds0 + (unsigned)(t) * 1024u))
    if (MIX >= 1)
        for (int k = 0; k < 4; k++) TILE_READ(k, k);
    // k-steps k0 .. k1-1 of row block `it`: wait for the tile, two MFMAs, read four steps ahead
#define KSTEPS(it, k0, k1)                                                                              \
    _Pragma("unroll") for (int k = k0; k < k1; k++) {                                                   \
        bf16x8 a = b0;                                                                                  \
        if (MIX >= 1) {                                                                                 \
            asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(w[k & 7]));                                      \
            a = __builtin_bit_cast(bf16x8, w[k & 7]);                                                   \
        }                                                                                               \
        acc[(2 * k) & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b0, acc[(2 * k) & 3], 0, 0, 0);    \
        acc[(2 * k + 1) & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b1, acc[(2 * k + 1) & 3], 0, 0, 0); \
        if (MIX >= 1) TILE_READ((k + 4) & 7, ((unsigned)((it) & 7) * 16 + k + 4) & 127);                \
    }
    // epilogue of a row block: relu, round to bf16, pack -> next layer's operand registers; bias initialiser
#define EPILOGUE()                                                                                      \
    if (MIX >= 2) {                                                                                     \
        keep += acc[0][0];                                                                              \
        _Pragma("unroll") for (int j = 0; j < 4; j++)                                                   \
            for (int e = 0; e < 4; e++) acc[j][e] = __builtin_amdgcn_fmed3f(acc[j][e], 0.0f, 3.0e38f);     \
        for (int e = 0; e < 4; e++) {                                                                   \
            b0[e] = (__bf16)acc[0][e], b0[4 + e] = (__bf16)acc[1][e];                                   \
            b1[e] = (__bf16)acc[2][e], b1[4 + e] = (__bf16)acc[3][e];                                   \
        }                                                                                               \
        for (int j = 0; j < 4; j++) acc[j] = f32x4{0.01f, -0.02f, 0.03f, -0.04f};                       \
    }
    if (STAG && wave >= 4) {
        // the SIMD partners of waves 0-3 run half a row block behind: their epilogue falls into the middle of the
        // partner's MFMA run, and the barrier (one per row block, as the weight ring's rendezvous) into the middle of theirs
        KSTEPS(0, 0, 8)
#pragma unroll 1
        for (int it = 0; it < iters; it++) {
            KSTEPS(it, 8, 16)
            EPILOGUE()
            KSTEPS(it + 1, 0, 8)
            if (MIX >= 3) __builtin_amdgcn_s_barrier();
        }
    } else {
#pragma unroll 1
        for (int it = 0; it < iters; it++) {
            KSTEPS(it, 0, 16)
            EPILOGUE()
            if (MIX >= 3) __builtin_amdgcn_s_barrier();
        }
    }
    if (MIX >= 1) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]));
    if (MIX >= 1) asm volatile("" : "+v"(w[4]), "+v"(w[5]), "+v"(w[6]), "+v"(w[7]));
    const unsigned long long t1 = __builtin_readcyclecounter(), r1 = wall_clock64();
    float s = keep;
    for (int j = 0; j < 4; j++) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
    if (s == 123.456f) out[0] = s;                        // keep the chain alive
    if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = t1 - t0, clk[1] = r1 - r0;
    (void)wave;
}

template <int MIX, int STAG>
void run(const char *name, int waves_per_cu, int zero_data, float *out, unsigned long long *clk) {
    const int cus = 256;
    for (int iters : {600, 60000}) {
        CHECK(hipFuncSetAttribute((const void *)&mix_kernel<MIX, STAG>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0));
        CHECK(hipEventCreate(&e1));
        const int reps = iters <= 600 ? 50 : (iters <= 6000 ? 10 : 3);
        for (int w = 0; w < 3; w++)
            hipLaunchKernelGGL((mix_kernel<MIX, STAG>), dim3(cus), dim3(64 * waves_per_cu), 131072, 0, iters, zero_data, out, clk);
        CHECK(hipEventRecord(e0));
        for (int r = 0; r < reps; r++)
            hipLaunchKernelGGL((mix_kernel<MIX, STAG>), dim3(cus), dim3(64 * waves_per_cu), 131072, 0, iters, zero_data, out, clk);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        ms /= reps;
        unsigned long long h[2];
        CHECK(hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost));
        const double flop = (double)cus * waves_per_cu * iters * 32.0 * 2.0 * 16 * 16 * 32;
        const double tf = flop / (ms * 1e-3) / 1e12, ghz = (double)h[0] / ((double)h[1] / 100e6) / 1e9;
        std::printf("{\"mix\": \"%s\", \"data\": \"%s\", \"waves_per_cu\": %d, \"launch_ms\": %.3f, \"tflops\": %.0f, \"frac_of_2500\": %.3f, "
                    "\"clock_ghz\": %.2f}\n", name, zero_data ? "zeros" : "random", waves_per_cu, ms, tf, tf / 2500.0, ghz);
        std::fflush(stdout);
    }
}

int main() {
    float *out;
    unsigned long long *clk;
    CHECK(hipMalloc(&out, 64));
    CHECK(hipMalloc(&clk, 64));
    // ~2 s of back-to-back launches first: the clock under load is a steady-state quantity
    for (int zero_data : {0, 1})
        for (int wpc : {8, 4}) {
            if (zero_data && wpc == 4) continue;
            run<0, 0>("mfma", wpc, zero_data, out, clk);
            run<1, 0>("mfma+lds", wpc, zero_data, out, clk);
            run<2, 0>("mfma+lds+epi", wpc, zero_data, out, clk);
            run<3, 0>("mfma+lds+epi+barrier", wpc, zero_data, out, clk);
            if (wpc == 8) {
                run<2, 1>("mfma+lds+epi, waves 4-7 half a block behind", wpc, zero_data, out, clk);
                run<3, 1>("mfma+lds+epi+barrier, waves 4-7 half a block behind", wpc, zero_data, out, clk);
            }
        }
    return 0;
}
