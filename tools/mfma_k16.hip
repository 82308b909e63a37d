// Issue cost of the K = 16 bf16 MFMA (v_mfma_f32_16x16x16_bf16) against the K = 32 one the engine uses
// (v_mfma_f32_16x16x32_bf16) on gfx950: one workgroup of 8 waves per CU, four independent accumulators per wave,
// cycles per instruction from s_memtime.  hipcc --offload-arch=gfx950 -O3 tools/mfma_k16.hip -o tools/bin/mfma_k16
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) short bf16x4;

template <int K>
__global__ __launch_bounds__(512) void loop(float *out, unsigned long long *cyc, int iters) {
    f32x4 acc[4] = {};
    bf16x8 a8 = {1, 2, 3, 4, 5, 6, 7, (short)threadIdx.x}, b8 = {7, 6, 5, 4, 3, 2, 1, (short)threadIdx.x};
    bf16x4 a4 = {1, 2, 3, (short)threadIdx.x}, b4 = {3, 2, 1, (short)threadIdx.x};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int j = 0; j < 16; j++) {
            if (K == 32) acc[j & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a8, b8, acc[j & 3], 0, 0, 0);
            else acc[j & 3] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a4, b4, acc[j & 3], 0, 0, 0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 512 + threadIdx.x] = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
    float *out;
    unsigned long long *cyc, h[256];
    hipMalloc(&out, 256 * 512 * 4);
    hipMalloc(&cyc, 256 * 8);
    const int iters = 20000;
    for (int K : {32, 16, 32, 16}) {
        if (K == 32) hipLaunchKernelGGL(loop<32>, dim3(256), dim3(512), 0, 0, out, cyc, iters);
        else hipLaunchKernelGGL(loop<16>, dim3(256), dim3(512), 0, 0, out, cyc, iters);
        hipDeviceSynchronize();
        hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
        double s = 0;
        for (int i = 0; i < 256; i++) s += (double)h[i];
        // two waves per SIMD share the pipe: cycles per MFMA per SIMD = wave cycles / (2 waves x instructions per wave)
        printf("K=%d: %.2f cycles per MFMA per SIMD (2 waves per SIMD, 4 accumulators each)\n", K,
               s / 256 / (2.0 * iters * 16));
    }
    return 0;
}
