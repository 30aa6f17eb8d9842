// Measures what the fp64 / fp32 16x16x4 MFMA pipe of this device actually sustains (registers only, no memory),
// and the clock it holds meanwhile -- the practical ceiling behind the nominal 78.6 / 157.3 TFLOP/s.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(1024) void k64(double* out, int iters, unsigned long long* clk) {
    extern __shared__ double pad[];   // large dynamic LDS: exactly one workgroup per CU
    if (iters < 0) pad[threadIdx.x] = 1.0;
    d4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = {0, 0, 0, 0};
    double a = threadIdx.x * 1e-3 + 0.5, b = 1.0 - threadIdx.x * 1e-3;
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}
template <int NACC>
__global__ __launch_bounds__(1024) void k32(float* out, int iters) {
    extern __shared__ double pad[];
    if (iters < 0) pad[threadIdx.x] = 1.0;
    f4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = {0, 0, 0, 0};
    float a = threadIdx.x * 1e-3f + 0.5f, b = 1.0f - threadIdx.x * 1e-3f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
    const int blocks = 256, iters = 20000;
    double* out; unsigned long long* clk;
    hipMalloc(&out, blocks * 1024 * 8); hipMalloc(&clk, blocks * 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipFuncSetAttribute((const void*)k64<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    hipFuncSetAttribute((const void*)k32<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    for (int rep = 0; rep < 6; ++rep) {
        const int threads = rep < 2 ? 256 : (rep < 4 ? 512 : 1024);   // 1, 2, 4 waves per SIMD
        hipEventRecord(e0);
        hipLaunchKernelGGL((k64<4>), dim3(blocks), dim3(threads), 100 * 1024, 0, out, iters, clk);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double fl = (double)blocks * (threads / 64) * iters * 4 * 2048.0;
        printf("waves/SIMD %d  ", threads / 256);
        unsigned long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
        printf("f64 16x16x4: %.2f ms  %.1f TFLOP/s   cycles/mfma/wave %.1f  clock %.0f MHz\n", ms, fl / ms / 1e9,
               (double)h[0] / (iters * 4.0), (double)h[0] / (double)h[1] * 100.0);
    }
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k32<4>), dim3(blocks), dim3(1024), 100 * 1024, 0, (float*)out, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double fl = (double)blocks * 16 * iters * 4 * 2048.0;
        printf("f32 16x16x4: %.2f ms  %.1f TFLOP/s\n", ms, fl / ms / 1e9);
    }
    return 0;
}
