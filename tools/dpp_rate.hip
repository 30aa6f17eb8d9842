// Issue rate of fp64 VALU instructions with and without the DPP row_newbcast modifier, one wave, s_memtime ticks per
// instruction (hipcc -O3 --offload-arch=gfx950 -o dpp_rate tools/dpp_rate.hip; run on the GPU box).  Measured: v_fmac_f64 5.1,
// v_fmac_f64_dpp 5.5, v_mov_b64(_dpp) 7.5, v_mul_f64 7.2; a dependent v_fmac_f64 chain 8.7 with or without DPP.
#include <hip/hip_runtime.h>
#include <cstdio>
// issue rate of fp64 DPP ops on one wave: cycles per instruction (s_memtime), 16 independent accumulators
template <int MODE>
__global__ void k(double* out, unsigned long long* cyc) {
    double a[16];
    for (int i = 0; i < 16; ++i) a[i] = out[threadIdx.x + 64 * i];
    double u = out[threadIdx.x + 2048], v = out[threadIdx.x + 4096];
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < 256; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (MODE == 0) asm volatile("v_fmac_f64_e32 %0, %1, %2" : "+v"(a[i]) : "v"(u), "v"(v));
            if (MODE == 1) asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(u), "v"(v));
            if (MODE == 2) asm volatile("v_mov_b64_dpp %0, %1 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "=v"(a[i]) : "v"(u));
            if (MODE == 3) asm volatile("v_mov_b64 %0, %1" : "=v"(a[i]) : "v"(u));
            if (MODE == 4) asm volatile("v_mul_f64 %0, %1, %2" : "=v"(a[i]) : "v"(u), "v"(v));
            if (MODE == 5) asm volatile("v_fmac_f64_e32 %0, %1, %2" : "+v"(a[0]) : "v"(u), "v"(v));   // dependent chain
            if (MODE == 6) asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(a[0]) : "v"(u), "v"(v));
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    double s = 0; for (int i = 0; i < 16; ++i) s += a[i];
    out[threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[MODE] = t1 - t0;
}
int main() {
    double* d; unsigned long long* c;
    hipMalloc(&d, 8192 * 8); hipMalloc(&c, 64); hipMemset(d, 0, 8192 * 8);
    k<0><<<1, 64>>>(d, c); k<1><<<1, 64>>>(d, c); k<2><<<1, 64>>>(d, c); k<3><<<1, 64>>>(d, c); k<4><<<1, 64>>>(d, c); k<5><<<1, 64>>>(d, c); k<6><<<1, 64>>>(d, c);
    k<0><<<1, 64>>>(d, c); k<1><<<1, 64>>>(d, c); k<2><<<1, 64>>>(d, c); k<3><<<1, 64>>>(d, c); k<4><<<1, 64>>>(d, c); k<5><<<1, 64>>>(d, c); k<6><<<1, 64>>>(d, c);
    unsigned long long h[8]; hipMemcpy(h, c, 64, hipMemcpyDeviceToHost);
    const char* names[] = {"v_fmac_f64 (independent)", "v_fmac_f64_dpp row_newbcast (independent)", "v_mov_b64_dpp row_newbcast", "v_mov_b64", "v_mul_f64", "v_fmac_f64 (dependent chain)", "v_fmac_f64_dpp (dependent chain)"};
    for (int m = 0; m < 7; ++m) printf("%-44s %.2f counter ticks per instruction\n", names[m], h[m] / (256.0 * 16));
    return 0;
}
