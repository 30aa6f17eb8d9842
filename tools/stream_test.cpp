// Times lcgp_potrf_logdet (n=4096, q=8, fp64) on the null stream vs a created stream, single-stream schedule,
// outside of torch (ROCm 7.2 runtime from /opt/rocm).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <random>
#include "../include/lcgp_hip.h"
int main() {
    const int n = 4096, d = 6, p = 64, q = 8;
    size_t wsb; lcgp_workspace_bytes(0, n, d, p, q, &wsb);
    void *ws, *dx; double* dth;
    hipMalloc(&ws, wsb); hipMalloc(&dx, n * d * 8);
    int tw = lcgp_theta_width(d, p);
    std::vector<double> x(n * d), th(q * tw);
    std::mt19937_64 g(1); std::uniform_real_distribution<double> U(0, 1);
    for (auto& v : x) v = U(g);
    for (int k = 0; k < q; ++k) { double* t = &th[k * tw]; for (int j = 0; j < d; ++j) t[j] = 0.5 + U(g); t[d] = 1; t[d + 1] = 1e-4; t[d + 2] = 1.0; for (int a = 0; a < p; ++a) t[d + 3 + a] = U(g); }
    hipMalloc(&dth, th.size() * 8);
    hipMemcpy(dx, x.data(), x.size() * 8, hipMemcpyHostToDevice);
    hipMemcpy(dth, th.data(), th.size() * 8, hipMemcpyHostToDevice);
    hipStream_t s1, s2; hipStreamCreateWithFlags(&s1, hipStreamNonBlocking); hipStreamCreate(&s2);
    hipStream_t streams[3] = {0, s1, s2};
    const char* names[3] = {"null stream", "hipStreamNonBlocking", "hipStreamCreate"};
    lcgp_set_tuning(3, 0);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep)
        for (int si = 0; si < 3; ++si) {
            hipStream_t st = streams[si];
            lcgp_kernel_build(st, 0, n, d, p, q, dx, nullptr, dth, ws);
            hipEventRecord(e0, st);
            lcgp_potrf_logdet(st, 0, n, d, p, q, ws, nullptr, nullptr);
            hipEventRecord(e1, st);
            hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("%-22s potrf %.3f ms\n", names[si], ms);
        }
    for (int la = 1; la <= 1; ++la) {
        lcgp_set_tuning(3, la);
        for (int rep = 0; rep < 2; ++rep) {
            lcgp_kernel_build(0, 0, n, d, p, q, dx, nullptr, dth, ws);
            hipEventRecord(e0, 0);
            lcgp_potrf_logdet(0, 0, n, d, p, q, ws, nullptr, nullptr);
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("lookahead=%d (null + internal chain stream) potrf %.3f ms\n", la, ms);
        }
    }
    return 0;
}
