// Does replaying the evaluation as a captured hipGraph shorten it?  Times lcgp_nll_grad (n=4096, fp64) launched
// kernel by kernel against the same launches captured once and replayed with hipGraphLaunch, for q = 1 and 8,
// plus a chain of 1000 dependent empty kernels both ways (the bare launch-boundary cost).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <random>
#include "../include/lcgp_hip.h"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)
__global__ void empty_kernel(int* p) { if (p && threadIdx.x == 9999) *p = 1; }
int main() {
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float ms;
    {   // bare boundary
        const int N = 1000;
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipEventRecord(e0, st));
            for (int i = 0; i < N; ++i) empty_kernel<<<256, 256, 0, st>>>(nullptr);
            CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
            printf("plain  : %d empty kernels %.3f ms = %.2f us each\n", N, ms, ms * 1e3 / N);
        }
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        for (int i = 0; i < N; ++i) empty_kernel<<<256, 256, 0, st>>>(nullptr);
        CK(hipStreamEndCapture(st, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0, st)); CK(hipGraphLaunch(ge, st));
            CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
            printf("graph  : %d empty kernels %.3f ms = %.2f us each\n", N, ms, ms * 1e3 / N);
        }
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    }
    const int n = 4096, d = 6, p = 64;
    for (int q : {1, 8}) {
        size_t wsb; lcgp_workspace_bytes(0, n, d, p, q, &wsb);
        void *ws, *dx, *dY; double *dth, *dout;
        int tw = lcgp_theta_width(d, p), ow = lcgp_out_width(d, p);
        CK(hipMalloc(&ws, wsb)); CK(hipMalloc(&dx, n * d * 8)); CK(hipMalloc(&dY, (size_t)p * n * 8));
        std::vector<double> x(n * d), Y((size_t)p * n), th(q * tw);
        std::mt19937_64 g(1); std::uniform_real_distribution<double> U(0, 1);
        for (auto& v : x) v = U(g);
        for (auto& v : Y) v = U(g) - 0.5;
        for (int k = 0; k < q; ++k) { double* t = &th[k * tw]; for (int j = 0; j < d; ++j) t[j] = 0.5 + U(g); t[d] = 1; t[d + 1] = 1e-4; t[d + 2] = 1.0; for (int a = 0; a < p; ++a) t[d + 3 + a] = U(g); }
        CK(hipMalloc(&dth, th.size() * 8)); CK(hipMalloc(&dout, (size_t)q * ow * 8));
        CK(hipMemcpy(dx, x.data(), x.size() * 8, hipMemcpyHostToDevice));
        CK(hipMemcpy(dY, Y.data(), Y.size() * 8, hipMemcpyHostToDevice));
        CK(hipMemcpy(dth, th.data(), th.size() * 8, hipMemcpyHostToDevice));
        std::vector<double> o1((size_t)q * ow), o2((size_t)q * ow);
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0, st));
            if (lcgp_nll_grad(st, 0, 0, n, d, p, q, dx, dY, nullptr, dth, ws, dout, nullptr, nullptr)) { printf("nll_grad failed: %s\n", lcgp_last_error()); return 1; }
            CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
            printf("q=%d plain  nll_grad %.3f ms\n", q, ms);
        }
        CK(hipMemcpy(o1.data(), dout, o1.size() * 8, hipMemcpyDeviceToHost));
        hipGraph_t gr; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        if (lcgp_nll_grad(st, 0, 0, n, d, p, q, dx, dY, nullptr, dth, ws, dout, nullptr, nullptr)) { printf("capture failed: %s\n", lcgp_last_error()); return 1; }
        CK(hipStreamEndCapture(st, &gr));
        size_t nn = 0; CK(hipGraphGetNodes(gr, nullptr, &nn));
        CK(hipGraphInstantiate(&ge, gr, nullptr, nullptr, 0));
        for (int rep = 0; rep < 4; ++rep) {
            CK(hipEventRecord(e0, st)); CK(hipGraphLaunch(ge, st));
            CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
            printf("q=%d graph  nll_grad %.3f ms (%zu nodes)\n", q, ms, nn);
        }
        CK(hipMemcpy(o2.data(), dout, o2.size() * 8, hipMemcpyDeviceToHost));
        double md = 0; for (size_t i = 0; i < o1.size(); ++i) md = std::max(md, std::abs(o1[i] - o2[i]));
        printf("q=%d max |plain - graph| = %.3e\n", q, md);
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(gr));
        hipFree(ws); hipFree(dx); hipFree(dY); hipFree(dth); hipFree(dout);
    }
    return 0;
}
