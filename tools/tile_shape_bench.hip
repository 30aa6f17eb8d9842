// Which fp64 MFMA tile shape holds its rate with ONE workgroup per CU?  
//
// C (M x N) (-)= A (M x K) B (N x K)^T on 16x16x4 fp64 MFMAs, operands row-major with k contiguous (the layout of the rank-k
// update of the factorisation), staged through LDS like lcgp_hip.hip's gemm_body (16 k rows per stage, [k][m] image with the
// column XOR-ed by k & 12, register prefetch).  Variants:
//   128x128, 8 waves (32x64 per wave), two workgroups per CU      -- what the library's wide launches are
//   128x128, 8 waves, ONE workgroup per CU (extra dynamic LDS)    -- what a launch that also hosts a 256-register body gets
//   256x128, 8 waves (64x64 per wave), one workgroup per CU       -- half the LDS fragment traffic per flop
// for K = 256 with C preloaded (read-modify-write, the trailing update) and K = 2048 (write only, the long products).
//   hipcc -O3 --offload-arch=gfx950 -o tools/tile_shape_bench tools/tile_shape_bench.hip && tools/tile_shape_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
#include <type_traits>

typedef double d4 __attribute__((ext_vector_type(4)));
constexpr int KT = 16;

template <int TMX, int NT>
__device__ __forceinline__ void load_stage(const double* __restrict__ P, int ld, int ks, double (&reg)[TMX * KT / NT], int tid) {
    constexpr int EPT = TMX * KT / NT, TPR = KT / EPT;
    const int m = tid / TPR, kk = (tid % TPR) * EPT;
    const double* src = P + (size_t)m * ld + ks + kk;
#pragma unroll
    for (int e = 0; e < EPT; ++e) reg[e] = src[e];
}
template <int TMX, int NT>
__device__ __forceinline__ void store_stage(double* __restrict__ S, const double (&reg)[TMX * KT / NT], int tid) {
    constexpr int EPT = TMX * KT / NT, TPR = KT / EPT, LD = TMX + 16;
    const int m = tid / TPR, kk = (tid % TPR) * EPT;
#pragma unroll
    for (int e = 0; e < EPT; ++e) S[(kk + e) * LD + (m ^ ((kk + e) & 12))] = reg[e];
}

// operands stored with m contiguous (element (m, k) at P[k ld + m]: the layout of W in W^T W): 16-byte pieces per lane
template <int TMX, int NT>
__device__ __forceinline__ void load_stage_km(const double* __restrict__ P, int ld, int ks, double (&reg)[TMX * KT / NT], int tid) {
    constexpr int EPT = TMX * KT / NT, TPK = TMX / EPT, NP = EPT / 2;
    const int kq = tid / TPK, j = tid % TPK;
    const double* src = P + (size_t)(ks + kq) * ld + j * 2;
#pragma unroll
    for (int pc = 0; pc < NP; ++pc)
#pragma unroll
        for (int e = 0; e < 2; ++e) reg[pc * 2 + e] = src[pc * (TMX / NP) + e];
}
template <int TMX, int NT>
__device__ __forceinline__ void store_stage_km(double* __restrict__ S, const double (&reg)[TMX * KT / NT], int tid) {
    constexpr int EPT = TMX * KT / NT, TPK = TMX / EPT, NP = EPT / 2, LD = TMX + 16;
    const int kq = tid / TPK, j = tid % TPK;
#pragma unroll
    for (int pc = 0; pc < NP; ++pc)
#pragma unroll
        for (int e = 0; e < 2; ++e) S[kq * LD + ((j * 2 + pc * (TMX / NP)) ^ (kq & 12)) + e] = reg[pc * 2 + e];
}

template <int TM, int TN, int WM, int WN, int PF, bool PRELOAD, int MINW, bool KMAJ = false>
__global__ __launch_bounds__(64 * WM * WN, MINW) void tile_kernel(const double* __restrict__ A, const double* __restrict__ B,
                                                                    double* __restrict__ C, int ldA, int ldB, int ldC, int nkt,
                                                                    int ntn) {
    constexpr int NT = 64 * WM * WN, LDA = TM + 16, LDB = TN + 16;
    constexpr int WTM = TM / WM, WTN = TN / WN, MIM = WTM / 16, MIN = WTN / 16;
    constexpr int EA = TM * KT / NT, EB = TN * KT / NT;
    extern __shared__ __align__(16) unsigned char lds_raw[];
    double* As = (double*)lds_raw;            // [2][KT * LDA]
    double* Bs = As + 2 * KT * LDA;           // [2][KT * LDB]
    const int tid = threadIdx.x;
    const int tm = blockIdx.x / ntn, tn = blockIdx.x % ntn;
    const double* A0 = KMAJ ? A + (size_t)tm * TM : A + (size_t)tm * TM * ldA;
    const double* B0 = KMAJ ? B + (size_t)tn * TN : B + (size_t)tn * TN * ldB;
    auto ldst = [&](const double* P, int ld, int ks, auto& reg, auto tag) {
        constexpr int X = decltype(tag)::value;
        if constexpr (KMAJ) load_stage_km<X, NT>(P, ld, ks, reg, tid); else load_stage<X, NT>(P, ld, ks, reg, tid);
    };
    auto stst = [&](double* S, auto& reg, auto tag) {
        constexpr int X = decltype(tag)::value;
        if constexpr (KMAJ) store_stage_km<X, NT>(S, reg, tid); else store_stage<X, NT>(S, reg, tid);
    };
    std::integral_constant<int, TM> tgm;
    std::integral_constant<int, TN> tgn;
    double* Ct = C + (size_t)tm * TM * ldC + (size_t)tn * TN;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm0 = (wave / WN) * WTM, wn0 = (wave % WN) * WTN;
    d4 acc[MIM][MIN];
#pragma unroll
    for (int i = 0; i < MIM; ++i)
#pragma unroll
        for (int j = 0; j < MIN; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e)
                acc[i][j][e] = PRELOAD ? Ct[(size_t)(wm0 + i * 16 + (lane >> 4) + 4 * e) * ldC + wn0 + j * 16 + (lane & 15)] : 0.0;
    const int nst = nkt;      // stages of 16 k
    double ra[PF][EA], rb[PF][EB];
#pragma unroll
    for (int h = 0; h < PF; ++h)
        if (h < nst) {
            ldst(A0, ldA, h * KT, ra[h], tgm);
            ldst(B0, ldB, h * KT, rb[h], tgn);
        }
    for (int s = 0; s < nst; s += PF) {
#pragma unroll
        for (int h = 0; h < PF; ++h) {
            if (s + h < nst) {
                const int buf = (PF & 1) ? ((s + h) & 1) : (h & 1);
                stst(As + buf * KT * LDA, ra[h], tgm);
                stst(Bs + buf * KT * LDB, rb[h], tgn);
                __syncthreads();
                if (s + h + PF < nst) {
                    ldst(A0, ldA, (s + h + PF) * KT, ra[h], tgm);
                    ldst(B0, ldB, (s + h + PF) * KT, rb[h], tgn);
                }
                const double* as = As + buf * KT * LDA;
                const double* bs = Bs + buf * KT * LDB;
#pragma unroll
                for (int kk = 0; kk < KT / 4; ++kk) {
                    const int kr = kk * 4 + (lane >> 4);
                    const int xc = (lane & 15) ^ (4 * kk);
                    double af[MIM], bf[MIN];
#pragma unroll
                    for (int i = 0; i < MIM; ++i) {
                        const double v = as[kr * LDA + wm0 + i * 16 + xc];
                        af[i] = PRELOAD ? -v : v;
                    }
#pragma unroll
                    for (int j = 0; j < MIN; ++j) bf[j] = bs[kr * LDB + wn0 + j * 16 + xc];
#pragma unroll
                    for (int i = 0; i < MIM; ++i)
#pragma unroll
                        for (int j = 0; j < MIN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[i], bf[j], acc[i][j], 0, 0, 0);
                }
            }
        }
    }
#pragma unroll
    for (int i = 0; i < MIM; ++i)
#pragma unroll
        for (int j = 0; j < MIN; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e)
                Ct[(size_t)(wm0 + i * 16 + (lane >> 4) + 4 * e) * ldC + wn0 + j * 16 + (lane & 15)] = acc[i][j][e];
}

__global__ void fill_kernel(double* p, size_t n, unsigned seed) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        p[i] = (double)((((unsigned)i + seed) * 2654435761u >> 8) & 0xffff) / 65536.0 - 0.5;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int TM, int TN, int WM, int WN, int PF, bool PRELOAD, int MINW, bool KMAJ = false>
double run(const char* name, const double* A, const double* B, double* C, int M, int N, int K, int extra_lds, int reps) {
    auto kern = tile_kernel<TM, TN, WM, WN, PF, PRELOAD, MINW, KMAJ>;
    const int ldA = KMAJ ? M : K, ldB = KMAJ ? N : K;
    const int lds = 2 * KT * (TM + 16 + TN + 16) * 8 + extra_lds;
    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    const int ntm = M / TM, ntn = N / TN;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(kern, dim3(ntm * ntn), dim3(64 * WM * WN), lds, 0, A, B, C, ldA, ldB, N, K / KT, ntn);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(kern, dim3(ntm * ntn), dim3(64 * WM * WN), lds, 0, A, B, C, ldA, ldB, N, K / KT, ntn);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    const double tf = 2.0 * M * N * (double)K / ms / 1e9;
    int nb = 0;
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kern, 64 * WM * WN, lds));
    printf("%-46s K=%5d  %8.3f ms  %6.1f TFLOP/s  (%d x %d tiles, %d workgroup(s) per CU, %d KB LDS)\n", name, K, ms, tf, ntm, ntn, nb,
           lds / 1024);
    return tf;
}

int main() {
    // correctness of every variant on a small problem first
    {
        const int M = 256, N = 256, K = 64;
        std::vector<double> hA(M * K), hB(N * K), hC(M * N), ref(M * N);
        for (int i = 0; i < M * K; ++i) hA[i] = sin(0.37 * i) * 0.5;
        for (int i = 0; i < N * K; ++i) hB[i] = cos(0.11 * i) * 0.5;
        for (int i = 0; i < M * N; ++i) hC[i] = 0.001 * (i % 97);
        for (int i = 0; i < M; ++i)
            for (int j = 0; j < N; ++j) {
                double s = 0;
                for (int k = 0; k < K; ++k) s += hA[i * K + k] * hB[j * K + k];
                ref[i * N + j] = hC[i * N + j] - s;
            }
        double *A, *B, *C;
        CK(hipMalloc(&A, M * K * 8)); CK(hipMalloc(&B, N * K * 8)); CK(hipMalloc(&C, M * N * 8));
        CK(hipMemcpy(A, hA.data(), M * K * 8, hipMemcpyHostToDevice));
        CK(hipMemcpy(B, hB.data(), N * K * 8, hipMemcpyHostToDevice));
        auto check = [&](const char* nm, auto kern, int tm, int tn, int threads, int lds) {
            CK(hipMemcpy(C, hC.data(), M * N * 8, hipMemcpyHostToDevice));
            CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
            hipLaunchKernelGGL(kern, dim3((M / tm) * (N / tn)), dim3(threads), lds, 0, A, B, C, K, K, N, K / KT, N / tn);
            std::vector<double> out(M * N);
            CK(hipMemcpy(out.data(), C, M * N * 8, hipMemcpyDeviceToHost));
            double err = 0;
            for (int i = 0; i < M * N; ++i) err = fmax(err, fabs(out[i] - ref[i]));
            printf("check %-28s max error %.2e\n", nm, err);
            if (!(err < 1e-12)) exit(2);
        };
        check("128x128 8 waves", tile_kernel<128, 128, 4, 2, 1, true, 4>, 128, 128, 512, 2 * KT * (144 + 144) * 8);
        check("256x128 8 waves", tile_kernel<256, 128, 4, 2, 1, true, 2>, 256, 128, 512, 2 * KT * (272 + 144) * 8);
        check("256x128 8 waves PF 2", tile_kernel<256, 128, 4, 2, 2, true, 2>, 256, 128, 512, 2 * KT * (272 + 144) * 8);
        CK(hipFree(A)); CK(hipFree(B)); CK(hipFree(C));
    }
    const int M = 8192, N = 8192;
    double *A, *B, *C;
    CK(hipMalloc(&A, (size_t)M * 2048 * 8)); CK(hipMalloc(&B, (size_t)N * 2048 * 8)); CK(hipMalloc(&C, (size_t)M * N * 8));
    // pseudo-random operands (zero-filled ones draw less power and read high); every variant sees the same data
    hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, A, (size_t)M * 2048, 1u);
    hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, B, (size_t)N * 2048, 7u);
    CK(hipMemset(C, 0, (size_t)M * N * 8));
    CK(hipDeviceSynchronize());
    // the read-modify-write update at other panel widths (what wider outer panels / deferred far columns could buy per flop)
    printf("---- C read-modify-write at other K\n");
    run<128, 128, 4, 2, 1, true, 4>("128x128, 8 waves, PF 1 (library shape)", A, B, C, M, N, 128, 0, 20);
    run<128, 128, 4, 2, 1, true, 4>("128x128, 8 waves, PF 1 (library shape)", A, B, C, M, N, 512, 0, 10);
    run<128, 128, 4, 2, 1, true, 4>("128x128, 8 waves, PF 1 (library shape)", A, B, C, M, N, 1024, 0, 6);
    run<256, 128, 4, 2, 1, true, 2>("256x128, 8 waves, PF 1", A, B, C, M, N, 512, 0, 10);
    run<256, 128, 4, 2, 1, true, 2>("256x128, 8 waves, PF 1", A, B, C, M, N, 1024, 0, 6);
    for (int pass = 0; pass < 2; ++pass) {
        const int K = pass == 0 ? 256 : 2048, reps = pass == 0 ? 20 : 4;
        printf("---- K = %d, %s\n", K, pass == 0 ? "C read-modify-write (the trailing update)" : "C written once (the long products)");
        if (pass == 0) {
            run<128, 128, 4, 2, 1, true, 4>("128x128, 8 waves, PF 1 (library shape)", A, B, C, M, N, K, 0, reps);
            run<128, 128, 4, 2, 1, true, 4>("128x128, 8 waves, PF 1, ONE workgroup per CU", A, B, C, M, N, K, 16 * 1024, reps);
            run<256, 128, 4, 2, 1, true, 2>("256x128, 8 waves, PF 1", A, B, C, M, N, K, 0, reps);
            run<256, 128, 4, 2, 2, true, 2>("256x128, 8 waves, PF 2", A, B, C, M, N, K, 0, reps);
        } else {
            run<128, 128, 4, 2, 2, false, 4>("128x128, 8 waves, PF 2 (library shape)", A, B, C, M, N, K, 0, reps);
            run<128, 128, 4, 2, 2, false, 4>("128x128, 8 waves, PF 2, ONE workgroup per CU", A, B, C, M, N, K, 16 * 1024, reps);
            run<256, 128, 4, 2, 1, false, 2>("256x128, 8 waves, PF 1", A, B, C, M, N, K, 0, reps);
            run<256, 128, 4, 2, 2, false, 2>("256x128, 8 waves, PF 2", A, B, C, M, N, K, 0, reps);
            run<128, 128, 4, 2, 2, false, 4, true>("128x128, 8 waves, PF 2, operands m-contiguous", A, B, C, M, N, K, 0, reps);
            run<256, 128, 4, 2, 2, false, 2, true>("256x128, 8 waves, PF 2, operands m-contiguous", A, B, C, M, N, K, 0, reps);
        }
    }
    return 0;
}
