// Feasibility probe (round 6): can ONE long-running kernel of a few LDS-resident workgroups live on a second HIP stream beside
// the launch-by-launch evaluation of the main stream without slowing either?  (Rounds 2 and 4 measured that a CHAIN OF LAUNCHES
// on a second stream pays 60-85 us per kernel boundary beside a saturating kernel; a single resident kernel has no boundary.)
//
//   probe_resident : nwg workgroups of 256 threads, `lds` bytes of dynamic LDS each (> 80 KB keeps a 73.7 KB tile workgroup
//                    off their CUs), each running `iters` rounds of: 64 x 64 x 64 fp64 MFMA product from LDS operands (the
//                    chain's kind of work), and every `chase_every` rounds one dependent global round trip (sc1 load of a
//                    word another kernel may have written).  Stamps (100 MHz counter): start, end, and the summed ticks of
//                    the product rounds and of the round trips.
//   probe_pingpong : the resident kernel answers `n` flag hand-offs from one-thread kernels of the main stream
//                    (signal_kernel / wait_kernel), bounded polls everywhere.
//   hipcc -O3 --offload-arch=gfx950 -fPIC -shared -o tools/libpersist_probe.so tools/persist_probe.hip
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef double d4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned long long rt() { return __builtin_amdgcn_s_memrealtime(); }

__device__ __forceinline__ unsigned ld_sc1(const unsigned* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_sc1(unsigned* p, unsigned v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

extern "C" __global__ __launch_bounds__(256) void resident_kernel(int iters, int chase_every, const unsigned* chase,
                                                                   unsigned long long* stamps /*[nwg][6]*/, double* sink) {
    extern __shared__ __align__(16) unsigned char lds_raw[];
    double* A = (double*)lds_raw;            // [64][80] k-major
    double* B = A + 64 * 80;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int e = tid; e < 64 * 80; e += 256) { A[e] = 1e-3 * (e % 17); B[e] = 1e-3 * (e % 13); }
    __syncthreads();
    const int wm0 = (wave >> 1) * 32, wn0 = (wave & 1) * 32;
    d4 acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) acc[i][j] = d4{0, 0, 0, 0};
    unsigned long long t_prod = 0, t_chase = 0, nchase = 0;
    const unsigned long long t0 = rt();
    unsigned idx = 0;
    for (int it = 0; it < iters; ++it) {
        const unsigned long long a0 = rt();
#pragma unroll 4
        for (int kk = 0; kk < 16; ++kk) {
            const int krow = (kk * 4 + (lane >> 4)) * 80;
            double af[2], bf[2];
            for (int i = 0; i < 2; ++i) af[i] = A[krow + wm0 + i * 16 + (lane & 15)];
            for (int j = 0; j < 2; ++j) bf[j] = B[krow + wn0 + j * 16 + (lane & 15)];
            for (int i = 0; i < 2; ++i)
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
        const unsigned long long a1 = rt();
        t_prod += a1 - a0;
        if (chase_every > 0 && (it % chase_every) == chase_every - 1) {
            // one dependent global round trip (L2-bypassing load), as a chain step's operand fetch would see it
            unsigned v = 0;
            if (tid == 0) v = ld_sc1(chase + (idx & 1023) * 16);
            v = __shfl(v, 0);
            idx = idx * 1664525u + 1013904223u + v;
            __syncthreads();
            t_chase += rt() - a1;
            ++nchase;
        }
    }
    const unsigned long long t1 = rt();
    double s = 0;
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 4; ++e) s += acc[i][j][e];
    if (s == 123.456) sink[0] = s + idx;
    if (tid == 0) {
        unsigned long long* o = stamps + (size_t)blockIdx.x * 6;
        o[0] = t0; o[1] = t1; o[2] = t_prod; o[3] = t_chase; o[4] = nchase;
        o[5] = __builtin_amdgcn_s_getreg(20 << 0 | (0 << 6) | (31 << 11));      // HW_REG_XCC_ID
    }
}

// ---- flag ping-pong: main-stream one-thread kernels <-> the resident kernel ----
extern "C" __global__ void signal_kernel(unsigned* flag, unsigned v, unsigned long long* stamp) {
    if (threadIdx.x == 0) { stamp[0] = rt(); st_sc1(flag, v); }
}
extern "C" __global__ void wait_kernel(const unsigned* flag, unsigned v, unsigned* fail, unsigned long long* stamp) {
    if (threadIdx.x == 0) {
        unsigned n = 0;
        while (ld_sc1(flag) < v) {
            __builtin_amdgcn_s_sleep(2);
            if (++n > 4000000u) { st_sc1(fail, 1u); break; }
        }
        stamp[0] = rt();
    }
}
extern "C" __global__ __launch_bounds__(256) void pong_kernel(int n, const unsigned* ping, unsigned* pong, unsigned* fail,
                                                               unsigned long long* stamps /*[n][2]*/) {
    extern __shared__ __align__(16) unsigned char lds_raw[];
    __shared__ int stop;
    if (threadIdx.x == 0) stop = 0;
    __syncthreads();
    for (int i = 1; i <= n; ++i) {
        if (threadIdx.x == 0) {
            unsigned c = 0;
            while (ld_sc1(ping) < (unsigned)i) {
                __builtin_amdgcn_s_sleep(1);
                if (++c > 8000000u) { st_sc1(fail, 2u); stop = 1; break; }
            }
            stamps[2 * (i - 1)] = rt();
        }
        __syncthreads();
        if (stop) return;
        // (a chain panel would run here)
        if (threadIdx.x == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            st_sc1(pong, (unsigned)i);
            stamps[2 * (i - 1) + 1] = rt();
        }
        __syncthreads();
    }
}

extern "C" int probe_resident(void* stream, int nwg, int lds, int iters, int chase_every, const unsigned* chase,
                              unsigned long long* stamps, double* sink) {
    hipError_t e = hipFuncSetAttribute((const void*)resident_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return -1;
    hipLaunchKernelGGL(resident_kernel, dim3(nwg), dim3(256), lds, (hipStream_t)stream, iters, chase_every, chase, stamps, sink);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
extern "C" int probe_pong(void* stream, int lds, int n, const unsigned* ping, unsigned* pong, unsigned* fail,
                          unsigned long long* stamps) {
    hipError_t e = hipFuncSetAttribute((const void*)pong_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return -1;
    hipLaunchKernelGGL(pong_kernel, dim3(1), dim3(256), lds, (hipStream_t)stream, n, ping, pong, fail, stamps);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
extern "C" int probe_signal(void* stream, unsigned* flag, unsigned v, unsigned long long* stamp) {
    hipLaunchKernelGGL(signal_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, flag, v, stamp);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
extern "C" int probe_wait(void* stream, const unsigned* flag, unsigned v, unsigned* fail, unsigned long long* stamp) {
    hipLaunchKernelGGL(wait_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, flag, v, fail, stamp);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
