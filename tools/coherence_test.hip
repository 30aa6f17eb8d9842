// Which in-kernel hand-off forms deliver fresh data to a consumer that has READ THE SAME LINES BEFORE (its CU's L1 and
// its XCD's L2 hold the old version)?  One writer workgroup per epoch rewrites a buffer, every workgroup of the grid
// (one per CU, all XCDs) then reads all of it and counts words that are not the epoch's.  Standalone measurement for
// the persistent factorisation launch (lcgp_hip.hip: dag_kernel); not part of the library.
//   hipcc -O3 --offload-arch=gfx950 -o tools/coherence_test tools/coherence_test.hip && tools/coherence_test
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

enum { ST_PLAIN_REL = 0, ST_SC1 = 1 };
enum { AQ_NONE = 0, AQ_AGENT = 1, AQ_SYSTEM = 2 };
enum { LD_PLAIN = 0, LD_AGENT = 1, LD_SYSTEM = 2 };

struct Ctl { int flag; int pad0[31]; int done; int pad1[31]; int fail; int pad2[31]; };

template <int ST, int AQ, int LD>
__global__ __launch_bounds__(256) void ring(unsigned long long* X, int words, Ctl* c, int epochs, unsigned long long* stale,
                                            unsigned limit) {
    const int tid = threadIdx.x, nwg = gridDim.x, me = blockIdx.x;
    unsigned long long bad = 0;
    for (int e = 1; e <= epochs; ++e) {
        const int writer = (int)(((long long)e * 37) % nwg);
        if (me == writer) {
            // everybody has read epoch e - 1
            if (tid == 0) {
                unsigned it = 0;
                while (__hip_atomic_load(&c->done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < nwg * (e - 1)) {
                    __builtin_amdgcn_s_sleep(2);
                    if (++it > limit) { __hip_atomic_store(&c->fail, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
                }
            }
            __syncthreads();
            for (int i = tid; i < words; i += 256) {
                if (ST == ST_SC1) __hip_atomic_store(X + i, (unsigned long long)e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                else X[i] = (unsigned long long)e;
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) {
                if (ST == ST_PLAIN_REL) {
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                __hip_atomic_store(&c->flag, e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        if (tid == 0) {
            unsigned it = 0;
            while (__hip_atomic_load(&c->flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < e) {
                __builtin_amdgcn_s_sleep(2);
                if (++it > limit || ((it & 1023) == 0 && __hip_atomic_load(&c->fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
                    __hip_atomic_store(&c->fail, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    break;
                }
            }
            if (AQ == AQ_AGENT) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            if (AQ == AQ_SYSTEM) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
        for (int i = tid; i < words; i += 256) {
            unsigned long long v;
            if (LD == LD_AGENT) v = __hip_atomic_load(X + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else if (LD == LD_SYSTEM) v = __hip_atomic_load(X + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            else v = X[i];
            bad += v != (unsigned long long)e;
        }
        __syncthreads();
        if (tid == 0) __hip_atomic_fetch_add(&c->done, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (bad) atomicAdd(stale, bad);
}

template <int ST, int AQ, int LD>
void run(const char* name, int words, int epochs) {
    unsigned long long *X, *stale;
    Ctl* c;
    CHECK(hipMalloc(&X, (size_t)words * 8));
    CHECK(hipMalloc(&stale, 8));
    CHECK(hipMalloc(&c, sizeof(Ctl)));
    CHECK(hipMemset(X, 0, (size_t)words * 8));
    CHECK(hipMemset(stale, 0, 8));
    CHECK(hipMemset(c, 0, sizeof(Ctl)));
    int ncu = 0;
    CHECK(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0));
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    CHECK(hipEventRecord(a));
    hipLaunchKernelGGL((ring<ST, AQ, LD>), dim3(ncu), dim3(256), 0, 0, X, words, c, epochs, stale, 4000000u);
    CHECK(hipEventRecord(b));
    CHECK(hipDeviceSynchronize());
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, a, b));
    unsigned long long h = 0;
    Ctl hc;
    CHECK(hipMemcpy(&h, stale, 8, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(&hc, c, sizeof(Ctl), hipMemcpyDeviceToHost));
    printf("%-44s %7d words x %4d epochs x %d readers: stale words %12llu (%.4f %%)  %s  %.2f us/epoch\n", name, words, epochs, ncu, h,
           100.0 * (double)h / ((double)words * epochs * ncu), hc.fail ? "TIMEOUT" : "", ms * 1e3 / epochs);
    CHECK(hipFree(X)); CHECK(hipFree(stale)); CHECK(hipFree(c));
}

int main() {
    for (int words : {512, 16384}) {
        const int ep = 400;
        run<ST_PLAIN_REL, AQ_NONE, LD_PLAIN>("plain+release | no acquire | plain loads", words, ep);
        run<ST_PLAIN_REL, AQ_AGENT, LD_PLAIN>("plain+release | agent acquire | plain loads", words, ep);
        run<ST_SC1, AQ_AGENT, LD_PLAIN>("sc1 stores | agent acquire | plain loads", words, ep);
        run<ST_SC1, AQ_SYSTEM, LD_PLAIN>("sc1 stores | system acquire | plain loads", words, ep);
        run<ST_SC1, AQ_NONE, LD_AGENT>("sc1 stores | no acquire | agent (sc1) loads", words, ep);
        run<ST_SC1, AQ_NONE, LD_SYSTEM>("sc1 stores | no acquire | system (sc0 sc1) loads", words, ep);
        run<ST_PLAIN_REL, AQ_NONE, LD_AGENT>("plain+release | no acquire | agent (sc1) loads", words, ep);
    }
    return 0;
}
