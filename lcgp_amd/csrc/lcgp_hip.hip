// lcgp_hip.hip -- gfx950 (MI355X) kernels + C ABI for the LCGP fit/predict hot path.
//
// What the reference does per L-BFGS-B evaluation (lcgp.py:635-666 / 554-630 + the gpflow tape):
// for each latent component k build C_k (covmat.py:31-55), decompose it, reduce to a scalar, and
// back-propagate.  Here, per component:  A = I + D (C o s s^T)  ->  A = L L^T (blocked Cholesky,
// 64x64 diagonal blocks in LDS, fp64 MFMA trailing updates)  ->  W = L^-1 (level-parallel TRMMs)
// ->  A^-1 = W^T W (one MFMA launch)  ->  z = A^-1 b  ->  fused contraction of
// G = s s^T o (D/2 A^-1 - z z^T/2) with dC/dtheta recomputed on the fly from x.
// All components of the rank are batched in every launch.  The library keeps no state: the launch schedule is a
// per-call argument (lcgp_sched), nothing is allocated, no stream or event is created.
//
// Layout in HBM: per component three npad x npad row-major matrices (npad = n rounded up to 128,
// the padding is the identity so every kernel works on whole 64x64 / 128x128 tiles):
//   M : A, then L (lower tiles)      W : L^-1 (lower tiles)      V : scratch, then A^-1 (lower tiles)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <math.h>
#include <type_traits>
#include <vector>

#include "../../include/lcgp_hip.h"
#include "fill_sched.h"

#define LCGP_VERSION 510

namespace {

constexpr int TS = 64;    // tile size (rows/cols of one tile)
constexpr int KT = 16;    // k extent of one LDS stage
constexpr int DMAX = 32;  // input dimensions staged in LDS at once (the fused kernels are instantiated for 2, 4, 6, 10, 16, 32)
constexpr int DWIDE = 126; // largest input dimension (beyond 32: chunks of 32 dimensions, grad_kernel_wide; the per-tile partial
                           // sums of the gradient contraction hold d + 2 <= 128 doubles)

typedef double d4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));

thread_local char g_err[256] = "";

inline int fail(const char* what, hipError_t e) {
    snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
    return -2;
}
inline int bad(const char* what) {
    snprintf(g_err, sizeof(g_err), "bad argument: %s", what);
    return -1;
}

inline int round_up(int a, int b) { return (a + b - 1) / b * b; }

// ---------------------------------------------------------------------------------------------------
// workspace carving (all offsets 256-byte aligned)
// ---------------------------------------------------------------------------------------------------
struct Ws {
    int n, npad, nb, d, p, q;
    int kern = 0;        // covariance kernel (lcgp_hip.h: LCGP_KERNEL_MATERN32 / LCGP_KERNEL_SE)
    size_t esz;
    size_t mat;          // elements per matrix
    char* base;
    size_t off_M, off_W, off_V, off_b, off_z, off_part, off_cpart, off_c, off_logdet, off_info, off_clock, total;
    int ntile_lower;
};

inline size_t align256(size_t x) { return (x + 255) & ~size_t(255); }

inline Ws carve(int dtype, int n, int d, int p, int q, void* base) {
    Ws w;
    w.n = n; w.d = d; w.p = p; w.q = q;
    w.npad = round_up(n, 2 * TS);   // whole 128x128 super-tiles (identity padding)
    w.nb = w.npad / TS;
    w.esz = dtype == LCGP_F64 ? 8 : 4;
    w.mat = (size_t)w.npad * w.npad;
    w.base = (char*)base;
    w.ntile_lower = w.nb * (w.nb + 1) / 2;
    size_t o = 0;
    w.off_M = o; o = align256(o + w.mat * q * w.esz);
    w.off_W = o; o = align256(o + w.mat * q * w.esz);
    w.off_V = o; o = align256(o + w.mat * q * w.esz);
    w.off_b = o; o = align256(o + (size_t)w.npad * q * w.esz);
    w.off_z = o; o = align256(o + (size_t)w.npad * q * w.esz);
    w.off_part = o; o = align256(o + (size_t)w.ntile_lower * q * 2 * TS * sizeof(double));   // symv partials (2 x 64 per
                                                                                          // tile), then gradient partials
    // float32 only: c = (C o s s^T) z in double (per-tile partials, then the vector), see grad_kernel
    w.off_cpart = o; o = align256(o + (dtype == LCGP_F64 ? 0 : (size_t)w.ntile_lower * q * 2 * TS * sizeof(double)));
    w.off_c = o; o = align256(o + (dtype == LCGP_F64 ? 0 : (size_t)w.npad * q * sizeof(double)));
    w.off_logdet = o; o = align256(o + (size_t)q * sizeof(double));
    w.off_info = o; o = align256(o + (size_t)q * sizeof(int));
    w.off_clock = o; o = align256(o + 4 * sizeof(unsigned long long));     // shader-clock / real-time stamps of the last A^-1 launch
    w.total = o;
    return w;
}

// ---------------------------------------------------------------------------------------------------
// MFMA wrappers.  A/B operand: lane l holds A[i = l & 15][k = l >> 4] / B[k = l >> 4][j = l & 15].
// C/D: col = l & 15;  row = (l >> 4) + 4 * reg for f64,  (l >> 4) * 4 + reg for f32.
// ---------------------------------------------------------------------------------------------------
template <typename T> struct Mfma;
template <> struct Mfma<double> {
    typedef d4 acc_t;
    static __device__ __forceinline__ acc_t run(double a, double b, acc_t c) {
        return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ int row(int lane, int reg) { return (lane >> 4) + 4 * reg; }
    static __device__ __forceinline__ int lane_row(int lane) { return lane >> 4; }      // row = lane_row + reg_row
    static __device__ __forceinline__ constexpr int reg_row(int reg) { return 4 * reg; }
};
template <> struct Mfma<float> {
    typedef f4 acc_t;
    static __device__ __forceinline__ acc_t run(float a, float b, acc_t c) {
        return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ int row(int lane, int reg) { return (lane >> 4) * 4 + reg; }
    static __device__ __forceinline__ int lane_row(int lane) { return (lane >> 4) * 4; }
    static __device__ __forceinline__ constexpr int reg_row(int reg) { return reg; }
};

// Buffer addressing of a tile: descriptor (base pointer, in scalar registers) + scalar byte offset + one per-lane byte
// offset register + immediate.  An accumulator tile addressed through 64-bit per-lane pointers costs two registers per
// distinct row of the MFMA lane map (16 of the 128 a wave of the fp64 128-tile kernels may use), held across the K loop.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t tile_rsrc(const void* base) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0x7fffffff, 0x00020000);
}
template <typename T> struct BufIo;
template <> struct BufIo<double> {
    typedef unsigned u2 __attribute__((ext_vector_type(2)));
    static __device__ __forceinline__ double load(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
        return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0));
    }
    static __device__ __forceinline__ void store(double v, __amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u2, v), r, voff, soff, 0);
    }
};
template <> struct BufIo<float> {
    static __device__ __forceinline__ float load(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
        return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
    }
    static __device__ __forceinline__ void store(float v, __amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, voff, soff, 0);
    }
};

// Thread index of a tile body.  Opaque to the optimiser on purpose: where bodies sit in a loop (the chain workgroup of
// host_kernel), every lane-dependent address computation of every body would otherwise be hoisted in front of that loop
// and kept alive across it (the diagonal-block code needs the whole register file itself).
__device__ __forceinline__ int body_tid() {
    int t = threadIdx.x;
    asm volatile("" : "+v"(t));
    __builtin_assume(t >= 0 && t < 1024);
    return t;
}

__device__ __forceinline__ void tri_decode(int t, int& r, int& c) {
    int rr = (int)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
    while ((rr + 1) * (rr + 2) / 2 <= t) ++rr;
    while (rr * (rr + 1) / 2 > t) --rr;
    r = rr;
    c = t - rr * (rr + 1) / 2;
}

// theta block accessors
__device__ __forceinline__ const double* th_row(const double* theta, int d, int p, int k) {
    return theta + (size_t)k * (d + 3 + p);
}

// exp(x) for x <= 0 (the kernel's -sum_j S_j): Cody-Waite reduction x = k ln2 + r, degree-13 Taylor polynomial on
// |r| <= ln2/2 (truncation 4e-18), scaling by v_ldexp_f64.  <= 1 ulp against libm over [-745, 0] (20 M samples on the
// host); about half the instructions of the library exp, whose overflow / NaN handling cannot occur here -- the
// kernel build and the gradient contraction are bound by fp64 VALU issue, not by HBM, with the library version.
// p * r + c with the constant c held in a SCALAR register pair: as a literal the compiler re-materialises every 64-bit
// constant with two v_mov per use (a quarter of the kernel-build instructions, which is bound by VALU issue)
__device__ __forceinline__ double fma_sc(double p, double r, double c) {
    double o;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(o) : "v"(p), "v"(r), "s"(c));
    return o;
}

__device__ __forceinline__ double exp_nonpos(double x) {
    const double kf = rint(x * 1.4426950408889634074);
    double r = fma(-kf, 6.93147180369123816490e-01, x);
    r = fma(-kf, 1.90821492927058770002e-10, r);
    double p = fma_sc(1.6059043836821613e-10, r, 2.08767569878681e-09);     // 1/13!, 1/12!
    p = fma_sc(p, r, 2.505210838544172e-08);
    p = fma_sc(p, r, 2.755731922398589e-07);
    p = fma_sc(p, r, 2.7557319223985893e-06);
    p = fma_sc(p, r, 2.48015873015873e-05);
    p = fma_sc(p, r, 1.984126984126984e-04);
    p = fma_sc(p, r, 1.388888888888889e-03);
    p = fma_sc(p, r, 8.333333333333333e-03);
    p = fma_sc(p, r, 4.1666666666666664e-02);
    p = fma_sc(p, r, 1.6666666666666666e-01);
    p = fma_sc(p, r, 0.5);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    return ldexp(p, (int)kf);
}
// float32 (no reference precision to match): v_exp_f32 on x log2(e); the absolute error stays below 1e-7
__device__ __forceinline__ float exp_nonpos(float x) { return __expf(x); }

// C0 = prod_j (1 + S_j) exp(-sum_j S_j): where the exponent is below the smallest normal number's logarithm the value is
// zero for every purpose of the path -- but the POLYNOMIAL there may have overflowed (lengthscales at their lower SoftClip
// bound 1e-6: S_j ~ 1e6, ten dimensions: 1e60, beyond float32's range), and inf x 0 is a NaN that takes the whole
// factorisation with it: this is what ended float32 fits of configs[3] (profiles/r06_fp32_breakdown.txt: status word 2 at
// lambda_max(A) ~ 5 .. 130, i.e. nothing to do with conditioning).  Above the threshold prod (1 + S_j) <= exp(sum S_j) is
// finite.  The matrices stay identical wherever they were finite before.
// (kernel build / cross covariance: the polynomial is capped instead -- one v_min per element in a kernel bound by VALU issue;
// prod (1 + S_j) <= exp(sum S_j), so a polynomial beyond the cap meets an exponential that has underflowed to zero long before)
template <typename T> __device__ __forceinline__ constexpr T poly_cap();
template <> __device__ __forceinline__ constexpr double poly_cap<double>() { return 1e300; }
template <> __device__ __forceinline__ constexpr float poly_cap<float>() { return 1e38f; }
template <typename T> __device__ __forceinline__ constexpr T exp_floor();
template <> __device__ __forceinline__ constexpr double exp_floor<double>() { return -708.0; }
template <> __device__ __forceinline__ constexpr float exp_floor<float>() { return -87.0f; }

// ---------------------------------------------------------------------------------------------------
// K1: kernel build.   A_ij = delta_ij + D sr_i sr_j s ((1 - nt) C0_ij + nt delta_ij)
//   C0 = prod_j (1 + S_j) exp(-sum_j S_j),  S_j = |x_i,j/ell_j - x_i',j/ell_j|      (covmat.py:35-53)
// One 64x64 lower tile per workgroup; x rows/cols staged in LDS already divided by ell.
// ---------------------------------------------------------------------------------------------------
// KERN: 0 = the reference's separable Matern-3/2 product (covmat.py:31-55), 1 = squared-exponential product kernel
//   C0 = exp(-1/2 sum_j S_j^2)   (no counterpart in the reference: BASELINE.json's north star names it; parity unpinned)
template <typename T, int DD /* >= d: the per-dimension loop is unrolled to DD */, int KERN>
__global__ __launch_bounds__(256) void build_kernel(T* __restrict__ M, size_t mat, int n, int npad, int d, int p,
                                                    const T* __restrict__ x, const T* __restrict__ sr,
                                                    const double* __restrict__ theta, int ntile,
                                                    const T* __restrict__ Y, T* __restrict__ bvec,
                                                    double* __restrict__ logdet, int* __restrict__ info) {
    // arithmetic in the storage type: the float32 variant is the HBM-bound regime (one float exp per element)
    __shared__ T xr[TS][DD + 1];
    __shared__ T xc[TS][DD + 1];
    __shared__ T srr[TS], src[TS];
    const int k = blockIdx.y;
    if ((int)blockIdx.x >= ntile) {
        // blocks past the tiles: b_k[i] = sum_a Y[a, i] psi_k[a]  (lcgp.py:646 + 657-658 collapsed; 608-610 for rep),
        // independent of the matrix, so it rides in this launch instead of a launch of its own
        const int i = (blockIdx.x - ntile) * 256 + threadIdx.x;
        if (i == 0 && logdet) { logdet[k] = 0.0; info[k] = 0; }      // the factorisation that follows starts from zero
        if (i >= npad) return;
        const double* psi = th_row(theta, d, p, k) + d + 3;
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
        if (i < n) {
            int a = 0;
            for (; a + 3 < p; a += 4) {
                s0 += (double)Y[(size_t)a * n + i] * psi[a];
                s1 += (double)Y[(size_t)(a + 1) * n + i] * psi[a + 1];
                s2 += (double)Y[(size_t)(a + 2) * n + i] * psi[a + 2];
                s3 += (double)Y[(size_t)(a + 3) * n + i] * psi[a + 3];
            }
            for (; a < p; ++a) s0 += (double)Y[(size_t)a * n + i] * psi[a];
        }
        bvec[(size_t)k * npad + i] = (T)((s0 + s1) + (s2 + s3));
        return;
    }
    int r, c;
    tri_decode(blockIdx.x, r, c);
    const double* th = th_row(theta, d, p, k);
    const double scale = th[d], nug = th[d + 1], D = th[d + 2];
    const double nt = nug / (1.0 + nug);
    const T c_off = (T)(D * scale * (1.0 - nt));        // multiplies C0
    const T c_diag = (T)(1.0 + D * scale * nt);         // extra term on the diagonal (times sr_i^2)
    const int tid = threadIdx.x;
    if (tid < TS) {
        int gi = r * TS + tid, gj = c * TS + tid;
        srr[tid] = (sr && gi < n) ? sr[gi] : (T)1;
        src[tid] = (sr && gj < n) ? sr[gj] : (T)1;
    }
    T* Mk = M + (size_t)k * mat;
    // 4 x 4 elements per thread (16 x 16 threads per tile): every LDS read of a scaled input row/column is used four
    // times, 16 independent exp chains per thread.  Columns per thread: in fp32 four consecutive ones (one 16-byte store
    // per row; the 16 lanes of a row write 256 contiguous bytes); in fp64 the pairs {2 tx, 2 tx + 1} and {32 + 2 tx, 33 + 2 tx},
    // so that EACH 16-byte store instruction of the 16 lanes covers 256 contiguous bytes (with four consecutive columns
    // per thread every instruction would write 16 of each 32 bytes: two partial passes over every line)
    const int tx = tid & 15, ty = tid >> 4;
    const int i0 = ty * 4;
    constexpr bool PAIRS = sizeof(T) == 8;
    auto colof = [&](int b) { return PAIRS ? (b >> 1) * 32 + 2 * tx + (b & 1) : 4 * tx + b; };
    T poly[4][4], ssum[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) { poly[a][b] = (T)1; ssum[a][b] = (T)0; }
    constexpr int UNR = DD <= 6 ? DD : 2;          // (fully unrolled the LDS reads of all dimensions are hoisted: registers)
    // one chunk of (at most) DD dimensions starting at d0: staged in LDS divided by ell, then accumulated
    auto chunk = [&](const int d0) {
        for (int e = tid; e < TS * DD; e += 256) {         // (columns beyond d are zero: they add |0 - 0| = 0)
            int i = e / DD, j = e - i * DD;
            int gi = r * TS + i, gj = c * TS + i;
            xr[i][j] = (gi < n && d0 + j < d) ? (T)((double)x[(size_t)gi * d + d0 + j] / th[d0 + j]) : (T)0;
            xc[i][j] = (gj < n && d0 + j < d) ? (T)((double)x[(size_t)gj * d + d0 + j] / th[d0 + j]) : (T)0;
        }
        __syncthreads();
#pragma unroll UNR
        for (int jj = 0; jj < DD; ++jj) {
            T xa[4], xb[4];
#pragma unroll
            for (int a = 0; a < 4; ++a) { xa[a] = xr[i0 + a][jj]; xb[a] = xc[colof(a)][jj]; }
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    if constexpr (KERN == 0) {
                        const T sd = fabs(xa[a] - xb[b]);
                        poly[a][b] = fma(poly[a][b], sd, poly[a][b]);
                        ssum[a][b] -= sd;
                    } else {
                        const T df = xa[a] - xb[b];
                        ssum[a][b] = fma((T)-0.5 * df, df, ssum[a][b]);
                    }
                }
        }
    };
    if constexpr (DD == DMAX) {
        // the widest instantiation also serves d > 32 (covmat.py:35-42 loops over any d): 32 dimensions at a time
        for (int d0 = 0; d0 < d; d0 += DD) {
            if (d0 > 0) __syncthreads();
            chunk(d0);
        }
    } else {
        chunk(0);
    }
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const int gi = r * TS + i0 + a;
        T v[4];
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int gj = c * TS + colof(b);
            if (gi < n && gj < n) {
                const T c0 = fmin(poly[a][b], poly_cap<T>()) * exp_nonpos(ssum[a][b]);
                const T ss = srr[i0 + a] * src[colof(b)];
                v[b] = ss * c_off * c0;
                if (gi == gj) v[b] += (T)1 + (c_diag - (T)1) * ss;
            } else {
                v[b] = gi == gj ? (T)1 : (T)0;
            }
        }
        T* dst = Mk + (size_t)gi * npad + c * TS;
        if constexpr (PAIRS) {
            typedef T pair_t __attribute__((ext_vector_type(2)));
            *(pair_t*)(dst + colof(0)) = pair_t{v[0], v[1]};
            *(pair_t*)(dst + colof(2)) = pair_t{v[2], v[3]};
        } else {
            typedef T quad_t __attribute__((ext_vector_type(4)));
            *(quad_t*)(dst + colof(0)) = quad_t{v[0], v[1], v[2], v[3]};
        }
    }
}

// rectangular Matern32 (covmat.py:31-55): out (n1 x n2) = scale ((1-nt) C0 + nt I[same]) o colscale^T.
// Parameters come either by value (host call, lcgp_matern32) or from a device theta row (predict).
struct ThetaArg { double v[DWIDE + 2]; };

template <typename T, int KERN>
__global__ __launch_bounds__(256) void cross_kernel(T* __restrict__ out, int ldo, int n1, int n2, int d,
                                                    const T* __restrict__ x1, const T* __restrict__ x2,
                                                    ThetaArg tv, const double* __restrict__ thp /*ell[d], scale, nug*/,
                                                    int same, const T* __restrict__ colscale, int n1pad, int n2pad,
                                                    int th_stride /*doubles between the theta rows of components*/,
                                                    size_t out_stride /*elements between the output slabs*/) {
    __shared__ double xr[TS][DMAX + 1];
    __shared__ double xc[TS][DMAX + 1];
    __shared__ double cs[TS];
    __shared__ double th[DWIDE + 2];
    const int r = blockIdx.y, c = blockIdx.x;
    const int tid = threadIdx.x;
    if (thp) thp += (size_t)blockIdx.z * th_stride;
    out += (size_t)blockIdx.z * out_stride;
    if (tid < d + 2) th[tid] = thp ? thp[tid] : tv.v[tid];
    if (tid < TS) {
        int gj = c * TS + tid;
        cs[tid] = (colscale && gj < n2) ? (double)colscale[gj] : 1.0;
    }
    __syncthreads();
    const double scale = th[d], nug = th[d + 1];
    const double nt = nug / (1.0 + nug);
    const int j = tid & 63;
    const int gj = c * TS + j;
    double poly[16], ssum[16];
#pragma unroll
    for (int m = 0; m < 16; ++m) { poly[m] = 1.0; ssum[m] = 0.0; }
    for (int d0 = 0; d0 < d; d0 += DMAX) {           // the dimensions in chunks of 32 (covmat.py:35-42 loops over any d)
        const int dc = d - d0 < DMAX ? d - d0 : DMAX;
        if (d0 > 0) __syncthreads();
        for (int e = tid; e < TS * dc; e += 256) {
            int i = e / dc, jj = e - i * dc;
            int gi = r * TS + i, gjj = c * TS + i;
            xr[i][jj] = gi < n1 ? (double)x1[(size_t)gi * d + d0 + jj] / th[d0 + jj] : 0.0;
            xc[i][jj] = gjj < n2 ? (double)x2[(size_t)gjj * d + d0 + jj] / th[d0 + jj] : 0.0;
        }
        __syncthreads();
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            const int i = (tid >> 6) * 16 + m;
            for (int jj = 0; jj < dc; ++jj) {
                if constexpr (KERN == 0) {
                    double sd = fabs(xr[i][jj] - xc[j][jj]);
                    poly[m] *= 1.0 + sd;
                    ssum[m] -= sd;
                } else {
                    const double df = xr[i][jj] - xc[j][jj];
                    ssum[m] = fma(-0.5 * df, df, ssum[m]);
                }
            }
        }
    }
#pragma unroll
    for (int m = 0; m < 16; ++m) {
        const int i = (tid >> 6) * 16 + m;
        const int gi = r * TS + i;
        if (gi >= n1pad || gj >= n2pad) continue;
        double v = 0.0;
        if (gi < n1 && gj < n2) {
            double c0 = fmin(poly[m], poly_cap<double>()) * exp_nonpos(ssum[m]);
            double dl = (same && gi + (same - 1) == gj) ? 1.0 : 0.0;   // same = 1 + row offset of x1 within x2
            v = scale * ((1.0 - nt) * c0 + nt * dl) * cs[j];
        }
        out[(size_t)gi * ldo + gj] = (T)v;
    }
}

// ---------------------------------------------------------------------------------------------------
// K2a: diagonal block.  Factorises the 64x64 block jb of M (L written back, upper part zeroed), writes its
// inverse into W (upper part zero), adds sum log L_ii to logdet[k], records info[k].  One workgroup per
// component.  The block lives in REGISTERS (thread (cj, rg) owns rows rg, rg+4, .. of column cj); per pivot
// the un-scaled pivot column goes through a double-buffered 64-entry LDS line, so the 64-step chain costs one
// barrier per step, and the update uses a_ij -= c_i (c_j / c_ss) so no sqrt sits on the chain.  The scaling
// by 1/sqrt(pivot), the log-determinant and the blocked (16) triangular inverse run after the chain.
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ double fast_rcp(double a) {
    // v_rcp_f64 (~26 bits) + two Newton steps: full double accuracy without the IEEE division sequence
    double x = __builtin_amdgcn_rcp(a);
    x = fma(fma(-a, x, 1.0), x, x);
    x = fma(fma(-a, x, 1.0), x, x);
    return x;
}

// ---- in-wave panel factorisation of the diagonal block ----
// The 64x64 block is split into four 16-column panels, one per wave.  Wave w keeps block column w as fp64 MFMA
// accumulators (lane (c = l & 15, g = l >> 4), reg e <-> row 16 rb + g + 4 e, column 16 w + c).  When its turn
// comes it re-reads the panel through a private LDS scratch into a layout made for the 16-pivot chain:
//   lane (i = l & 15, j = l >> 4):  rD[c] = row i of the panel's 16x16 DIAGONAL block (the same in all four lane rows),
//                                   rL[c] = row i of the j-th 16x16 block BELOW it (j < 3 - kb)
// so that everything a pivot step broadcasts lives in its own 16-lane row: the update
//   a_ic -= (u_i / d) u_c        u = the un-scaled pivot column (u = l sqrt(d)), u_c held by lane c of the row
// is ONE v_fmac_f64 with the DPP modifier row_newbcast:c on u (fp64 DPP exists for exactly this control) -- no
// v_readlane pair, no SGPR hazard nop, no LDS, no barrier.  The 1/sqrt(d) scaling runs after the chain.  The finished
// panel goes to LDS as lt[col][row]; after ONE barrier the waves to its right apply it with 16x16x4 MFMAs.
// Chain per panel: 16 x (broadcast, rcp + 2 Newton steps, 2 products) with the 2 (15 - s) updates filling the shadow.
template <int C>
__device__ __forceinline__ double row_bcast(double x) {        // lane C of every 16-lane row -> the whole row
    double o;
    // s_nop 1: a DPP read needs two wait states after a VALU write of its source (x has usually just been produced)
    asm("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(o) : "v"(x), "n"(C));
    return o;
}

template <int C>
__device__ __forceinline__ void fmac_row_bcast(double& acc, double u, double v) {   // acc += u[lane C of the row] * v
    asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(u), "v"(v), "n"(C));
}

template <int S, int C>
struct PivotCols {      // columns C .. 15 of pivot step S
    static __device__ __forceinline__ void run(double (&rD)[16], double (&rL)[16], double vD, double vL) {
        if constexpr (C < 16) {
            fmac_row_bcast<C>(rD[C], rD[S], vD);
            fmac_row_bcast<C>(rL[C], rD[S], vL);
            PivotCols<S, C + 1>::run(rD, rL, vD, vL);
        }
    }
};

template <int S>
struct PivotSteps {     // pivot steps S .. 15; dmine collects the pivot of row i (lane i of each row)
    static __device__ __forceinline__ void run(double (&rD)[16], double (&rL)[16], double& dmine, int i) {
        if constexpr (S < 16) {
            // rD[S] was last written by the previous step's first update: the broadcast carries the DPP wait states,
            // and every fmac of this step depends on it through vD / vL
            const double dp = row_bcast<S>(rD[S]);
            dmine = i == S ? rD[S] : dmine;
            const double rdp = fast_rcp(dp);
            const double vD = -rD[S] * rdp, vL = -rL[S] * rdp;
            PivotCols<S, S + 1>::run(rD, rL, vD, vL);
            PivotSteps<S + 1>::run(rD, rL, dmine, i);
        }
    }
};

template <int S>
struct ScaleCols {      // l = u / sqrt(pivot): column S times the reciprocal root held by lane S of the row
    static __device__ __forceinline__ void run(double (&rD)[16], double (&rL)[16], double rsl) {
        if constexpr (S < 16) {
            const double f = row_bcast<S>(rsl);
            rD[S] *= f;
            rL[S] *= f;
            ScaleCols<S + 1>::run(rD, rL, rsl);
        }
    }
};

__device__ __forceinline__ double fast_rsqrt(double a) {
    double y = __builtin_amdgcn_rsq(a);
    double e = fma(-(a * y), y, 1.0);
    y = fma(0.5 * y, e, y);
    e = fma(-(a * y), y, 1.0);
    y = fma(0.5 * y, e, y);
    return y;
}

constexpr int LEAF_LDT = TS + 1;
constexpr int LEAF_SCR = TS * 17;                                         // ONE [64][17] transposition scratch: only one wave factors at a time
constexpr int LEAF_LDS_BYTES = (2 * TS * LEAF_LDT + LEAF_SCR + 2 * TS) * 8 + 16;    // lt + w + scratch + dinv + pivs + bad

// Inverse of the lower-triangular 64x64 block, one 16-row block ROW at a time, by ONE wave (no workgroup barrier inside):
//   W_aa by substitution (lanes 0..15: one column each, solve L_aa w = e_c), then for b < a
//   T_ab = sum_{m=b}^{a-1} L_am W_mb ;  W_ab = -W_aa T_ab   on the fp64 MFMA.  The accumulator of T (lane: col = l & 15,
//   rows (l >> 4) + 4 reg) is exactly the B-operand fragment of the second product (k = 4 step + (l >> 4)).
// Needs L rows of block a and the rows < a of W: row a of the inverse can be formed as soon as panel a is factored.
__device__ __forceinline__ void leaf_inverse_diag(double (*lt)[LEAF_LDT], double (*w)[LEAF_LDT], const double* dinv, int a,
                                                  int lane) {
    if (lane < 16) {
        const int b0 = a * 16, cl = lane;
        double wc[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            double sacc = i == cl ? -1.0 : 0.0;
#pragma unroll
            for (int m = 0; m < i; ++m) sacc = fma(lt[b0 + m][b0 + i], wc[m], sacc);
            wc[i] = -sacc * dinv[b0 + i];
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) w[b0 + i][b0 + cl] = wc[i];
    }
}

__device__ __forceinline__ d4 leaf_inverse_t(double (*lt)[LEAF_LDT], double (*w)[LEAF_LDT], int a, int b, int lane) {
    const int li = lane & 15, lq = lane >> 4;
    d4 tacc = {0.0, 0.0, 0.0, 0.0};
    for (int mb = b; mb < a; ++mb) {
#pragma unroll
        for (int st = 0; st < 4; ++st)
            tacc = __builtin_amdgcn_mfma_f64_16x16x4f64(lt[mb * 16 + 4 * st + lq][a * 16 + li], w[mb * 16 + 4 * st + lq][b * 16 + li],
                                                        tacc, 0, 0, 0);
    }
    return tacc;
}

__device__ __forceinline__ void leaf_inverse_w(double (*w)[LEAF_LDT], const d4& tacc, int a, int b, int lane) {
    const int li = lane & 15, lq = lane >> 4;
    d4 wacc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int st = 0; st < 4; ++st)
        wacc = __builtin_amdgcn_mfma_f64_16x16x4f64(w[a * 16 + li][a * 16 + 4 * st + lq], tacc[st], wacc, 0, 0, 0);
#pragma unroll
    for (int e = 0; e < 4; ++e) w[a * 16 + lq + 4 * e][b * 16 + li] = -wacc[e];
}

// Factor AND inverse of the diagonal block.  While wave kb factors its 16-column panel (the 16-pivot chain, alone on its
// SIMD), wave kb-1 -- idle otherwise -- forms row kb-1 of the inverse from the panels that are already final; only row 3
// is left when the last panel is done, and that one is spread over all four waves.
// The results leave for memory as they become final, on waves that would otherwise wait at the panel's barrier: L panel
// kb-1 and row block kb-2 of the inverse during panel kb (and the zero quadrant beside W during panel 0), so that only
// the last panel, the last two row blocks and the log-determinant are left after the chain.
template <typename T>
__device__ __forceinline__ void leaf_store_l_panel(T* __restrict__ Mb, int npad, double (*lt)[LEAF_LDT], int pb, int lane) {
    const int col = pb * 16 + (lane & 15);
#pragma unroll
    for (int m = 0; m < 16; ++m) {
        const int i = (lane >> 4) + 4 * m;
        Mb[(size_t)i * npad + col] = (T)lt[col][i];
    }
}

template <typename T>
__device__ __forceinline__ void leaf_store_w_rows(T* __restrict__ Wb, int npad, double (*w)[LEAF_LDT], int a, int lane) {
#pragma unroll
    for (int m = 0; m < 16; ++m) {
        const int i = a * 16 + m;
        Wb[(size_t)i * npad + lane] = (T)(lane <= i ? w[i][lane] : 0.0);
    }
}

template <typename T, bool FROM_LDS>
__device__ __forceinline__ void leaf_factor_invert(T* __restrict__ Mb, T* __restrict__ Wb, int npad, double (*lt)[LEAF_LDT],
                                                   double (*w)[LEAF_LDT], double* scratch, double* dinv, double* pivs,
                                                   int* bad, int jb) {
    const int tid = body_tid(), lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lq = lane >> 4;
    d4 acc[4];
#pragma unroll
    for (int rb = 0; rb < 4; ++rb)
#pragma unroll
        for (int e = 0; e < 4; ++e)
            // FROM_LDS: the caller left the block in the w area (w itself is first written after the first barrier below)
            acc[rb][e] = rb < wv ? 0.0 : FROM_LDS ? w[rb * 16 + lq + 4 * e][wv * 16 + li]
                                                  : (double)Mb[(size_t)(rb * 16 + lq + 4 * e) * npad + wv * 16 + li];
    double (*S)[17] = (double (*)[17])scratch;
    int first_bad = 0;
#pragma unroll 1
    for (int kb = 0; kb < 4; ++kb) {
        if (wv == kb) {
#pragma unroll
            for (int rb = 0; rb < 4; ++rb)
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (rb >= kb) S[rb * 16 + lq + 4 * e][li] = acc[rb][e];
            const int nlow = 3 - kb;                  // 16-row blocks below the diagonal block of this panel
            const int base = kb * 16;
            double rD[16], rL[16];
            const int lrow = lq < nlow ? base + 16 * (1 + lq) + li : base + li;      // (unconditional read, then select)
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                rD[c] = S[base + li][c];
                const double below = S[lrow][c];
                rL[c] = lq < nlow ? below : 0.0;
            }
            double dmine = 1.0;                       // the pivot of row li of the diagonal block
            PivotSteps<0>::run(rD, rL, dmine, li);
            const double rsl = fast_rsqrt(dmine);     // off the chain
            ScaleCols<0>::run(rD, rL, rsl);
            if (lq == 0) { dinv[base + li] = rsl; pivs[base + li] = dmine; }
            const unsigned long long bm = __ballot(!(dmine > 0.0)) & 0xffffull;
            if (bm) first_bad = jb * TS + base + __ffsll((long long)bm);
            // lt[col][row], all 64 rows of the panel's 16 columns in ONE pass of 16 stores: the four lane rows are exactly the
            // nlow blocks below the diagonal block (lane rows 0 .. nlow-1), the diagonal block itself (lane row nlow: rD is
            // the same in every lane row; zero above the diagonal) and the kb zero blocks above it (the remaining lane rows)
            const int row = lq < nlow ? base + 16 * (1 + lq) + li : lq == nlow ? base + li : 16 * (lq - nlow - 1) + li;
#pragma unroll
            for (int m = 0; m < 16; ++m)
                lt[base + m][row] = lq < nlow ? rL[m] : (lq == nlow && li >= m) ? rD[m] : 0.0;
            if (lane == 0) bad[kb] = first_bad;
        } else if (wv == kb - 1) {
            // row kb-1 of the inverse (panel kb-1 and everything it needs became visible at the last barrier)
            const int a = kb - 1;
            leaf_inverse_diag(lt, w, dinv, a, lane);
            for (int b = 0; b < a; ++b) {
                const d4 tacc = leaf_inverse_t(lt, w, a, b, lane);
                leaf_inverse_w(w, tacc, a, b, lane);
            }
        } else if (kb == 0) {
            // the 128x128 tile kernels read whole diagonal 128-blocks of W: keep the quadrant above this block zero
            if ((jb & 1) == 0)
                for (int i = wv - 1; i < TS; i += 3) Wb[(size_t)i * npad + TS + lane] = (T)0;
        } else if (wv == ((kb + 1) & 3)) {
            leaf_store_l_panel<T>(Mb, npad, lt, kb - 1, lane);
        } else if (kb >= 2 && wv == ((kb + 2) & 3)) {
            leaf_store_w_rows<T>(Wb, npad, w, kb - 2, lane);
        }
        __syncthreads();
        if (wv > kb) {
#pragma unroll
            for (int st = 0; st < 4; ++st) {
                const double bv = lt[kb * 16 + 4 * st + lq][wv * 16 + li];
#pragma unroll
                for (int rb = 1; rb < 4; ++rb)
                    if (rb >= wv)
                        acc[rb] = __builtin_amdgcn_mfma_f64_16x16x4f64(-lt[kb * 16 + 4 * st + lq][rb * 16 + li], bv, acc[rb],
                                                                       0, 0, 0);
            }
        }
    }
    // row 3 of the inverse: W_33 on wave 3, T_3b = sum_m L_3m W_mb on wave b (b = 0, 1, 2), then W_3b = -W_33 T_3b
    d4 tacc = {0.0, 0.0, 0.0, 0.0};
    if (wv == 3) leaf_inverse_diag(lt, w, dinv, 3, lane);
    else tacc = leaf_inverse_t(lt, w, 3, wv, lane);
    __syncthreads();
    if (wv < 3) leaf_inverse_w(w, tacc, 3, wv, lane);
    else leaf_store_l_panel<T>(Mb, npad, lt, 3, lane);
    __syncthreads();
    if (wv == 1) leaf_store_w_rows<T>(Wb, npad, w, 2, lane);
    else if (wv == 2) leaf_store_w_rows<T>(Wb, npad, w, 3, lane);
}

template <typename T, bool FROM_LDS = false>
__device__ __forceinline__ void leaf_body(unsigned char* lds, int k, T* __restrict__ M, T* __restrict__ W, size_t mat,
                                          int npad, int jb, double* __restrict__ logdet, int* __restrict__ info) {
    double (*lt)[LEAF_LDT] = (double (*)[LEAF_LDT])lds;                   // lt[col][row] = L[row][col]
    double (*w)[LEAF_LDT] = (double (*)[LEAF_LDT])((double*)lds + TS * LEAF_LDT);     // w[row][col] = (L^-1)[row][col]
    double* scratch = (double*)lds + 2 * TS * LEAF_LDT;
    double* dinv = scratch + LEAF_SCR;
    double* pivs = dinv + TS;
    int* bad = (int*)(pivs + TS);
    const int tid = body_tid();
    T* Mb = M + (size_t)k * mat + (size_t)jb * TS * npad + (size_t)jb * TS;
    T* Wb = W + (size_t)k * mat + (size_t)jb * TS * npad + (size_t)jb * TS;
    // the running log-determinant and status of the component are fetched now, so that the end is a store, not a round trip
    double ld_prev = 0.0;
    int info_prev = 0;
    if (tid == 0) { ld_prev = logdet[k]; info_prev = info[k]; }
    leaf_factor_invert<T, FROM_LDS>(Mb, Wb, npad, lt, w, scratch, dinv, pivs, bad, jb);
    if (tid < TS) {   // wave 0 (the other waves are storing the last rows): 1/2 sum log(pivot)
        double lg = 0.5 * log(pivs[tid]);
        for (int off = 32; off > 0; off >>= 1) lg += __shfl_xor(lg, off);
        if (tid == 0) {
            logdet[k] = ld_prev + lg;
            const int fb = bad[0] ? bad[0] : bad[1] ? bad[1] : bad[2] ? bad[2] : bad[3];
            if (fb && info_prev == 0) info[k] = fb;
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256, 2) void leaf_kernel(T* __restrict__ M, T* __restrict__ W, size_t mat, int npad, int jb,
                                                   double* __restrict__ logdet, int* __restrict__ info) {
    __shared__ __align__(16) unsigned char lds[LEAF_LDS_BYTES];
    leaf_body<T>(lds, blockIdx.x, M, W, mat, npad, jb, logdet, info);
}

// ---------------------------------------------------------------------------------------------------
// Tile GEMM on MFMA:  C_tile (op)= alpha * sum_kt  Aop(kt) * Bop(kt)^T,   64x64 output per workgroup,
// 4 waves each owning a 32x32 quadrant (2x2 MFMA 16x16x4 accumulators).  Operand tiles are 64 x 64
// sub-blocks of M / W / V, read in either orientation:
//   MK : element (m, k) at P[m * ld + k]      KM : element (m, k) at P[k * ld + m]
// and staged in LDS as [k][m] (KT = 16 k rows per stage, double buffered through registers).
// ---------------------------------------------------------------------------------------------------
#ifndef LCGP_PAIR_TILES_DEFAULT
#define LCGP_PAIR_TILES_DEFAULT 4000
#endif
#ifndef LCGP_EXP
#define LCGP_EXP 0      // destructive timing experiments (tools/build_variant.sh ... -DLCGP_EXP=n); 0 in every product build
#endif
enum GemmOp { OP_SYRK = 1, OP_TRTRI_T = 2, OP_TRTRI_W = 3, OP_LAUUM = 4, OP_PRED_U = 5 };
enum Lay { MK = 0, KM = 1 };

struct GemmArgs {
    const void* A; const void* B; void* C;     // component-0 bases
    size_t sA, sB, sC;                          // per-component strides (elements)
    int ldA, ldB, ldC;
    int nb;                                     // number of 64-blocks
    int p0, p1, p2, p3;                         // op specific
    int q;                                      // components in this launch
    int t0;                                     // first tile of this launch (OP_SYRK)
    int skipq;                                  // OP_SYRK on 128-tiles: tile 0 leaves its top-left 64x64 quadrant alone
                                                // (the diagonal block there is factored by the same launch, wide_leaf_kernel)
    const void* bvec = nullptr;                 // OP_LAUUM on 128-tiles: b (npad per component) and the partial buffer of
    double* part = nullptr;                     // z = A^-1 b, [component][tile][2][128]; null = no fused product
    unsigned long long* clk = nullptr;          // OP_LAUUM: the first block (the longest K loop of the launch) leaves its duration
                                                // in shader-clock cycles and in 10 ns ticks here: the clock the chip held (lcgp_lauum_clock)
};

// one K-stage (KT = 16 k values) of a TM-row operand tile: global -> registers -> LDS [k][m], ld = TM + 16;
// NT threads move TM * KT elements, EPT = TM * KT / NT consecutive ones each
// LDS image of a stage: [k][m] with row length LD = TM + 16 (= 16 mod 32 doubles: the two k rows a 32-lane ds_read_b64
// group touches lie in disjoint bank halves) and the column index XOR-ed with (k & 12).  The XOR is what makes the
// TRANSPOSING store of a k-contiguous operand (MK: a lane holds 4 consecutive k of one row m) conflict-free: a
// ds_write_b64 is served in groups of 16 contiguous lanes over 32 banks (MI355X_MICROARCH.md, LDS), and those 16 lanes
// are 4 rows m x 4 k-quads -- without the XOR all four quads of a row hit one bank pair (4-way conflict on every
// store of every stage; same-box A/B: 11.20 -> 10.95 ms per evaluation).  Within one k row the XOR only permutes each
// aligned group of 16 columns, so the fragment reads (16 consecutive columns of one row) stay conflict-free, and for
// the m-contiguous operand layout (KM: 16-byte pieces per lane) it moves whole aligned pieces, so the 16-byte stores stay.
// In float the banks are per 4 bytes and a ds_write_b32 group is 32 lanes = 8 rows x 4 k-quads: the XOR constants are
// 0, 8, 16, 24 there (they exchange whole 16-column groups inside a wave's 32-aligned sub-tile: the read side applies
// the XOR to the column index relative to the sub-tile origin, which is a multiple of 32).
template <typename T>
__device__ __forceinline__ constexpr int lds_swz(int k) { return (k & 12) * (int)(8 / sizeof(T)); }   // float: 0, 8, 16, 24
static_assert(KT == 16, "lds_swz assumes 16 k rows per stage");
// Swizzled column of a fragment read: 16-group `g16` (a multiple of 16, compile-time after unrolling) + lane column l15 of
// k step kk.  The XOR splits into a part that exchanges whole 16-groups (compile-time: folds into the instruction's
// offset) and a part inside the group (lane-dependent: one register per distinct constant).
template <typename T>
__device__ __forceinline__ int swz_col(int g16, int l15, int kk) {
    return (g16 ^ (lds_swz<T>(4 * kk) & ~15)) + (l15 ^ (lds_swz<T>(4 * kk) & 15));
}

template <typename T, int L, int TM, int NT>
__device__ __forceinline__ void load_stage(const T* __restrict__ P, int ld, int ks, T (&reg)[TM * KT / NT], int tid) {
    constexpr int EPT = TM * KT / NT;
    if (L == MK) {
        constexpr int TPR = KT / EPT;                       // threads per operand row (a row holds 16 k values)
        const int m = tid / TPR, kk = (tid % TPR) * EPT;
        const T* src = P + (size_t)m * ld + ks + kk;
#pragma unroll
        for (int e = 0; e < EPT; ++e) reg[e] = src[e];
    } else {
        // a lane holds EPT / PE pieces of 16 bytes; piece p of the lanes of one k row is one contiguous run (so that the
        // 8-lane groups of the ds_write_b128 that stores it cover 128 contiguous bytes = all 32 banks once; with four
        // consecutive doubles per lane, lanes l and l + 4 of a group shared their banks: 2-way conflict on every store)
        constexpr int TPK = TM / EPT;                       // threads per k row
        constexpr int PE = 16 / (int)sizeof(T), NP = EPT / PE;
        const int kq = tid / TPK, j = tid % TPK;
        const T* src = P + (size_t)(ks + kq) * ld + j * PE;
#pragma unroll
        for (int pc = 0; pc < NP; ++pc)
#pragma unroll
            for (int e = 0; e < PE; ++e) reg[pc * PE + e] = src[pc * (TM / NP) + e];
    }
}

template <typename T, int L, int TM, int NT>
__device__ __forceinline__ void store_stage(T* __restrict__ S, const T (&reg)[TM * KT / NT], int tid) {
    constexpr int EPT = TM * KT / NT;
    constexpr int LD = TM + 16;
    if (L == MK) {
        constexpr int TPR = KT / EPT;
        const int m = tid / TPR, kk = (tid % TPR) * EPT;
#pragma unroll
        for (int e = 0; e < EPT; ++e) S[(kk + e) * LD + (m ^ lds_swz<T>(kk + e))] = reg[e];
    } else {
        constexpr int TPK = TM / EPT;
        constexpr int PE = 16 / (int)sizeof(T), NP = EPT / PE;
        const int kq = tid / TPK, j = tid % TPK;
#pragma unroll
        for (int pc = 0; pc < NP; ++pc)
#pragma unroll
            for (int e = 0; e < PE; ++e) S[kq * LD + ((j * PE + pc * (TM / NP)) ^ lds_swz<T>(kq)) + e] = reg[pc * PE + e];
    }
}

// ---- hand-counted operand prefetch (the triangular products of the inverse) ----
// hipcc's wait-count pass puts `s_waitcnt vmcnt(3..0)` in front of the first LDS store of every double stage of the register
// prefetch loop below, although the set being stored is the OLDER of two outstanding ones (it needs vmcnt(4..7)): the stage
// fetched one compute stage ago is waited for as well and the second stage of prefetch covers no latency (the pass is
// conservative at the loop header whether or not the loads sit under a condition; the same loop with unconditional loads
// compiles to the same waits).  So the loads of that loop are issued through inline asm, which the pass does not track, and
// the waits are written out: vector-memory operations retire in order, so with PF sets of NL loads outstanding the oldest
// set is complete at vmcnt((PF - 1) NL).  The asm that waits takes the set's registers as read-write operands, so every
// use of them is ordered behind it.  Compiler-generated waits stay correct beside this (they can only wait for more).
template <typename T> struct Piece;
template <> struct Piece<double> { typedef double v __attribute__((ext_vector_type(2))); };
template <> struct Piece<float> { typedef float v __attribute__((ext_vector_type(4))); };

// one 16-byte piece: address = wave-uniform base (SGPR pair) + per-lane byte offset (one VGPR, loop-invariant) + immediate.
// The scalar base keeps the whole address arithmetic of a stage on the scalar unit: the fp64 128-tile kernels sit at the
// 128-register limit of four waves per SIMD, and 64-bit per-lane pointers spilled there (a spill reload inside the K loop
// is a scratch load, i.e. a vmcnt(0) wait for every prefetched stage).
template <int IMM, typename P>
__device__ __forceinline__ void gload_piece(P& dst, unsigned voff, const void* sbase) {
    asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(dst) : "v"(voff), "s"(sbase), "n"(IMM) : "memory");
}

template <int N, typename P>
__device__ __forceinline__ void vm_wait_set(P (&a)[1], P (&b)[1]) {
    asm volatile("s_waitcnt vmcnt(%2)" : "+v"(a[0]), "+v"(b[0]) : "n"(N) : "memory");
}
template <int N, typename P>
__device__ __forceinline__ void vm_wait_set(P (&a)[4], P (&b)[4]) {
    asm volatile("s_waitcnt vmcnt(%8)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3])
                 : "n"(N) : "memory");
}
template <int N, typename P>
__device__ __forceinline__ void vm_wait_set(P (&a)[2], P (&b)[2]) {
    asm volatile("s_waitcnt vmcnt(%4)" : "+v"(a[0]), "+v"(a[1]), "+v"(b[0]), "+v"(b[1]) : "n"(N) : "memory");
}

// the same stage image as load_stage / store_stage, the registers as 16-byte pieces (NPC = EPT sizeof(T) / 16 per lane)
// byte offset of this lane's first piece within a stage of an operand tile
template <typename T, int L, int TM, int NT>
__device__ __forceinline__ unsigned stage_lane_offset(int ld, int tid) {
    constexpr int EPT = TM * KT / NT;
    constexpr int PE = 16 / (int)sizeof(T);
    if (L == MK) {
        constexpr int TPR = KT / EPT;
        const int m = tid / TPR, kk = (tid % TPR) * EPT;
        return (unsigned)((m * ld + kk) * (int)sizeof(T));
    } else {
        constexpr int TPK = TM / EPT;
        const int kq = tid / TPK, j = tid % TPK;
        return (unsigned)((kq * ld + j * PE) * (int)sizeof(T));
    }
}

// load_stage with the address split into a wave-uniform base (scalar registers) and this lane's offset within a stage
// (stage_lane_offset, bytes; loop-invariant): the compiler then uses the scalar-base form of global_load and no per-lane
// 64-bit pointer lives across the K loop
template <typename T, int L, int TM, int NT>
__device__ __forceinline__ void load_stage_u(const T* __restrict__ P /*wave-uniform*/, int ld, int ks, T (&reg)[TM * KT / NT],
                                             unsigned voff) {
    constexpr int EPT = TM * KT / NT;
    const T* src = (const T*)((const char*)(L == MK ? P + ks : P + (size_t)ks * ld) + voff);
    if (L == MK) {
#pragma unroll
        for (int e = 0; e < EPT; ++e) reg[e] = src[e];
    } else {
        constexpr int PE = 16 / (int)sizeof(T), NP = EPT / PE;
#pragma unroll
        for (int pc = 0; pc < NP; ++pc)
#pragma unroll
            for (int e = 0; e < PE; ++e) reg[pc * PE + e] = src[pc * (TM / NP) + e];
    }
}

template <typename T, int L, int TM, int NT, int PC = 0>
__device__ __forceinline__ void load_stage_p(const T* __restrict__ P /*wave-uniform*/, int ld, int ks,
                                             typename Piece<T>::v (&reg)[TM * KT / NT * (int)sizeof(T) / 16], unsigned voff) {
    constexpr int EPT = TM * KT / NT;
    constexpr int PE = 16 / (int)sizeof(T), NP = EPT / PE;
    const T* base = L == MK ? P + ks : P + (size_t)ks * ld;
    constexpr int STEP = (L == MK ? PE : TM / NP) * (int)sizeof(T);      // bytes between the pieces of a lane
    if constexpr (PC < NP) {
        gload_piece<PC * STEP>(reg[PC], voff, base);
        load_stage_p<T, L, TM, NT, PC + 1>(P, ld, ks, reg, voff);
    }
}

template <typename T, int L, int TM, int NT>
__device__ __forceinline__ void store_stage_p(T* __restrict__ S, const typename Piece<T>::v (&reg)[TM * KT / NT * (int)sizeof(T) / 16],
                                              int tid) {
    constexpr int EPT = TM * KT / NT;
    constexpr int LD = TM + 16;
    constexpr int PE = 16 / (int)sizeof(T), NP = EPT / PE;
    if (L == MK) {
        constexpr int TPR = KT / EPT;
        const int m = tid / TPR, kk = (tid % TPR) * EPT;
#pragma unroll
        for (int e = 0; e < EPT; ++e) S[(kk + e) * LD + (m ^ lds_swz<T>(kk + e))] = reg[e / PE][e % PE];
    } else {
        constexpr int TPK = TM / EPT;
        const int kq = tid / TPK, j = tid % TPK;
#pragma unroll
        for (int pc = 0; pc < NP; ++pc)
            *(typename Piece<T>::v*)&S[kq * LD + ((j * PE + pc * (TM / NP)) ^ lds_swz<T>(kq))] = reg[pc];
    }
}

// ---- row image of a k-contiguous fp64 operand (gemm_body only) ----
// A k-contiguous operand (MK: a lane holds 4 consecutive k of one row m) went into the [k][m] image through a TRANSPOSING
// store: four ds_write_b64 per lane, operand and stage, behind the XOR that makes them conflict-free.  Kept as it lies in
// memory instead -- S[m * LDK + k], LDK = KT + 2 doubles -- a lane stores its 32 bytes as two 16-byte pieces (half the LDS
// store instructions, whose issue + wait + barrier in front of every stage cost the rank-256 update 18 % of its time:
// profiles/r06_syrk_destructive.txt) and the TRANSPOSITION moves to the fragment read, where it is free: lane (i, kq) of an
// MFMA operand reads S[(m0 + i) LDK + k0 + kq], and with 36 dwords per row the 16 rows of a 32-lane ds_read_b64 group start
// on 16 different multiples of 4 banks (36 i mod 64), two banks each, the second k of the group two banks further: all 64
// banks once.  One address register per operand (the swizzled [k][m] image needs one per k step).
constexpr int LDK = KT + 2;
template <int TM, int NT>
__device__ __forceinline__ void store_stage_rows(double* __restrict__ S, const double (&reg)[TM * KT / NT], int tid) {
    constexpr int EPT = TM * KT / NT, TPR = KT / EPT;
    typedef double d2 __attribute__((ext_vector_type(2)));
    const int m = tid / TPR, kk = (tid % TPR) * EPT;
#pragma unroll
    for (int pc = 0; pc < EPT / 2; ++pc) *(d2*)&S[m * LDK + kk + 2 * pc] = d2{reg[2 * pc], reg[2 * pc + 1]};
}
template <int TM, int NT>
__device__ __forceinline__ void store_stage_rows_p(double* __restrict__ S, const Piece<double>::v (&reg)[TM * KT / NT / 2], int tid) {
    constexpr int EPT = TM * KT / NT, TPR = KT / EPT;
    const int m = tid / TPR, kk = (tid % TPR) * EPT;
#pragma unroll
    for (int pc = 0; pc < EPT / 2; ++pc) *(Piece<double>::v*)&S[m * LDK + kk + 2 * pc] = reg[pc];
}

// Workgroups are dealt round-robin over the 8 XCDs (each with its own L2).  Remap the linear tile id so that
// every XCD works on one contiguous run of tiles (neighbouring tiles share operand panels): bijective for any
// grid size (speed only, never correctness).
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, x = bid & 7, i = bid >> 3;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
}

// TM x TM output tile per workgroup of NW waves arranged (NW/2) x 2:
//   TM = 64,  NW = 4: 32x32 per wave (2x2 MFMA accumulators), 4 workgroups per CU
//   TM = 128, NW = 8: 32x64 per wave (2x4 accumulators), 2 workgroups per CU = 4 waves per SIMD, half the
//                     operand traffic per flop of the 64-tile
// All tile coordinates (g.nb, g.p0..p3) are in units of TM.
template <typename T, int OP, int TM, int NW>
__device__ __forceinline__ void gemm_body(const GemmArgs& g, const int lin /*block index within this descriptor*/,
                                          unsigned char* lds) {
    constexpr int LA = (OP == OP_LAUUM) ? KM : MK;
    constexpr int LB = (OP == OP_TRTRI_T || OP == OP_TRTRI_W || OP == OP_LAUUM) ? KM : MK;
    constexpr int NT = NW * 64;
    constexpr int LD = TM + 16;     // = 16 (mod 32): the two k rows a 32-lane group reads hit disjoint banks
    constexpr int WTM = TM / (NW / 2), WTN = TM / 2;   // per-wave sub-tile
    constexpr int MIM = WTM / 16, MIN = WTN / 16;      // MFMA tiles per wave
    constexpr int EPT = TM * KT / NT;
    T* As = (T*)lds;                  // [2][KT * LD]
    T* Bs = As + 2 * KT * LD;         // [2][KT * LD]

    // (tile, component) with the component as the FAST index: tiles are enumerated heaviest first, so the heaviest
    // tiles of every component start at once instead of component by component
    int lin_ = lin;
    if constexpr (OP == OP_TRTRI_T || OP == OP_TRTRI_W || OP == OP_LAUUM) {
        // With few components a launch of the triangular products is one or two rounds of resident workgroups, so its
        // time is the k length a CU collects from the tiles it hosts together.  Workgroups go to the CUs round-robin
        // (workgroup b and b + 256 share a CU): every other group of 256 is enumerated backwards, which pairs the
        // heaviest tiles with the lightest ones (first round at n = 4096, one component: 80 k tiles on the fullest CU
        // in plain heaviest-first order, 66 on every CU this way; same-box A/B: 2.80 -> 2.73 ms at one component, 3.93 ->
        // 3.91 at two, nothing at four).  With many components the launches have many rounds, and with 8 the component
        // index doubles as the XCD index (L2 locality), which the reversal would break.
        if (g.q < 4 && !g.skipq && ((lin >> 8) & 1)) {      // (skipq: the body runs as a filler job, gridDim is not its own)
            const int base = lin & ~255, nblk = (int)gridDim.x;
            const int top = base + 255 < nblk ? base + 255 : nblk - 1;      // last index of this (possibly short) group
            lin_ = top - (lin - base);
        }
    }
    const int k = lin_ % g.q;
    const int bid = lin_ / g.q;
    const T* Ab = (const T*)g.A + (size_t)k * g.sA;
    const T* Bb = (const T*)g.B + (size_t)k * g.sB;
    T* Cb = (T*)g.C + (size_t)k * g.sC;

    // ---- per-op tile decode: A0/B0 = first operand tiles, dA/dB = pointer step per kt, nkt, C tile ----
    const T* A0; const T* B0; T* Ct;
    ptrdiff_t dA, dB;           // signed: some ops walk their k tiles downwards (see below)
    int nkt;
    double alpha = 1.0;
    bool accumulate = false;
    bool tri_b = false;         // OP_LAUUM: the B operand of the last k tile is triangular too (diagonal tile)
    if constexpr (OP == OP_SYRK) {
        // M[r, c] -= sum_{kt in [p0, p1)} M[r, kt] M[c, kt]^T over the tiles c in [p2, p3), r in [c, nb)
        // (p3 == nb: the whole trailing triangle; p3 < nb: the rest of the current panel), column-major
#ifndef LCGP_SYRK_BAND
#define LCGP_SYRK_BAND 8
#endif
        int t = bid + g.t0, c = g.p2, r;
        if constexpr (LCGP_SYRK_BAND == 0) {
            while (t >= g.nb - c) { t -= g.nb - c; ++c; }
            r = c + t;
        } else {
            // Band-major: the rows p2 + i in bands of SB; inside a band column by column.  The workgroups resident on an XCD at
            // one time (with eight components the component IS the XCD) then cover SB row tiles x a dozen column tiles instead
            // of ~100 row tiles of one column: their operand panels (SB + a dozen of them) stay in the XCD's 4 MB L2, where the
            // column-major order re-fetched a row panel per tile (profiles/r06_hbm_traffic_per_kernel.txt: 7.2 GB per
            // evaluation through the fabric for 1.07 GB of matrix).  Same tiles, same arithmetic per tile.
            constexpr int SB = LCGP_SYRK_BAND;
            int i0 = 0;                                   // first row of the band, relative to p2
            for (;;) {
                // tiles of the band [i0, i0 + SB): row p2 + i holds the columns p2 .. min(p2 + i, p3 - 1)
                const int rows = g.nb - g.p2 - i0 < SB ? g.nb - g.p2 - i0 : SB;
                int cnt = 0;
                for (int i = i0; i < i0 + rows; ++i) cnt += (i < g.p3 - g.p2 ? i : g.p3 - g.p2 - 1) + 1;
                if (t < cnt) break;
                t -= cnt;
                i0 += SB;
            }
            const int rows = g.nb - g.p2 - i0 < SB ? g.nb - g.p2 - i0 : SB;
            // column j (relative) of the band holds the rows max(j, i0) .. i0 + rows - 1
            int j = 0;
            for (;;) {
                const int lo = j > i0 ? j : i0;
                const int cnt = i0 + rows - lo;
                if (t < cnt) { r = g.p2 + lo + t; break; }
                t -= cnt;
                ++j;
            }
            c = g.p2 + j;
        }
        A0 = Ab + (size_t)r * TM * g.ldA + (size_t)g.p0 * TM; dA = TM;
        B0 = Bb + (size_t)c * TM * g.ldB + (size_t)g.p0 * TM; dB = TM;
        nkt = g.p1 - g.p0;
        Ct = Cb + (size_t)r * TM * g.ldC + (size_t)c * TM;
        alpha = -1.0; accumulate = true;
    } else if constexpr (OP == OP_TRTRI_T || OP == OP_TRTRI_W) {
        // level with block size mb = p0: pair pr covers block rows [2 pr mb, 2 pr mb + 2 mb).
        // Tiles are enumerated longest-k-loop first (T: cl ascending, W21: rl descending; the pair index is the
        // fastest one) so that the long tiles start early and the short ones fill the tail.
        const int mb = g.p0;
        const int npair = g.p1;
        const int pr = g.p2 + bid % npair, rem = bid / npair;      // p2 = first pair (0 for a whole level)
        int rl, cl;
        if constexpr (OP == OP_TRTRI_T) { cl = rem / mb; rl = rem - cl * mb; }
        else { rl = mb - 1 - rem / mb; cl = rem % mb; }
        const int C0 = 2 * pr * mb, R0 = C0 + mb;
        if (R0 + rl >= g.nb) return;
        if constexpr (OP == OP_TRTRI_T) {
            // T[rl, cl] = sum_{kt = cl}^{mb-1} L21[rl, kt] W11[kt, cl]          (A from M, B from W, C into V)
            // walked from kt = mb-1 DOWN to cl: all tiles of the launch start on the same block row of W11 and
            // the same block column of L21 and stay in step, so the per-XCD L2 serves the re-reads
            A0 = Ab + (size_t)(R0 + rl) * TM * g.ldA + (size_t)(C0 + mb - 1) * TM; dA = -(ptrdiff_t)TM;
            B0 = Bb + (size_t)(C0 + mb - 1) * TM * g.ldB + (size_t)(C0 + cl) * TM; dB = -(ptrdiff_t)TM * g.ldB;
            nkt = mb - cl;
        } else {
            // W21[rl, cl] = - sum_{kt = 0}^{rl} W22[rl, kt] T[kt, cl]            (A from W, B from V, C into W)
            A0 = Ab + (size_t)(R0 + rl) * TM * g.ldA + (size_t)R0 * TM; dA = TM;
            B0 = Bb + (size_t)R0 * TM * g.ldB + (size_t)(C0 + cl) * TM; dB = (ptrdiff_t)TM * g.ldB;
            nkt = rl + 1;
            alpha = -1.0;
        }
        Ct = Cb + (size_t)(R0 + rl) * TM * g.ldC + (size_t)(C0 + cl) * TM;
    } else if constexpr (OP == OP_LAUUM) {
        // V[r, c] = sum_{kt = r}^{nb-1} W[kt, r]^T W[kt, c]
        int r, c;
        tri_decode(bid, r, c);   // ascending r = longest k loops first
        // k tiles walked from nb-1 DOWN to r: every tile starts on the last block row of W and they stay in step
        A0 = Ab + (size_t)(g.nb - 1) * TM * g.ldA + (size_t)r * TM; dA = -(ptrdiff_t)TM * g.ldA;
        B0 = Bb + (size_t)(g.nb - 1) * TM * g.ldB + (size_t)c * TM; dB = -(ptrdiff_t)TM * g.ldB;
        nkt = g.nb - r;
        tri_b = r == c;
        Ct = Cb + (size_t)r * TM * g.ldC + (size_t)c * TM;
    } else {
        // OP_PRED_U: U[m, r] = sum_{kt = 0}^{r} X[m, kt] W[r, kt]^T    (X = scaled cross covariance, n0pad x npad)
        const int r = g.nb - 1 - bid / g.p0, m = bid % g.p0;       // p0 = row tiles of X; longest k loops (large r) first
        A0 = Ab + (size_t)m * TM * g.ldA; dA = TM;
        B0 = Bb + (size_t)r * TM * g.ldB; dB = TM;
        nkt = r + 1;
        Ct = Cb + (size_t)m * TM * g.ldC + (size_t)r * TM;
    }

    const int tid = body_tid();
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // wave-uniform: the sub-tile origin (wm0, wn0) stays in SGPRs
    const int wm0 = (wave >> 1) * WTM, wn0 = (wave & 1) * WTN;
    typedef typename Mfma<T>::acc_t acc_t;
    acc_t acc[MIM][MIN];
    // address of accumulator element (i, j, e) in the C tile = wave-uniform part (scalar registers) + this lane's element offset
    // (byte offsets: a 128-row tile of the largest matrix spans 128 x 16384 x 8 bytes)
    const __amdgpu_buffer_rsrc_t crs = tile_rsrc(Ct);
    const unsigned cvoff = (unsigned)(Mfma<T>::lane_row(lane) * g.ldC + (lane & 15)) * (unsigned)sizeof(T);
    auto c_soff = [&](int i, int e) -> unsigned {
        return (unsigned)((wm0 + i * 16 + Mfma<T>::reg_row(e)) * g.ldC + wn0) * (unsigned)sizeof(T);
    };
    // C -= A B^T: the accumulators start from the C tile (its load overlaps the first operand loads) and the A
    // fragments are negated, so the epilogue is stores only
    constexpr bool PRELOAD_C = (OP == OP_SYRK);
#pragma unroll
    for (int i = 0; i < MIM; ++i)
#pragma unroll
        for (int j = 0; j < MIN; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if constexpr (PRELOAD_C && !(LCGP_EXP & 16)) {
                    acc[i][j][e] = BufIo<T>::load(crs, cvoff + j * 16 * (unsigned)sizeof(T), c_soff(i, e));
                } else {
                    acc[i][j][e] = 0;
                }
            }

    constexpr int SPT = TM / KT;   // stages per k tile
    const int nst = nkt * SPT;
    // (measurement support: where the host passes the two clock words -- the launch that forms A^-1 = W^T W, in either tile
    // size, and the first wide trailing update of the factorisation, which is what is left to stamp where A^-1 is accumulated
    // behind the chain -- wave 0 of the first block (the longest K loop of such a launch) stamps its K loop and epilogue
    // with the shader-clock counter and the 100 MHz real-time counter, both scalar: the ratio is the clock the chip held.
    // The window is approximate: the stamps are scalar instructions the compiler may schedule a few instructions into the
    // prologue / epilogue either way)
    unsigned long long clk0 = 0, rt0 = 0;
    if constexpr (OP == OP_LAUUM || OP == OP_SYRK) {
        if (g.clk && lin == 0 && wave == 0) { clk0 = __builtin_amdgcn_s_memtime(); rt0 = __builtin_amdgcn_s_memrealtime(); }
    }
    // The LAST k tile of the triangular products holds a triangular TM x TM block of W (LAUUM: W[r,r] as A, and as B too
    // on a diagonal tile; TRTRI_T: W11[cl,cl] as B; TRTRI_W: W22[rl,rl] as A; PRED_U: W[r,r] as B).  In the stage that
    // covers its k rows [ks, ks + 16) a wave whose rows (columns) of that operand lie wholly on the zero side would
    // only add exact zeros: it skips the stage's fragment reads and MFMAs (one wave-uniform test per stage, nothing
    // else changes; bit-identical results: the zeros are stored zeros).  LAUUM / TRTRI_W: 24 of the 64 (wave, stage)
    // pairs of such a tile, TRTRI_T / PRED_U: 16.
    constexpr bool HAS_TRI = OP != OP_SYRK;
    const int tri_first = HAS_TRI ? (nkt - 1) * SPT : nst;
    // the wave is idle in the stages [dead_lo, dead_hi) of the k loop (two scalars per wave)
    int dead_lo = nst, dead_hi = nst;
    if constexpr (OP == OP_LAUUM) { dead_lo = tri_first; dead_hi = tri_first + (tri_b && wn0 > wm0 ? wn0 : wm0) / KT; }
    else if constexpr (OP == OP_TRTRI_T) { dead_lo = tri_first; dead_hi = tri_first + wn0 / KT; }
    else if constexpr (OP == OP_TRTRI_W) dead_lo = tri_first + (wm0 + WTM) / KT;
    else if constexpr (OP == OP_PRED_U) dead_lo = tri_first + (wn0 + WTN) / KT;
    auto wave_live = [&](int sg) { return !HAS_TRI || sg < dead_lo || sg >= dead_hi; };
#ifndef LCGP_NO_ROW_IMAGE
    constexpr bool ROWS_A = LA == MK && sizeof(T) == 8, ROWS_B = LB == MK && sizeof(T) == 8;     // (store_stage_rows)
#else
    constexpr bool ROWS_A = false, ROWS_B = false;
#endif
    static_assert(TM * LDK <= KT * LD, "the row image fits the stage buffer");
    auto put_a = [&](T* S, const T (&reg)[EPT]) {
        if constexpr (ROWS_A) store_stage_rows<TM, NT>((double*)S, (const double (&)[EPT])reg, tid);
        else store_stage<T, LA, TM, NT>(S, reg, tid);
    };
    auto put_b = [&](T* S, const T (&reg)[EPT]) {
        if constexpr (ROWS_B) store_stage_rows<TM, NT>((double*)S, (const double (&)[EPT])reg, tid);
        else store_stage<T, LB, TM, NT>(S, reg, tid);
    };
    auto compute_stage = [&](int buf) {
        const T* as = As + buf * KT * LD;
        const T* bs = Bs + buf * KT * LD;
        const int l15 = lane & 15;
#pragma unroll
        for (int kk = 0; kk < KT / 4; ++kk) {
            const int krow = (kk * 4 + (lane >> 4)) * LD;
            const int kcol = kk * 4 + (lane >> 4);
            T af[MIM], bf[MIN];
#pragma unroll
            for (int i = 0; i < MIM; ++i) {
                const T v = ROWS_A ? as[(wm0 + i * 16 + l15) * LDK + kcol] : as[krow + wm0 + swz_col<T>(i * 16, l15, kk)];
                af[i] = PRELOAD_C ? -v : v;
            }
#pragma unroll
            for (int j = 0; j < MIN; ++j)
                bf[j] = ROWS_B ? bs[(wn0 + j * 16 + l15) * LDK + kcol] : bs[krow + wn0 + swz_col<T>(j * 16, l15, kk)];
#pragma unroll
            for (int i = 0; i < MIM; ++i)
#pragma unroll
                for (int j = 0; j < MIN; ++j) acc[i][j] = Mfma<T>::run(af[i], bf[j], acc[i][j]);
        }
    };
    bool done = false;
    if constexpr (TM == 64) {
        // K = 64: such a launch is one or two rounds of tiles and its time is the latency of ONE tile -- fetch all
        // four stages up front (one memory latency instead of four)
        if (nkt == 1) {
            T pa[SPT][EPT], pb[SPT][EPT];
#pragma unroll
            for (int s = 0; s < SPT; ++s) {
                load_stage<T, LA, TM, NT>(A0, g.ldA, s * KT, pa[s], tid);
                load_stage<T, LB, TM, NT>(B0, g.ldB, s * KT, pb[s], tid);
            }
#pragma unroll
            for (int s = 0; s < SPT; ++s) {
                put_a(As + (s & 1) * KT * LD, pa[s]);
                put_b(Bs + (s & 1) * KT * LD, pb[s]);
                __syncthreads();
                if (wave_live(s)) compute_stage(s & 1);
            }
            done = true;
        }
    }
    if (!done) {
        // register prefetch PF stages ahead (the loads of stage s+PF are issued while stage s is multiplied): with
        // few workgroups per CU one stage of MFMAs (~0.5 us) does not cover an HBM/L2 round trip; the 64-tile kernel,
        // whose launches are often a fraction of a round, looks a whole k tile ahead
        // Depth by register budget (128 VGPRs per lane at the occupancy the launches need): two stages everywhere since
        // round 6 -- the fp64 rank-k update, which holds its C tile in the accumulators from the start, looked ONE stage
        // ahead until the C tile moved to buffer addressing (one offset register instead of eight 64-bit row pointers) and
        // the k-contiguous operands to their row image (one fragment address instead of four): 119 registers at one stage,
        // 128 without a spill at two (profiles/r06_syrk_ab.txt: -0.09 ms per evaluation).  The hand-counted form of the loop
        // is not used for it: beside the 64 loads of the C tile the compiler spills accumulators around the loop.
        constexpr bool F64 = sizeof(T) == 8;
#ifndef LCGP_SYRK_PF
#define LCGP_SYRK_PF 2
#endif
        constexpr int PF = OP == OP_SYRK ? (F64 ? LCGP_SYRK_PF : 2) : (TM == 64 ? (F64 ? 2 : 4) : 2);
        constexpr int NPC = EPT * (int)sizeof(T) / 16;      // 16-byte pieces per lane, operand and stage
#ifndef LCGP_NO_COUNTED_PREFETCH
#ifdef LCGP_SYRK_COUNTED
        constexpr bool COUNTED = PF == 2 && (NPC == 1 || NPC == 2) && (!PRELOAD_C || F64);
#else
        constexpr bool COUNTED = PF == 2 && (NPC == 1 || NPC == 2 || NPC == 4) && !PRELOAD_C;
#endif
#else
        constexpr bool COUNTED = false;
#endif
        if constexpr (COUNTED) {
            // the hand-counted form of the loop below (see gload_piece): same stage images, same order of arithmetic
            typedef typename Piece<T>::v pc_t;
            constexpr int NL = 2 * NPC;                     // loads per register set
            pc_t qa[PF][NPC], qb[PF][NPC];
            auto put_ap = [&](T* S, const pc_t (&reg)[NPC]) {
                if constexpr (ROWS_A) store_stage_rows_p<TM, NT>((double*)S, (const Piece<double>::v (&)[NPC])reg, tid);
                else store_stage_p<T, LA, TM, NT>(S, reg, tid);
            };
            auto put_bp = [&](T* S, const pc_t (&reg)[NPC]) {
                if constexpr (ROWS_B) store_stage_rows_p<TM, NT>((double*)S, (const Piece<double>::v (&)[NPC])reg, tid);
                else store_stage_p<T, LB, TM, NT>(S, reg, tid);
            };
            const unsigned voffA = stage_lane_offset<T, LA, TM, NT>(g.ldA, tid), voffB = stage_lane_offset<T, LB, TM, NT>(g.ldB, tid);
            // NO control flow between an asm load and the wait that covers it: at a join the compiler may move a value to
            // another register, and a move placed behind the asm copies the register before the data has arrived.  The number
            // of stages is a positive multiple of SPT >= 4, so the first PF stages exist and the loop ends with exactly PF
            // stages that have nothing left to fetch.
            static_assert(SPT % PF == 0 && SPT >= 2 * PF, "stage count of a k tile");
#pragma unroll
            for (int h = 0; h < PF; ++h) {
                const int kt = h / SPT, ks = (h % SPT) * KT;
                load_stage_p<T, LA, TM, NT>(A0 + (ptrdiff_t)kt * dA, g.ldA, ks, qa[h], voffA);
                load_stage_p<T, LB, TM, NT>(B0 + (ptrdiff_t)kt * dB, g.ldB, ks, qb[h], voffB);
            }
            int s = 0;
            // main part: every stage of a pass exists and has a stage PF ahead to fetch: PF sets are outstanding whenever
            // one is stored, the oldest of them is complete at vmcnt((PF - 1) NL)
            for (; s + 2 * PF <= nst; s += PF) {
#pragma unroll
                for (int h = 0; h < PF; ++h) {
                    const int buf = (PF & 1) ? ((s + h) & 1) : (h & 1);
                    // (LCGP_EXP: destructive timing experiments, never in a product build -- 1: no operand loads, 2: no LDS
                    // stores / barrier, 8: no MFMA stage; results are garbage)
                    if constexpr (!(LCGP_EXP & 1)) vm_wait_set<(PF - 1) * NL>(qa[h], qb[h]);
                    if constexpr (!(LCGP_EXP & 2)) {
                        put_ap(As + buf * KT * LD, qa[h]);
                        put_bp(Bs + buf * KT * LD, qb[h]);
                        __syncthreads();
                    }
                    const int kt = (s + h + PF) / SPT, ks = ((s + h + PF) % SPT) * KT;
                    if constexpr (!(LCGP_EXP & 1)) {
                        load_stage_p<T, LA, TM, NT>(A0 + (ptrdiff_t)kt * dA, g.ldA, ks, qa[h], voffA);
                        load_stage_p<T, LB, TM, NT>(B0 + (ptrdiff_t)kt * dB, g.ldB, ks, qb[h], voffB);
                    }
                    if constexpr (!(LCGP_EXP & 8))
                    if (wave_live(s + h)) compute_stage(buf);
                }
            }
            // the last PF stages (s = nst - PF here): nothing left to fetch, the sets in flight are waited for together
#pragma unroll
            for (int h = 0; h < PF; ++h) {
                const int buf = (PF & 1) ? ((s + h) & 1) : (h & 1);
                vm_wait_set<0>(qa[h], qb[h]);
                put_ap(As + buf * KT * LD, qa[h]);
                put_bp(Bs + buf * KT * LD, qb[h]);
                __syncthreads();
                if (wave_live(s + h)) compute_stage(buf);
            }
            // (nothing the asm loaded is outstanding here: the last stored set was waited for with vmcnt(0))
        } else {
        T ra[PF][EPT], rb[PF][EPT];
        const unsigned uoffA = stage_lane_offset<T, LA, TM, NT>(g.ldA, tid), uoffB = stage_lane_offset<T, LB, TM, NT>(g.ldB, tid);
#pragma unroll
        for (int h = 0; h < PF; ++h) {
            if (h < nst) {
                const int kt = h / SPT, ks = (h % SPT) * KT;
                load_stage_u<T, LA, TM, NT>(A0 + (ptrdiff_t)kt * dA, g.ldA, ks, ra[h], uoffA);
                load_stage_u<T, LB, TM, NT>(B0 + (ptrdiff_t)kt * dB, g.ldB, ks, rb[h], uoffB);
            }
        }
        for (int s = 0; s < nst; s += PF) {
#pragma unroll
            for (int h = 0; h < PF; ++h) {       // static register index h; LDS buffer (s + h) & 1
                if (s + h < nst) {
                    const int buf = (PF & 1) ? ((s + h) & 1) : (h & 1);
                    if constexpr (!(LCGP_EXP & 128)) {
                    put_a(As + buf * KT * LD, ra[h]);
                    put_b(Bs + buf * KT * LD, rb[h]);
                    __syncthreads();
                    }
                    if (!(LCGP_EXP & 64) && s + h + PF < nst) {
                        const int kt = (s + h + PF) / SPT, ks = ((s + h + PF) % SPT) * KT;
                        load_stage_u<T, LA, TM, NT>(A0 + (ptrdiff_t)kt * dA, g.ldA, ks, ra[h], uoffA);
                        load_stage_u<T, LB, TM, NT>(B0 + (ptrdiff_t)kt * dB, g.ldB, ks, rb[h], uoffB);
                    }
                    if (wave_live(s + h)) compute_stage(buf);
                }
            }
        }
        }
    }
#pragma unroll
    for (int mi = 0; mi < MIM; ++mi)
#pragma unroll
        for (int ni = 0; ni < MIN; ++ni)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const unsigned dvoff = cvoff + ni * 16 * (unsigned)sizeof(T), dsoff = c_soff(mi, e);
                if constexpr (OP == OP_SYRK && TM == 128) {
                    if (g.skipq && bid + g.t0 == 0 && wm0 + mi * 16 + Mfma<T>::row(lane, e) < TS && wn0 + ni * 16 + (lane & 15) < TS)
                        continue;
                }
                if constexpr (PRELOAD_C) {
                    if constexpr (LCGP_EXP & 32) { if (acc[mi][ni][e] == (T)1.2345e-300) BufIo<T>::store((T)acc[mi][ni][e], crs, dvoff, dsoff); } else
                    BufIo<T>::store((T)acc[mi][ni][e], crs, dvoff, dsoff);
                } else {
                    double v = alpha * (double)acc[mi][ni][e];
                    if (accumulate) v += (double)BufIo<T>::load(crs, dvoff, dsoff);
                    BufIo<T>::store((T)v, crs, dvoff, dsoff);
                }
            }
    if constexpr (OP == OP_LAUUM || OP == OP_SYRK) {
        if (g.clk && lin == 0 && wave == 0) {
            const unsigned long long c1 = __builtin_amdgcn_s_memtime() - clk0, r1 = __builtin_amdgcn_s_memrealtime() - rt0;
            if (lane == 0) { g.clk[0] = c1; g.clk[1] = r1; }
        }
    }
    if constexpr (OP == OP_LAUUM && TM == 128) {
        // z = A^-1 b rides on the tiles of A^-1 while they are in registers (a pass over A^-1 of its own costs 0.1 ms at
        // the headline size): tile (r, c) contributes p1 = V_rc b_c to z_r and, off the diagonal, p2 = V_rc^T b_r to z_c.
        // Per lane the products over its own accumulators, then the 16 lanes of a row group (p1) / the four row groups
        // (p2) by butterflies, then the two column halves / four row quarters of the waves through LDS; fixed orders.
        if (g.part) {
            __syncthreads();                                  // the stage buffers are free now
            double* bsh = (double*)lds;                       // [0,128): b over the tile's rows, [128,256): over its columns
            double* p1s = bsh + 256;                          // [2][128]
            double* p2s = p1s + 256;                          // [4][128]
            int r, c;
            tri_decode(bid, r, c);
            const T* bk = (const T*)g.bvec + (size_t)k * g.ldC;
            if (tid < 256) bsh[tid] = (double)bk[(size_t)(tid < 128 ? r : c) * 128 + (tid & 127)];
            __syncthreads();
            const int l15 = lane & 15;
            double s2[MIN];
#pragma unroll
            for (int j = 0; j < MIN; ++j) s2[j] = 0.0;
#pragma unroll
            for (int i = 0; i < MIM; ++i)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int row = wm0 + i * 16 + Mfma<T>::row(lane, e);
                    const double br = bsh[row];
                    double s1 = 0.0;
#pragma unroll
                    for (int j = 0; j < MIN; ++j) {
                        const double v = (double)acc[i][j][e];
                        s1 = fma(v, bsh[128 + wn0 + j * 16 + l15], s1);
                        s2[j] = fma(v, br, s2[j]);
                    }
                    s1 += __shfl_xor(s1, 1);
                    s1 += __shfl_xor(s1, 2);
                    s1 += __shfl_xor(s1, 4);
                    s1 += __shfl_xor(s1, 8);
                    if (l15 == 0) p1s[(wave & 1) * 128 + row] = s1;
                }
#pragma unroll
            for (int j = 0; j < MIN; ++j) {
                double v = s2[j];
                v += __shfl_xor(v, 16);
                v += __shfl_xor(v, 32);
                if ((lane >> 4) == 0) p2s[(wave >> 1) * 128 + wn0 + j * 16 + l15] = v;
            }
            __syncthreads();
            double* pt = g.part + ((size_t)k * (g.nb * (g.nb + 1) / 2) + bid) * 256;
            if (tid < 128) pt[tid] = p1s[tid] + p1s[128 + tid];
            else if (tid < 256) {
                const int cc = tid - 128;
                pt[128 + cc] = (p2s[cc] + p2s[128 + cc]) + (p2s[256 + cc] + p2s[384 + cc]);
            }
        }
    }
}

template <typename T, int OP, int TM, int NW>
__global__ __launch_bounds__(NW * 64, TM == 128 ? (NW == 8 ? 4 : 2) : 4) void tile_gemm(GemmArgs g) {
    __shared__ __align__(16) unsigned char lds[4 * KT * (TM + 16) * sizeof(T)];
    gemm_body<T, OP, TM, NW>(g, blockIdx.x, lds);
}

// ---------------------------------------------------------------------------------------------------
// Filler tiles (fill_sched.h): 128 x 64 outputs on 4 waves (64 x 32 per wave, 4x2 accumulators), K = nst stages of 16.
// A filler workgroup shares its launch with the diagonal-block kernel, i.e. 256 threads and two workgroups per CU; a
// 64x64 tile then leaves the MFMA pipe half idle, this shape fills it (57 KB of LDS).  Operands in either orientation
// (MK: element (m, k) at P[m ld + k];  KM: at P[k ld + m]); C = (first ? 0 : C) +- A B^T(-like) product.
// ---------------------------------------------------------------------------------------------------
using lcgp_fill::FillJob;
using lcgp_fill::FillSet;

template <typename T, int LA, int LB, bool NEG>
__device__ __forceinline__ void rect_tile(const T* __restrict__ A0, int ldA, const T* __restrict__ B0, int ldB,
                                          T* __restrict__ Ct, int ldC, int nst, bool first, unsigned char* lds) {
    constexpr int TMR = 128, TNC = 64, NT = 256, NJ_ = TNC / 32;
    constexpr int LDA = TMR + 16, LDB = TNC + 16;
    constexpr int EA = TMR * KT / NT, EB = TNC * KT / NT;
    T* As = (T*)lds;                   // [2][KT * LDA]
    T* Bs = As + 2 * KT * LDA;         // [2][KT * LDB]
    const int tid = body_tid(), lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm0 = (wave >> 1) * 64, wn0 = (wave & 1) * (TNC / 2);
    typedef typename Mfma<T>::acc_t acc_t;
    acc_t acc[4][NJ_];
    if (first) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < NJ_; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[i][j][e] = 0;
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < NJ_; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    acc[i][j][e] = Ct[(size_t)(wm0 + i * 16 + Mfma<T>::row(lane, e)) * ldC + wn0 + j * 16 + (lane & 15)];
    }
    // Software pipeline over the K stages, one barrier per stage: while the MFMAs of stage s run from LDS buffer s & 1, the
    // registers of stage s + 1 are written to the other buffer (last read in stage s - 1, i.e. before the previous
    // barrier) and the global loads of stage s + 3 are issued.  A filler workgroup is alone on its SIMDs (one wave
    // each), so nothing else hides the LDS writes: issued behind the MFMAs of the same stage they cost 0.4 us per stage.
    T ra[2][EA], rb[2][EB];
    if (nst > 0) {
        load_stage<T, LA, TMR, NT>(A0, ldA, 0, ra[0], tid);
        load_stage<T, LB, TNC, NT>(B0, ldB, 0, rb[0], tid);
    }
    if (nst > 1) {
        load_stage<T, LA, TMR, NT>(A0, ldA, KT, ra[1], tid);
        load_stage<T, LB, TNC, NT>(B0, ldB, KT, rb[1], tid);
    }
    if (nst > 0) {
        store_stage<T, LA, TMR, NT>(As, ra[0], tid);
        store_stage<T, LB, TNC, NT>(Bs, rb[0], tid);
        if (nst > 2) {
            load_stage<T, LA, TMR, NT>(A0, ldA, 2 * KT, ra[0], tid);
            load_stage<T, LB, TNC, NT>(B0, ldB, 2 * KT, rb[0], tid);
        }
        __syncthreads();
    }
    for (int s = 0; s < nst; s += 2) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if (s + h < nst) {
                const T* as = As + h * KT * LDA;
                const T* bs = Bs + h * KT * LDB;
#pragma unroll
                for (int kk = 0; kk < KT / 4; ++kk) {
                    const int kr = kk * 4 + (lane >> 4);
                    T af[4], bf[NJ_];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const T v = as[kr * LDA + wm0 + swz_col<T>(i * 16, lane & 15, kk)];
                        af[i] = NEG ? -v : v;
                    }
#pragma unroll
                    for (int j = 0; j < NJ_; ++j) bf[j] = bs[kr * LDB + wn0 + swz_col<T>(j * 16, lane & 15, kk)];
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int j = 0; j < NJ_; ++j) acc[i][j] = Mfma<T>::run(af[i], bf[j], acc[i][j]);
                    if (kk == 0 && s + h + 1 < nst) {
                        // stage s + h + 1 (register set (h + 1) & 1) into the other buffer, behind the first MFMAs
                        store_stage<T, LA, TMR, NT>(As + (h ^ 1) * KT * LDA, ra[h ^ 1], tid);
                        store_stage<T, LB, TNC, NT>(Bs + (h ^ 1) * KT * LDB, rb[h ^ 1], tid);
                        if (s + h + 3 < nst) {
                            load_stage<T, LA, TMR, NT>(A0, ldA, (s + h + 3) * KT, ra[h ^ 1], tid);
                            load_stage<T, LB, TNC, NT>(B0, ldB, (s + h + 3) * KT, rb[h ^ 1], tid);
                        }
                    }
                }
                __syncthreads();
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NJ_; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e)
                Ct[(size_t)(wm0 + i * 16 + Mfma<T>::row(lane, e)) * ldC + wn0 + j * 16 + (lane & 15)] = (T)acc[i][j][e];
}

// One block of a filler set: block b of the launch's filler range -> (job, component, tile) -> operands (fill_sched.h).
template <typename T>
__device__ __forceinline__ void fill_dispatch(const FillSet& fs, int b, unsigned char* lds) {
    int ji = 0;
    while (ji + 1 < fs.njobs && b >= fs.job[ji].nblk) { b -= fs.job[ji].nblk; ++ji; }
    const FillJob jb = fs.job[ji];
    const int ld = fs.npad;
    if (jb.type == lcgp_fill::FILL_TRI_T || jb.type == lcgp_fill::FILL_TRI_W) {
        GemmArgs g;
        g.sA = g.sB = g.sC = fs.mat; g.ldA = g.ldB = g.ldC = ld; g.nb = fs.nb;
        g.p0 = jb.R0; g.p1 = jb.R1; g.p2 = jb.j0; g.p3 = 0; g.q = fs.q; g.t0 = 0; g.skipq = 1;
        if (jb.type == lcgp_fill::FILL_TRI_T) {
            g.A = fs.M; g.B = fs.W; g.C = fs.V;
            gemm_body<T, OP_TRTRI_T, 64, 4>(g, b + jb.t0 * fs.q, lds);       // (a job may be split over launches)
        } else {
            g.A = fs.W; g.B = fs.V; g.C = fs.W;
            gemm_body<T, OP_TRTRI_W, 64, 4>(g, b + jb.t0 * fs.q, lds);
        }
        return;
    }
    const int k = b % fs.q;
    int t = b / fs.q + jb.t0;
    T* M = (T*)fs.M + (size_t)k * fs.mat;
    T* W = (T*)fs.W + (size_t)k * fs.mat;
    T* V = (T*)fs.V + (size_t)k * fs.mat;
    const int kb0 = jb.kb0, kb1 = jb.kb1;
    if (jb.type == lcgp_fill::FILL_SYRK) {
        // M[R, j] -= sum_k M[R, k] M[j, k]^T over the block columns [kb0, kb1)
        int j = jb.j0;
        while (t >= jb.R1 - (j >> 1)) { t -= jb.R1 - (j >> 1); ++j; }
        const int R = (j >> 1) + t;
        rect_tile<T, MK, MK, true>(M + (size_t)R * 128 * ld + (size_t)kb0 * TS, ld, M + (size_t)j * TS * ld + (size_t)kb0 * TS, ld,
                                   M + (size_t)R * 128 * ld + (size_t)j * TS, ld, (kb1 - kb0) * (TS / KT), false, lds);
    } else if (jb.type == lcgp_fill::FILL_BROW) {
        // W[R, j] = -W[R, kb0 .. ] V[kb0 .., j]: row block R of the panel's block inverse (lower triangular: k < 2R + 2)
        const int nc = jb.j1 - jb.j0;
        const int R = jb.R0 + t / nc, j = jb.j0 + t % nc;
        const int ke = kb1 < 2 * R + 2 ? kb1 : 2 * R + 2;
        rect_tile<T, MK, KM, true>(W + (size_t)R * 128 * ld + (size_t)kb0 * TS, ld, V + (size_t)kb0 * TS * ld + (size_t)j * TS, ld,
                                   W + (size_t)R * 128 * ld + (size_t)j * TS, ld, (ke - kb0) * (TS / KT), true, lds);
    } else if (jb.type == lcgp_fill::FILL_CUPD) {
        // V[R, j] (+)= M[R, kb0 ..] W[kb0 .., j]; a column inside the panel starts at its own 128-aligned block row (zeros
        // above) and is the first contribution to the tile
        const int nc = jb.j1 - jb.j0;
        const int R = jb.R0 + t / nc, j = jb.j0 + t % nc;
        const bool own = j >= kb0;
        const int ks = kb0 + (own ? ((j - kb0) & ~1) : 0);
        rect_tile<T, MK, KM, false>(M + (size_t)R * 128 * ld + (size_t)ks * TS, ld, W + (size_t)ks * TS * ld + (size_t)j * TS, ld,
                                    V + (size_t)R * 128 * ld + (size_t)j * TS, ld, (kb1 - ks) * (TS / KT), own, lds);
    } else {
        // FILL_DUPD: V[R, j] (+)= W[kb0 .., R]^T W[kb0 .., j]; a row block inside (or below) the K range starts at its own
        // block row and is written for the first time
        int R = (int)((sqrt(4.0 * (double)t + 1.0) - 1.0) * 0.5);
        while ((R + 1) * (R + 2) <= t) ++R;
        while (R * (R + 1) > t) --R;
        const int j = t - R * (R + 1);
        const bool own = 2 * R >= kb0;
        const int ks = own ? 2 * R : kb0;
        rect_tile<T, KM, KM, false>(W + (size_t)ks * TS * ld + (size_t)R * 128, ld, W + (size_t)ks * TS * ld + (size_t)j * TS, ld,
                                    V + (size_t)R * 128 * ld + (size_t)j * TS, ld, (kb1 - ks) * (TS / KT), own, lds);
    }
}

constexpr int FILL_LDS_BYTES = 2 * KT * (128 + 16 + 64 + 16) * 8;

// filler jobs on their own (what the chain launches could not carry, and the tail of the progressive inverse)
template <typename T>
__global__ __launch_bounds__(256, 2) void fill_kernel(FillSet fs) {
    __shared__ __align__(16) unsigned char lds[FILL_LDS_BYTES];
    fill_dispatch<T>(fs, blockIdx.x, lds);
}

// Heterogeneous launches.  The panel chain of the Cholesky (diagonal block -> panel TRMM -> panel update, 64 times)
// is a sequence of small dependent launches that leave most CUs idle, and two HIP streams cannot overlap them with
// the wide trailing update on this platform.  So the chain launches CARRY independent work: blocks
// beyond the chain's own are filler tiles (fill_sched.h): the previous panel's trailing update on columns the chain of
// the current panel neither reads nor writes, and the jobs of the progressive inverse.  No inter-workgroup dependency
// exists inside such a launch; stream order between launches provides all the ordering.
template <typename T>
__global__ __launch_bounds__(256, 2) void leaf_fill_kernel(T* __restrict__ M, T* __restrict__ W, size_t mat, int npad, int jb,
                                                        double* __restrict__ logdet, int* __restrict__ info,
                                                        int q, FillSet fs) {
    __shared__ __align__(16) unsigned char lds[LEAF_LDS_BYTES];
    if ((int)blockIdx.x < q) leaf_body<T>(lds, blockIdx.x, M, W, mat, npad, jb, logdet, info);
    else fill_dispatch<T>(fs, blockIdx.x - q, lds);
}

// ---------------------------------------------------------------------------------------------------
// One launch per 64-column step of the panel chain.
// The three dependent launches of a step (diagonal block -> panel TRMM -> rank-64 update of the rest of the panel)
// are re-cut so that a step needs ONE launch X_c with no dependency between its workgroups:
//   * TRMM tiles (r, c), r > c:   L[r,c] = (A[r,c] - L[r,c-1] L[c,c-1]^T) W_cc^T   -- the contribution of the
//     PREVIOUS column is applied by the tile itself, the older ones arrived through the delayed updates below;
//     a tile whose row lies inside the panel also applies  A[r,r] -= L[r,c] L[r,c]^T  to its row's diagonal tile
//     (so the panel's diagonal tiles have exactly one writer per launch);
//   * the tile (c+1, c) is "special": after the two steps above its workgroup factors and inverts the diagonal
//     block c+1 (leaf_body), which is what launch X_{c+1} needs;
//   * delayed updates, one visit per tile: the tiles (r, c+1), r > c+1, of the next column receive the columns
//     J .. c-1 in one K loop (all of them final before this launch; column c is folded into the next step's TRMM);
//   * filler tiles of the previous panel's trailing update (syrk_rect_body) as before.
// Per panel of 4 columns: 1 diagonal-block launch + 4 step launches instead of 12 launches, and the chain of a step
// is  2-3 K=64 products + one diagonal block  on a single workgroup.
// ---------------------------------------------------------------------------------------------------
struct StepArgs {
    void* M; void* W; size_t mat; int npad; int nb;
    int c, J, pe;        // this step's column, the panel [J, pe)  (64-block units)
    int diag_end;        // TRMM tiles of rows < diag_end also update their row's diagonal tile (pe, or pe + 1 when the
                         // next panel's first diagonal block is factored by the trailing-update launch)
    int q;
    int has_special;     // tile (c+1, c) continues with the diagonal block c+1  (c + 1 < pe)
    int n_trmm;          // TRMM tiles per component INCLUDING the special one: rows trmm_r0 .. trmm_r0 + n_trmm - 1
    int n_upd;           // delayed-update tiles per component: rows upd_r0 .. upd_r0 + n_upd - 1
    int trmm_r0, upd_r0; // c + 1 / c + 2 for a whole step; the persistent launch cuts a step into the rows the next diagonal
                         // blocks need and the rows below (fill_sched.h: run_interleaved)
    double* logdet; int* info;
    FillSet fs;          // filler jobs (fs.nblk blocks)
};

template <typename T>
struct Tile64 {          // 64x64 tile on 256 threads: wave (wm, wn) owns a 32x32 quadrant = 2x2 MFMA accumulators
    typedef typename Mfma<T>::acc_t acc_t;
    static constexpr int LD = TS + 16, NT = 256, EPT = TS * KT / NT, SPT = TS / KT;
    static __device__ __forceinline__ void load(acc_t (&acc)[2][2], const T* Ct, int ld, int lane, int wm0, int wn0) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    acc[i][j][e] = Ct[(size_t)(wm0 + i * 16 + Mfma<T>::row(lane, e)) * ld + wn0 + j * 16 + (lane & 15)];
    }
    static __device__ __forceinline__ void zero(acc_t (&acc)[2][2]) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[i][j][e] = 0;
    }
    static __device__ __forceinline__ void store(const acc_t (&acc)[2][2], T* Ct, int ld, int lane, int wm0, int wn0) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    Ct[(size_t)(wm0 + i * 16 + Mfma<T>::row(lane, e)) * ld + wn0 + j * 16 + (lane & 15)] = (T)acc[i][j][e];
    }
    // One 64x64 operand (element (m, k) at P[m * ld + k]) into registers: all four K stages at once = one memory latency
    static __device__ __forceinline__ void fetch(T (&p)[SPT][EPT], const T* P, int ld, int tid) {
#pragma unroll
        for (int s = 0; s < SPT; ++s) load_stage<T, MK, TS, NT>(P, ld, s * KT, p[s], tid);
    }
    // acc += (NEG ? -1 : 1) * A B^T from fetched operands, staged through the two LDS buffers of each
    template <bool NEG>
    static __device__ __forceinline__ void mma_regs(acc_t (&acc)[2][2], const T (&pa)[SPT][EPT], const T (&pb)[SPT][EPT],
                                                    T* lds, int tid, int lane, int wm0, int wn0) {
        T* As = lds;
        T* Bs = As + 2 * KT * LD;
#pragma unroll
        for (int s = 0; s < SPT; ++s) {
            store_stage<T, MK, TS, NT>(As + (s & 1) * KT * LD, pa[s], tid);
            store_stage<T, MK, TS, NT>(Bs + (s & 1) * KT * LD, pb[s], tid);
            __syncthreads();
            const T* as = As + (s & 1) * KT * LD;
            const T* bs = Bs + (s & 1) * KT * LD;
#pragma unroll
            for (int kk = 0; kk < KT / 4; ++kk) {
                const int krow = (kk * 4 + (lane >> 4)) * LD;
                T af[2], bf[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const T v = as[krow + wm0 + swz_col<T>(i * 16, lane & 15, kk)];
                    af[i] = NEG ? -v : v;
                }
#pragma unroll
                for (int j = 0; j < 2; ++j) bf[j] = bs[krow + wn0 + swz_col<T>(j * 16, lane & 15, kk)];
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = Mfma<T>::run(af[i], bf[j], acc[i][j]);
            }
        }
    }
    template <bool NEG>
    static __device__ __forceinline__ void mma(acc_t (&acc)[2][2], const T* A0, int ldA, const T* B0, int ldB, T* lds,
                                               int tid, int lane, int wm0, int wn0) {
        T pa[SPT][EPT], pb[SPT][EPT];
        fetch(pa, A0, ldA, tid);
        fetch(pb, B0, ldB, tid);
        mma_regs<NEG>(acc, pa, pb, lds, tid, lane, wm0, wn0);
    }
    // ---- products of the panel chain whose operand is a tile this workgroup has just computed: it stays on chip ----
    // The accumulator tile as a complete k-major LDS operand, F[k * LD + (m ^ (k & 15))] = tile(m, k)   (TS * LD elements).
    // The 16 contiguous lanes of a ds_write_b64 group hold 16 consecutive k of ONE row m here, i.e. 16 addresses LD
    // apart = one bank pair (16-way conflict, 1.6 us per tile -- more than the L2 round trip it is meant to replace);
    // XOR-ing the column with the low four bits of k gives the 16 lanes 16 different bank pairs, and a fragment read
    // (16 consecutive columns of one k row) only sees a permutation of its aligned group.
    static __device__ __forceinline__ void to_operand(const acc_t (&acc)[2][2], T* F, int lane, int wm0, int wn0) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    F[(wn0 + j * 16 + (lane & 15)) * LD + ((wm0 + i * 16 + Mfma<T>::row(lane, e)) ^ (lane & 15))] = acc[i][j][e];
    }
    // acc += (NEG ? -1 : 1) * A B^T,  A = F (to_operand), B fetched; Bs = two staging buffers of KT * LD elements.
    // The first barrier inside also orders the writes of F.
    template <bool NEG>
    static __device__ __forceinline__ void mma_a_lds(acc_t (&acc)[2][2], const T* F, const T (&pb)[SPT][EPT], T* Bs, int tid,
                                                     int lane, int wm0, int wn0) {
#pragma unroll
        for (int s = 0; s < SPT; ++s) {
            store_stage<T, MK, TS, NT>(Bs + (s & 1) * KT * LD, pb[s], tid);
            __syncthreads();
            const T* as = F + s * KT * LD;
            const T* bs = Bs + (s & 1) * KT * LD;
#pragma unroll
            for (int kk = 0; kk < KT / 4; ++kk) {
                const int krow = (kk * 4 + (lane >> 4)) * LD;
                T af[2], bf[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const T v = as[krow + wm0 + i * 16 + ((lane & 15) ^ (4 * kk + (lane >> 4)))];
                    af[i] = NEG ? -v : v;
                }
#pragma unroll
                for (int j = 0; j < 2; ++j) bf[j] = bs[krow + wn0 + swz_col<T>(j * 16, lane & 15, kk)];
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = Mfma<T>::run(af[i], bf[j], acc[i][j]);
            }
        }
    }
    // acc += (NEG ? -1 : 1) * F^T-free form  X X^T  with X = the tile in F  (no staging, no barrier inside)
    template <bool NEG>
    static __device__ __forceinline__ void mma_ab_lds(acc_t (&acc)[2][2], const T* F, int lane, int wm0, int wn0) {
#pragma unroll
        for (int kk = 0; kk < TS / 4; ++kk) {
            const int krow = (kk * 4 + (lane >> 4)) * LD;
            T af[2], bf[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const T v = F[krow + wm0 + i * 16 + ((lane & 15) ^ ((4 * kk + (lane >> 4)) & 15))];
                af[i] = NEG ? -v : v;
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) bf[j] = F[krow + wn0 + j * 16 + ((lane & 15) ^ ((4 * kk + (lane >> 4)) & 15))];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = Mfma<T>::run(af[i], bf[j], acc[i][j]);
        }
    }
};

template <typename T>
__device__ __forceinline__ void chain_step_body(const StepArgs& a, int b, unsigned char* lds) {
    typedef Tile64<T> TL;
    const int tid = body_tid(), lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm0 = (wave >> 1) * 32, wn0 = (wave & 1) * 32;
    const int ld = a.npad;
    // block order = longest first: the special tiles (chain of the step), the filler tiles, then the short ones
    const int nspecial = a.has_special * a.q;
    int t = -1, k = 0;
    if (b < nspecial) { t = 0; k = b; }
    else if (b < nspecial + a.fs.nblk) { fill_dispatch<T>(a.fs, b - nspecial, lds); return; }
    else {
        b -= nspecial + a.fs.nblk;
        if (b < (a.n_trmm - a.has_special) * a.q) { k = b % a.q; t = b / a.q + a.has_special; }
        else b -= (a.n_trmm - a.has_special) * a.q;
    }
    if (t >= 0) {
        if (t == 0 && a.has_special) __builtin_amdgcn_s_setprio(3);   // the chain of the step shares its CU with other tiles
        const int c = a.c, r = a.trmm_r0 + t;
        T* Mk = (T*)a.M + (size_t)k * a.mat;
        const T* Wk = (const T*)a.W + (size_t)k * a.mat;
        T* Ct = Mk + (size_t)r * TS * ld + (size_t)c * TS;
        // The tile stays on chip between its products: after the previous column's contribution it goes to LDS as a
        // complete k-major operand (F), L[r,c] = tile W_cc^T reads it from there, and so does the update of the row's
        // diagonal block (X X^T with X = L[r,c]); only L[r,c] itself and the diagonal block travel to memory.  W_cc is
        // fetched before the first product, so the second one starts without a memory latency of its own.
        T* F = (T*)lds;                        // TS * LD elements (= the four staging buffers of TL::mma)
        T* Bst = F + TS * TL::LD;              // two B staging buffers behind it
        typename TL::acc_t acc[2][2];
        T pw[TL::SPT][TL::EPT];
        TL::fetch(pw, Wk + (size_t)c * TS * ld + (size_t)c * TS, ld, tid);
        TL::load(acc, Ct, ld, lane, wm0, wn0);
        if (c > a.J) {       // the previous column's contribution to this tile
            TL::template mma<true>(acc, Mk + (size_t)r * TS * ld + (size_t)(c - 1) * TS, ld,
                                   Mk + (size_t)c * TS * ld + (size_t)(c - 1) * TS, ld, (T*)lds, tid, lane, wm0, wn0);
            __syncthreads();                   // the staging buffers become F
        }
        TL::to_operand(acc, F, lane, wm0, wn0);
        TL::zero(acc);
        TL::template mma_a_lds<false>(acc, F, pw, Bst, tid, lane, wm0, wn0);
        TL::store(acc, Ct, ld, lane, wm0, wn0);         // L[r, c]
        if (r < a.diag_end) {
            T* Dt = Mk + (size_t)r * TS * ld + (size_t)r * TS;
            typename TL::acc_t dacc[2][2];
            TL::load(dacc, Dt, ld, lane, wm0, wn0);     // in flight across the barriers
            __syncthreads();                            // every read of the old F has been issued and consumed
            TL::to_operand(acc, F, lane, wm0, wn0);
            __syncthreads();
            TL::template mma_ab_lds<true>(dacc, F, lane, wm0, wn0);
            if (a.has_special && t == 0) {
                // the updated diagonal block goes to the factorisation through LDS (leaf_body's w area, which that routine
                // does not write before its first barrier); the block's L and inverse are what memory gets
                __syncthreads();
                double (*blk)[LEAF_LDT] = (double (*)[LEAF_LDT])((double*)lds + TS * LEAF_LDT);
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            blk[wm0 + i * 16 + Mfma<T>::row(lane, e)][wn0 + j * 16 + (lane & 15)] = (double)dacc[i][j][e];
                __syncthreads();
                leaf_body<T, true>(lds, k, (T*)a.M, (T*)a.W, a.mat, a.npad, c + 1, a.logdet, a.info);
            } else {
                TL::store(dacc, Dt, ld, lane, wm0, wn0);
            }
        }
        return;
    }
    {
        // delayed update, one visit per tile: the tiles (r, c + 1), r > c + 1, of the NEXT column receive the columns
        // J .. c-1 in one K loop (column c itself is folded into the next step's TRMM tiles)
        k = b % a.q;
        t = b / a.q;
        const int cc = a.c + 1;
        const int r = a.upd_r0 + t;
        T* Mk = (T*)a.M + (size_t)k * a.mat;
        T* Ct = Mk + (size_t)r * TS * ld + (size_t)cc * TS;
        typename TL::acc_t acc[2][2];
        TL::load(acc, Ct, ld, lane, wm0, wn0);
        // two operand sets: the next product's tiles are in flight while the current one runs
        T pa[2][TL::SPT][TL::EPT], pb[2][TL::SPT][TL::EPT];
        const T* Ar = Mk + (size_t)r * TS * ld;
        const T* Br = Mk + (size_t)cc * TS * ld;
        TL::fetch(pa[0], Ar + (size_t)a.J * TS, ld, tid);
        TL::fetch(pb[0], Br + (size_t)a.J * TS, ld, tid);
        for (int j = a.J; j < a.c; j += 2) {
            if (j + 1 < a.c) {
                TL::fetch(pa[1], Ar + (size_t)(j + 1) * TS, ld, tid);
                TL::fetch(pb[1], Br + (size_t)(j + 1) * TS, ld, tid);
            }
            TL::template mma_regs<true>(acc, pa[0], pb[0], (T*)lds, tid, lane, wm0, wn0);
            if (j + 1 < a.c) {
                if (j + 2 < a.c) {
                    TL::fetch(pa[0], Ar + (size_t)(j + 2) * TS, ld, tid);
                    TL::fetch(pb[0], Br + (size_t)(j + 2) * TS, ld, tid);
                }
                TL::template mma_regs<true>(acc, pa[1], pb[1], (T*)lds, tid, lane, wm0, wn0);
            }
        }
        TL::store(acc, Ct, ld, lane, wm0, wn0);
    }
}

template <typename T>
__global__ __launch_bounds__(256, 2) void chain_step_kernel(StepArgs a) {
    __shared__ __align__(16) unsigned char lds[LEAF_LDS_BYTES];
    chain_step_body<T>(a, blockIdx.x, lds);
}

// Trailing update of a panel that also factors the FIRST diagonal block of the next panel: the first q workgroups run
// leaf_body on that 64x64 block (the chain steps of the panel have already applied the panel to it: diag_end = pe + 1),
// the others are the tiles of the wide update (tile 0 of the 128-tile form leaves that quadrant alone; the 64-tile form
// starts at tile 1).  The next panel's chain then starts with its first step launch -- one dependent launch less per
// panel, and the diagonal block hides under the update.
template <typename T, int TM>
__global__ __launch_bounds__(256, 2) void wide_leaf_kernel(GemmArgs g, T* __restrict__ M, T* __restrict__ W, size_t mat,
                                                           int npad, int jb, double* __restrict__ logdet,
                                                           int* __restrict__ info) {
    constexpr int WIDE_LDS = 4 * KT * (TM + 16) * (int)sizeof(T);
    __shared__ __align__(16) unsigned char lds[WIDE_LDS > LEAF_LDS_BYTES ? WIDE_LDS : LEAF_LDS_BYTES];
    const int q = g.q;
    if ((int)blockIdx.x < q) {
        leaf_body<T>(lds, blockIdx.x, M, W, mat, npad, jb, logdet, info);
        return;
    }
    gemm_body<T, OP_SYRK, TM, 4>(g, blockIdx.x - q, lds);
}

// ---------------------------------------------------------------------------------------------------
// z = A^-1 b with A^-1 stored as lower 64x64 tiles, every tile read ONCE:
//   symv_tile_kernel : tile (r, c) -> p1 = V_tile b_c (64 values, belongs to z_r) and, off the diagonal,
//                      p2 = V_tile^T b_r (belongs to z_c); written to the partial buffer [tile][2][64]
//   symv_reduce_kernel: z_R = sum_{c <= R} p1(R, c) + sum_{r > R} p2(r, R)      (fixed summation order)
// ---------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void symv_tile_kernel(const T* __restrict__ V, size_t mat, int npad,
                                                        const T* __restrict__ b, double* __restrict__ part, int ntile) {
    __shared__ double vt[TS][TS + 1];
    __shared__ double red[4][TS];
    __shared__ double br[TS], bc[TS];
    const int k = blockIdx.y;
    int r, c;
    tri_decode(blockIdx.x, r, c);
    const T* Vt = V + (size_t)k * mat + (size_t)r * TS * npad + (size_t)c * TS;
    const int tid = threadIdx.x, j = tid & 63, g = tid >> 6;
    if (tid < TS) {
        br[tid] = (double)b[(size_t)k * npad + r * TS + tid];
        bc[tid] = (double)b[(size_t)k * npad + c * TS + tid];
    }
    double v[16];
#pragma unroll
    for (int m = 0; m < 16; ++m) v[m] = (double)Vt[(size_t)(g * 16 + m) * npad + j];
    __syncthreads();
    double* dst = part + ((size_t)k * ntile + blockIdx.x) * 2 * TS;
    // p2[j] = sum_i V[i][j] b_r[i]: each thread sums its 16 rows, then the 4 row groups
    double s2 = 0.0;
#pragma unroll
    for (int m = 0; m < 16; ++m) {
        s2 = fma(v[m], br[g * 16 + m], s2);
        vt[g * 16 + m][j] = v[m];
    }
    red[g][j] = s2;
    __syncthreads();
    if (tid < TS) dst[TS + tid] = r == c ? 0.0 : (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
    // p1[i] = sum_j V[i][j] b_c[j]: thread (i = tid & 63, g) sums columns 16 g .. 16 g + 15 from the LDS copy
    double s1 = 0.0;
#pragma unroll
    for (int m = 0; m < 16; ++m) s1 = fma(vt[j][g * 16 + m], bc[g * 16 + m], s1);
    __syncthreads();
    red[g][j] = s1;
    __syncthreads();
    if (tid < TS) dst[tid] = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
}

template <typename T, int TZ>
__global__ __launch_bounds__(256) void symv_reduce_kernel(const double* __restrict__ part, int ntile, int npad, int nb,
                                                          T* __restrict__ z) {
    // TZ = tile size of the partials (64: symv_tile_kernel, 128: the epilogue of the 128-tile LAUUM); nb in TZ units.
    // 256 / TZ groups of TZ threads take every NG-th term; the partial sums are combined in a fixed order
    constexpr int NG = 256 / TZ;
    __shared__ double sh[NG][TZ];
    const int k = blockIdx.y, R = blockIdx.x, i = threadIdx.x % TZ, g = threadIdx.x / TZ;
    const double* pk = part + (size_t)k * ntile * 2 * TZ;
    double s = 0.0;
    for (int c = g; c <= R; c += NG) s += pk[((size_t)(R * (R + 1) / 2 + c)) * 2 * TZ + i];
    for (int r = R + 1 + g; r < nb; r += NG) s += pk[((size_t)(r * (r + 1) / 2 + R)) * 2 * TZ + TZ + i];
    sh[g][i] = s;
    __syncthreads();
    if (g == 0) {
        double v;
        if constexpr (NG == 4) v = (sh[0][i] + sh[1][i]) + (sh[2][i] + sh[3][i]);
        else v = sh[0][i] + sh[1][i];
        z[(size_t)k * npad + R * TZ + i] = (T)v;
    }
}

// ---------------------------------------------------------------------------------------------------
// K5: fused gradient contraction over the lower tiles of A^-1 (HBM read once):
//   G_ij = sr_i sr_j (D/2 Ainv_ij - z_i z_j / 2),  weight 2 off the diagonal,
//   part[0..d-1] = sum w G C0 S_j^2/(1+S_j),  part[d] = sum w G C0,  part[d+1] = sum_i G_ii
// C0 and S_j are recomputed from x in LDS.  Per-tile partial sums, reduced later in fixed order.
// ---------------------------------------------------------------------------------------------------
// ---------------------------------------------------------------------------------------------------
// finalize: per component, reduce the tile partials, quad = b.(b - z), pack output (gsig_a = Y[a,:].(b - z): gsig_body).
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ double block_sum(double v, double* sh /*>= 4*/, int tid) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    __syncthreads();
    if ((tid & 63) == 0) sh[tid >> 6] = v;
    __syncthreads();
    return (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

// gsig_a = sum_i Y[a, i] (b_i - z_i): one workgroup per (output a, component); rides in the launch of the gradient
// contraction (blocks past the tiles), like b in the kernel-build launch
template <typename T>
__device__ __forceinline__ void gsig_body(int a, int k, int n, int npad, int d, int p, const T* __restrict__ Y,
                                          const T* __restrict__ b, const T* __restrict__ z, double* __restrict__ out) {
    __shared__ double sh[4];
    const int tid = threadIdx.x;
    const T* bk = b + (size_t)k * npad;
    const T* zk = z + (size_t)k * npad;
    double s = 0.0;
    for (int i = tid; i < n; i += 256) s += (double)Y[(size_t)a * n + i] * ((double)bk[i] - (double)zk[i]);
    s = block_sum(s, sh, tid);
    if (tid == 0) out[(size_t)k * (d + 5 + p) + 5 + d + a] = s;
}

// float32 (CZ): the quadratic form b^T (b - z) and the noise gradient Y (b - z) cancel when A is close to the identity
// (z ~ b), and an fp32 z carries an error of 6e-8 |b| -- the same size as b - z itself in that regime (the NLL of the
// n = 16384 configuration moved by 1.6e-4 relative, enough to end an L-BFGS-B run early or late).  A z = b gives
//   b - z = D (C o s s^T) z   exactly,
// whose error is D C (error of z): small exactly where the difference cancels.  The kernel matrix is recomputed tile by
// tile here anyway, so the tiles also emit the symmetric matrix-vector partials of c = (C o s s^T) z in double:
//   cpart[tile][0][i] = sum_j Cs_ij z_j  (rows of the tile),  cpart[tile][1][j] = sum_{i != j} Cs_ij z_i  (its columns)
template <typename T, int DD, int KERN>
__global__ __launch_bounds__(256) void grad_kernel(const T* __restrict__ V, size_t mat, int n, int npad, int d, int p,
                                                   const T* __restrict__ x, const T* __restrict__ sr,
                                                   const T* __restrict__ z, const double* __restrict__ theta,
                                                   double* __restrict__ part, int ntile, const T* __restrict__ Y,
                                                   const T* __restrict__ bvec, double* __restrict__ out,
                                                   double* __restrict__ cpart) {
    constexpr bool CZ = sizeof(T) == 4;
    if ((int)blockIdx.x >= ntile) {
        gsig_body<T>(blockIdx.x - ntile, blockIdx.y, n, npad, d, p, Y, bvec, z, out);
        return;
    }
    __shared__ double xr[TS][DD + 1];
    __shared__ double xc[TS][DD + 1];
    __shared__ double zr[TS], zc[TS], srr[TS], src[TS];
    __shared__ double red[4][DD + 2];
    const int k = blockIdx.y;
    int r, c;
    tri_decode(blockIdx.x, r, c);
    const double* th = th_row(theta, d, p, k);
    const double D = th[d + 2];
    const int tid = threadIdx.x;
    // scaled inputs of the tile's rows and columns; the dimensions d .. DD-1 of the instantiation are zero-filled, so the
    // loops below run over DD without a test (a zero distance leaves every product and sum unchanged)
    for (int e = tid; e < TS * DD; e += 256) {
        int i = e / DD, j = e - i * DD;
        int gi = r * TS + i, gj = c * TS + i;
        xr[i][j] = (j < d && gi < n) ? (double)x[(size_t)gi * d + j] / th[j] : 0.0;
        xc[i][j] = (j < d && gj < n) ? (double)x[(size_t)gj * d + j] / th[j] : 0.0;
    }
    if (tid < TS) {
        int gi = r * TS + tid, gj = c * TS + tid;
        zr[tid] = (double)z[(size_t)k * npad + gi];
        zc[tid] = (double)z[(size_t)k * npad + gj];
        srr[tid] = (sr && gi < n) ? (double)sr[gi] : 1.0;
        src[tid] = (sr && gj < n) ? (double)sr[gj] : 1.0;
    }
    __syncthreads();
    double acc[DD + 2];
#pragma unroll
    for (int e = 0; e < DD + 2; ++e) acc[e] = 0.0;
    double cz1[CZ ? 8 : 1], cz2[CZ ? 2 : 1];      // c partials: the thread's 8 rows (over its 2 columns), its 2 columns (over its rows)
    if constexpr (CZ) {
#pragma unroll
        for (int m = 0; m < 8; ++m) cz1[m] = 0.0;
        cz2[0] = cz2[1] = 0.0;
    }
    const double scale_c = th[d], nt_c = th[d + 1] / (1.0 + th[d + 1]);
    const T* Vk = V + (size_t)k * mat;
    // thread = two adjacent columns (one 16-byte load per row in fp64) x 8 rows: two independent chains per load
    const int j0 = (tid & 31) * 2;
    typedef T pair_t __attribute__((ext_vector_type(2)));
    // all eight rows of the thread are requested up front (the padded matrix has every row of the tile), and the scaled
    // inputs of its two columns stay in registers when the instantiation is narrow enough
    pair_t avs[8];
#pragma unroll
    for (int m = 0; m < 8; ++m)
        avs[m] = *(const pair_t*)(Vk + (size_t)(r * TS + (tid >> 5) * 8 + m) * npad + c * TS + j0);
    constexpr bool HOIST = DD <= 10;
    double xcv[2][HOIST ? DD : 1];
    if constexpr (HOIST) {
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int jj = 0; jj < DD; ++jj) xcv[h][jj] = xc[j0 + h][jj];
    }
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        const int i = (tid >> 5) * 8 + m;
        const int gi = r * TS + i;
        if (gi >= n) continue;
        const pair_t av = avs[m];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int j = j0 + h;
            const int gj = c * TS + j;
            if (gj >= n || gj > gi) continue;
            const double ainv = (double)av[h];
            const double wgt = gi == gj ? 1.0 : 2.0;
            const double G = wgt * srr[i] * src[j] * (0.5 * D * ainv - 0.5 * zr[i] * zc[j]);
            // C0 S_j^2 / (1 + S_j) = exp(-sum S) S_j^2 prod_{i != j} (1 + S_i): prefix / suffix products, no division
            double sv[DD], pre[DD];
            double prod = 1.0, ssum = 0.0;
#pragma unroll
            for (int jj = 0; jj < DD; ++jj) {
                double xcj;
                if constexpr (HOIST) xcj = xcv[h][jj]; else xcj = xc[j][jj];
                if constexpr (KERN == 0) {
                    const double s = fabs(xr[i][jj] - xcj);
                    sv[jj] = s;
                    pre[jj] = prod;
                    prod = fma(prod, s, prod);
                    ssum -= s;
                } else {            // squared exponential: dC0/d ell_j = C0 S_j^2 / ell_j, no polynomial factor
                    const double s = xr[i][jj] - xcj;
                    sv[jj] = s;
                    ssum = fma(-0.5 * s, s, ssum);
                }
            }
            // C0 = 0 where its exponential underflows, and the polynomial beside it may have overflowed (inf x 0): no contribution.
            // (Only the widest instantiation can get there: (1 + S)^16 stays finite for every S below 1e19.)
            if constexpr (DD > 16) { if (ssum < exp_floor<double>()) continue; }
            const double ex = exp_nonpos(ssum);
            const double ge = G * ex;
            if constexpr (CZ) {
                const double cs = srr[i] * src[j] * scale_c * ((1.0 - nt_c) * (ex * prod) + (gi == gj ? nt_c : 0.0));
                cz1[m] = fma(cs, zc[j], cz1[m]);
                if (gi != gj) cz2[h] = fma(cs, zr[i], cz2[h]);
            }
            double suf = 1.0;
#pragma unroll
            for (int jj = DD - 1; jj >= 0; --jj) {
                if constexpr (KERN == 0) {
                    acc[jj] = fma(ge * (sv[jj] * sv[jj]), pre[jj] * suf, acc[jj]);
                    suf = fma(suf, sv[jj], suf);
                } else {
                    acc[jj] = fma(ge, sv[jj] * sv[jj], acc[jj]);
                }
            }
            acc[DD] = fma(ge, prod, acc[DD]);
            if (gi == gj) acc[DD + 1] += G;
        }
    }
    // deterministic block reduction: wave butterfly, then 4 waves through LDS
#pragma unroll
    for (int e = 0; e < DD + 2; ++e) {
        double v = acc[e];
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
        acc[e] = v;
    }
    if ((tid & 63) == 0)
#pragma unroll
        for (int e = 0; e < DD + 2; ++e) red[tid >> 6][e] = acc[e];
    __syncthreads();
    if (tid < d + 2) {
        const int e = tid < d ? tid : (DD + tid - d);
        double* dst = part + ((size_t)k * ntile + blockIdx.x) * (DMAX + 2);       // (stride of the narrow kernels: d <= 32)
        dst[tid] = (red[0][e] + red[1][e]) + (red[2][e] + red[3][e]);
    }
    if constexpr (CZ) {
        // rows: the 32 lanes of a half wave share the thread's 8 rows (butterfly); columns: the 8 row groups through LDS
        __shared__ double c2s[8][TS];
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            double v = cz1[m];
            for (int off = 16; off > 0; off >>= 1) v += __shfl_xor(v, off);
            cz1[m] = v;
        }
        c2s[tid >> 5][j0] = cz2[0];
        c2s[tid >> 5][j0 + 1] = cz2[1];
        __syncthreads();
        double* dst = cpart + ((size_t)k * ntile + blockIdx.x) * 2 * TS;
        if ((tid & 31) == 0) {
#pragma unroll
            for (int m = 0; m < 8; ++m) dst[(tid >> 5) * 8 + m] = cz1[m];
        }
        if (tid < TS)
            dst[TS + tid] = ((c2s[0][tid] + c2s[1][tid]) + (c2s[2][tid] + c2s[3][tid])) +
                            ((c2s[4][tid] + c2s[5][tid]) + (c2s[6][tid] + c2s[7][tid]));
    }
}

// The same contraction for d > 32 (the reference's kernel loops over any number of input dimensions, covmat.py:35-42;
// its examples stop at 10): the dimensions are staged in LDS 32 at a time.  A first sweep over the chunks gives every
// element its total product and exponent; then one sweep per chunk with the dimension as the OUTER loop,
//   sum_ij G_ij e^{-sum S} S_j^2 prod_{i != j} (1 + S_i)  =  sum_ij ge_ij S_j^2 (prod_ij / (1 + S_j)),
// one accumulator at a time (per-thread accumulators for all d dimensions would not fit the register file; the division
// replaces the prefix/suffix products of the narrow kernels).  Per-tile partial sums have stride d + 2.
template <typename T, int KERN>
__global__ __launch_bounds__(256) void grad_kernel_wide(const T* __restrict__ V, size_t mat, int n, int npad, int d, int p,
                                                        const T* __restrict__ x, const T* __restrict__ sr,
                                                        const T* __restrict__ z, const double* __restrict__ theta,
                                                        double* __restrict__ part, int ntile, const T* __restrict__ Y,
                                                        const T* __restrict__ bvec, double* __restrict__ out,
                                                        double* __restrict__ cpart) {
    constexpr bool CZ = sizeof(T) == 4;
    if ((int)blockIdx.x >= ntile) {
        gsig_body<T>(blockIdx.x - ntile, blockIdx.y, n, npad, d, p, Y, bvec, z, out);
        return;
    }
    __shared__ double xr[TS][DMAX + 1];
    __shared__ double xc[TS][DMAX + 1];
    __shared__ double zr[TS], zc[TS], srr[TS], src[TS];
    __shared__ double red[4][DWIDE + 2];
    const int k = blockIdx.y;
    int r, c;
    tri_decode(blockIdx.x, r, c);
    const double* th = th_row(theta, d, p, k);
    const double D = th[d + 2];
    const double scale_c = th[d], nt_c = th[d + 1] / (1.0 + th[d + 1]);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid < TS) {
        int gi = r * TS + tid, gj = c * TS + tid;
        zr[tid] = (double)z[(size_t)k * npad + gi];
        zc[tid] = (double)z[(size_t)k * npad + gj];
        srr[tid] = (sr && gi < n) ? (double)sr[gi] : 1.0;
        src[tid] = (sr && gj < n) ? (double)sr[gj] : 1.0;
    }
    const T* Vk = V + (size_t)k * mat;
    const int j0 = (tid & 31) * 2;
    typedef T pair_t __attribute__((ext_vector_type(2)));
    pair_t avs[8];
#pragma unroll
    for (int m = 0; m < 8; ++m)
        avs[m] = *(const pair_t*)(Vk + (size_t)(r * TS + (tid >> 5) * 8 + m) * npad + c * TS + j0);
    auto stage = [&](int d0) {       // scaled inputs of the dimensions d0 .. d0 + 31 (zero beyond d)
        __syncthreads();
        for (int e = tid; e < TS * DMAX; e += 256) {
            int i = e / DMAX, jj = e - i * DMAX;
            int gi = r * TS + i, gj = c * TS + i;
            xr[i][jj] = (d0 + jj < d && gi < n) ? (double)x[(size_t)gi * d + d0 + jj] / th[d0 + jj] : 0.0;
            xc[i][jj] = (d0 + jj < d && gj < n) ? (double)x[(size_t)gj * d + d0 + jj] / th[d0 + jj] : 0.0;
        }
        __syncthreads();
    };
    double prodT[16], geT[16];          // element e = 2 m + h: total product, then G e^{-sum S}
#pragma unroll
    for (int e = 0; e < 16; ++e) { prodT[e] = 1.0; geT[e] = 0.0; }
    for (int d0 = 0; d0 < d; d0 += DMAX) {
        stage(d0);
#pragma unroll
        for (int m = 0; m < 8; ++m)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int i = (tid >> 5) * 8 + m, j = j0 + h;
                double pr = prodT[2 * m + h], ss = geT[2 * m + h];
                for (int jj = 0; jj < DMAX; ++jj) {
                    if constexpr (KERN == 0) {
                        const double s = fabs(xr[i][jj] - xc[j][jj]);
                        pr = fma(pr, s, pr);
                        ss -= s;
                    } else {
                        const double s = xr[i][jj] - xc[j][jj];
                        ss = fma(-0.5 * s, s, ss);
                    }
                }
                prodT[2 * m + h] = pr;
                geT[2 * m + h] = ss;
            }
    }
    double a_scale = 0.0, a_nug = 0.0;
    double cz1[CZ ? 8 : 1], cz2[CZ ? 2 : 1];
    if constexpr (CZ) {
#pragma unroll
        for (int m = 0; m < 8; ++m) cz1[m] = 0.0;
        cz2[0] = cz2[1] = 0.0;
    }
#pragma unroll
    for (int m = 0; m < 8; ++m)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int i = (tid >> 5) * 8 + m, j = j0 + h;
            const int gi = r * TS + i, gj = c * TS + j;
            double ge = 0.0;
            if (gi < n && gj < n && gj <= gi) {
                const double wgt = gi == gj ? 1.0 : 2.0;
                const double G = wgt * srr[i] * src[j] * (0.5 * D * (double)avs[m][h] - 0.5 * zr[i] * zc[j]);
                const bool zero = geT[2 * m + h] < exp_floor<double>();       // C0 = 0 (its polynomial may be inf)
                if (zero) prodT[2 * m + h] = 0.0;
                const double ex = zero ? 0.0 : exp_nonpos(geT[2 * m + h]);
                ge = G * ex;
                a_scale = fma(ge, prodT[2 * m + h], a_scale);
                if (gi == gj) a_nug += G;
                if constexpr (CZ) {
                    const double cs = srr[i] * src[j] * scale_c * ((1.0 - nt_c) * (ex * prodT[2 * m + h]) + (gi == gj ? nt_c : 0.0));
                    cz1[m] = fma(cs, zc[j], cz1[m]);
                    if (gi != gj) cz2[h] = fma(cs, zr[i], cz2[h]);
                }
            }
            geT[2 * m + h] = ge;
        }
    auto wave_sum = [&](double v) {
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
        return v;
    };
    for (int d0 = 0; d0 < d; d0 += DMAX) {
        stage(d0);
        const int dc = d - d0 < DMAX ? d - d0 : DMAX;
        for (int jj = 0; jj < dc; ++jj) {
            double a = 0.0;
#pragma unroll
            for (int m = 0; m < 8; ++m)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const double s = fabs(xr[(tid >> 5) * 8 + m][jj] - xc[j0 + h][jj]);
                    if constexpr (KERN == 0) a = fma(geT[2 * m + h] * (s * s), prodT[2 * m + h] / (1.0 + s), a);
                    else a = fma(geT[2 * m + h], s * s, a);
                }
            a = wave_sum(a);
            if (lane == 0) red[wave][d0 + jj] = a;
        }
    }
    a_scale = wave_sum(a_scale);
    a_nug = wave_sum(a_nug);
    if (lane == 0) { red[wave][d] = a_scale; red[wave][d + 1] = a_nug; }
    __syncthreads();
    if (tid < d + 2) {
        double* dst = part + ((size_t)k * ntile + blockIdx.x) * (d + 2);
        dst[tid] = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
    }
    if constexpr (CZ) {
        __shared__ double c2s[8][TS];
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            double v = cz1[m];
            for (int off = 16; off > 0; off >>= 1) v += __shfl_xor(v, off);
            cz1[m] = v;
        }
        c2s[tid >> 5][j0] = cz2[0];
        c2s[tid >> 5][j0 + 1] = cz2[1];
        __syncthreads();
        double* dst = cpart + ((size_t)k * ntile + blockIdx.x) * 2 * TS;
        if ((tid & 31) == 0) {
#pragma unroll
            for (int m = 0; m < 8; ++m) dst[(tid >> 5) * 8 + m] = cz1[m];
        }
        if (tid < TS)
            dst[TS + tid] = ((c2s[0][tid] + c2s[1][tid]) + (c2s[2][tid] + c2s[3][tid])) +
                            ((c2s[4][tid] + c2s[5][tid]) + (c2s[6][tid] + c2s[7][tid]));
    }
}

// c_R = sum_{c <= R} cpart(R, c)[0] + sum_{r >= R} cpart(r, R)[1]   (the diagonal tile's column part holds its strictly lower
// elements only), fixed order; one workgroup per 64-row block and component
__global__ __launch_bounds__(256) void cvec_reduce_kernel(const double* __restrict__ cpart, int ntile, int npad, int nb,
                                                          double* __restrict__ c) {
    __shared__ double sh[4][TS];
    const int k = blockIdx.y, R = blockIdx.x, i = threadIdx.x & 63, g = threadIdx.x >> 6;
    const double* pk = cpart + (size_t)k * ntile * 2 * TS;
    double s = 0.0;
    for (int cc = g; cc <= R; cc += 4) s += pk[((size_t)(R * (R + 1) / 2 + cc)) * 2 * TS + i];
    for (int r = R + g; r < nb; r += 4) s += pk[((size_t)(r * (r + 1) / 2 + R)) * 2 * TS + TS + i];
    sh[g][i] = s;
    __syncthreads();
    if (g == 0) c[(size_t)k * npad + R * TS + i] = (sh[0][i] + sh[1][i]) + (sh[2][i] + sh[3][i]);
}

// float32: gsig_a = D sum_i Y[a, i] c_i   (= sum_i Y[a, i] (b_i - z_i) without the cancellation)
template <typename T>
__global__ __launch_bounds__(256) void gsig_c_kernel(int n, int npad, int d, int p, const T* __restrict__ Y,
                                                     const double* __restrict__ c, const double* __restrict__ theta,
                                                     double* __restrict__ out) {
    __shared__ double sh[4];
    const int a = blockIdx.x, k = blockIdx.y, tid = threadIdx.x;
    const double* ck = c + (size_t)k * npad;
    double s = 0.0;
    for (int i = tid; i < n; i += 256) s += (double)Y[(size_t)a * n + i] * ck[i];
    s = block_sum(s, sh, tid);
    if (tid == 0) out[(size_t)k * (d + 5 + p) + 5 + d + a] = th_row(theta, d, p, k)[d + 2] * s;
}


template <typename T>
__global__ __launch_bounds__(256) void finalize_kernel(int n, int npad, int d, int p, int ntile,
                                                       const T* __restrict__ Y, const T* __restrict__ b,
                                                       const T* __restrict__ z, const double* __restrict__ part,
                                                       const double* __restrict__ logdet, const int* __restrict__ info,
                                                       const double* __restrict__ theta, double* __restrict__ out,
                                                       const double* __restrict__ cvec /*float32: (C o s s^T) z, else null*/) {
    __shared__ double sh[4];
    __shared__ double grp[256];
    __shared__ double sums[DWIDE + 2];
    const int pstride = (d > DMAX ? d : DMAX) + 2;       // doubles per tile in `part` (grad_kernel / grad_kernel_wide)
    const int k = blockIdx.x;
    const int tid = threadIdx.x;
    const double* th = th_row(theta, d, p, k);
    double* o = out + (size_t)k * (d + 5 + p);
    const T* bk = b + (size_t)k * npad;
    const T* zk = z + (size_t)k * npad;
    // tile partials: thread = (entry e = tid % ne, group g = tid / ne) sums its entry over the tiles g, g + ng, ...;
    // entry e then adds its ng group sums in a fixed order (one pass over the partials, deterministic)
    const int ne = d + 2, ng = 256 / ne;
    const int e = tid % ne, g = tid / ne;
    double acc = 0.0;
    if (g < ng) {
        // four independent chains (the loads of one chain would otherwise wait for each other), combined in a fixed order
        double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
        const double* pk = part + (size_t)k * ntile * pstride + e;
        int t = g;
        for (; t + 3 * ng < ntile; t += 4 * ng) {
            a0 += pk[(size_t)t * pstride];
            a1 += pk[(size_t)(t + ng) * pstride];
            a2 += pk[(size_t)(t + 2 * ng) * pstride];
            a3 += pk[(size_t)(t + 3 * ng) * pstride];
        }
        for (; t < ntile; t += ng) a0 += pk[(size_t)t * pstride];
        acc = (a0 + a1) + (a2 + a3);
    }
    grp[tid] = acc;
    double v = 0.0;
    if (cvec) {
        // b^T (b - z) = D b^T (C o s s^T) z: no cancellation (see grad_kernel)
        const double* ck = cvec + (size_t)k * npad;
#pragma unroll 4
        for (int i = tid; i < n; i += 256) v += (double)bk[i] * ck[i];
        v *= th[d + 2];
    } else {
#pragma unroll 4
        for (int i = tid; i < n; i += 256) v += (double)bk[i] * ((double)bk[i] - (double)zk[i]);
    }
    __syncthreads();
    if (tid < ne) {
        double s = 0.0;
        for (int gg = 0; gg < ng; ++gg) s += grp[gg * ne + tid];
        sums[tid] = s;
    }
    const double quad = block_sum(v, sh, tid);          // (its barriers also publish sums[])
    const double scale = th[d], nug = th[d + 1];
    const double nt = nug / (1.0 + nug);
    if (tid == 0) {
        o[0] = logdet[k];
        o[1] = quad;
        o[2] = (double)info[k];
        for (int j = 0; j < d; ++j) o[3 + j] = scale * (1.0 - nt) / th[j] * sums[j];
        o[3 + d] = (1.0 - nt) * sums[d] + nt * sums[d + 1];
        o[4 + d] = scale * (sums[d + 1] - sums[d]) / ((1.0 + nug) * (1.0 + nug));
    }
}


// small helpers ----------------------------------------------------------------------------------------
__global__ void zero_stats_kernel(double* logdet, int* info, int q) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < q) { logdet[i] = 0.0; info[i] = 0; }
}

__global__ void copy_stats_kernel(const double* logdet, const int* info, double* ld_out, int* info_out, int q) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < q) { if (ld_out) ld_out[i] = logdet[i]; if (info_out) info_out[i] = info[i]; }
}

template <typename T>
__global__ void fetch_kernel(const T* __restrict__ src, int npad, int n, T* __restrict__ dst) {
    int j = blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y;
    if (j >= n) return;
    dst[(size_t)i * n + j] = j <= i ? src[(size_t)i * npad + j] : src[(size_t)j * npad + i];
}

// ghat[k, m] = sum_i X_k[m, i] z_k[i] ;  gvar[k, m] = scale_k - D_k * sum_i U_k[m, i]^2        (one wave per row m)
template <typename T>
__global__ __launch_bounds__(64) void pred_reduce_kernel(const T* __restrict__ X, const T* __restrict__ U, size_t slab, int ld,
                                                         int n, const T* __restrict__ z, int npad,
                                                         const double* __restrict__ theta, int tw, int d, int ldo,
                                                         double* __restrict__ ghat, double* __restrict__ gvar) {
    const int m = blockIdx.x, k = blockIdx.y, lane = threadIdx.x;
    const double* th = theta + (size_t)k * tw;
    const double scale = th[d], D = th[d + 2];
    const T* Xr = X + (size_t)k * slab + (size_t)m * ld;
    const T* Ur = U + (size_t)k * slab + (size_t)m * ld;
    const T* zk = z + (size_t)k * npad;
    double s1 = 0.0, s2 = 0.0;
    for (int i = lane; i < n; i += 64) {
        const double u = (double)Ur[i];
        s1 += (double)Xr[i] * (double)zk[i];
        s2 += u * u;
    }
    for (int off = 32; off > 0; off >>= 1) { s1 += __shfl_xor(s1, off); s2 += __shfl_xor(s2, off); }
    if (lane == 0) { ghat[(size_t)k * ldo + m] = s1; gvar[(size_t)k * ldo + m] = scale - D * s2; }
}

// ---------------------------------------------------------------------------------------------------
// host-side drivers (enqueue only)
// ---------------------------------------------------------------------------------------------------
#define CHECK_LAUNCH(what)                                  \
    do {                                                    \
        hipError_t e__ = hipGetLastError();                 \
        if (e__ != hipSuccess) return fail(what, e__);      \
    } while (0)

template <typename T, int DD>
void launch_build(hipStream_t st, const Ws& w, dim3 grid, const void* x, const void* sr, const double* theta, const void* Y) {
    // with Y (the NLL path) the launch also zeroes the log-determinant and status words of the components
    auto go = [&](auto kern) {
        hipLaunchKernelGGL((build_kernel<T, DD, decltype(kern)::value>), grid, dim3(256), 0, st, (T*)(w.base + w.off_M), w.mat, w.n,
                           w.npad, w.d, w.p, (const T*)x, (const T*)sr, theta, w.ntile_lower, (const T*)Y, (T*)(w.base + w.off_b),
                           Y ? (double*)(w.base + w.off_logdet) : nullptr, Y ? (int*)(w.base + w.off_info) : nullptr);
    };
    if (w.kern == 0) go(std::integral_constant<int, 0>{}); else go(std::integral_constant<int, 1>{});
}

template <typename T>
int do_build(hipStream_t st, const Ws& w, const void* x, const void* sr, const double* theta, const void* Y = nullptr) {
    // with Y the launch also computes b_k = Y^T psi_k into the workspace (extra blocks past the tiles)
    dim3 grid(w.ntile_lower + (Y ? (w.npad + 255) / 256 : 0), w.q);
    if (w.d <= 2) launch_build<T, 2>(st, w, grid, x, sr, theta, Y);
    else if (w.d <= 4) launch_build<T, 4>(st, w, grid, x, sr, theta, Y);
    else if (w.d <= 6) launch_build<T, 6>(st, w, grid, x, sr, theta, Y);
    else if (w.d <= 10) launch_build<T, 10>(st, w, grid, x, sr, theta, Y);
    else if (w.d <= 16) launch_build<T, 16>(st, w, grid, x, sr, theta, Y);
    else launch_build<T, DMAX>(st, w, grid, x, sr, theta, Y);
    CHECK_LAUNCH("build_kernel");
    return 0;
}

template <typename T, int OP, int TM = TS>
int launch_gemm(hipStream_t st, const GemmArgs& g, int ntiles, int q) {
    if (ntiles <= 0) return 0;
    // 128x128 tiles: eight waves (32x64 each, 128 registers, four waves per SIMD) for the long products of the inverse; FOUR
    // waves (64x64 each, 233 registers, two per SIMD, still two workgroups per CU) for the rank-256 update, whose 16-stage
    // tiles spend relatively more time at their barriers: half as many waves to synchronise, twice the MFMAs between two
    // barriers, half the fragment reads per MFMA (profiles/r06_syrk_ab.txt: 1294 -> 1242 us per evaluation; the long products
    // lose on four waves: A^-1 2.79 -> 2.83 ms)
    constexpr int NW = TM == 128 ? (OP == OP_SYRK ? 4 : 8) : 4;
    GemmArgs h = g;
    h.q = q;
    h.t0 = 0;
    h.skipq = 0;
    hipLaunchKernelGGL((tile_gemm<T, OP, TM, NW>), dim3((unsigned)ntiles * q), dim3(NW * 64), 0, st, h);
    CHECK_LAUNCH("tile_gemm");
    return 0;
}

// ---------------------------------------------------------------------------------------------------
// Schedule parameters.  They travel with each call (lcgp_sched in the header; NULL = lcgp_sched_default):
// the library keeps no mutable state.  Results do not depend on them beyond rounding.
// ---------------------------------------------------------------------------------------------------
inline lcgp_sched default_sched() {
    lcgp_sched s;
    s.outer_blocks = 0;            // 0 = automatic: 4 (fp64) / 8 (fp32) 64-blocks per outer Cholesky panel
    s.syrk_small_tiles = 3000;     // below this many 128x128 tiles (x components) a trailing update runs on 64x64 tiles (2000 until
                                   // round 6: with two prefetch stages and the row image the 64-tile update caught up with the 128-tile one)
    s.trtri_small_tiles = 4200;    // the same switch for the whole triangular inverse ...
    s.lauum_small_tiles = 2048;    // ... and for A^-1 = W^T W
    s.trtri_level_small = 600;     // a single level of the triangular inverse below this many 128x128 tiles: 64x64 tiles
    s.fill_leaf = 248;             // filler blocks (128x64 tiles) carried by a diagonal-block launch
    s.fill_step = 248;             // ... and by a chain-step launch that ends in a diagonal block
    s.leaf_in_wide = 2048;         // a trailing update of at most this many 64x64 tiles also factors the next diagonal block
    s.progressive_tiles = 600;     // L^-1 and A^-1 formed behind the chain up to this many 128x128 lower tiles x components
                                   // (n = 4096: one component per rank; measured 2.72 -> 2.55 ms there, slower from two on)
    s.progressive_far = 1;         // ... with the far columns of the trailing updates still riding on the chain
    s.progressive_lauum = 48;      // ... and A^-1 = W^T W accumulated behind the chain as well up to this many 64-blocks per side
                                   // (n = 2048: 1.12 -> 0.98 ms; at n = 4096 its tail is one ragged launch of long K loops
                                   // that loses to the one-launch W^T W: 2.54 vs 2.43 ms)
    s.pair_tiles = LCGP_PAIR_TILES_DEFAULT;   // paired panels: one K = 2 ob update of the columns between the second panel and the far ones
    return s;
}

inline int check_sched(const lcgp_sched& s) {
    if (s.outer_blocks < 0 || s.outer_blocks > 64) return bad("sched.outer_blocks must be in [0, 64]");
    if (s.syrk_small_tiles < 0 || s.trtri_small_tiles < 0 || s.lauum_small_tiles < 0 || s.trtri_level_small < 0 ||
        s.fill_leaf < 0 || s.fill_step < 0 || s.leaf_in_wide < 0 || s.progressive_tiles < 0 || s.progressive_lauum < 0 ||
        s.pair_tiles < 0)
        return bad("sched fields must be >= 0");
    return 0;
}

using lcgp_fill::trapezoid_tiles;

// launches the filler set on its own
template <typename T>
int launch_fill(hipStream_t st, const FillSet& fs) {
    if (fs.nblk <= 0) return 0;
    hipLaunchKernelGGL((fill_kernel<T>), dim3((unsigned)fs.nblk), dim3(256), 0, st, fs);
    CHECK_LAUNCH("fill_kernel");
    return 0;
}

// trailing update with the panel [J, pe) of the tile columns [c_lo, c_hi) (64-block units, all rows below)
template <typename T>
int potrf_trailing(hipStream_t st, const Ws& w, int J, int pe, int c_lo, int c_hi, bool tiles128, bool with_leaf,
                   unsigned long long* clk = nullptr) {
    if (c_lo >= c_hi) return 0;
    T* M = (T*)(w.base + w.off_M);
    GemmArgs g;
    g.sA = g.sB = g.sC = w.mat; g.ldA = g.ldB = g.ldC = w.npad;
    g.A = M; g.B = M; g.C = M;
    g.clk = clk;
    // 128x128 tiles when the panel boundaries are 128-aligned AND the launch has enough of them to fill the chip (the
    // plan decides); a launch with few tiles is bounded by the duration of one tile, which is 4x shorter on 64x64 tiles
    if (tiles128) {
        g.nb = w.nb / 2; g.p0 = J / 2; g.p1 = pe / 2; g.p2 = c_lo / 2; g.p3 = c_hi / 2;
        const int nt = trapezoid_tiles(w.nb / 2, c_lo / 2, c_hi / 2);
        if (with_leaf) {
            g.q = w.q; g.t0 = 0; g.skipq = 1;
            hipLaunchKernelGGL((wide_leaf_kernel<T, 128>), dim3((unsigned)(nt + 1) * w.q), dim3(256), 0, st, g, M,
                               (T*)(w.base + w.off_W), w.mat, w.npad, c_lo, (double*)(w.base + w.off_logdet),
                               (int*)(w.base + w.off_info));
            CHECK_LAUNCH("wide_leaf_kernel");
            return 0;
        }
        return launch_gemm<T, OP_SYRK, 128>(st, g, nt, w.q);
    }
    g.nb = w.nb; g.p0 = J; g.p1 = pe; g.p2 = c_lo; g.p3 = c_hi;
    const int nt = trapezoid_tiles(w.nb, c_lo, c_hi);
    if (with_leaf) {
        // tile 0 = the diagonal block itself: it belongs to the special workgroups (the chain steps of the panel have
        // already applied the panel to it)
        g.q = w.q; g.t0 = 1; g.skipq = 0;
        hipLaunchKernelGGL((wide_leaf_kernel<T, 64>), dim3((unsigned)nt * w.q), dim3(256), 0, st, g, M,
                           (T*)(w.base + w.off_W), w.mat, w.npad, c_lo, (double*)(w.base + w.off_logdet),
                           (int*)(w.base + w.off_info));
        CHECK_LAUNCH("wide_leaf_kernel");
        return 0;
    }
    return launch_gemm<T, OP_SYRK>(st, g, nt, w.q);
}

// ---- the plan of a factorisation as a caller-owned, position-independent block of bytes (lcgp_plan_build) ----
// header | Launch[nlaunch].  It depends on (dtype, n, q_local, with_inverse, sched) only, so a caller builds it once and
// passes it with every evaluation: no planning in the evaluation loop.
constexpr unsigned PLAN_MAGIC = 0x4c43504cu;
struct PlanHeader {
    unsigned magic;
    int version;
    int dtype, n, nb, q, with_inverse;
    int nlaunch;
    int inverse_done;          // what the plan leaves behind the factorisation: 0 = L, 1 = and L^-1, 2 = and A^-1
    int reserved;              // (zero; the plan is host-only: nothing in it depends on the device)
    lcgp_sched sched;
    size_t off_launch, bytes;
};

inline lcgp_fill::PlanParams plan_params(int dtype, int nb, int q, bool with_inverse, const lcgp_sched& sc, int* inverse_done) {
    lcgp_fill::PlanParams pp;
    pp.nb = nb; pp.q = q;
    pp.ob = sc.outer_blocks < 1 ? (dtype == LCGP_F32 ? 8 : 4) : sc.outer_blocks;
    pp.syrk_small_tiles = sc.syrk_small_tiles; pp.fill_leaf = sc.fill_leaf; pp.fill_step = sc.fill_step;
    pp.leaf_in_wide = sc.leaf_in_wide;
    bool prog = false;
    if (with_inverse && sc.progressive_tiles > 0 && pp.ob >= 2 && (pp.ob & (pp.ob - 1)) == 0) {
        const int nb2 = nb / 2;
        prog = (long long)q * (nb2 * (nb2 + 1) / 2) <= sc.progressive_tiles;
    }
    pp.progressive = prog;
    pp.far_rides = !(pp.progressive && sc.progressive_far == 0);
    pp.with_dupd = nb <= sc.progressive_lauum;
    pp.pair_tiles = sc.pair_tiles;
    if (inverse_done) *inverse_done = pp.progressive ? (pp.with_dupd ? 2 : 1) : 0;
    return pp;
}

// builds the plan into `out` (NULL: only the size is computed) or into `vec` (resized); returns the bytes, 0 on failure
inline size_t make_plan(int dtype, int n, int q, bool with_inverse, const lcgp_sched& sc, void* out,
                        std::vector<char>* vec = nullptr) {
    const int npad = round_up(n, 2 * TS), nb = npad / TS;
    int inverse_done = 0;
    lcgp_fill::Planner plan(plan_params(dtype, nb, q, with_inverse, sc, &inverse_done));
    plan.run();
    if (plan.failed) { bad("internal: the filler queue did not drain"); return 0; }
    PlanHeader h;
    memset(&h, 0, sizeof(h));
    h.magic = PLAN_MAGIC; h.version = LCGP_VERSION;
    h.dtype = dtype; h.n = n; h.nb = nb; h.q = q; h.with_inverse = with_inverse ? 1 : 0;
    h.nlaunch = (int)plan.launches.size();
    h.inverse_done = inverse_done;
    h.sched = sc;
    h.off_launch = (sizeof(PlanHeader) + 255) & ~size_t(255);
    h.bytes = (h.off_launch + sizeof(lcgp_fill::Launch) * h.nlaunch + 255) & ~size_t(255);
    if (vec) { vec->resize(h.bytes); out = vec->data(); }
    if (out) {
        memset(out, 0, h.bytes);
        memcpy(out, &h, sizeof(h));
        memcpy((char*)out + h.off_launch, plan.launches.data(), sizeof(lcgp_fill::Launch) * h.nlaunch);
    }
    return h.bytes;
}

inline int check_plan(const void* plan_host, int dtype, int n, int q, bool with_inverse) {
    const PlanHeader* h = (const PlanHeader*)plan_host;
    if (h->magic != PLAN_MAGIC || h->version != LCGP_VERSION) return bad("plan: not a plan of this library version");
    if (h->dtype != dtype || h->n != n || h->q != q || h->with_inverse != (with_inverse ? 1 : 0))
        return bad("plan: built for another (dtype, n, q_local, with_inverse)");
    return 0;
}

// Two-level right-looking Cholesky.  Outer panels of `ob` 64-blocks: inside a panel every 64-column step is ONE launch
// (chain_step_kernel) that only touches the panel's block column and the rest of the panel; the trailing matrix is read
// and written once per outer panel with K = 64 ob.  The trailing update of panel J is split by columns into one wide
// launch (the columns of panel J+1 and as many more as do not fit below) and its right-most columns, which the chain
// launches of panel J+1 carry as filler tiles -- the chain leaves >= 97 % of the CUs idle, and a second HIP stream
// cannot fill them on this platform.  The launch sequence is PLANNED first (fill_sched.h: Planner, host
// only, replayed on the CPU by tests/test_fill_sched.py through tests/native/dump_plan.cpp) -- by the caller, once
// (lcgp_plan_build), or here per call when no plan is passed -- and then enqueued launch by launch.
template <typename T>
int do_potrf(hipStream_t st, const Ws& w, const lcgp_sched& sc, bool stats_zeroed = false, bool with_inverse = false,
             int* inverse_done = nullptr /* 0 = nothing, 1 = L^-1, 2 = L^-1 and A^-1 */, const void* plan_host = nullptr) {
    T* M = (T*)(w.base + w.off_M);
    T* W = (T*)(w.base + w.off_W);
    double* logdet = (double*)(w.base + w.off_logdet);
    int* info = (int*)(w.base + w.off_info);
    if (!stats_zeroed) {       // (the NLL path's kernel-build launch has done it)
        hipLaunchKernelGGL(zero_stats_kernel, dim3((w.q + 63) / 64), dim3(64), 0, st, logdet, info, w.q);
        CHECK_LAUNCH("zero_stats");
    }
    const int dtype = sizeof(T) == 4 ? LCGP_F32 : LCGP_F64;
    std::vector<char> local;
    if (!plan_host) {
        if (!make_plan(dtype, w.n, w.q, with_inverse, sc, nullptr, &local)) return -1;
        plan_host = local.data();
    }
    const PlanHeader* h = (const PlanHeader*)plan_host;
    if (inverse_done) *inverse_done = h->inverse_done;
    const lcgp_fill::Launch* launches = (const lcgp_fill::Launch*)((const char*)plan_host + h->off_launch);
    const int nlaunch = h->nlaunch;
    bool first_trail = true;
    for (int li = 0; li < nlaunch; ++li) {
        const lcgp_fill::Launch& l = launches[li];
        FillSet fs = l.fs;
        fs.M = w.base + w.off_M; fs.W = w.base + w.off_W; fs.V = w.base + w.off_V;
        fs.mat = w.mat; fs.npad = w.npad; fs.nb = w.nb; fs.q = w.q;
        int rc = 0;
        switch (l.kind) {
            case lcgp_fill::L_LEAF:
                if (fs.nblk > 0)
                    hipLaunchKernelGGL((leaf_fill_kernel<T>), dim3(w.q + fs.nblk), dim3(256), 0, st, M, W, w.mat, w.npad, l.J,
                                       logdet, info, w.q, fs);
                else
                    hipLaunchKernelGGL((leaf_kernel<T>), dim3(w.q), dim3(256), 0, st, M, W, w.mat, w.npad, l.J, logdet, info);
                CHECK_LAUNCH("leaf_kernel");
                break;
            case lcgp_fill::L_STEP: {
                StepArgs a;
                a.M = M; a.W = W; a.mat = w.mat; a.npad = w.npad; a.nb = w.nb;
                a.c = l.c; a.J = l.J; a.pe = l.pe; a.q = w.q;
                a.diag_end = l.diag_end; a.has_special = l.has_special; a.n_trmm = l.n_trmm; a.n_upd = l.n_upd;
                a.trmm_r0 = l.c + 1;
                a.upd_r0 = l.c + 2;
                a.logdet = logdet; a.info = info;
                a.fs = fs;
                const long nblk = (long)(a.n_trmm + a.n_upd) * w.q + fs.nblk;
                hipLaunchKernelGGL((chain_step_kernel<T>), dim3((unsigned)nblk), dim3(256), 0, st, a);
                CHECK_LAUNCH("chain_step_kernel");
                break;
            }
            case lcgp_fill::L_TRAIL:
                // (the first trailing update -- the widest -- leaves the clock words of lcgp_lauum_clock; a later A^-1 launch
                // overwrites them)
                rc = potrf_trailing<T>(st, w, l.J, l.pe, l.c_lo, l.c_hi, l.tiles128 != 0, l.with_leaf != 0,
                                       first_trail ? (unsigned long long*)(w.base + w.off_clock) : nullptr);
                first_trail = false;
                break;
            default:
                rc = launch_fill<T>(st, fs);
        }
        if (rc) return rc;
    }
    return 0;
}

// With few components in flight the 128x128 launches of the inverse are bounded by their LONGEST tile (one tile with
// K = 4096 keeps a CU busy for ~0.5 ms while the rest of the chip idles): below a threshold of 128-tiles per launch
// the same products run on 64x64 tiles (4x more, 4x shorter tiles).
inline bool use_small_tiles(const Ws& w, int threshold) {
    const int nb2 = w.nb / 2;
    return (long long)w.q * (nb2 * (nb2 + 1) / 2) < threshold;
}

template <typename T, int TM>
int trtri_level(hipStream_t st, const Ws& w, int mb) {      // one level: pairs of blocks of mb tiles (TM units)
    GemmArgs g;
    g.sA = g.sB = g.sC = w.mat; g.ldA = g.ldB = g.ldC = w.npad; g.p2 = g.p3 = 0;
    const int nbt = w.npad / TM;
    g.nb = nbt;
    const int pairs = (nbt + 2 * mb - 1) / (2 * mb);
    g.p0 = mb; g.p1 = pairs;
    g.A = (T*)(w.base + w.off_M); g.B = (T*)(w.base + w.off_W); g.C = (T*)(w.base + w.off_V);
    int rc = launch_gemm<T, OP_TRTRI_T, TM>(st, g, pairs * mb * mb, w.q);
    if (rc) return rc;
    g.A = (T*)(w.base + w.off_W); g.B = (T*)(w.base + w.off_V); g.C = (T*)(w.base + w.off_W);
    return launch_gemm<T, OP_TRTRI_W, TM>(st, g, pairs * mb * mb, w.q);
}

template <typename T>
int do_trtri(hipStream_t st, const Ws& w, const lcgp_sched& sc) {
    const bool all_small = use_small_tiles(w, sc.trtri_small_tiles);
    // levels in 64-block units: mb64 = 1 joins pairs of 64-blocks (always 64x64 tiles); a further level works on 128x128
    // tiles unless the whole inverse or this level is too small to fill the chip with them
    for (int mb64 = 1; mb64 < w.nb; mb64 *= 2) {
        bool small = all_small || mb64 == 1;
        if (!small) {
            const int mb = mb64 / 2, nbt = w.npad / 128;
            const long long tiles = (long long)((nbt + 2 * mb - 1) / (2 * mb)) * mb * mb * w.q;
            small = tiles < sc.trtri_level_small;
        }
        const int rc = small ? trtri_level<T, 64>(st, w, mb64) : trtri_level<T, 128>(st, w, mb64 / 2);
        if (rc) return rc;
    }
    return 0;
}

template <typename T>
int do_lauum(hipStream_t st, const Ws& w, const lcgp_sched& sc, bool* z_partials = nullptr) {
    // z_partials: the caller also wants z = A^-1 b (b in the workspace).  On 128-tiles the launch leaves the per-tile
    // partial products in the workspace (epilogue of gemm_body) and sets *z_partials; on 64-tiles it does not.
    GemmArgs g;
    g.sA = g.sB = g.sC = w.mat; g.ldA = g.ldB = g.ldC = w.npad; g.p0 = g.p1 = g.p2 = g.p3 = 0;
    g.A = (T*)(w.base + w.off_W); g.B = g.A; g.C = (T*)(w.base + w.off_V);
    if (z_partials) *z_partials = false;
    g.clk = (unsigned long long*)(w.base + w.off_clock);      // (either tile size stamps its first block: lcgp_lauum_clock)
    if (use_small_tiles(w, sc.lauum_small_tiles)) {
        g.nb = w.nb;
        return launch_gemm<T, OP_LAUUM, 64>(st, g, w.nb * (w.nb + 1) / 2, w.q);
    }
    const int nb2 = w.nb / 2;
    g.nb = nb2;
    // the epilogue runs in every 128-tile launch, also when only A^-1 is asked for (lcgp_lauum / lcgp_potri: b in the
    // workspace may then be stale and the partials are never read): one launch shape to measure and to maintain
    g.bvec = w.base + w.off_b;
    g.part = (double*)(w.base + w.off_part);
    if (z_partials) *z_partials = true;
    return launch_gemm<T, OP_LAUUM, 128>(st, g, nb2 * (nb2 + 1) / 2, w.q);
}

template <typename T>
int do_potri(hipStream_t st, const Ws& w, const lcgp_sched& sc, bool* z_partials = nullptr) {
    int rc = do_trtri<T>(st, w, sc);
    if (rc) return rc;
    return do_lauum<T>(st, w, sc, z_partials);
}

template <typename T, int DD>
void launch_grad(hipStream_t st, const Ws& w, const void* x, const void* sr, const double* theta, const void* Y,
                 double* out) {
    // (float32: the noise gradient comes from c = (C o s s^T) z after this launch, so no extra blocks here)
    auto go = [&](auto kern) {
        hipLaunchKernelGGL((grad_kernel<T, DD, decltype(kern)::value>), dim3(w.ntile_lower + (sizeof(T) == 4 ? 0 : w.p), w.q),
                           dim3(256), 0, st, (const T*)(w.base + w.off_V), w.mat, w.n, w.npad, w.d, w.p, (const T*)x, (const T*)sr,
                           (const T*)(w.base + w.off_z), theta, (double*)(w.base + w.off_part), w.ntile_lower, (const T*)Y,
                           (const T*)(w.base + w.off_b), out, (double*)(w.base + w.off_cpart));
    };
    if (w.kern == 0) go(std::integral_constant<int, 0>{}); else go(std::integral_constant<int, 1>{});
}

template <typename T>
int do_nll_grad(hipStream_t st, const Ws& w, const lcgp_sched& sc, const void* x, const void* Y, const void* sr,
                const double* theta, double* out, const void* plan_host) {
    int rc = do_build<T>(st, w, x, sr, theta, Y);
    if (rc) return rc;
    T* b = (T*)(w.base + w.off_b);
    T* z = (T*)(w.base + w.off_z);
    int inverse_done = 0;          // what the progressive inverse has left behind the factorisation: 1 = L^-1, 2 = and A^-1
    rc = do_potrf<T>(st, w, sc, true, true, &inverse_done, plan_host);
    if (rc) return rc;
    bool z_partials = false;       // z = A^-1 b: per-tile partials from the 128-tile LAUUM's epilogue, or a pass of its own
    if (inverse_done == 0) rc = do_potri<T>(st, w, sc, &z_partials);
    else if (inverse_done == 1) rc = do_lauum<T>(st, w, sc, &z_partials);
    if (rc) return rc;
    if (z_partials) {
        const int nb2 = w.nb / 2;
        hipLaunchKernelGGL((symv_reduce_kernel<T, 128>), dim3(nb2, w.q), dim3(256), 0, st,
                           (const double*)(w.base + w.off_part), nb2 * (nb2 + 1) / 2, w.npad, nb2, z);
    } else {
        hipLaunchKernelGGL((symv_tile_kernel<T>), dim3(w.ntile_lower, w.q), dim3(256), 0, st, (const T*)(w.base + w.off_V),
                           w.mat, w.npad, (const T*)b, (double*)(w.base + w.off_part), w.ntile_lower);
        CHECK_LAUNCH("symv_tile_kernel");
        hipLaunchKernelGGL((symv_reduce_kernel<T, TS>), dim3(w.nb, w.q), dim3(256), 0, st,
                           (const double*)(w.base + w.off_part), w.ntile_lower, w.npad, w.nb, z);
    }
    CHECK_LAUNCH("symv_reduce_kernel");
    if (w.d <= 2) launch_grad<T, 2>(st, w, x, sr, theta, Y, out);
    else if (w.d <= 4) launch_grad<T, 4>(st, w, x, sr, theta, Y, out);
    else if (w.d <= 6) launch_grad<T, 6>(st, w, x, sr, theta, Y, out);
    else if (w.d <= 10) launch_grad<T, 10>(st, w, x, sr, theta, Y, out);
    else if (w.d <= 16) launch_grad<T, 16>(st, w, x, sr, theta, Y, out);
    else if (w.d <= DMAX) launch_grad<T, DMAX>(st, w, x, sr, theta, Y, out);
    else {
        auto go = [&](auto kern) {
            hipLaunchKernelGGL((grad_kernel_wide<T, decltype(kern)::value>), dim3(w.ntile_lower + (sizeof(T) == 4 ? 0 : w.p), w.q),
                               dim3(256), 0, st, (const T*)(w.base + w.off_V), w.mat, w.n, w.npad, w.d, w.p, (const T*)x,
                               (const T*)sr, (const T*)(w.base + w.off_z), theta, (double*)(w.base + w.off_part), w.ntile_lower,
                               (const T*)Y, (const T*)(w.base + w.off_b), out, (double*)(w.base + w.off_cpart));
        };
        if (w.kern == 0) go(std::integral_constant<int, 0>{}); else go(std::integral_constant<int, 1>{});
    }
    CHECK_LAUNCH("grad_kernel");
    const double* cvec = nullptr;
    if constexpr (sizeof(T) == 4) {
        double* c = (double*)(w.base + w.off_c);
        hipLaunchKernelGGL(cvec_reduce_kernel, dim3(w.nb, w.q), dim3(256), 0, st, (const double*)(w.base + w.off_cpart),
                           w.ntile_lower, w.npad, w.nb, c);
        CHECK_LAUNCH("cvec_reduce_kernel");
        hipLaunchKernelGGL((gsig_c_kernel<T>), dim3(w.p, w.q), dim3(256), 0, st, w.n, w.npad, w.d, w.p, (const T*)Y,
                           (const double*)c, theta, out);
        CHECK_LAUNCH("gsig_c_kernel");
        cvec = c;
    }
    hipLaunchKernelGGL((finalize_kernel<T>), dim3(w.q), dim3(256), 0, st, w.n, w.npad, w.d, w.p, w.ntile_lower,
                       (const T*)Y, (const T*)b, (const T*)z, (const double*)(w.base + w.off_part),
                       (const double*)(w.base + w.off_logdet), (const int*)(w.base + w.off_info), theta, out, cvec);
    CHECK_LAUNCH("finalize_kernel");
    return 0;
}

// The rank's share of the reduced vector, assembled on the device (one workgroup, fixed summation order):
//   vec = [ sum_k (half_logdet_k - quad_k / (2 D_k)) | sum_k info_k | g_ell (q x d) | g_scale (q) | g_nug (q) | g_sigma (p) | guard ]
// component-local slots are written at the GLOBAL component index comp[i]; g_sigma_a = sum_k psi_k[a] gsig_k[a] / (2 D_k).
// guard: a word of the caller's (a hash of the parameter vector the rank evaluated) that travels through the all-reduce in
// the last slot, so that ranks which have drifted apart are detected at the first evaluation instead of diverging.
__global__ __launch_bounds__(256) void pack_partial_kernel(int d, int p, int q_local, int q_total,
                                                           const int* __restrict__ comp, const double* __restrict__ theta,
                                                           const double* __restrict__ out, const double* __restrict__ guard,
                                                           double* __restrict__ vec) {
    const int tid = threadIdx.x;
    const int tw = d + 3 + p, ow = d + 5 + p;
    const int off_s = 2 + q_total * d, off_n = off_s + q_total, off_g = off_n + q_total;
    for (int e = tid; e < off_g; e += 256) vec[e] = 0.0;
    __syncthreads();
    if (tid == 0) {
        vec[off_g + p] = guard ? guard[0] : 0.0;
        double v = 0.0, bad_sum = 0.0;
        for (int i = 0; i < q_local; ++i) {
            const double* o = out + (size_t)i * ow;
            v += o[0] - o[1] / (2.0 * theta[(size_t)i * tw + d + 2]);
            bad_sum += o[2];
        }
        vec[0] = v;
        vec[1] = bad_sum;
    }
    for (int e = tid; e < q_local * (d + 2); e += 256) {
        const int i = e / (d + 2), j = e - i * (d + 2);
        const int k = comp[i];
        const double val = out[(size_t)i * ow + 3 + j];
        if (j < d) vec[2 + k * d + j] = val;
        else if (j == d) vec[off_s + k] = val;
        else vec[off_n + k] = val;
    }
    for (int a = tid; a < p; a += 256) {
        double s = 0.0;
        for (int i = 0; i < q_local; ++i) {
            const double* th = theta + (size_t)i * tw;
            s += 0.5 * th[d + 3 + a] * out[(size_t)i * ow + 5 + d + a] / th[d + 2];
        }
        vec[off_g + a] = s;
    }
}

int check_common(int dtype, int n, int d, int p, int q, int kernel_id = 0) {
    if (dtype != LCGP_F64 && dtype != LCGP_F32) return bad("dtype must be 0 (f64) or 1 (f32)");
    if (kernel_id != LCGP_KERNEL_MATERN32 && kernel_id != LCGP_KERNEL_SE) return bad("kernel_id must be 0 (Matern-3/2) or 1 (squared exponential)");
    if (n < 1) return bad("n < 1");
    if (d < 1 || d > DWIDE) return bad("d must be in [1, 126]");
    if (p < 1) return bad("p < 1");
    if (q < 1 || q > 65535) return bad("q_local must be in [1, 65535]");
    return 0;
}

inline int resolve_sched(const lcgp_sched* in, lcgp_sched& out) {
    out = in ? *in : default_sched();
    return check_sched(out);
}

template <typename T>
int do_matern(hipStream_t st, int kern, int n1, int n2, int d, const void* x1, const void* x2, const ThetaArg& th, int same,
              void* out) {
    dim3 grid((n2 + TS - 1) / TS, (n1 + TS - 1) / TS);
    if (kern == 0)
        hipLaunchKernelGGL((cross_kernel<T, 0>), grid, dim3(256), 0, st, (T*)out, n2, n1, n2, d, (const T*)x1, (const T*)x2, th,
                           (const double*)nullptr, same, (const T*)nullptr, n1, n2, 0, (size_t)0);
    else
        hipLaunchKernelGGL((cross_kernel<T, 1>), grid, dim3(256), 0, st, (T*)out, n2, n1, n2, d, (const T*)x1, (const T*)x2, th,
                           (const double*)nullptr, same, (const T*)nullptr, n1, n2, 0, (size_t)0);
    CHECK_LAUNCH("cross_kernel");
    return 0;
}

// rows of the padded cross-covariance block: whole 128-tiles when there are at least 128 new inputs (the MFMA products
// then run on the 128x128 8-wave tile kernel), whole 64-tiles otherwise
inline int predict_pad(int n0) { return n0 >= 128 ? round_up(n0, 2 * TS) : round_up(n0, TS); }

// K6, all local components in every launch:  X_k = c0k o sr^T (one launch), U_k = X_k W_k^T (one launch of the tile
// kernel, k tiles only up to the diagonal: W is lower triangular), then the row reductions.
template <typename T>
int do_predict(hipStream_t st, const Ws& w, const void* x, const void* sr, const double* theta, int n0, const void* x0,
               int same, void* scratch, double* ghat, double* gvar, int ldo) {
    const int n0pad = predict_pad(n0);
    const size_t slab = (size_t)n0pad * w.npad;
    T* X = (T*)scratch;                 // q slabs n0pad x npad : c0k o sr^T (zero padded)
    T* U = X + slab * w.q;              // q slabs n0pad x npad : X W^T = (L^-1 X^T)^T
    const int tw = w.d + 3 + w.p;
    ThetaArg dummy;
    memset(&dummy, 0, sizeof(dummy));
    dim3 grid(w.nb, n0pad / TS, w.q);
    if (w.kern == 0)
        hipLaunchKernelGGL((cross_kernel<T, 0>), grid, dim3(256), 0, st, X, w.npad, n0, w.n, w.d, (const T*)x0, (const T*)x, dummy,
                           theta, same, (const T*)sr, n0pad, w.npad, tw, slab);
    else
        hipLaunchKernelGGL((cross_kernel<T, 1>), grid, dim3(256), 0, st, X, w.npad, n0, w.n, w.d, (const T*)x0, (const T*)x, dummy,
                           theta, same, (const T*)sr, n0pad, w.npad, tw, slab);
    CHECK_LAUNCH("cross_kernel");
    GemmArgs g;
    g.A = X; g.B = (const T*)(w.base + w.off_W); g.C = U;
    g.sA = slab; g.sB = w.mat; g.sC = slab; g.ldA = g.ldB = g.ldC = w.npad; g.p1 = g.p2 = g.p3 = 0;
    int rc;
    if (n0pad % (2 * TS) == 0) {
        g.nb = w.nb / 2; g.p0 = n0pad / (2 * TS);
        rc = launch_gemm<T, OP_PRED_U, 128>(st, g, g.p0 * g.nb, w.q);
    } else {
        g.nb = w.nb; g.p0 = n0pad / TS;
        rc = launch_gemm<T, OP_PRED_U, 64>(st, g, g.p0 * g.nb, w.q);
    }
    if (rc) return rc;
    hipLaunchKernelGGL((pred_reduce_kernel<T>), dim3(n0, w.q), dim3(64), 0, st, (const T*)X, (const T*)U, slab, w.npad, w.n,
                       (const T*)(w.base + w.off_z), w.npad, theta, tw, w.d, ldo, ghat, gvar);
    CHECK_LAUNCH("pred_reduce_kernel");
    return 0;
}

}  // namespace

// ---------------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------------
#ifndef LCGP_SRC_HASH
#define LCGP_SRC_HASH "unknown"
#endif

extern "C" {

int lcgp_version(void) { return LCGP_VERSION; }
const char* lcgp_source_hash(void) { return LCGP_SRC_HASH; }
const char* lcgp_last_error(void) { return g_err; }
int lcgp_theta_width(int d, int p) { return d + 3 + p; }
int lcgp_out_width(int d, int p) { return d + 5 + p; }
int lcgp_partial_width(int d, int p, int q_total) { return 3 + q_total * d + 2 * q_total + p; }

int lcgp_sched_default(lcgp_sched* sched) {
    if (!sched) return bad("sched is NULL");
    *sched = default_sched();
    return 0;
}

int lcgp_workspace_bytes(int dtype, int n, int d, int p, int q_local, size_t* bytes) {
    int rc = check_common(dtype, n, d, p, q_local);
    if (rc) return rc;
    if (!bytes) return bad("bytes is NULL");
    *bytes = carve(dtype, n, d, p, q_local, nullptr).total;
    return 0;
}

int lcgp_predict_scratch_bytes(int dtype, int n, int q_local, int n0, size_t* bytes) {
    if (dtype != LCGP_F64 && dtype != LCGP_F32) return bad("dtype must be 0 (f64) or 1 (f32)");
    if (n < 1 || n0 < 1 || q_local < 1) return bad("n, n0, q_local must be >= 1");
    if (!bytes) return bad("bytes is NULL");
    const size_t npad = round_up(n, 2 * TS), n0pad = predict_pad(n0);
    *bytes = 2 * (size_t)q_local * n0pad * npad * (dtype == LCGP_F64 ? 8 : 4);
    return 0;
}

int lcgp_covmat(void* stream, int dtype, int kernel_id, int n1, int n2, int d, const void* x1, const void* x2, const double* ell,
                double scale, double nug, int same, void* out) {
    if (dtype != LCGP_F64 && dtype != LCGP_F32) return bad("dtype must be 0 (f64) or 1 (f32)");
    if (kernel_id != LCGP_KERNEL_MATERN32 && kernel_id != LCGP_KERNEL_SE) return bad("kernel_id must be 0 (Matern-3/2) or 1 (squared exponential)");
    if (n1 < 1 || n2 < 1) return bad("n1/n2 < 1");
    if (d < 1 || d > DWIDE) return bad("d must be in [1, 126]");
    if (!x1 || !x2 || !ell || !out) return bad("NULL pointer");
    ThetaArg th;
    memset(&th, 0, sizeof(th));
    for (int j = 0; j < d; ++j) th.v[j] = ell[j];
    th.v[d] = scale;
    th.v[d + 1] = nug;
    hipStream_t st = (hipStream_t)stream;
    return dtype == LCGP_F64 ? do_matern<double>(st, kernel_id, n1, n2, d, x1, x2, th, same, out)
                             : do_matern<float>(st, kernel_id, n1, n2, d, x1, x2, th, same, out);
}
int lcgp_matern32(void* stream, int dtype, int n1, int n2, int d, const void* x1, const void* x2, const double* ell,
                  double scale, double nug, int same, void* out) {
    return lcgp_covmat(stream, dtype, LCGP_KERNEL_MATERN32, n1, n2, d, x1, x2, ell, scale, nug, same, out);
}

int lcgp_kernel_build(void* stream, int dtype, int kernel_id, int n, int d, int p, int q_local, const void* x, const void* sr,
                      const double* theta, void* workspace) {
    int rc = check_common(dtype, n, d, p, q_local, kernel_id);
    if (rc) return rc;
    if (!x || !theta || !workspace) return bad("NULL pointer");
    Ws w = carve(dtype, n, d, p, q_local, workspace);
    w.kern = kernel_id;
    return dtype == LCGP_F64 ? do_build<double>((hipStream_t)stream, w, x, sr, theta)
                             : do_build<float>((hipStream_t)stream, w, x, sr, theta);
}

int lcgp_potrf_logdet(void* stream, int dtype, int n, int d, int p, int q_local, void* workspace, double* half_logdet,
                      int* info, const lcgp_sched* sched, const void* plan) {
    const void* plan_host = plan;
    int rc = check_common(dtype, n, d, p, q_local);
    if (rc) return rc;
    if (!workspace) return bad("NULL workspace");
    lcgp_sched sc;
    if ((rc = resolve_sched(sched, sc))) return rc;
    if (plan_host && (rc = check_plan(plan_host, dtype, n, q_local, false))) return rc;
    Ws w = carve(dtype, n, d, p, q_local, workspace);
    hipStream_t st = (hipStream_t)stream;
    rc = dtype == LCGP_F64 ? do_potrf<double>(st, w, sc, false, false, nullptr, plan_host)
                           : do_potrf<float>(st, w, sc, false, false, nullptr, plan_host);
    if (rc) return rc;
    if (half_logdet || info) {
        hipLaunchKernelGGL(copy_stats_kernel, dim3((q_local + 63) / 64), dim3(64), 0, st,
                           (const double*)(w.base + w.off_logdet), (const int*)(w.base + w.off_info), half_logdet, info,
                           q_local);
        CHECK_LAUNCH("copy_stats");
    }
    return 0;
}

int lcgp_potri(void* stream, int dtype, int n, int d, int p, int q_local, void* workspace, const lcgp_sched* sched) {
    int rc = check_common(dtype, n, d, p, q_local);
    if (rc) return rc;
    if (!workspace) return bad("NULL workspace");
    lcgp_sched sc;
    if ((rc = resolve_sched(sched, sc))) return rc;
    Ws w = carve(dtype, n, d, p, q_local, workspace);
    return dtype == LCGP_F64 ? do_potri<double>((hipStream_t)stream, w, sc) : do_potri<float>((hipStream_t)stream, w, sc);
}

int lcgp_trtri(void* stream, int dtype, int n, int d, int p, int q_local, void* workspace, const lcgp_sched* sched) {
    int rc = check_common(dtype, n, d, p, q_local);
    if (rc) return rc;
    if (!workspace) return bad("NULL workspace");
    lcgp_sched sc;
    if ((rc = resolve_sched(sched, sc))) return rc;
    Ws w = carve(dtype, n, d, p, q_local, workspace);
    return dtype == LCGP_F64 ? do_trtri<double>((hipStream_t)stream, w, sc) : do_trtri<float>((hipStream_t)stream, w, sc);
}

int lcgp_lauum(void* stream, int dtype, int n, int d, int p, int q_local, void* workspace, const lcgp_sched* sched) {
    int rc = check_common(dtype, n, d, p, q_local);
    if (rc) return rc;
    if (!workspace) return bad("NULL workspace");
    lcgp_sched sc;
    if ((rc = resolve_sched(sched, sc))) return rc;
    Ws w = carve(dtype, n, d, p, q_local, workspace);
    return dtype == LCGP_F64 ? do_lauum<double>((hipStream_t)stream, w, sc) : do_lauum<float>((hipStream_t)stream, w, sc);
}

int lcgp_lauum_clock(void* stream, int dtype, int n, int d, int p, int q_local, const void* workspace,
                     unsigned long long* out) {
    int rc = check_common(dtype, n, d, p, q_local);
    if (rc) return rc;
    if (!workspace || !out) return bad("NULL pointer");
    Ws w = carve(dtype, n, d, p, q_local, (void*)workspace);
    hipError_t e = hipMemcpyAsync(out, w.base + w.off_clock, 2 * sizeof(unsigned long long), hipMemcpyDeviceToDevice,
                                  (hipStream_t)stream);
    if (e != hipSuccess) return fail("hipMemcpyAsync", e);
    // read and clear: a second read without a stamped launch in between returns zeros (the 64x64-tile form of the launch,
    // which small problems and single components use, does not stamp), and so does a read after one call on a fresh workspace
    e = hipMemsetAsync(w.base + w.off_clock, 0, 2 * sizeof(unsigned long long), (hipStream_t)stream);
    if (e != hipSuccess) return fail("hipMemsetAsync", e);
    return 0;
}

int lcgp_fetch_matrix(void* stream, int dtype, int n, int d, int p, int q_local, const void* workspace, int which, int k,
                      void* out) {
    int rc = check_common(dtype, n, d, p, q_local);
    if (rc) return rc;
    if (!workspace || !out) return bad("NULL pointer");
    if (which < 0 || which > 2 || k < 0 || k >= q_local) return bad("which/k out of range");
    Ws w = carve(dtype, n, d, p, q_local, (void*)workspace);
    size_t off = which == 0 ? w.off_M : (which == 1 ? w.off_W : w.off_V);
    dim3 grid((n + 255) / 256, n);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == LCGP_F64)
        hipLaunchKernelGGL((fetch_kernel<double>), grid, dim3(256), 0, st,
                           (const double*)(w.base + off) + (size_t)k * w.mat, w.npad, n, (double*)out);
    else
        hipLaunchKernelGGL((fetch_kernel<float>), grid, dim3(256), 0, st,
                           (const float*)(w.base + off) + (size_t)k * w.mat, w.npad, n, (float*)out);
    CHECK_LAUNCH("fetch_kernel");
    return 0;
}

int lcgp_fetch_vector(void* stream, int dtype, int n, int d, int p, int q_local, const void* workspace, int which, int k,
                      void* out) {
    int rc = check_common(dtype, n, d, p, q_local);
    if (rc) return rc;
    if (!workspace || !out) return bad("NULL pointer");
    if (which < 0 || which > 1 || k < 0 || k >= q_local) return bad("which/k out of range");
    Ws w = carve(dtype, n, d, p, q_local, (void*)workspace);
    const char* src = w.base + (which == 0 ? w.off_b : w.off_z) + (size_t)k * w.npad * w.esz;
    hipError_t e = hipMemcpyAsync(out, src, (size_t)n * w.esz, hipMemcpyDeviceToDevice, (hipStream_t)stream);
    if (e != hipSuccess) return fail("hipMemcpyAsync", e);
    return 0;
}

int lcgp_nll_grad(void* stream, int dtype, int kernel_id, int n, int d, int p, int q_local, const void* x, const void* Y,
                  const void* sr, const double* theta, void* workspace, double* out, const lcgp_sched* sched,
                  const void* plan) {
    const void* plan_host = plan;
    int rc = check_common(dtype, n, d, p, q_local, kernel_id);
    if (rc) return rc;
    if (!x || !Y || !theta || !workspace || !out) return bad("NULL pointer");
    lcgp_sched sc;
    if (plan_host) {
        // the plan carries the schedule it was built for (the stages behind the factorisation read their thresholds there)
        if ((rc = check_plan(plan_host, dtype, n, q_local, true))) return rc;
        sc = ((const PlanHeader*)plan_host)->sched;
    } else if ((rc = resolve_sched(sched, sc))) {
        return rc;
    }
    Ws w = carve(dtype, n, d, p, q_local, workspace);
    w.kern = kernel_id;
    return dtype == LCGP_F64 ? do_nll_grad<double>((hipStream_t)stream, w, sc, x, Y, sr, theta, out, plan_host)
                             : do_nll_grad<float>((hipStream_t)stream, w, sc, x, Y, sr, theta, out, plan_host);
}

int lcgp_plan_bytes(int dtype, int n, int q_local, int with_inverse, const lcgp_sched* sched, size_t* bytes) {
    int rc = check_common(dtype, n, 1, 1, q_local);
    if (rc) return rc;
    if (!bytes) return bad("bytes is NULL");
    lcgp_sched sc;
    if ((rc = resolve_sched(sched, sc))) return rc;
    *bytes = make_plan(dtype, n, q_local, with_inverse != 0, sc, nullptr);
    return *bytes ? 0 : -1;
}

int lcgp_plan_build(int dtype, int n, int q_local, int with_inverse, const lcgp_sched* sched, void* plan, size_t bytes) {
    int rc = check_common(dtype, n, 1, 1, q_local);
    if (rc) return rc;
    if (!plan) return bad("plan is NULL");
    lcgp_sched sc;
    if ((rc = resolve_sched(sched, sc))) return rc;
    const size_t need = make_plan(dtype, n, q_local, with_inverse != 0, sc, nullptr);
    if (!need) return -1;
    if (bytes < need) return bad("plan buffer too small (lcgp_plan_bytes)");
    return make_plan(dtype, n, q_local, with_inverse != 0, sc, plan) ? 0 : -1;
}

int lcgp_plan_info(const void* plan, int* nlaunch, int* inverse_done) {
    if (!plan) return bad("plan is NULL");
    const PlanHeader* h = (const PlanHeader*)plan;
    if (h->magic != PLAN_MAGIC || h->version != LCGP_VERSION) return bad("plan: not a plan of this library version");
    if (nlaunch) *nlaunch = h->nlaunch;
    if (inverse_done) *inverse_done = h->inverse_done;
    return 0;
}

int lcgp_pack_partial(void* stream, int d, int p, int q_local, int q_total, const int* comp, const double* theta,
                      const double* out, const double* guard, double* vec) {
    if (d < 1 || d > DWIDE || p < 1) return bad("d must be in [1, 126], p >= 1");
    if (q_local < 0 || q_total < 1 || q_local > q_total) return bad("need 0 <= q_local <= q_total, q_total >= 1");
    if (!vec || (q_local > 0 && (!comp || !theta || !out))) return bad("NULL pointer");
    hipLaunchKernelGGL(pack_partial_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, d, p, q_local, q_total, comp, theta,
                       out, guard, vec);
    CHECK_LAUNCH("pack_partial_kernel");
    return 0;
}

int lcgp_predict(void* stream, int dtype, int kernel_id, int n, int d, int p, int q_local, const void* x, const void* sr,
                 const double* theta, const void* workspace, int n0, const void* x0, int same, void* scratch,
                 double* ghat, double* gvar, int out_stride) {
    int rc = check_common(dtype, n, d, p, q_local, kernel_id);
    if (rc) return rc;
    if (n0 < 1) return bad("n0 < 1");
    if (!x || !theta || !workspace || !x0 || !scratch || !ghat || !gvar) return bad("NULL pointer");
    if (out_stride != 0 && out_stride < n0) return bad("out_stride must be 0 (= n0) or >= n0");
    const int ldo = out_stride ? out_stride : n0;
    Ws w = carve(dtype, n, d, p, q_local, (void*)workspace);
    w.kern = kernel_id;
    hipStream_t st = (hipStream_t)stream;
    return dtype == LCGP_F64 ? do_predict<double>(st, w, x, sr, theta, n0, x0, same, scratch, ghat, gvar, ldo)
                             : do_predict<float>(st, w, x, sr, theta, n0, x0, same, scratch, ghat, gvar, ldo);
}

}  // extern "C"
