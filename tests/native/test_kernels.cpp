// Native parity driver for liblcgp_hip.so: runs every C-ABI entry point on the GPU and checks it against a
// straightforward host computation of the same formulas.  No torch, no Python: meant
// for fast iterations on the GPU box.   usage: test_kernels [n ...]
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <cstring>
#include <vector>

#include "../../include/lcgp_hip.h"

#define HIPCHK(x)                                                                       \
    do {                                                                                \
        hipError_t e = (x);                                                             \
        if (e != hipSuccess) {                                                          \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); \
            exit(2);                                                                    \
        }                                                                               \
    } while (0)
#define LCHK(x)                                                         \
    do {                                                                \
        int rc = (x);                                                   \
        if (rc) {                                                       \
            printf("lcgp error %d (%s) at %s:%d\n", rc, lcgp_last_error(), __FILE__, __LINE__); \
            exit(3);                                                    \
        }                                                               \
    } while (0)

typedef std::vector<double> vec;
static int g_fail = 0;

static void report(const char* what, double err, double tol) {
    bool ok = err <= tol && std::isfinite(err);
    printf("  %-34s err %.3e (tol %.1e) %s\n", what, err, tol, ok ? "ok" : "FAIL");
    if (!ok) ++g_fail;
}

static double relmax(const vec& a, const vec& b) {
    double m = 0, s = 0;
    for (size_t i = 0; i < a.size(); ++i) { m = fmax(m, fabs(a[i] - b[i])); s = fmax(s, fabs(b[i])); }
    return m / fmax(s, 1e-300);
}

struct Problem {
    int n, d, p, q;
    vec x, Y, sr, theta;  // theta rows: ell[d], scale, nug, D, psi[p]
    bool rep;
};

static Problem make_problem(int n, int d, int p, int q, bool rep, unsigned seed) {
    std::mt19937_64 g(seed);
    std::uniform_real_distribution<double> U(0, 1);
    std::normal_distribution<double> N(0, 1);
    Problem P{n, d, p, q, {}, {}, {}, {}, rep};
    P.x.resize((size_t)n * d);
    for (auto& v : P.x) v = U(g);
    P.Y.resize((size_t)p * n);
    for (auto& v : P.Y) v = N(g);
    P.sr.resize(n);
    for (auto& v : P.sr) v = sqrt(1.0 + floor(U(g) * 5));
    int tw = lcgp_theta_width(d, p);
    P.theta.resize((size_t)q * tw);
    for (int k = 0; k < q; ++k) {
        double* t = &P.theta[(size_t)k * tw];
        for (int j = 0; j < d; ++j) t[j] = 0.3 + U(g);
        t[d] = 0.5 + U(g);
        t[d + 1] = 1e-4 + 1e-3 * U(g);
        t[d + 2] = 0.5 + 2 * U(g);
        for (int a = 0; a < p; ++a) t[d + 3 + a] = N(g);
    }
    return P;
}

// host reference ------------------------------------------------------------------------------------
static void host_c0(const Problem& P, int k, vec& c0, std::vector<vec>& S) {
    int n = P.n, d = P.d, tw = lcgp_theta_width(P.d, P.p);
    const double* t = &P.theta[(size_t)k * tw];
    c0.assign((size_t)n * n, 0);
    S.assign(d, vec((size_t)n * n));
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) {
            double poly = 1, ss = 0;
            for (int jj = 0; jj < d; ++jj) {
                double s = fabs(P.x[(size_t)i * d + jj] / t[jj] - P.x[(size_t)j * d + jj] / t[jj]);
                S[jj][(size_t)i * n + j] = s;
                poly *= 1 + s;
                ss -= s;
            }
            c0[(size_t)i * n + j] = poly * exp(ss);
        }
}

static bool host_chol(vec& a, int n) {
    for (int j = 0; j < n; ++j) {
        double s = a[(size_t)j * n + j];
        for (int k = 0; k < j; ++k) s -= a[(size_t)j * n + k] * a[(size_t)j * n + k];
        if (!(s > 0)) return false;
        double l = sqrt(s);
        a[(size_t)j * n + j] = l;
        for (int i = j + 1; i < n; ++i) {
            double t = a[(size_t)i * n + j];
            for (int k = 0; k < j; ++k) t -= a[(size_t)i * n + k] * a[(size_t)j * n + k];
            a[(size_t)i * n + j] = t / l;
        }
        for (int i = 0; i < j; ++i) a[(size_t)i * n + j] = 0;
    }
    return true;
}

static void host_trinv(const vec& L, vec& W, int n) {
    W.assign((size_t)n * n, 0);
    for (int c = 0; c < n; ++c) {
        W[(size_t)c * n + c] = 1 / L[(size_t)c * n + c];
        for (int i = c + 1; i < n; ++i) {
            double s = 0;
            for (int m = c; m < i; ++m) s += L[(size_t)i * n + m] * W[(size_t)m * n + c];
            W[(size_t)i * n + c] = -s / L[(size_t)i * n + i];
        }
    }
}

template <typename T>
static std::vector<T> cast_vec(const vec& v) { return std::vector<T>(v.begin(), v.end()); }

template <typename T>
static void run_case(int n, int d, int p, int q, bool rep, unsigned seed) {
    const int dtype = sizeof(T) == 8 ? LCGP_F64 : LCGP_F32;
    const double eps = sizeof(T) == 8 ? 1e-11 : 2e-3;
    printf("case n=%d d=%d p=%d q=%d %s %s\n", n, d, p, q, rep ? "rep" : "full", dtype == LCGP_F64 ? "f64" : "f32");
    Problem P = make_problem(n, d, p, q, rep, seed);
    const int tw = lcgp_theta_width(d, p), ow = lcgp_out_width(d, p);
    size_t wsb = 0;
    LCHK(lcgp_workspace_bytes(dtype, n, d, p, q, &wsb));
    void *ws, *dx, *dY, *dsr, *dmat;
    double *dtheta, *dout, *dld;
    int* dinfo;
    HIPCHK(hipMalloc(&ws, wsb));
    HIPCHK(hipMemset(ws, 0xff, wsb));  // poison: NaN everywhere
    std::vector<T> xs = cast_vec<T>(P.x), Ys = cast_vec<T>(P.Y), srs = cast_vec<T>(P.sr);
    HIPCHK(hipMalloc(&dx, xs.size() * sizeof(T)));
    HIPCHK(hipMalloc(&dY, Ys.size() * sizeof(T)));
    HIPCHK(hipMalloc(&dsr, srs.size() * sizeof(T)));
    HIPCHK(hipMalloc(&dmat, (size_t)n * n * sizeof(T)));
    HIPCHK(hipMalloc(&dtheta, P.theta.size() * 8));
    HIPCHK(hipMalloc(&dout, (size_t)q * ow * 8));
    HIPCHK(hipMalloc(&dld, q * 8));
    HIPCHK(hipMalloc(&dinfo, q * 4));
    HIPCHK(hipMemcpy(dx, xs.data(), xs.size() * sizeof(T), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(dY, Ys.data(), Ys.size() * sizeof(T), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(dsr, srs.data(), srs.size() * sizeof(T), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(dtheta, P.theta.data(), P.theta.size() * 8, hipMemcpyHostToDevice));
    // use the host-rounded inputs for the reference so that f32 compares like with like
    for (size_t i = 0; i < P.x.size(); ++i) P.x[i] = (double)xs[i];
    for (size_t i = 0; i < P.Y.size(); ++i) P.Y[i] = (double)Ys[i];
    for (size_t i = 0; i < P.sr.size(); ++i) P.sr[i] = rep ? (double)srs[i] : 1.0;
    const void* srp = rep ? dsr : nullptr;

    auto fetch = [&](int which, int k) {
        LCHK(lcgp_fetch_matrix(nullptr, dtype, n, d, p, q, ws, which, k, dmat));
        std::vector<T> h((size_t)n * n);
        HIPCHK(hipMemcpy(h.data(), dmat, h.size() * sizeof(T), hipMemcpyDeviceToHost));
        return vec(h.begin(), h.end());
    };

    LCHK(lcgp_kernel_build(nullptr, dtype, LCGP_KERNEL_MATERN32, n, d, p, q, dx, srp, dtheta, ws));
    HIPCHK(hipDeviceSynchronize());
    std::vector<vec> Aall(q), C0all(q);
    std::vector<std::vector<vec>> Sall(q);
    double e_build = 0;
    for (int k = 0; k < q; ++k) {
        const double* t = &P.theta[(size_t)k * tw];
        host_c0(P, k, C0all[k], Sall[k]);
        vec& A = Aall[k];
        A.assign((size_t)n * n, 0);
        double nt = t[d + 1] / (1 + t[d + 1]);
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < n; ++j) {
                double dl = i == j;
                A[(size_t)i * n + j] = dl + t[d + 2] * P.sr[i] * P.sr[j] * t[d] * ((1 - nt) * C0all[k][(size_t)i * n + j] + nt * dl);
            }
        e_build = fmax(e_build, relmax(fetch(0, k), A));
    }
    report("kernel_build A", e_build, sizeof(T) == 8 ? 1e-14 : 1e-6);

    LCHK(lcgp_potrf_logdet(nullptr, dtype, n, d, p, q, ws, dld, dinfo, nullptr, nullptr));
    HIPCHK(hipDeviceSynchronize());
    vec hld(q);
    std::vector<int> hinfo(q);
    HIPCHK(hipMemcpy(hld.data(), dld, q * 8, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(hinfo.data(), dinfo, q * 4, hipMemcpyDeviceToHost));
    std::vector<vec> Lall(q), Wall(q), Vall(q);
    double e_l = 0, e_ld = 0;
    for (int k = 0; k < q; ++k) {
        Lall[k] = Aall[k];
        if (!host_chol(Lall[k], n)) printf("host chol failed\n");
        vec g = fetch(0, k);
        for (int i = 0; i < n; ++i)
            for (int j = i + 1; j < n; ++j) g[(size_t)i * n + j] = 0;  // fetch mirrors; compare lower only
        e_l = fmax(e_l, relmax(g, Lall[k]));
        double ld = 0;
        for (int i = 0; i < n; ++i) ld += log(Lall[k][(size_t)i * n + i]);
        e_ld = fmax(e_ld, fabs(hld[k] - ld) / fabs(ld));
        if (hinfo[k]) { printf("  info[%d] = %d\n", k, hinfo[k]); ++g_fail; }
    }
    report("potrf L", e_l, eps);
    report("potrf half_logdet", e_ld, eps);

    LCHK(lcgp_potri(nullptr, dtype, n, d, p, q, ws, nullptr));
    HIPCHK(hipDeviceSynchronize());
    double e_w = 0, e_v = 0;
    for (int k = 0; k < q; ++k) {
        host_trinv(Lall[k], Wall[k], n);
        vec g = fetch(1, k);
        for (int i = 0; i < n; ++i)
            for (int j = i + 1; j < n; ++j) g[(size_t)i * n + j] = 0;
        e_w = fmax(e_w, relmax(g, Wall[k]));
        vec& V = Vall[k];
        V.assign((size_t)n * n, 0);
        for (int i = 0; i < n; ++i)
            for (int j = 0; j <= i; ++j) {
                double s = 0;
                for (int m = i; m < n; ++m) s += Wall[k][(size_t)m * n + i] * Wall[k][(size_t)m * n + j];
                V[(size_t)i * n + j] = V[(size_t)j * n + i] = s;
            }
        e_v = fmax(e_v, relmax(fetch(2, k), V));
    }
    report("potri L^-1", e_w, eps * 10);
    report("potri A^-1", e_v, eps * 10);

    // whole path
    HIPCHK(hipMemset(ws, 0xff, wsb));
    LCHK(lcgp_nll_grad(nullptr, dtype, LCGP_KERNEL_MATERN32, n, d, p, q, dx, dY, srp, dtheta, ws, dout, nullptr, nullptr));
    HIPCHK(hipDeviceSynchronize());
    vec hout((size_t)q * ow);
    HIPCHK(hipMemcpy(hout.data(), dout, hout.size() * 8, hipMemcpyDeviceToHost));
    double e_nll = 0, e_g = 0, e_sig = 0;
    std::vector<vec> zall(q);
    for (int k = 0; k < q; ++k) {
        const double* t = &P.theta[(size_t)k * tw];
        const double* o = &hout[(size_t)k * ow];
        vec b(n, 0), z(n, 0);
        for (int i = 0; i < n; ++i)
            for (int a = 0; a < p; ++a) b[i] += P.Y[(size_t)a * n + i] * t[d + 3 + a];
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < n; ++j) z[i] += Vall[k][(size_t)i * n + j] * b[j];
        zall[k] = z;
        double ld = 0, quad = 0;
        for (int i = 0; i < n; ++i) { ld += log(Lall[k][(size_t)i * n + i]); quad += b[i] * (b[i] - z[i]); }
        double D = t[d + 2], nug = t[d + 1], scale = t[d], nt = nug / (1 + nug);
        double nll_ref = ld - quad / (2 * D), nll_gpu = o[0] - o[1] / (2 * D);
        e_nll = fmax(e_nll, fabs(nll_ref - nll_gpu) / fabs(nll_ref));
        vec gref(d + 2, 0), ggpu(d + 2);
        double sc0 = 0, tr = 0;
        vec sd(d, 0);
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < n; ++j) {
                double G = P.sr[i] * P.sr[j] * (0.5 * D * Vall[k][(size_t)i * n + j] - 0.5 * z[i] * z[j]);
                double c0 = C0all[k][(size_t)i * n + j];
                sc0 += G * c0;
                if (i == j) tr += G;
                for (int jj = 0; jj < d; ++jj) {
                    double s = Sall[k][jj][(size_t)i * n + j];
                    sd[jj] += G * c0 * s * s / (1 + s);
                }
            }
        for (int jj = 0; jj < d; ++jj) gref[jj] = scale * (1 - nt) / t[jj] * sd[jj];
        gref[d] = (1 - nt) * sc0 + nt * tr;
        gref[d + 1] = scale * (tr - sc0) / ((1 + nug) * (1 + nug));
        for (int jj = 0; jj < d + 2; ++jj) ggpu[jj] = o[3 + jj];
        e_g = fmax(e_g, relmax(ggpu, gref));
        vec sref(p, 0), sgpu(p);
        for (int a = 0; a < p; ++a) {
            for (int i = 0; i < n; ++i) sref[a] += P.Y[(size_t)a * n + i] * (b[i] - z[i]);
            sgpu[a] = o[5 + d + a];
        }
        e_sig = fmax(e_sig, relmax(sgpu, sref));
        if (o[2] != 0) { printf("  out info[%d] = %g\n", k, o[2]); ++g_fail; }
    }
    report("nll_grad NLL_k", e_nll, sizeof(T) == 8 ? 1e-10 : 5e-3);
    report("nll_grad kernel-param grads", e_g, sizeof(T) == 8 ? 1e-9 : 2e-2);
    report("nll_grad gsig", e_sig, sizeof(T) == 8 ? 1e-9 : 2e-2);

    // the same call with a caller-owned plan: identical bits
    {
        lcgp_sched sc;
        LCHK(lcgp_sched_default(&sc));
        size_t pb = 0;
        LCHK(lcgp_plan_bytes(dtype, n, q, 1, &sc, &pb));
        std::vector<char> plan(pb);
        LCHK(lcgp_plan_build(dtype, n, q, 1, &sc, plan.data(), pb));
        int nl = 0, inv = 0;
        LCHK(lcgp_plan_info(plan.data(), &nl, &inv));
        HIPCHK(hipMemset(ws, 0xff, wsb));
        HIPCHK(hipMemset(dout, 0, (size_t)q * ow * 8));
        LCHK(lcgp_nll_grad(nullptr, dtype, LCGP_KERNEL_MATERN32, n, d, p, q, dx, dY, srp, dtheta, ws, dout, nullptr, plan.data()));
        HIPCHK(hipDeviceSynchronize());
        vec hout2((size_t)q * ow);
        HIPCHK(hipMemcpy(hout2.data(), dout, hout2.size() * 8, hipMemcpyDeviceToHost));
        double e = 0;
        for (size_t i = 0; i < hout2.size(); ++i) {
            const double df = fabs(hout2[i] - hout[i]) / (1e-300 + fabs(hout[i]));
            e = fmax(e, std::isfinite(df) ? df : 1e300);
        }
        char what[112];
        snprintf(what, sizeof(what), "caller-owned plan (%d launches) vs per-call plan", nl);
        report(what, e, 0.0);
    }

    // predict
    {
        const int n0 = 37;
        std::mt19937_64 g(seed + 99);
        std::uniform_real_distribution<double> U(0, 1);
        vec x0((size_t)n0 * d);
        for (auto& v : x0) v = U(g);
        std::vector<T> x0s = cast_vec<T>(x0);
        for (size_t i = 0; i < x0.size(); ++i) x0[i] = (double)x0s[i];
        void *dx0, *dscr;
        double *dgh, *dgv;
        size_t scr_bytes = 0;
        LCHK(lcgp_predict_scratch_bytes(dtype, n, q, n0, &scr_bytes));
        HIPCHK(hipMalloc(&dx0, x0s.size() * sizeof(T)));
        HIPCHK(hipMalloc(&dscr, scr_bytes));
        HIPCHK(hipMalloc(&dgh, (size_t)q * n0 * 8));
        HIPCHK(hipMalloc(&dgv, (size_t)q * n0 * 8));
        HIPCHK(hipMemcpy(dx0, x0s.data(), x0s.size() * sizeof(T), hipMemcpyHostToDevice));
        LCHK(lcgp_predict(nullptr, dtype, LCGP_KERNEL_MATERN32, n, d, p, q, dx, srp, dtheta, ws, n0, dx0, 0, dscr, dgh, dgv, 0));
        HIPCHK(hipDeviceSynchronize());
        vec gh((size_t)q * n0), gv((size_t)q * n0), ghr((size_t)q * n0), gvr((size_t)q * n0);
        HIPCHK(hipMemcpy(gh.data(), dgh, gh.size() * 8, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(gv.data(), dgv, gv.size() * 8, hipMemcpyDeviceToHost));
        for (int k = 0; k < q; ++k) {
            const double* t = &P.theta[(size_t)k * tw];
            double nt = t[d + 1] / (1 + t[d + 1]);
            for (int m = 0; m < n0; ++m) {
                vec c(n);
                for (int i = 0; i < n; ++i) {
                    double poly = 1, ss = 0;
                    for (int jj = 0; jj < d; ++jj) {
                        double s = fabs(x0[(size_t)m * d + jj] / t[jj] - P.x[(size_t)i * d + jj] / t[jj]);
                        poly *= 1 + s;
                        ss -= s;
                    }
                    c[i] = t[d] * (1 - nt) * poly * exp(ss) * P.sr[i];
                }
                double s1 = 0, s2 = 0;
                for (int i = 0; i < n; ++i) {
                    s1 += c[i] * zall[k][i];
                    for (int j = 0; j < n; ++j) s2 += c[i] * Vall[k][(size_t)i * n + j] * c[j];
                }
                ghr[(size_t)k * n0 + m] = s1;
                gvr[(size_t)k * n0 + m] = t[d] - t[d + 2] * s2;
            }
        }
        report("predict ghat", relmax(gh, ghr), sizeof(T) == 8 ? 1e-9 : 1e-2);
        report("predict gvar", relmax(gv, gvr), sizeof(T) == 8 ? 1e-9 : 1e-2);
        // rectangular Matern32
        LCHK(lcgp_matern32(nullptr, dtype, n0, n, d, dx0, dx, &P.theta[0], P.theta[d], P.theta[d + 1], 0, dscr));
        std::vector<T> hm((size_t)n0 * n);
        HIPCHK(hipMemcpy(hm.data(), dscr, hm.size() * sizeof(T), hipMemcpyDeviceToHost));
        vec mg(hm.begin(), hm.end()), mr((size_t)n0 * n);
        {
            const double* t = &P.theta[0];
            double nt = t[d + 1] / (1 + t[d + 1]);
            for (int m = 0; m < n0; ++m)
                for (int i = 0; i < n; ++i) {
                    double poly = 1, ss = 0;
                    for (int jj = 0; jj < d; ++jj) {
                        double s = fabs(x0[(size_t)m * d + jj] / t[jj] - P.x[(size_t)i * d + jj] / t[jj]);
                        poly *= 1 + s;
                        ss -= s;
                    }
                    mr[(size_t)m * n + i] = t[d] * (1 - nt) * poly * exp(ss);
                }
        }
        report("matern32 rectangular", relmax(mg, mr), sizeof(T) == 8 ? 1e-14 : 1e-6);
        (void)hipFree(dx0); (void)hipFree(dscr); (void)hipFree(dgh); (void)hipFree(dgv);
    }
    (void)hipFree(ws); (void)hipFree(dx); (void)hipFree(dY); (void)hipFree(dsr); (void)hipFree(dmat); (void)hipFree(dtheta); (void)hipFree(dout);
    (void)hipFree(dld); (void)hipFree(dinfo);
}

int main(int argc, char** argv) {
    int dev = 0;
    HIPCHK(hipSetDevice(dev));
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, dev));
    printf("device: %s (%s), %d CUs, lcgp version %d\n", prop.name, prop.gcnArchName, prop.multiProcessorCount,
           lcgp_version());
    std::vector<int> ns;
    for (int i = 1; i < argc; ++i) ns.push_back(atoi(argv[i]));
    if (ns.empty()) ns = {64, 100, 192, 300, 512};
    unsigned seed = 1;
    for (int n : ns) {
        run_case<double>(n, 3, 5, 2, false, seed++);
        run_case<double>(n, 2, 4, 3, true, seed++);
    }
    run_case<double>(130, 6, 7, 1, false, seed++);
    run_case<double>(70, 11, 3, 2, false, seed++);
    run_case<float>(200, 3, 5, 2, false, seed++);
    run_case<float>(320, 2, 4, 2, true, seed++);
    printf(g_fail ? "FAILED (%d)\n" : "ALL OK\n", g_fail);
    return g_fail ? 1 : 0;
}
