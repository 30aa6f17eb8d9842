/* lcgp_hip.h -- C ABI of liblcgp_hip.so: the MI355X (gfx950) hot path of LCGP.
 *
 * The reference (mosesyhc/LCGP) is pure Python on TensorFlow; it has no FFI.  The "interface" each
 * entry point below replaces is therefore a span of reference Python (file:line cited per function).
 * The Python host side (lcgp_amd/lcgp.py) binds these with ctypes; INTEGRATION.md shows the stub a
 * reference maintainer would add.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer on the current device unless marked "host";
 *  - matrices are row-major;  `dtype`: 0 = float64, 1 = float32 (the reference is float64 only);
 *  - `stream` is a hipStream_t passed as void* (NULL = the null stream); every call only ENQUEUES
 *    work ordered after / before that stream and returns; no device memory is allocated, freed or
 *    synchronised inside (lcgp_nll_grad keeps a few internal streams and events, see lcgp_shutdown);
 *  - the caller owns all memory, including `workspace` (size from lcgp_workspace_bytes);
 *  - return value 0 = enqueued; < 0 = bad argument / HIP error (see lcgp_last_error()).
 *    A non positive-definite matrix is reported through the `info` word of the output block,
 *    not through the return value (the call is asynchronous).
 *
 * theta block (host computes it, one H2D copy per evaluation), per local component k, doubles:
 *     [ ell_0 .. ell_{d-1} | scale | nug | D_k | psi_0 .. psi_{p-1} ]        width d + 3 + p
 *   ell/scale/nug are the CONSTRAINED values (what the reference calls lLmb[k], lLmb0[k],
 *   lnugGPs[k]; lcgp.py:515-532), D_k = diag_D[k] (lcgp.py:480), psi = phi[:,k]/sigma
 *   (lcgp.py:646) -- for the replicated path sigma is sigma_used (lcgp.py:576-584).
 *
 * output block, per local component k, doubles:
 *     [ half_logdet | quad | info | g_ell_0 .. g_ell_{d-1} | g_scale | g_nug | gsig_0 .. gsig_{p-1} ]
 *                                                                         width d + 5 + p
 *   half_logdet = sum_i log L_ii            (= 1/2 log det A_k, A_k = I + D_k (C_k o s s^T))
 *   quad        = b^T (b - A_k^-1 b)        (so NLL_k = half_logdet - quad / (2 D_k))
 *   info        = 0, or 1 + index of the first non-positive pivot
 *   g_*         = d NLL_k / d (constrained ell, scale, nug)
 *   gsig_a      = sum_i Y[a,i] (b_i - z_i)  (host turns it into d NLL / d lsigma2s)
 */
#ifndef LCGP_HIP_H
#define LCGP_HIP_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LCGP_F64 0
#define LCGP_F32 1

/* library version (major*10000 + minor*100 + patch) and last error text (host string). */
int lcgp_version(void);
const char* lcgp_last_error(void);

/* width of the theta / output blocks described above. */
int lcgp_theta_width(int d, int p);
int lcgp_out_width(int d, int p);

/* performance knobs (process-wide, for experiments; results do not depend on them beyond rounding):
 *   key 0: width of the outer Cholesky panel in 64-column blocks (default 0 = automatic: 4 in fp64, 8 in fp32);
 *   key 5: width of the Cholesky super-panel in 64-column blocks (default 0 = the panel width, i.e. off): trailing updates of the panels stop at
 *          the super-panel boundary, the rest of the matrix is updated once per super-panel;
 *   key 6 / key 7: below this many 128x128 tiles per launch (components x tiles) the triangular inverse (6, default
 *          4200) / A^-1 = W^T W (7, default 2048) run on 64x64 tiles: with few components a launch is bounded by
 *          its longest tile (0 = always 128x128);
 *   key 8: the same switch for the trailing update of the Cholesky (default 2000);
 *   key 11: filler blocks (128x64 tiles of the previous panel's trailing update) carried by each diagonal-block launch
 *          (default 248 = one per otherwise idle CU; 0 = no filler);
 *   key 12: 1 (default) = one launch per 64-column step of the Cholesky panel chain (panel TRMM with the previous
 *          column's update folded in, the next diagonal block factored by the workgroup of the tile above it);
 *          0 = diagonal block / panel TRMM / panel update as three dependent launches;
 *   key 13: filler blocks carried by each chain-step launch that ends in a diagonal block (default 248);
 *   key 14: a trailing-update launch of at most this many 64x64 tiles (x components) also factors the next panel's
 *          first diagonal block, so that panel's chain starts one launch earlier (default 1024; 0 = never);
 *   key 15: a level of the triangular inverse with fewer than this many 128x128 tiles (x components) runs on 64x64
 *          tiles even when the large levels use 128x128 ones (default 600);
 *   key 2: bit 2 (value 4) = barrier-per-pivot-pair variant of the diagonal-block kernel instead of the in-wave
 *          16-column panels (bits 0 and 1 skip work for timing experiments and give wrong results);
 *   key 3: 1 = look-ahead Cholesky (panel chain on an internal stream), 0 (default) = single stream;
 *   key 4: 1 = create that internal stream with the highest priority;
 *   key 1: number of component groups whose factorisation chains run on internal streams (default 1 = off, max 8). */
int lcgp_set_tuning(int key, int value);

/* lcgp_nll_grad overlaps the dependent launch chains of different components on a few internal HIP streams
 * (created on first use, one set per device, kept for the life of the process).  lcgp_shutdown destroys them. */
int lcgp_shutdown(void);

/* bytes of `workspace` needed by lcgp_nll_grad / lcgp_potrf_logdet / lcgp_potri for q_local components. */
int lcgp_workspace_bytes(int dtype, int n, int d, int p, int q_local, size_t* bytes /*host out*/);

/* Matern32(x1, x2, llmb, llmb0, lnug)  -- covmat.py:5-55 (build branch 31-55).
 * out (n1 x n2, row-major).  `same` != 0 adds the nugget term on the diagonal (the reference adds it
 * iff x1 and x2 have equal shape and bitwise-equal values, covmat.py:46-51; the caller decides). */
int lcgp_matern32(void* stream, int dtype, int n1, int n2, int d,
                  const void* x1, const void* x2,
                  const double* ell /*host, d*/, double scale, double nug, int same, void* out);

/* K1: A_k = I + D_k * (C_k o sr sr^T) for all local components, into the workspace
 * (lcgp.py:651 for the full path; lcgp.py:606 + 616 for the replicated path, sr = sqrt(r)).
 * Only the lower-triangular 64x64 tiles are written.  sr may be NULL (all ones). */
int lcgp_kernel_build(void* stream, int dtype, int n, int d, int p, int q_local,
                      const void* x /*n x d*/, const void* sr /*n or NULL*/,
                      const double* theta, void* workspace);

/* K2: blocked Cholesky A_k = L_k L_k^T of the matrices left by lcgp_kernel_build, in place, plus
 * W = L^-1 blocks needed later.  Replaces tf.linalg.eigh (lcgp.py:652) / tf.linalg.cholesky
 * (lcgp.py:617) and the log-determinant (lcgp.py:660 / 624).
 * half_logdet (q_local doubles) and info (q_local ints) are written on the device. */
int lcgp_potrf_logdet(void* stream, int dtype, int n, int d, int p, int q_local, void* workspace,
                      double* half_logdet, int* info);

/* K4: A_k^-1 (lower tiles) from the factor left by lcgp_potrf_logdet.  Replaces the dense
 * U diag(.) U^T products of lcgp.py:654 / 705-715 and cholesky_solve with identity (lcgp.py:785). */
int lcgp_potri(void* stream, int dtype, int n, int d, int p, int q_local, void* workspace);

/* the two stages of lcgp_potri on their own (for per-kernel timing): W = L^-1 (level-parallel triangular
 * products), then A^-1 = W^T W (a single launch of the MFMA tile kernel; n^3/3 flops per component). */
int lcgp_trtri(void* stream, int dtype, int n, int d, int p, int q_local, void* workspace);
int lcgp_lauum(void* stream, int dtype, int n, int d, int p, int q_local, void* workspace);

/* copies matrix `which` (0 = A/L, 1 = L^-1, 2 = A^-1) of local component k out of the workspace as a
 * dense n x n row-major matrix (lower triangle valid, upper triangle mirrored); for tests. */
int lcgp_fetch_matrix(void* stream, int dtype, int n, int d, int p, int q_local, const void* workspace,
                      int which, int k, void* out /*n x n*/);

/* copies vector `which` (0 = b_k, 1 = z_k = A_k^-1 b_k) of local component k (n elements of dtype). */
int lcgp_fetch_vector(void* stream, int dtype, int n, int d, int p, int q_local, const void* workspace,
                      int which, int k, void* out /*n*/);

/* The whole hot path for q_local components: build + Cholesky + inverse + z = A^-1 b + fused gradient
 * contraction.  Replaces one call of LCGP.neglpost (lcgp.py:635-666) or LCGP.neglpost_rep
 * (lcgp.py:554-630) TOGETHER WITH the tf.GradientTape backward pass gpflow runs around it
 * (lcgp.py:538-539), for the components held by this rank.
 *   x  : n x d standardised inputs (x_unique_s for the replicated path)
 *   Y  : p x n outputs the latent targets are projected from (standardised y; sqrt(r) o ybar for rep)
 *   sr : NULL for the full path; sqrt(r) (n) for the replicated path
 *   theta, out : device blocks described at the top (q_local rows each) */
int lcgp_nll_grad(void* stream, int dtype, int n, int d, int p, int q_local,
                  const void* x, const void* Y, const void* sr,
                  const double* theta, void* workspace, double* out);

/* K6 prediction (lcgp.py:808-859 / 864-930 with the caches of 685-803): for local component k and
 * n0 new inputs x0 (already standardised) computes
 *     ghat[k, :] = c0k (sr o z_k)                       (lcgp.py:831 / 888)
 *     gvar[k, :] = scale_k - D_k rowsum((c0k o sr) A_k^-1 (c0k o sr)^T)   (lcgp.py:832 / 891-894)
 * using A_k^-1 and z_k left in the workspace by the last lcgp_nll_grad call with the same theta.
 * `same` as in lcgp_matern32 (nugget on the diagonal when x0 is the training set itself).
 * scratch: 2 * n0pad * npad elements of dtype (n0pad = n0 rounded up to 64, npad = n rounded up to 128). */
int lcgp_predict(void* stream, int dtype, int n, int d, int p, int q_local,
                 const void* x, const void* sr, const double* theta, const void* workspace,
                 int n0, const void* x0, int same, void* scratch,
                 double* ghat /*q_local x n0*/, double* gvar /*q_local x n0*/);

#ifdef __cplusplus
}
#endif
#endif /* LCGP_HIP_H */
