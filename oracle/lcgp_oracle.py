"""CPU oracle for the LCGP fit/predict hot path  --  TEST INFRASTRUCTURE ONLY.

This file is a plain numpy/scipy restatement of the reference algorithm
(mosesyhc/LCGP, `src/lcgp/covmat.py` and `src/lcgp/lcgp.py`).  It exists so that
the HIP path in `lcgp_amd/` can be checked against an independent CPU
implementation.  Only `tests/`, `__graft_entry__.smoke()` and the
`cpu_baseline` leg of `bench.py` may import it; nothing under `lcgp_amd/`
does, and the product path never falls back to it.

Parity pin: the reference cannot be imported in the build container (it needs
tensorflow / tensorflow_probability / gpflow, none installed; an ordinary
ModuleNotFoundError, nothing was refused) and its test-suite pins no numeric
value of the hot path.  The oracle is therefore pinned by the only numbers the
reference stores: the cell outputs of
`illustration-examples/lcgp-rep-1d-illustration.ipynb` (KAT-1: diag_D and
var(g) to 8 printed digits; KAT-2: fitted lengthscales / noise / RMSE /
coverage / DSS after fit+predict) -- see `tests/test_oracle_kat.py` -- plus
cross-identities (eigh form == Cholesky form, full == n * rep when r = 1,
closed-form gradient == finite differences).

Each function cites the reference lines it restates.  Third-party behaviour
restated here because the reference only calls it:
  * tfp.stats.percentile(..., 50.0) default interpolation='nearest'
    (tensorflow-probability >= 0.25, unpinned): sorted ascending, index
    round_half_even(0.5 * (m - 1)).
  * tfp.bijectors.SoftClip(low, high), hinge_softness = 1.
  * gpflow.optimizers.Scipy().minimize == scipy L-BFGS-B, default options, on
    the flat unconstrained vector ordered lLmb, lLmb0, lnugGPs, lsigma2s.
"""
from __future__ import annotations

import numpy as np
import scipy.linalg as sla
import scipy.optimize as sopt

F64 = np.float64

# bounds of the three SoftClip transforms, lcgp.py:181-211
LLMB_BOUNDS = (1e-6, 1e4)
LLMB0_BOUNDS = (1e-4, 1e4)
LNUG_BOUNDS = (float(np.exp(-16.0)), float(np.exp(-2.0)))


# --------------------------------------------------------------------------------------
# third-party pieces restated
# --------------------------------------------------------------------------------------
def percentile50_nearest(a, axis=1):
    """tfp.stats.percentile(a, 50.0, axis, keepdims=True), interpolation='nearest'.

    Used by lcgp.py:317-318 and 388-389.  NOT np.median: for an even count m it
    picks sorted[round_half_even(0.5*(m-1))], e.g. m = 40 -> index 20.
    """
    a = np.asarray(a, dtype=F64)
    m = a.shape[axis]
    idx = int(np.round(0.5 * (m - 1)))  # np.round is round-half-to-even, like tf.round
    s = np.sort(a, axis=axis)
    return np.take(s, [idx], axis=axis)


def _softplus(t):
    t = np.asarray(t, dtype=F64)
    return np.logaddexp(0.0, t)


def _softplus_inv(t):
    t = np.asarray(t, dtype=F64)
    return t + np.log(-np.expm1(-t))


def _sigmoid(t):
    t = np.asarray(t, dtype=F64)
    return 0.5 * (1.0 + np.tanh(0.5 * t))


def softclip_forward(u, lo, hi):
    """constrained value of a SoftClip(low, high) parameter."""
    w = hi - lo
    return hi - w / _softplus(w) * _softplus(w - _softplus(np.asarray(u, F64) - lo))


def softclip_inverse(v, lo, hi):
    w = hi - lo
    inner = _softplus_inv((hi - np.asarray(v, F64)) * _softplus(w) / w)
    return lo + _softplus_inv(w - inner)


def softclip_grad(u, lo, hi):
    """d constrained / d unconstrained."""
    w = hi - lo
    s1 = _softplus(np.asarray(u, F64) - lo)
    return w / _softplus(w) * _sigmoid(w - s1) * _sigmoid(np.asarray(u, F64) - lo)


# --------------------------------------------------------------------------------------
# covmat.py
# --------------------------------------------------------------------------------------
def matern32(x1, x2, llmb, llmb0, lnug, diag_only=False, kernel='matern32'):
    """covmat.py:5-55.  `llmb`, `llmb0`, `lnug` are the constrained values.

    kernel='se': the squared-exponential product kernel C0 = exp(-1/2 sum_j S_j^2) with the same nugget / scale structure.
    The reference has NO such kernel: this branch restates nothing, it defines the extension the HIP path offers as
    LCGP_KERNEL_SE (parity unpinned; checked through identities only, tests/test_oracle_identities.py)."""
    x1 = np.asarray(x1, F64)
    x2 = np.asarray(x2, F64)
    assert x1.ndim == 2, 'input x1 should be 2-dimensional, (n_param, dim_param)'
    assert x2.ndim == 2, 'input x2 should be 2-dimensional, (n_param, dim_param)'
    assert x1.shape[1] == x2.shape[1], 'the dim_param of input x1 and x2 should be the same.'
    llmb = np.asarray(llmb, F64).reshape(-1)
    if diag_only:  # covmat.py:23-29
        assert np.all(np.abs(x1 - x2) <= (1e-6 + 1e-6 * np.abs(x2)))
        return float(llmb0) * np.ones(x1.shape[0], F64)
    a = x1 / llmb
    b = x2 / llmb
    logpart = np.zeros((x1.shape[0], x2.shape[0]), F64)
    poly = np.ones_like(logpart)
    for j in range(x1.shape[1]):  # covmat.py:37-41
        s = np.abs(a[:, j][:, None] - b[:, j][None, :])
        if kernel == 'se':
            logpart -= 0.5 * s * s
        else:
            poly *= 1.0 + s
            logpart -= s
    c0 = poly * np.exp(logpart)
    nt = lnug / (1.0 + lnug)
    same = (x1.shape == x2.shape) and bool(np.all(x1 == x2))  # covmat.py:46-53
    if same:
        c = (1.0 - nt) * c0 + nt * np.eye(x1.shape[0])
    else:
        c = (1.0 - nt) * c0
    return float(llmb0) * c


def matern32_c0_and_s(x, ell, kernel='matern32'):
    """C0 = prod_j (1+S_j) exp(-sum S_j) and the list of S_j; kernel='se': C0 = exp(-1/2 sum S_j^2)."""
    a = x / ell
    n, d = x.shape
    s_all = np.empty((d, n, n), F64)
    for j in range(d):
        s_all[j] = np.abs(a[:, j][:, None] - a[:, j][None, :])
    if kernel == 'se':
        c0 = np.exp(-0.5 * (s_all * s_all).sum(axis=0))
    else:
        c0 = np.prod(1.0 + s_all, axis=0) * np.exp(-s_all.sum(axis=0))
    return c0, s_all


# --------------------------------------------------------------------------------------
# one-off preprocessing, lcgp.py:295-324, 349-395, 454-513
# --------------------------------------------------------------------------------------
def standardize_x(x):
    """lcgp.py:295-310 (xnorm omitted here; see `xnorm_pairs`)."""
    x = np.asarray(x, F64)
    x_min = x.min(axis=0)
    x_max = x.max(axis=0)
    return (x - x_min) / (x_max - x_min), x_min, x_max


def xnorm_pairs(x):
    """lcgp.py:304-309: mean of the strictly positive |x_ij - x_i'j| (O(n^2) form)."""
    x = np.asarray(x, F64)
    out = np.zeros(x.shape[1], F64)
    for j in range(x.shape[1]):
        dist = np.abs(x[:, j][:, None] - x[:, j][None, :])
        pos = dist[dist > 0]
        out[j] = pos.mean() if pos.size else np.nan
    return out


def center_spread(y, robust, guard_zero):
    """lcgp.py:312-321 (guard_zero=False) and 383-395 (guard_zero=True)."""
    y = np.asarray(y, F64)
    if robust:
        c = percentile50_nearest(y, axis=1)
        s = percentile50_nearest(np.abs(y - c), axis=1)
    else:
        c = y.mean(axis=1, keepdims=True)
        s = y.std(axis=1, keepdims=True)
    if guard_zero:
        s = np.where(s > 0, s, 1.0)
    return c, s


def group_replicates(x_raw, y_raw):
    """lcgp.py:349-367: unique rows, inverse, counts, replicate means (raw scale)."""
    xu, inv, cnt = np.unique(np.asarray(x_raw, F64), axis=0, return_inverse=True, return_counts=True)
    inv = np.asarray(inv).reshape(-1)
    n = xu.shape[0]
    p = y_raw.shape[0]
    ybar = np.zeros((p, n), F64)
    for i in range(n):
        ybar[:, i] = y_raw[:, inv == i].mean(axis=1)
    return xu, inv, cnt.astype(np.int32), ybar


def init_basis(ymat, n, q=None, var_threshold=None):
    """lcgp.py:454-485: thin SVD -> phi (p,q), diag_D (q,), g (q,n), q."""
    ymat = np.asarray(ymat, F64)
    p = ymat.shape[0]
    u, s, _ = np.linalg.svd(ymat, full_matrices=False)
    if q is None and var_threshold is None:
        q = p
    elif q is None:
        cum = np.cumsum(s ** 2) / np.sum(s ** 2)
        q = int(np.argmax(cum > var_threshold) + 1) if np.any(cum > var_threshold) else p
    else:
        q = int(q)
    phi = u[:, :q] * np.sqrt(float(n)) / s[:q]
    diag_d = np.sum(phi ** 2, axis=0)
    g = phi.T @ ymat
    return g, phi, diag_d, q


def init_param_values(xs_all, y_for_var, d, q, err_struct):
    """lcgp.py:490-513: constrained initial values (lLmb, lLmb0, lnugGPs, lsigma2s)."""
    llmb_row = np.exp(0.5 * np.log(d) + np.log(np.std(xs_all, axis=0)))
    lLmb = np.tile(llmb_row, q).reshape(q, d)
    lLmb0 = np.ones(q, F64)
    lnug = np.exp(-10.0) * np.ones(q, F64)
    ls2 = np.zeros(len(err_struct), F64)
    col = 0
    for g, w in enumerate(err_struct):
        ls2[g] = np.log(np.var(y_for_var[col:col + w]))
        col += w
    return lLmb, lLmb0, lnug, ls2


def expand_lsigma2s(ls2, err_struct):
    """lcgp.py:521-530."""
    return np.repeat(np.asarray(ls2, F64), np.asarray(err_struct, int))


# --------------------------------------------------------------------------------------
# full path NLL, two algebraically identical forms
# --------------------------------------------------------------------------------------
def nll_full_eigh(x, y, phi, diag_d, lLmb, lLmb0, ls2_built, lnug, kernel='matern32'):
    """lcgp.py:635-666 restated literally (eigh + dense products) -- the reference algorithm."""
    n = x.shape[0]
    q = phi.shape[1]
    psi_c = phi.T / np.sqrt(np.exp(ls2_built))
    nlp = 0.0
    for k in range(q):
        ck = matern32(x, x, lLmb[k], lLmb0[k], lnug[k], kernel=kernel)
        wk, uk = np.linalg.eigh(ck)
        qk = uk @ (np.diag(1.0 / (diag_d[k] + 1.0 / wk)) @ uk.T)
        pk = psi_c[k][:, None] @ psi_c[k][None, :]
        yqk = y @ qk
        ypk = y.T @ pk.T
        nlp += 0.5 * np.sum(np.log(1.0 + diag_d[k] * wk))
        nlp += -0.5 * np.sum(yqk * ypk.T)
    nlp += n / 2.0 * np.sum(ls2_built)
    nlp += 0.5 * np.sum((y.T / np.sqrt(np.exp(ls2_built))) ** 2)
    return float(nlp)


def _chol_component(x, ell, scale, nug, dk, b, sr=None, kernel='matern32'):
    """A = I + D (C o sr sr^T); returns L, C0, S, half log det, z = A^-1 b."""
    c0, s_all = matern32_c0_and_s(x, ell, kernel)
    nt = nug / (1.0 + nug)
    n = x.shape[0]
    c = scale * ((1.0 - nt) * c0 + nt * np.eye(n))
    if sr is not None:
        c_scaled = c * sr[:, None] * sr[None, :]
    else:
        c_scaled = c
    a = np.eye(n) + dk * c_scaled
    low = np.linalg.cholesky(a)
    half_logdet = float(np.sum(np.log(np.diag(low))))
    z = sla.cho_solve((low, True), b)
    return low, c0, s_all, half_logdet, z


def _kernel_param_grads(low, c0, s_all, z, dk, ell, scale, nug, sr=None, kernel='matern32'):
    """sum_ij G_ij dC_ij/dtheta with G = sr sr^T o (D/2 A^-1 - z z^T / 2)  (SURVEY A.5)."""
    n = low.shape[0]
    ainv = sla.cho_solve((low, True), np.eye(n))
    gmat = 0.5 * dk * ainv - 0.5 * np.outer(z, z)
    if sr is not None:
        gmat = gmat * sr[:, None] * sr[None, :]
    nt = nug / (1.0 + nug)
    g_ell = np.empty(len(ell), F64)
    for j in range(len(ell)):
        sj = s_all[j]
        if kernel == 'se':      # d/d ell_j exp(-1/2 sum S^2) = C0 S_j^2 / ell_j
            g_ell[j] = np.sum(gmat * (scale * (1.0 - nt) * c0 * sj * sj / ell[j]))
        else:
            g_ell[j] = np.sum(gmat * (scale * (1.0 - nt) * c0 * sj * sj / ((1.0 + sj) * ell[j])))
    tr_g = np.trace(gmat)
    g_c0 = np.sum(gmat * c0)
    g_scale = (1.0 - nt) * g_c0 + nt * tr_g
    g_nug = scale * (tr_g - g_c0) / (1.0 + nug) ** 2
    return g_ell, g_scale, g_nug


def nll_grad_full_chol(x, y, phi, diag_d, err_struct, lLmb, lLmb0, ls2, lnug, want_grad=True, kernel='matern32'):
    """Cholesky form of lcgp.py:635-666 + closed-form gradient (A.5).

    Returns (nll, dict of gradients w.r.t. the CONSTRAINED parameters).
    """
    x = np.asarray(x, F64)
    y = np.asarray(y, F64)
    n, d = x.shape
    p, q = phi.shape
    ls2_b = expand_lsigma2s(ls2, err_struct)
    sig = np.exp(0.5 * ls2_b)
    ysq = np.sum(y * y, axis=1)
    nll = n / 2.0 * np.sum(ls2_b) + 0.5 * np.sum(ysq / sig ** 2)
    g_lLmb = np.zeros((q, d), F64)
    g_lLmb0 = np.zeros(q, F64)
    g_lnug = np.zeros(q, F64)
    g_ls2_b = n / 2.0 - 0.5 * ysq / sig ** 2
    for k in range(q):
        b = y.T @ (phi[:, k] / sig)
        low, c0, s_all, half_logdet, z = _chol_component(x, lLmb[k], lLmb0[k], lnug[k], diag_d[k], b, kernel=kernel)
        nll += half_logdet - float(b @ (b - z)) / (2.0 * diag_d[k])
        if want_grad:
            ge, gs, gn = _kernel_param_grads(low, c0, s_all, z, diag_d[k], lLmb[k], lLmb0[k], lnug[k], kernel=kernel)
            g_lLmb[k], g_lLmb0[k], g_lnug[k] = ge, gs, gn
            gb = -(b - z) / diag_d[k]
            g_ls2_b += -0.5 / sig * phi[:, k] * (y @ gb)
    if not want_grad:
        return float(nll), None
    g_ls2 = np.add.reduceat(g_ls2_b, np.r_[0, np.cumsum(err_struct)[:-1]])
    return float(nll), dict(lLmb=g_lLmb, lLmb0=g_lLmb0, lnugGPs=g_lnug, lsigma2s=g_ls2)


# --------------------------------------------------------------------------------------
# replicated path NLL
# --------------------------------------------------------------------------------------
def nll_rep_literal(xu_s, ybar_used, ybar_std, use_std, r, phi, diag_d, n, p, lLmb, lLmb0, ls2_built, lnug, kernel='matern32'):
    """lcgp.py:554-630 restated step by step (Cholesky of I + d_k R^1/2 C_k R^1/2)."""
    r = np.asarray(r, F64)
    sigma_var_raw = np.exp(ls2_built)
    sis_raw = np.sqrt(1.0 / sigma_var_raw)
    if use_std:
        std = ybar_std[:, 0]
        sigma_var_used = sigma_var_raw / std ** 2
        sis = sis_raw * std
    else:
        sigma_var_used = sigma_var_raw
        sis = sis_raw
    nlp = 0.5 * np.sum(r * np.sum((ybar_used * sis[:, None]) ** 2, axis=0))
    nlp += 0.5 * n * np.sum(np.log(sigma_var_used))
    nlp += -0.5 * p * np.sum(np.log(r))
    sr = np.sqrt(r)
    bsb = 0.0
    logdet = 0.0
    for k in range(phi.shape[1]):
        ck = matern32(xu_s, xu_s, lLmb[k], lLmb0[k], lnug[k], kernel=kernel)
        b = r * (ybar_used.T @ (sis * phi[:, k]))
        dk = diag_d[k]
        cb = ck @ b
        a = np.eye(int(n)) + dk * ((ck * sr[None, :]) * sr[:, None])
        la = np.linalg.cholesky(a)
        u = np.sqrt(dk) * (sr * cb)
        z = sla.cho_solve((la, True), u)
        sb = cb - ck @ (np.sqrt(dk) * (sr * z))
        bsb += float(b @ sb)
        logdet += 2.0 * float(np.sum(np.log(np.diag(la))))
    nlp += -0.5 * bsb + 0.5 * logdet
    return float(nlp / n)


def nll_grad_rep_chol(xu_s, ybar_used, ybar_std, use_std, r, phi, diag_d, err_struct,
                      lLmb, lLmb0, ls2, lnug, want_grad=True, kernel='matern32'):
    """Same value as `nll_rep_literal`, through the identity
    b^T (C^-1 + D R)^-1 b = (beta^T beta - beta^T A^-1 beta)/D with beta = b / sqrt(r),
    A = I + D (C o sqrt(r) sqrt(r)^T); closed-form gradient w.r.t. constrained parameters.
    """
    xu_s = np.asarray(xu_s, F64)
    r = np.asarray(r, F64)
    n, d = xu_s.shape
    p, q = phi.shape
    sr = np.sqrt(r)
    ls2_b = expand_lsigma2s(ls2, err_struct)
    std = ybar_std[:, 0] if use_std else np.ones(p, F64)
    sig_eff = np.exp(0.5 * ls2_b) / std              # sqrt(sigma_var_used)
    yeff = ybar_used * sr[None, :]                    # sqrt(r_i) * ybar
    ysq = np.sum(yeff * yeff, axis=1)
    nll = 0.5 * np.sum(ysq / sig_eff ** 2) + n / 2.0 * np.sum(ls2_b - 2.0 * np.log(std)) \
        - 0.5 * p * np.sum(np.log(r))
    g_lLmb = np.zeros((q, d), F64)
    g_lLmb0 = np.zeros(q, F64)
    g_lnug = np.zeros(q, F64)
    g_ls2_b = n / 2.0 - 0.5 * ysq / sig_eff ** 2
    for k in range(q):
        beta = yeff.T @ (phi[:, k] / sig_eff)
        low, c0, s_all, half_logdet, z = _chol_component(xu_s, lLmb[k], lLmb0[k], lnug[k], diag_d[k], beta, sr, kernel=kernel)
        nll += half_logdet - float(beta @ (beta - z)) / (2.0 * diag_d[k])
        if want_grad:
            ge, gs, gn = _kernel_param_grads(low, c0, s_all, z, diag_d[k], lLmb[k], lLmb0[k], lnug[k], sr, kernel=kernel)
            g_lLmb[k], g_lLmb0[k], g_lnug[k] = ge, gs, gn
            gb = -(beta - z) / diag_d[k]
            g_ls2_b += -0.5 / sig_eff * phi[:, k] * (yeff @ gb)
    if not want_grad:
        return float(nll / n), None
    g_ls2 = np.add.reduceat(g_ls2_b, np.r_[0, np.cumsum(err_struct)[:-1]])
    return float(nll / n), dict(lLmb=g_lLmb / n, lLmb0=g_lLmb0 / n, lnugGPs=g_lnug / n, lsigma2s=g_ls2 / n)


# --------------------------------------------------------------------------------------
# the model (reference API restated on the CPU)
# --------------------------------------------------------------------------------------
class OracleLCGP:
    """CPU restatement of `lcgp.LCGP` (lcgp.py:19-930) for parity checks only."""

    def __init__(self, y, x, q=None, var_threshold=None, diag_error_structure=None,
                 parameter_clamp_flag=False, robust_mean=True, submethod='full',
                 rep_standardize_ybar=True, verbose=False, kernel='matern32'):
        self.kernel = kernel            # 'matern32' = the reference; 'se' = the extension (see matern32())
        self.robust_mean = robust_mean
        self.rep_standardize_ybar = rep_standardize_ybar
        x = np.asarray(x, F64)
        y = np.asarray(y, F64)
        if x.ndim < 2:
            x = x[:, None]
        if y.ndim < 2:
            y = y[:, None]
        if submethod not in ('full', 'rep'):
            raise ValueError("Invalid submethod. Choices are 'full' or 'rep'.")
        self.submethod = submethod
        if q is not None and var_threshold is not None:
            raise ValueError('Include only q or var_threshold but not both.')
        assert y.shape[1] == x.shape[0]
        self.x_orig, self.y_orig = x, y
        self.n, self.d, self.p = x.shape[0], x.shape[1], y.shape[0]
        self.x, self.x_min, self.x_max = standardize_x(x)
        self.y = y
        if submethod == 'rep':
            xu, inv, cnt, ybar = group_replicates(x, y)
            self.x_unique, self.group_ids, self.r, self.ybar = xu, inv, cnt, ybar
            self.x_unique_s = (xu - self.x_min) / (self.x_max - self.x_min)
            self.ybar_mean, self.ybar_std = center_spread(ybar, robust_mean, guard_zero=True)
            self.ybar_s = (ybar - self.ybar_mean) / self.ybar_std
            self.n = xu.shape[0]
            basis_in = self.ybar_s if rep_standardize_ybar else self.ybar
        else:
            self.ymean, self.ystd = center_spread(y, robust_mean, guard_zero=False)
            self.y = (y - self.ymean) / self.ystd
            basis_in = self.y
        self.g, self.phi, self.diag_D, self.q = init_basis(basis_in, self.n, q, var_threshold)
        self.diag_error_structure = [1] * self.p if diag_error_structure is None else list(diag_error_structure)
        assert sum(self.diag_error_structure) == self.y.shape[0]
        self.lLmb, self.lLmb0, self.lnugGPs, self.lsigma2s = init_param_values(
            self.x, self.y, self.d, self.q, self.diag_error_structure)
        self._aux = None
        # gpflow.Parameter.assign stores the UNCONSTRAINED value and re-applies the bijector on every read
        # (lcgp.py:509-512): the constrained values carry that round trip (~1e-12 for bounds of 1e4)
        self.set_unconstrained(self.get_unconstrained())

    # --- parameter vector seen by L-BFGS-B ------------------------------------------------
    def _sizes(self):
        return self.q * self.d, self.q, self.q, len(self.diag_error_structure)

    def get_unconstrained(self):
        return np.concatenate([
            softclip_inverse(self.lLmb, *LLMB_BOUNDS).reshape(-1),
            softclip_inverse(self.lLmb0, *LLMB0_BOUNDS),
            softclip_inverse(self.lnugGPs, *LNUG_BOUNDS),
            np.asarray(self.lsigma2s, F64)])

    def _split(self, u):
        a, b, c, e = self._sizes()
        return u[:a].reshape(self.q, self.d), u[a:a + b], u[a + b:a + b + c], u[a + b + c:a + b + c + e]

    def set_unconstrained(self, u):
        u1, u2, u3, u4 = self._split(np.asarray(u, F64))
        self.lLmb = softclip_forward(u1, *LLMB_BOUNDS)
        self.lLmb0 = softclip_forward(u2, *LLMB0_BOUNDS)
        self.lnugGPs = softclip_forward(u3, *LNUG_BOUNDS)
        self.lsigma2s = np.array(u4, F64)
        self._aux = None

    def get_param(self):
        return self.lLmb, self.lLmb0, expand_lsigma2s(self.lsigma2s, self.diag_error_structure), self.lnugGPs

    # --- objective ---------------------------------------------------------------------
    def _value_and_constrained_grad(self, want_grad=True):
        if self.submethod == 'full':
            return nll_grad_full_chol(self.x, self.y, self.phi, self.diag_D, self.diag_error_structure,
                                      self.lLmb, self.lLmb0, self.lsigma2s, self.lnugGPs, want_grad, kernel=self.kernel)
        ybar_used = self.ybar_s if self.rep_standardize_ybar else self.ybar
        return nll_grad_rep_chol(self.x_unique_s, ybar_used, self.ybar_std, self.rep_standardize_ybar,
                                 self.r, self.phi, self.diag_D, self.diag_error_structure,
                                 self.lLmb, self.lLmb0, self.lsigma2s, self.lnugGPs, want_grad, kernel=self.kernel)

    def loss(self):
        return self._value_and_constrained_grad(False)[0]

    def loss_reference_form(self):
        """The literal restatement (eigh form / step-by-step rep form)."""
        lLmb, lLmb0, ls2b, lnug = self.get_param()
        if self.submethod == 'full':
            return nll_full_eigh(self.x, self.y, self.phi, self.diag_D, lLmb, lLmb0, ls2b, lnug, kernel=self.kernel)
        ybar_used = self.ybar_s if self.rep_standardize_ybar else self.ybar
        return nll_rep_literal(self.x_unique_s, ybar_used, self.ybar_std, self.rep_standardize_ybar, self.r,
                               self.phi, self.diag_D, self.n, self.p, lLmb, lLmb0, ls2b, lnug, kernel=self.kernel)

    def loss_and_grad_unconstrained(self, u=None):
        if u is not None:
            self.set_unconstrained(u)
        uu = self.get_unconstrained() if u is None else np.asarray(u, F64)
        u1, u2, u3, _ = self._split(uu)
        val, g = self._value_and_constrained_grad(True)
        grad = np.concatenate([
            (g['lLmb'] * softclip_grad(u1, *LLMB_BOUNDS)).reshape(-1),
            g['lLmb0'] * softclip_grad(u2, *LLMB0_BOUNDS),
            g['lnugGPs'] * softclip_grad(u3, *LNUG_BOUNDS),
            g['lsigma2s']])
        return val, grad

    def fit(self, verbose=False):
        """lcgp.py:537-540: scipy L-BFGS-B, all defaults, jac=True."""
        u0 = self.get_unconstrained()
        res = sopt.minimize(lambda u: self.loss_and_grad_unconstrained(u), u0, jac=True, method='L-BFGS-B')
        self.set_unconstrained(res.x)
        self.opt_result = res
        return None

    # --- prediction, lcgp.py:685-930 ------------------------------------------------------
    def _aux_full(self):
        lLmb, lLmb0, ls2b, lnug = self.get_param()
        bmat = (self.y.T / np.sqrt(np.exp(ls2b))) @ self.phi
        cinvm = np.zeros((self.q, self.n))
        ths = np.zeros((self.q, self.n, self.n))
        for k in range(self.q):
            ck = matern32(self.x, self.x, lLmb[k], lLmb0[k], lnug[k], kernel=self.kernel)
            wk, uk = np.linalg.eigh(ck)
            dk = self.diag_D[k]
            ipd = uk @ (np.diag(1.0 / (1.0 + dk * wk)) @ uk.T)
            cinvm[k] = ipd @ bmat[:, k]
            ths[k] = uk @ (np.diag(np.sqrt((dk * wk ** 2) / (wk ** 2 + dk * wk ** 3))) @ uk.T)
        return dict(CinvMs=cinvm, Ths=ths)

    def _aux_rep(self):
        lLmb, lLmb0, ls2b, lnug = self.get_param()
        r = np.asarray(self.r, F64)
        use_std = self.rep_standardize_ybar
        ybar = self.ybar_s if use_std else self.ybar
        sis = np.exp(-0.5 * ls2b) * (self.ybar_std[:, 0] if use_std else 1.0)
        sr = np.sqrt(r)
        n = self.n
        cinvm = np.zeros((self.q, n))
        tks = np.zeros((self.q, n, n))
        mks = np.zeros((self.q, n))
        for k in range(self.q):
            ck = matern32(self.x_unique_s, self.x_unique_s, lLmb[k], lLmb0[k], lnug[k], kernel=self.kernel)
            b = r * (ybar.T @ (sis * self.phi[:, k]))
            dk = self.diag_D[k]
            cb = ck @ b
            a = np.eye(n) + dk * ((ck * sr[None, :]) * sr[:, None])
            la = np.linalg.cholesky(a)
            z = sla.cho_solve((la, True), np.sqrt(dk) * (sr * cb))
            mk = cb - ck @ (np.sqrt(dk) * (sr * z))
            cinvm[k] = b - dk * (r * mk)
            lc = np.linalg.cholesky(ck)
            invc = sla.cho_solve((lc, True), np.eye(n))
            vk = np.linalg.inv(invc + dk * np.diag(r))
            tks[k] = invc - invc @ vk @ invc
            mks[k] = mk
        return dict(CinvMs=cinvm, Tks=tks, mks=mks)

    def predict(self, x0, return_fullcov=False):
        x0 = np.asarray(x0, F64)
        if x0.ndim < 2:
            x0 = x0[:, None]
        lLmb, lLmb0, ls2b, lnug = self.get_param()
        x0s = (x0 - self.x_min) / (self.x_max - self.x_min)
        n0 = x0s.shape[0]
        ghat = np.zeros((self.q, n0))
        gvar = np.zeros((self.q, n0))
        if self.submethod == 'full':
            if self._aux is None:
                self._aux = self._aux_full()
            for k in range(self.q):
                c00 = matern32(x0s, x0s, lLmb[k], lLmb0[k], lnug[k], diag_only=True)
                c0k = matern32(x0s, self.x, lLmb[k], lLmb0[k], lnug[k], kernel=self.kernel)
                ghat[k] = c0k @ self._aux['CinvMs'][k]
                gvar[k] = c00 - np.sum((c0k @ self._aux['Ths'][k]) ** 2, axis=1)
            self.ghat, self.gvar = ghat, gvar
            psi = self.phi.T * np.sqrt(np.exp(ls2b))
            predmean = psi.T @ ghat
            confvar = gvar.T @ psi ** 2
            predvar = confvar + np.exp(ls2b)
            ypred = predmean * self.ystd + self.ymean
            yconfvar = confvar.T * self.ystd ** 2
            ypredvar = predvar.T * self.ystd ** 2
            if return_fullcov:
                ch = np.einsum('kn,kp->npk', np.sqrt(gvar), psi)
                cov = ch @ np.transpose(ch, (0, 2, 1)) + np.diag(np.exp(ls2b))[None]
                sv = self.ystd[:, 0]
                cov = cov * (sv[:, None] * sv[None, :])[None]
                return ypred, ypredvar, yconfvar, cov
            return ypred, ypredvar, yconfvar
        if self._aux is None:
            self._aux = self._aux_rep()
        for k in range(self.q):
            c00 = matern32(x0s, x0s, lLmb[k], lLmb0[k], lnug[k], diag_only=True)
            c0k = matern32(x0s, self.x_unique_s, lLmb[k], lLmb0[k], lnug[k], kernel=self.kernel)
            ghat[k] = c0k @ self._aux['CinvMs'][k]
            gvar[k] = c00 - np.sum((c0k @ self._aux['Tks'][k]) * c0k, axis=1)
        self.ghat, self.gvar = ghat, gvar
        use_std = self.rep_standardize_ybar
        std = self.ybar_std[:, 0] if use_std else np.ones(self.p)
        s_sqrt = np.sqrt(np.exp(ls2b)) / std
        s_var = np.exp(ls2b) / std ** 2
        psi_m = self.phi * s_sqrt[:, None]
        pm = psi_m @ ghat
        cv = (psi_m ** 2) @ gvar
        pv = cv + s_var[:, None]
        if use_std:
            out = (pm * self.ybar_std + self.ybar_mean, pv * self.ybar_std ** 2, cv * self.ybar_std ** 2)
        else:
            out = (pm, pv, cv)
        if return_fullcov:
            return out + (None,)
        return out


# --------------------------------------------------------------------------------------
# metrics (evaluation.py:5-63) -- used by the KAT only
# --------------------------------------------------------------------------------------
def rmse(y, m):
    return float(np.sqrt(np.mean((y - m) ** 2)))


def normalized_rmse(y, m):
    rng = (np.max(y, axis=1) - np.min(y, axis=1)).reshape(y.shape[0], 1)
    return float(np.sqrt(np.mean(((y - m) / rng) ** 2)))


def intervalstats(y, m, v):
    import scipy.stats as sps
    lo = m + np.sqrt(v) * sps.norm.ppf(0.025)
    hi = m + np.sqrt(v) * sps.norm.ppf(0.975)
    return float(np.mean(np.logical_and(y <= hi, y >= lo))), float(np.mean(hi - lo))


def dss_diag(y, m, v):
    n = y.shape[1]
    return float(sum(np.log(v[:, i]).sum() + ((y[:, i] - m[:, i]) ** 2 / v[:, i]).sum() for i in range(n)) / n)
