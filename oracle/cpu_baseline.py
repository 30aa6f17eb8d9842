"""CPU timing legs for bench.py's `cpu_baseline`  --  TEST / MEASUREMENT INFRASTRUCTURE ONLY (see lcgp_oracle.py).

`eigh_form_component` is the reference's algorithm for ONE latent component: Matern32 build ->
tf.linalg.eigh -> two dense n x n x n products -> reductions (lcgp.py:651-661), differentiated by reverse-mode
autodiff (what gpflow's Scipy wrapper does with tf.GradientTape, lcgp.py:538-539); here with torch on the host
cores because TensorFlow is not installable.  `chol_form_component` is the same quantity through the Cholesky
form with the closed-form gradient (the algorithm the GPU path runs), timed so that the algorithmic and the
hardware speed-up can be told apart."""
import time

import numpy as np

from . import lcgp_oracle as orc


def eigh_form_component(x, y, phi_k, d_k, ell, scale, nug, ls2_built, threads=None):
    """One NLL+grad evaluation of component k in the reference's eigh form; returns (seconds, value, grads)."""
    import torch
    if threads:
        torch.set_num_threads(int(threads))
    xt = torch.as_tensor(np.asarray(x, np.float64))
    yt = torch.as_tensor(np.asarray(y, np.float64))
    n = xt.shape[0]
    ell_t = torch.tensor(np.asarray(ell, np.float64), requires_grad=True)
    s_t = torch.tensor(float(scale), dtype=torch.float64, requires_grad=True)
    v_t = torch.tensor(float(nug), dtype=torch.float64, requires_grad=True)
    e_t = torch.tensor(np.asarray(ls2_built, np.float64), requires_grad=True)
    phi_t = torch.as_tensor(np.asarray(phi_k, np.float64))
    t0 = time.perf_counter()
    a = xt / ell_t
    c0 = torch.ones((n, n), dtype=torch.float64)
    v = torch.zeros((n, n), dtype=torch.float64)
    for j in range(xt.shape[1]):                               # covmat.py:37-41
        s = (a[:, j].reshape(-1, 1) - a[:, j]).abs()
        c0 = c0 * (1 + s)
        v = v - s
    c0 = c0 * torch.exp(v)
    nt = v_t / (1 + v_t)
    ck = s_t * ((1 - nt) * c0 + nt * torch.eye(n, dtype=torch.float64))
    wk, uk = torch.linalg.eigh(ck)                              # lcgp.py:652
    qk = uk @ (torch.diag(1 / (d_k + 1 / wk)) @ uk.T)           # lcgp.py:654
    psi = phi_t / torch.sqrt(torch.exp(e_t))
    pk = psi.reshape(-1, 1) @ psi.reshape(1, -1)
    yqk = yt @ qk
    ypk = yt.T @ pk.T
    val = 0.5 * torch.sum(torch.log(1 + d_k * wk)) - 0.5 * torch.sum(yqk * ypk.T)
    val.backward()
    dt = time.perf_counter() - t0
    return dt, float(val.detach()), dict(ell=ell_t.grad.numpy(), scale=float(s_t.grad), nug=float(v_t.grad),
                                         ls2=e_t.grad.numpy())


def chol_form_component(x, y, phi_k, d_k, ell, scale, nug, ls2_built, threads=None):
    """Same component in the Cholesky form with closed-form gradients, on all host threads
    (torch CPU: MKL potrf / potri, threaded elementwise passes); returns (seconds, pieces)."""
    import torch
    if threads:
        torch.set_num_threads(int(threads))
    xt = torch.as_tensor(np.asarray(x, np.float64))
    yt = torch.as_tensor(np.asarray(y, np.float64))
    ell = np.asarray(ell, np.float64)
    n, d = xt.shape
    t0 = time.perf_counter()
    sig = np.exp(0.5 * np.asarray(ls2_built, np.float64))
    b = yt.T @ torch.as_tensor(np.asarray(phi_k, np.float64) / sig)
    a = xt / torch.as_tensor(ell)
    c0 = torch.ones((n, n), dtype=torch.float64)
    v = torch.zeros((n, n), dtype=torch.float64)
    for j in range(d):                                         # covmat.py:37-41
        s = (a[:, j].reshape(-1, 1) - a[:, j]).abs_()
        c0.mul_(1 + s)
        v.sub_(s)
    c0.mul_(v.exp_())
    nt = nug / (1.0 + nug)
    amat = (d_k * scale * (1.0 - nt)) * c0
    amat.diagonal().add_(1.0 + d_k * scale * nt)
    low = torch.linalg.cholesky(amat)
    half_logdet = float(torch.log(low.diagonal()).sum())
    ainv = torch.cholesky_inverse(low)
    z = ainv @ b
    gmat = (0.5 * d_k) * ainv - 0.5 * torch.outer(z, z)
    gc0 = gmat * c0                                            # shared by every parameter's contraction
    g_ell = np.empty(d)
    for j in range(d):
        sj = (a[:, j].reshape(-1, 1) - a[:, j]).abs_()
        g_ell[j] = scale * (1.0 - nt) / ell[j] * float((gc0 * (sj * sj / (1.0 + sj))).sum())
    tr_g = float(gmat.diagonal().sum())
    g_c0 = float(gc0.sum())
    g_scale = (1.0 - nt) * g_c0 + nt * tr_g
    g_nug = scale * (tr_g - g_c0) / (1.0 + nug) ** 2
    quad = float(b @ (b - z))
    gsig = (yt @ (b - z)).numpy()
    dt = time.perf_counter() - t0
    return dt, dict(half_logdet=half_logdet, quad=quad, g_ell=g_ell, g_scale=g_scale, g_nug=g_nug, gsig=gsig,
                    value=half_logdet - quad / (2.0 * d_k))
