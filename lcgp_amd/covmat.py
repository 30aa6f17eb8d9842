"""Matern32 with the reference's signature (covmat.py:5-55), evaluated by the HIP kernel `cross_kernel`; and the
squared-exponential product kernel BASELINE.json's north star names (the reference has none: an extension, parity unpinned)."""
from __future__ import annotations

import numpy as np
import torch


def _as2d(a):
    if isinstance(a, torch.Tensor):
        a = a.detach().cpu().numpy()
    return np.asarray(a, dtype=np.float64)


def Matern32(x1, x2, llmb, llmb0, lnug, diag_only: bool = False, *, _kernel: str = 'matern32'):
    """
    Returns the Matern 3/2 covariance matrix (separable product form, no sqrt(3) factor).

    :param x1: (n1, d)   :param x2: (n2, d)
    :param llmb: lengthscale per dimension (constrained value)   :param llmb0: scale
    :param lnug: nugget parameter, nugget = lnug / (1 + lnug)
    :param diag_only: return only the diagonal (requires x1 ~ x2)
    :return: CPU float64 tensor (n1, n2), or (n1,) for diag_only
    """
    x1 = _as2d(x1)
    x2 = _as2d(x2)
    assert x1.ndim == 2, 'input x1 should be 2-dimensional, (n_param, dim_param)'
    assert x2.ndim == 2, 'input x2 should be 2-dimensional, (n_param, dim_param)'
    assert x1.shape[1] == x2.shape[1], 'the dim_param of input x1 and x2 should be the same.'
    d = x1.shape[1]
    if diag_only:
        assert np.all(np.abs(x1 - x2) <= (1e-6 + 1e-6 * np.abs(x2))), \
            'diag_only should only be called when x1 and x2 are identical.'
        return torch.as_tensor(float(np.asarray(_as2d(llmb0)).reshape(-1)[0]) * np.ones(x1.shape[0]))
    ell = np.broadcast_to(np.asarray(_as2d(llmb), np.float64).reshape(-1), (d,)).copy()
    scale = float(np.asarray(_as2d(llmb0)).reshape(-1)[0])
    nug = float(np.asarray(_as2d(lnug)).reshape(-1)[0])
    same = (x1.shape == x2.shape) and bool(np.all(x1 == x2))
    from .engine import matern32_device
    return torch.as_tensor(matern32_device(x1, x2, ell, scale, nug, same, kernel=_kernel))


def SquaredExponential(x1, x2, llmb, llmb0, lnug, diag_only: bool = False):
    """
    The squared-exponential product kernel with Matern32's signature and nugget / scale structure:
        C = scale ((1 - nt) exp(-1/2 sum_j ((x1_j - x2_j) / llmb_j)^2) + nt [x1 is x2]),   nt = lnug / (1 + lnug).
    The reference has no such kernel (covmat.py:5-55 holds Matern32 only); `LCGP(..., kernel='se')` uses it.
    """
    return Matern32(x1, x2, llmb, llmb0, lnug, diag_only, _kernel='se')
