"""Prediction metrics (numpy only), same definitions as the reference's evaluation.py:5-63."""
import numpy as np
import scipy.stats as sps


def rmse(y, ypredmean):
    """Root mean squared error."""
    return np.sqrt(np.mean((np.asarray(y) - np.asarray(ypredmean)) ** 2))


def normalized_rmse(y, ypredmean):
    """RMSE with every output row normalised by its range."""
    y = np.asarray(y)
    span = (y.max(axis=1) - y.min(axis=1))[:, None]
    return np.sqrt(np.mean(((y - np.asarray(ypredmean)) / span) ** 2))


def dss(y, ypredmean, ypredcov, use_diag):
    """Dawid-Sebastiani score (Gneiting et al. 2007, Eq. 25), averaged over the n points.
    use_diag: ypredcov is (p, n) variances; otherwise (p, p, n) covariances."""
    y, mu, cov = np.asarray(y), np.asarray(ypredmean), np.asarray(ypredcov)
    n = y.shape[1]
    total = 0.0
    for i in range(n):
        res = y[:, i] - mu[:, i]
        if use_diag:
            total += np.log(cov[:, i]).sum() + (res * res / cov[:, i]).sum()
        else:
            w, u = np.linalg.eigh(cov[:, :, i])
            total += np.linalg.slogdet(cov[:, :, i])[1] + (((res @ u) / np.sqrt(w)) ** 2).sum()
    return total / n


def intervalstats(y, ypredmean, ypredvar):
    """Empirical coverage and mean length of the central 95 % interval."""
    y, mu, sd = np.asarray(y), np.asarray(ypredmean), np.sqrt(np.asarray(ypredvar))
    lo = mu + sd * sps.norm.ppf(0.025)
    hi = mu + sd * sps.norm.ppf(0.975)
    return np.mean(np.logical_and(y <= hi, y >= lo)), np.mean(hi - lo)
