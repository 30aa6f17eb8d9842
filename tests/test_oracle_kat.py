"""Pins the CPU oracle against the reference's stored notebook outputs."""
import numpy as np
import pytest

from oracle import lcgp_oracle as orc
from tests import kat_data as kd


@pytest.fixture(scope="module")
def kat_model():
    xtr, ytr, xte, ytrue = kd.kat_dataset()
    m = orc.OracleLCGP(y=ytr, x=xtr, q=3, diag_error_structure=[1, 1, 1], robust_mean=True, submethod='rep')
    return m, xtr, ytr, xte, ytrue


def test_kat_dataset_matches_notebook_cell12():
    xtr, ytr, _, _ = kd.kat_dataset()
    assert xtr.shape == (kd.KAT_N_TOTAL, 1) and ytr.shape == (3, kd.KAT_N_TOTAL)
    xu, cnt = np.unique(xtr[:, 0], return_counts=True)
    assert len(xu) == kd.KAT_N_UNIQUE
    assert list(cnt[:8]) == kd.KAT_FIRST_COUNTS
    assert cnt.min() == 1 and cnt.max() == 20 and abs(cnt.mean() - 4.85) < 1e-12


def test_kat1_basis_matches_notebook_digits(kat_model):
    m = kat_model[0]
    # notebook prints 8 significant decimals
    np.testing.assert_allclose(m.diag_D, kd.KAT_DIAG_D, rtol=0, atol=5e-9)
    np.testing.assert_allclose(np.var(m.g, axis=1), kd.KAT_VAR_G, rtol=0, atol=5e-9)


def test_np_median_would_not_match(kat_model):
    """The 'nearest' percentile is what pins KAT-1; np.median gives different diag_D."""
    m = kat_model[0]
    c = np.median(m.ybar, axis=1, keepdims=True)
    s = np.median(np.abs(m.ybar - c), axis=1, keepdims=True)
    _, _, dd, _ = orc.init_basis((m.ybar - c) / s, m.n, 3, None)
    assert np.max(np.abs(dd - kd.KAT_DIAG_D)) > 1e-3


def test_kat2_fit_predict_matches_notebook(kat_model):
    m, xtr, ytr, xte, ytrue = kat_model
    before = m.loss()
    m.fit()
    after = m.loss()
    assert after < before
    lLmb, _, ls2, _ = m.get_param()
    np.testing.assert_allclose(lLmb[:, 0], kd.KAT_LENGTHSCALES, rtol=1e-3)
    np.testing.assert_allclose(ls2, kd.KAT_LSIGMA2S, atol=1e-3)
    mean, pvar, cvar = m.predict(xte)
    assert abs(orc.rmse(ytrue, mean) - kd.KAT_RMSE) < 5e-5
    assert abs(orc.normalized_rmse(ytrue, mean) - kd.KAT_NRMSE) < 5e-5
    cover, width = orc.intervalstats(ytrue, mean, cvar)
    assert abs(cover - kd.KAT_COVER) < 5e-4
    assert abs(width - kd.KAT_WIDTH) < 5e-5
    assert abs(orc.dss_diag(ytrue, mean, cvar) - kd.KAT_DSS) < 2e-4
