"""The example harness (reference docs/call_model.py:5-126) on CPU: call pattern, return shapes, the harness's own
summary functions, and the KAT-2 notebook run driven THROUGH the harness with `lcgp_amd.evaluation` as the metrics
(the device engine is replaced by the test-only oracle stand-in; the GPU version is in tests/test_gpu_parity.py)."""
import numpy as np
import pytest

from lcgp_amd import evaluation, harness, synth
from tests import kat_data as kd
from tests.helpers import patch_engine


def _kat_run(**kw):
    xtr, ytr, xte, ytrue = kd.kat_dataset()
    data = dict(xtrain=xtr, ytrain=ytr, xtest=xte, ytest=ytrue, ytrue=ytrue)
    run = harness.LCGPRun(runno='kat2', data=data, submethod='rep', num_latent=3, err_struct=[1, 1, 1], **kw)
    return run, ytrue


def test_call_pattern_and_attributes():
    run, ytrue = _kat_run()
    assert run.model is None and run.modelname == 'LCGP_robust' and run.runno == 'kat2'
    assert run.n == kd.KAT_N_TOTAL and run.num_output == 3 and run.ytrue is ytrue and not hasattr(run, 'ystd')
    assert harness.LCGPRun(runno='x', data=run.data, robust=False).modelname == 'LCGP'
    run.define_model()
    assert run.model.submethod == 'rep' and run.model.q == 3 and run.model.robust_mean
    assert harness.SuperRun(runno='s', data=run.data).train() is None


def test_kat2_through_the_harness_matches_the_reference_notebook():
    run, ytrue = _kat_run()
    run.define_model()
    patch_engine(run.model)
    run.train()
    mean, pvar, cvar = run.predict()
    assert mean.shape == pvar.shape == cvar.shape == (3, 400) and isinstance(mean, np.ndarray)
    assert abs(evaluation.rmse(ytrue, mean) - kd.KAT_RMSE) < 5e-5
    assert abs(evaluation.normalized_rmse(ytrue, mean) - kd.KAT_NRMSE) < 5e-5
    cover, width = evaluation.intervalstats(ytrue, mean, cvar)
    assert abs(cover - kd.KAT_COVER) < 5e-4 and abs(width - kd.KAT_WIDTH) < 5e-5
    assert abs(evaluation.dss(ytrue, mean, cvar, use_diag=True) - kd.KAT_DSS) < 2e-4
    # variants of predict(): training inputs, transposed output, full covariance slot (None on the replicated path)
    tr = run.predict(train=True, as_pxn=True)
    assert tr[0].shape == (kd.KAT_N_TOTAL, 3)
    out = run.predict(return_fullcov=True)
    assert len(out) == 4 and out[3] is None


def test_full_path_through_the_harness_returns_the_full_covariance():
    x, y = synth.make_full(77, 40, 2, 3, 2)
    data = dict(xtrain=x[:30], ytrain=y[:, :30], xtest=x[30:], ytest=y[:, 30:], ystd=np.ones((3, 1)))
    run = harness.LCGPRun(runno=1, data=data, num_latent=2, robust=False)
    assert run.ystd.shape == (3, 1)
    run.define_model()
    patch_engine(run.model)
    run.train()
    mean, pvar, cvar, cov = run.predict(return_fullcov=True)
    assert mean.shape == (3, 10) and cov.shape == (10, 3, 3)
    np.testing.assert_allclose(np.diagonal(cov, axis1=1, axis2=2).T, pvar, rtol=1e-8)


def test_harness_summaries():
    rng = np.random.default_rng(0)
    y = rng.standard_normal((3, 50)) * np.array([[1.0], [5.0], [0.2]])
    m = y + 0.1 * rng.standard_normal((3, 50))
    v = np.full((3, 50), 0.02)
    assert harness.rmse(y, y) == 0.0 and abs(harness.rmse(y, m) - np.sqrt(np.mean((y - m) ** 2))) < 1e-15
    per_row = np.sqrt(np.mean((y - m) ** 2, axis=1))
    assert abs(harness.normalized_rmse(y, m) - np.mean(per_row / np.ptp(y, axis=1))) < 1e-15
    assert abs(harness.normalized_rmse(y, m, method='std') - np.mean(per_row / np.std(y, axis=1))) < 1e-15
    const = np.ones((2, 5))
    assert harness.normalized_rmse(const, const + 1.0) == 1.0          # zero spread counts as 1
    with pytest.raises(ValueError):
        harness.normalized_rmse(y, m, method='iqr')
    cover, width = harness.intervalstats(y, m, v)
    assert 0.0 <= cover <= 1.0 and abs(width - 2 * 1.96 * np.sqrt(0.02)) < 1e-15
    assert harness.intervalstats(y, y, v, z=0.0) == (1.0, 0.0)
    assert abs(harness.dss(y, m, v) - np.mean((y - m) ** 2 / v + np.log(v))) < 1e-15
    assert np.isfinite(harness.dss(y, m, np.zeros_like(v)))             # variance floored at 1e-12
