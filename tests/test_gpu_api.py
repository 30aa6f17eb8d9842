"""The reference's own behavioural tests (test_training.py, test_rep.py sections 3-4, test_coverage_gaps.py
prediction parts) run against the HIP path."""
import copy

import numpy as np
import pytest

from lcgp_amd import LCGP

pytestmark = pytest.mark.gpu


def _rep_data(seed=0, n_unique=20, p=4, d=2, reps=3):
    rng = np.random.default_rng(seed)
    xu = rng.uniform(0, 1, (n_unique, d))
    return np.tile(xu, (reps, 1)), rng.standard_normal((p, n_unique * reps)), xu


def test_fit_predict_get_param_full():
    x = np.linspace(0, 1, 40)
    y = np.reshape(copy.copy(x), (1, 40))
    model = LCGP(y=y, x=x, submethod='full')
    model.fit()
    out = model.predict(x0=x)
    assert out[0].shape == (1, 40)
    assert len(model.get_param()) == 4


@pytest.mark.parametrize('n_unique,reps,p,d', [(20, 3, 4, 2), (15, 4, 3, 1)])
def test_rep_fit_and_predict(n_unique, reps, p, d):
    x, y, xu = _rep_data(n_unique=n_unique, p=p, d=d, reps=reps, seed=42)
    model = LCGP(y=y, x=x, submethod='rep')
    before = float(model.loss())
    model.fit()
    assert float(model.loss()) <= before + 1e-3
    for var in model.trainable_variables:
        assert np.all(np.isfinite(var.numpy())), var.name
    x0 = np.random.default_rng(5).uniform(0, 1, (10, d))
    ypred, ypredvar, yconfvar = model.predict(x0)
    assert ypred.shape == (p, 10) and ypredvar.shape == (p, 10) and yconfvar.shape == (p, 10)
    assert np.all(ypredvar.numpy() > 0)
    assert all(np.all(np.isfinite(t.numpy())) for t in (ypred, ypredvar, yconfvar))
    assert np.all(yconfvar.numpy() <= ypredvar.numpy() + 1e-9)
    ybar = model.ybar.numpy()
    pred_rmse = np.sqrt(np.mean((model.predict(xu)[0].numpy() - ybar) ** 2))
    assert pred_rmse < 2 * np.sqrt(np.mean((ybar - ybar.mean()) ** 2))
    assert model.ghat.shape == (model.q, xu.shape[0]) and model.gvar.shape == (model.q, xu.shape[0])
    out = model.predict(x0, return_fullcov=True)
    assert out[3] is None


def test_rep_non_standardized():
    x, y, _ = _rep_data()
    model = LCGP(y=y, x=x, submethod='rep', rep_standardize_ybar=False)
    assert np.isfinite(float(model.neglpost_rep()))
    model.fit()
    out = model.predict(np.random.default_rng(12).uniform(0, 1, (10, 2)), return_fullcov=True)
    assert out[3] is None and all(np.all(np.isfinite(t.numpy())) for t in out[:3])


def test_full_cov_diagonal_equals_predvar():
    rng = np.random.default_rng(0)
    x = rng.uniform(0, 1, (40, 2))
    y = rng.standard_normal((3, 40))
    model = LCGP(y=y, x=x, submethod='full')
    model.fit()
    x0 = np.random.default_rng(11).uniform(0, 1, (8, 2))
    ypred, ypredvar, yconfvar, cov = model.predict(x0, return_fullcov=True)
    assert cov.shape == (8, 3, 3) and np.all(np.isfinite(cov.numpy()))
    np.testing.assert_allclose(np.diagonal(cov.numpy(), axis1=1, axis2=2).T, ypredvar.numpy(), rtol=1e-5, atol=1e-6)


def test_second_fit_invalidates_prediction_caches():
    rng = np.random.default_rng(1)
    x = rng.uniform(0, 1, (50, 2))
    y = rng.standard_normal((3, 50))
    model = LCGP(y=y, x=x)
    p0 = model.predict(x[:5])[0].numpy()
    model.fit()
    p1 = model.predict(x[:5])[0].numpy()
    assert np.max(np.abs(p0 - p1)) > 1e-8


def test_rccl_collectives_on_a_one_rank_group():
    """The N > 1 bench runs over RCCL (backend "nccl"), which this one-GPU box cannot form with several ranks; a 1-rank
    RCCL group still goes through the same code: host vector -> device -> all_reduce / broadcast -> host."""
    import os
    import tempfile
    import torch
    import torch.distributed as dist
    from lcgp_amd import dist as ldist
    if dist.is_initialized():
        pytest.skip('a process group already exists in this process')
    store = tempfile.NamedTemporaryFile(delete=False)
    store.close()
    try:
        dist.init_process_group('nccl', init_method='file://' + store.name, rank=0, world_size=1,
                                device_id=torch.device('cuda', torch.cuda.current_device()))
        v = np.arange(131, dtype=np.float64) / 7.0
        got = ldist._all_reduce_impl(v)
        assert got.dtype == np.float64 and np.array_equal(got, v)
        got = ldist._broadcast_impl(v.reshape(1, -1), 0)
        assert got.shape == (1, 131) and np.array_equal(got.ravel(), v)
        dist.barrier()
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()
        if os.path.exists(store.name):
            os.unlink(store.name)
