"""CPU tests of the host side of lcgp_amd: constructor contract of the reference's tests
(test_initialize.py, test_standardization.py, test_rep.py sections 1-2, test_coverage_gaps.py), exact
preprocessing vs the oracle and the notebook KAT-1, and the assembly / chain rule around the hot path
(checked through the test-only OracleEngine stand-in)."""
import copy

import numpy as np
import pytest
import torch

from lcgp_amd import LCGP
from lcgp_amd import synth
from lcgp_amd.params import SoftClip
from oracle import lcgp_oracle as orc
from tests import kat_data as kd
from tests.helpers import patch_engine


def _rep_data(seed=0, n_unique=20, p=4, d=2, reps=3):
    rng = np.random.default_rng(seed)
    xu = rng.uniform(0, 1, (n_unique, d))
    return np.tile(xu, (reps, 1)), rng.standard_normal((p, n_unique * reps)), xu


# ---- constructor contract (reference test_initialize.py) -------------------------------------------------
def test_1d_y_rejected_1d_x_accepted():
    x = np.linspace(0, 1, 40)
    with pytest.raises(AssertionError):
        LCGP(y=copy.copy(x), x=x)
    m = LCGP(y=x.reshape(1, 40), x=x)
    assert m.x.shape == (40, 1) and int(m.n.numpy()) == 40 and m.q == 1
    m.tx_x(m.x)
    m.tx_y(m.y)
    print(m)


@pytest.mark.parametrize('err', [[2, 1], [1, 1, 1], None, [1, 2]])
def test_valid_error_structures(err):
    LCGP(y=np.random.randn(3, 40), x=np.random.randn(40, 5), diag_error_structure=err)


@pytest.mark.parametrize('err', [[1, 1], [0, 1, 1], [2, 2]])
def test_invalid_error_structures(err):
    with pytest.raises(AssertionError):
        LCGP(y=np.random.randn(3, 40), x=np.random.randn(40, 5), diag_error_structure=err)


def test_q_and_threshold_exclusive_and_threshold_rule():
    x = np.linspace(0, 1, 40)
    y = np.random.default_rng(0).standard_normal((3, 40))
    with pytest.raises(ValueError):
        LCGP(y=y, x=x, q=2, var_threshold=0.9)
    m = LCGP(y=y, x=x, var_threshold=0.9)
    s = np.linalg.svd(m.y.numpy(), compute_uv=False)
    cum = np.cumsum(s ** 2) / np.sum(s ** 2)
    assert m.q == int(np.argmax(cum > 0.9) + 1)


def test_mismatch_and_bad_submethod():
    with pytest.raises(AssertionError):
        LCGP(y=np.random.randn(3, 25), x=np.linspace(0, 1, 40))
    with pytest.raises(ValueError):
        LCGP(y=np.random.randn(3, 40), x=np.linspace(0, 1, 40), submethod='null')


def test_dtype_aliases_are_normalised_once():
    """'f32' / 'f64' are accepted like the engine accepts them and mean the same model (the float32 logic of fit() and of the
    evaluation is keyed on the NORMALISED name); anything else is refused at construction"""
    x, y = synth.make_full(12, 30, 2, 3, 2)
    assert LCGP(y=y, x=x, dtype='f32')._dtype == 'float32' == LCGP(y=y, x=x, dtype='float32')._dtype
    assert LCGP(y=y, x=x, dtype='f64')._dtype == 'float64' == LCGP(y=y, x=x)._dtype
    with pytest.raises(ValueError):
        LCGP(y=y, x=x, dtype='float16')


def test_accepts_torch_inputs():
    x = torch.rand(30, 2, dtype=torch.float32)
    y = torch.randn(3, 30)
    m = LCGP(y=y, x=x)
    assert m.x.dtype == torch.float64 and m.y.dtype == torch.float64


# ---- standardisation (reference test_standardization.py) ----------------------------------------------------
def test_standard_x_and_xnorm_match_reference_definition():
    x = np.random.default_rng(1).uniform(-2, 5, (60, 3))
    x[7] = x[3]                                   # duplicated rows -> zero distances are excluded
    xs, xmin, xmax, xo, xnorm = LCGP.init_standard_x(torch.as_tensor(x))
    assert xs.shape == (60, 3) and float(xs.min()) == 0.0 and float(xs.max()) == 1.0
    np.testing.assert_allclose(xnorm.numpy(), orc.xnorm_pairs(x), rtol=1e-12)
    assert np.all(xnorm.numpy() > 0)


@pytest.mark.parametrize('robust', [True, False])
def test_standard_y_roundtrip_and_oracle(robust):
    x, y = synth.make_full(3, 51, 2, 4, 3)
    m = LCGP(y=y, x=x, robust_mean=robust)
    np.testing.assert_allclose(m.tx_y(m.y).numpy(), y, atol=1e-10)
    c, s = orc.center_spread(y, robust, guard_zero=False)
    np.testing.assert_array_equal(m.ymean.numpy(), c)
    np.testing.assert_array_equal(m.ystd.numpy(), s)


# ---- replication preprocessing (reference test_rep.py 1-2, test_coverage_gaps.py) ------------------------------
def test_rep_structures():
    x, y, xu = _rep_data(n_unique=15, reps=4, p=3, d=2)
    m = LCGP(y=y, x=x, submethod='rep')
    for attr in ['x_unique', 'x_unique_s', 'ybar', 'ybar_s', 'ybar_mean', 'ybar_std', 'r', 'R', 'group_ids']:
        assert hasattr(m, attr)
    assert int(m.n.numpy()) == 15 and m._rep_initialized is True
    assert np.all(m.r.numpy() == 4)
    np.testing.assert_array_equal(torch.diagonal(m.R).numpy(), m.r.numpy().astype(float))
    _, inv, _ = np.unique(x, axis=0, return_inverse=True, return_counts=True)
    for i in range(15):
        np.testing.assert_allclose(m.ybar.numpy()[:, i], y[:, np.asarray(inv).reshape(-1) == i].mean(axis=1), atol=1e-10)
    xs = m.x_unique_s.numpy()
    assert xs.min() >= -1e-9 and xs.max() <= 1 + 1e-9
    out = m.preprocess(x_raw=x, y_raw=y)
    assert len(out) == 12 and int(out[9].numpy()) == 15 and int(out[10].numpy()) == 2 and int(out[11].numpy()) == 3
    assert out[4].shape == (15, 15) and out[6].shape == (3, 15)
    assert int(m.preprocess()[9].numpy()) == 15


def test_ensure_replication_spy():
    x, y, _ = _rep_data(n_unique=15, reps=4, p=3)
    m = LCGP(y=y, x=x, submethod='rep')
    calls = {'n': 0}
    orig = m.preprocess

    def spy(*a, **k):
        calls['n'] += 1
        return orig(*a, **k)
    m.preprocess = spy
    m._ensure_replication()
    assert calls['n'] == 0
    m._rep_initialized = False
    m._ensure_replication()
    assert calls['n'] == 1 and m._rep_initialized is True


def test_phi_input_fallbacks_and_nonrobust_center():
    x, y, _ = _rep_data(n_unique=15, reps=4, p=3)
    m = LCGP(y=y, x=x, submethod='rep', rep_standardize_ybar=False, robust_mean=False)
    np.testing.assert_allclose(m._get_phi_input().numpy(), m.ybar.numpy())
    c, s = m._compute_center_spread_tf(m.ybar)
    np.testing.assert_allclose(c.numpy()[:, 0], m.ybar.numpy().mean(axis=1))
    np.testing.assert_allclose(s.numpy()[:, 0], m.ybar.numpy().std(axis=1))
    m2 = LCGP(y=y, x=x, submethod='rep')
    del m2.ybar_s
    del m2.ybar
    np.testing.assert_allclose(m2._get_phi_input().numpy(), m2.y.numpy())


# ---- exact preprocessing vs the oracle and the notebook -------------------------------------------------------
def test_kat1_through_the_product_host_side():
    xtr, ytr, _, _ = kd.kat_dataset()
    m = LCGP(y=ytr, x=xtr, q=3, diag_error_structure=[1, 1, 1], submethod='rep')
    np.testing.assert_allclose(m.diag_D.numpy(), kd.KAT_DIAG_D, atol=5e-9, rtol=0)
    np.testing.assert_allclose(np.var(m.g.numpy(), axis=1), kd.KAT_VAR_G, atol=5e-9, rtol=0)
    assert int(m.n.numpy()) == kd.KAT_N_UNIQUE and int(m.r.numpy().sum()) == kd.KAT_N_TOTAL


@pytest.mark.parametrize('mode', ['full', 'rep'])
def test_initial_state_equals_oracle(mode):
    if mode == 'full':
        x, y = synth.make_full(4, 37, 3, 5, 3)
        kw = dict(q=3, diag_error_structure=[2, 3])
    else:
        x, y = synth.make_rep(4, 21, 3, 2, 4, 4)
        kw = {}
    m = LCGP(y=y, x=x, submethod=mode, **kw)
    o = orc.OracleLCGP(y=y, x=x, submethod=mode, **kw)
    np.testing.assert_allclose(m.diag_D.numpy(), o.diag_D, rtol=1e-12)
    np.testing.assert_allclose(np.abs(m.phi.numpy()), np.abs(o.phi), rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(m._get_flat(), o.get_unconstrained(), rtol=1e-12, atol=1e-12)
    for a, b in zip(m.get_param(), o.get_param()):
        np.testing.assert_allclose(a.numpy(), b, rtol=1e-12)
    names = [v.name for v in m.trainable_variables]
    assert len(names) == 4 and all(np.all(np.isfinite(v.numpy())) for v in m.trainable_variables)


def test_softclip_matches_oracle_formulas():
    for lo, hi in (orc.LLMB_BOUNDS, orc.LLMB0_BOUNDS, orc.LNUG_BOUNDS):
        sc = SoftClip(lo, hi)
        u = np.linspace(-4.0, 6.0, 21)
        np.testing.assert_allclose(sc.forward(u), orc.softclip_forward(u, lo, hi), rtol=1e-13, atol=hi * 1e-15)
        np.testing.assert_allclose(sc.dforward(u), orc.softclip_grad(u, lo, hi), rtol=1e-12)
        v = sc.forward(u)
        np.testing.assert_allclose(sc.forward(sc.inverse(v)), v, rtol=1e-9, atol=hi * 1e-15)


def test_flat_softclip_pass_equals_the_per_parameter_transforms_bit_for_bit():
    """The evaluation loop forms the constrained values and their derivatives of the three bounded blocks in one vectorised pass
    (LCGP._flat_transform); element by element it is the arithmetic of SoftClip.forward / .dforward (lcgp.py:181-211 through the
    bijector), so the results are the same bits -- and a model whose transform a caller has replaced takes the general path."""
    x, y = synth.make_full(8, 40, 3, 5, 3)
    m = LCGP(y=y, x=x, q=3)
    rng = np.random.default_rng(3)
    for par in (m.lLmb, m.lLmb0, m.lnugGPs):
        par.unconstrained = rng.normal(0.0, 3.0, par.unconstrained.shape)
    v, dv = m._flat_transform()
    want_v = np.concatenate([m.lLmb.numpy().reshape(-1), m.lLmb0.numpy(), m.lnugGPs.numpy()])
    want_d = np.concatenate([par.transform.dforward(par.unconstrained).reshape(-1) for par in (m.lLmb, m.lLmb0, m.lnugGPs)])
    assert np.array_equal(v, want_v) and np.array_equal(dv, want_d)

    class Shifted(SoftClip):          # any other bijector type
        pass
    m.lLmb0.transform = Shifted(m.lLmb0.transform.low, m.lLmb0.transform.high)
    assert m._flat_transform() is None


# ---- host assembly around the hot path, through the stand-in engine -----------------------------------------------
@pytest.mark.parametrize('mode,kw', [('full', dict(q=3)), ('full', dict(q=2, diag_error_structure=[1, 3], robust_mean=False)),
                                     ('rep', {}), ('rep', dict(rep_standardize_ybar=False))])
def test_assembly_and_chain_rule_equal_oracle(mode, kw):
    if mode == 'full':
        x, y = synth.make_full(6, 45, 2, 4, 3)
    else:
        x, y = synth.make_rep(6, 18, 3, 2, 4, 4)
    m = patch_engine(LCGP(y=y, x=x, submethod=mode, **kw))
    o = orc.OracleLCGP(y=y, x=x, submethod=mode, **kw)
    o.phi = m.phi.numpy().copy()
    for u in synth.param_points(6, o.get_unconstrained()):
        v1, g1 = m.loss_and_grad(u)
        v2, g2 = o.loss_and_grad_unconstrained(u)
        assert abs(v1 - v2) <= 1e-12 * max(1.0, abs(v2))
        np.testing.assert_allclose(g1, g2, rtol=1e-10, atol=1e-12 * np.max(np.abs(g2)))
    assert abs(float(m.loss()) - o.loss()) <= 1e-12 * abs(o.loss())


@pytest.mark.parametrize('mode', ['full', 'rep'])
def test_predict_and_caches_equal_oracle_through_stand_in(mode):
    if mode == 'full':
        x, y = synth.make_full(8, 40, 2, 3, 3)
    else:
        x, y = synth.make_rep(8, 16, 3, 2, 4, 4)
    m = patch_engine(LCGP(y=y, x=x, submethod=mode))
    o = orc.OracleLCGP(y=y, x=x, submethod=mode)
    o.phi = m.phi.numpy().copy()
    u = synth.param_points(8, o.get_unconstrained())[1]
    m._set_flat(u)
    o.set_unconstrained(u)
    x0 = np.random.default_rng(3).uniform(0, 1, (9, 2))
    got = m.predict(x0, return_fullcov=True)
    want = o.predict(x0, return_fullcov=True)
    for a, b in zip(got[:3], want[:3]):
        assert a.shape == (m.p, 9)
        np.testing.assert_allclose(a.numpy(), b, rtol=1e-8, atol=1e-10)
    if mode == 'full':
        np.testing.assert_allclose(got[3].numpy(), want[3], rtol=1e-8, atol=1e-10)
        diag = np.diagonal(got[3].numpy(), axis1=1, axis2=2).T
        np.testing.assert_allclose(diag, got[1].numpy(), rtol=1e-5, atol=1e-6)
        aux = o._aux_full()          # the caches the reference keeps as attributes (lcgp.py:708-715), its own matrices
        np.testing.assert_allclose(m.CinvMs.numpy(), aux['CinvMs'], rtol=1e-7, atol=1e-9)
        np.testing.assert_allclose(m.Ths.numpy(), aux['Ths'], rtol=1e-7, atol=1e-9)
    else:
        assert got[3] is None
        aux = o._aux_rep()
        np.testing.assert_allclose(m.CinvMs.numpy(), aux['CinvMs'], rtol=1e-7, atol=1e-9)
        np.testing.assert_allclose(m.mks.numpy(), aux['mks'], rtol=1e-7, atol=1e-9)
        np.testing.assert_allclose(m.Tks.numpy(), aux['Tks'], rtol=1e-5, atol=1e-7)
    # predicting at the training inputs adds the nugget on the diagonal (covmat.py:46-51)
    xt = (m.x_unique if mode == 'rep' else m.x_orig).numpy()
    a = m.predict(xt)[0].numpy()
    b = o.predict(xt)[0]
    np.testing.assert_allclose(a, b, rtol=1e-8, atol=1e-10)
    # cache invalidation: new parameters -> predictions change and equal the oracle's again
    u2 = synth.param_points(8, o.get_unconstrained())[2]
    m._set_flat(u2)
    o.set_unconstrained(u2)
    np.testing.assert_allclose(m.predict(x0)[0].numpy(), o.predict(x0)[0], rtol=1e-8, atol=1e-10)


def test_psi_c_value_as_the_reference_verifier_checks_it():
    """test_verification.py:138-183 (LCGPVerifier step 3): psi_c == phi^T / (exp(-lsigma2s/2) * ybar_std)[:, None]
    to 1e-10 (relative Frobenius norm), on the verifier's own problem shape (n_unique=50, reps=3, d=2, p=3 = q)."""
    rng = np.random.default_rng(0)
    xu = rng.uniform(0, 1, (50, 2))
    x = np.tile(xu, (3, 1))
    y = np.vstack([np.sin(2 * np.pi * x[:, 0]), np.cos(2 * np.pi * x[:, 1]), x[:, 0] * x[:, 1]]) \
        + 0.05 * rng.standard_normal((3, 150))
    m = patch_engine(LCGP(y=y, x=x, submethod='rep'))
    assert m.q == int(m.p) == 3
    m._compute_aux_predictive_quantities_rep()
    phi = m.phi.numpy()
    sis = np.exp(-0.5 * m.lsigma2s.numpy()) * m.ybar_std[:, 0].numpy()
    manual = phi.T / sis[:, None]
    got = m.psi_c.numpy()
    assert got.shape == (m.q, int(m.p))
    assert np.linalg.norm(got - manual) / np.linalg.norm(manual) < 1e-10
    # q != p: the reference's expression cannot broadcast (it raises there); this build scales per output instead
    m2 = patch_engine(LCGP(y=y, x=x, q=2, submethod='rep'))
    m2._compute_aux_predictive_quantities_rep()
    sis2 = np.exp(-0.5 * m2.lsigma2s.numpy()) * m2.ybar_std[:, 0].numpy()
    np.testing.assert_allclose(m2.psi_c.numpy(), m2.phi.numpy().T / sis2[None, :], rtol=1e-13)


def test_basis_reconstruction_as_the_reference_verifier_checks_it():
    """test_verification.py:89-136 (LCGPVerifier step 2) on the verifier's own generator (test_verification.py:331-341:
    np.random.seed(0), n_unique = 50, 3 replicates, d = 2, p = 3 = q).  The verifier forms phi @ g and compares it with
    ybar_s at 1e-8 -- but it only RETURNS the verdict (test_run_all never asserts, :300-328), and with the reference's own
    definitions (lcgp.py:477-479: phi = U sqrt(n) / S, g = phi^T Y) that product is U diag(n / S^2) U^T Y, not Y: the step
    reports FAIL on the reference itself.  What the definitions do guarantee, and what is pinned here at the verifier's
    1e-8: phi diag(1 / diag_D) g = Y at q = p (diag_D = n / S^2, lcgp.py:478), and the verifier's literal quantity has
    exactly the value the singular values dictate -- so this build's phi, g, diag_D are the reference's."""
    np.random.seed(0)
    x_unique = np.random.rand(50, 2)
    x = np.repeat(x_unique, 3, axis=0)
    weights = np.random.randn(2, 3)
    y = np.sin(x @ weights).T + 0.1 * np.random.randn(3, 150)
    m = patch_engine(LCGP(y=y, x=x, submethod='rep'))
    assert m.q == int(m.p) == 3
    ybar_s, phi, g, D = m.ybar_s.numpy(), m.phi.numpy(), m.g.numpy(), m.diag_D.numpy()
    n = ybar_s.shape[1]
    assert phi.shape == (3, 3) and g.shape == (3, 50) and n == 50
    sv = np.linalg.svd(ybar_s, compute_uv=False)
    np.testing.assert_allclose(D, n / sv ** 2, rtol=1e-10)
    # the identity behind the basis, at the verifier's tolerance
    assert np.linalg.norm(ybar_s - phi @ (g / D[:, None])) / np.linalg.norm(ybar_s) < 1e-8
    # the verifier's literal quantity: ||Y - phi g|| / ||Y|| = ||(n / S^2 - 1) S|| / ||S||
    literal = np.linalg.norm(ybar_s - phi @ g) / np.linalg.norm(ybar_s)
    np.testing.assert_allclose(literal, np.linalg.norm((n / sv ** 2 - 1.0) * sv) / np.linalg.norm(sv), rtol=1e-8)
    assert literal > 1e-8                                 # (the reference's step 2 prints FAIL for its own basis)
    # reduced basis (q < p): the discarded singular directions are the whole error of the corrected reconstruction
    m2 = patch_engine(LCGP(y=y, x=x, q=2, submethod='rep'))
    Y2, D2 = m2.ybar_s.numpy(), m2.diag_D.numpy()
    err = np.linalg.norm(Y2 - m2.phi.numpy() @ (m2.g.numpy() / D2[:, None])) / np.linalg.norm(Y2)
    np.testing.assert_allclose(err, np.sqrt(1.0 - np.sum(sv[:2] ** 2) / np.sum(sv ** 2)), rtol=1e-8)
    # the full path builds the same basis from the standardised outputs
    mf = patch_engine(LCGP(y=y, x=x))
    Yf, Df = mf.y.numpy(), mf.diag_D.numpy()
    assert np.linalg.norm(Yf - mf.phi.numpy() @ (mf.g.numpy() / Df[:, None])) / np.linalg.norm(Yf) < 1e-8


def test_predict_bad_submethod_keyerror_and_cache_reset():
    x, y, _ = _rep_data()
    m = patch_engine(LCGP(y=y, x=x, submethod='rep'))
    m.CinvMs = torch.full((m.q, int(m.n)), float('nan'), dtype=torch.float64)
    m.Tks = None
    m.compute_aux_predictive_quantities()
    assert m.Tks is not None and not np.any(np.isnan(m.CinvMs.numpy()))
    assert m.psi_c.shape == (m.q, int(m.p))
    m.submethod = 'bogus'
    with pytest.raises(KeyError):
        m.predict(x0=m.x_unique)
    with pytest.raises(ValueError):
        m.loss()


def test_fit_through_stand_in_reduces_loss():
    x, y, _ = _rep_data()
    m = patch_engine(LCGP(y=y, x=x, submethod='rep'))
    before = float(m.loss())
    m.fit()
    assert float(m.loss()) <= before + 1e-3
    for v in m.trainable_variables:
        assert np.all(np.isfinite(v.numpy()))


# ---- the C ABI library ---------------------------------------------------------------------------------------------
def test_c_abi_library_exports_every_declared_symbol():
    import re
    from lcgp_amd import _hip
    _hip.build_library()
    lib = _hip.load()
    header = open(_hip.HDR_PATH).read()
    declared = set(re.findall(r'\b(lcgp_[a-z0-9_]+)\s*\(', header))
    assert declared == set(_hip.SIGNATURES), declared ^ set(_hip.SIGNATURES)
    for name in declared:
        assert hasattr(lib, name)
    assert lib.lcgp_version() >= 100
    assert lib.lcgp_theta_width(6, 64) == 6 + 3 + 64 and lib.lcgp_out_width(6, 64) == 6 + 5 + 64
    import ctypes as C
    nbytes = C.c_size_t(0)
    assert lib.lcgp_workspace_bytes(0, 4096, 6, 64, 8, C.byref(nbytes)) == 0
    assert nbytes.value >= 3 * 8 * 4096 * 4096 * 8
    assert lib.lcgp_workspace_bytes(0, 4096, 127, 64, 8, C.byref(nbytes)) < 0     # d > 126 refused
    assert b'd must be' in lib.lcgp_last_error()
    assert lib.lcgp_partial_width(6, 64, 8) == 3 + 8 * 6 + 2 * 8 + 64      # ... + the lock-step guard word
    sc = _hip.default_sched()                   # schedule parameters travel per call: the library has no setters
    assert sc.outer_blocks == 0 and sc.fill_leaf > 0 and sc.syrk_small_tiles > 0
    assert not any(n.startswith(('lcgp_set', 'lcgp_shutdown')) for n in declared)


def test_launch_plan_is_built_once_by_the_caller():
    """lcgp_plan_bytes / lcgp_plan_build / lcgp_plan_info are host-only: a position-independent block that depends on
    (dtype, n, q_local, with_inverse, schedule) and nothing else"""
    import ctypes as C
    from lcgp_amd import _hip
    lib = _hip.load()
    assert lib.lcgp_version() >= 400                      # the ABI of round 5 (one plan argument, lcgp_sched without the persistent launch)
    def build(n, q, inv, **fields):
        sc = _hip.default_sched()
        for k, v in fields.items():
            setattr(sc, k, v)
        nb = C.c_size_t(0)
        assert lib.lcgp_plan_bytes(0, n, q, inv, C.byref(sc), C.byref(nb)) == 0
        buf = np.zeros(nb.value, np.uint8)
        assert lib.lcgp_plan_build(0, n, q, inv, C.byref(sc), C.c_void_p(buf.ctypes.data), nb) == 0
        v = [C.c_int(0) for _ in range(2)]
        assert lib.lcgp_plan_info(C.c_void_p(buf.ctypes.data), *[C.byref(t) for t in v]) == 0
        return buf, [t.value for t in v]
    a, ia = build(4096, 8, 1)
    b, ib = build(4096, 8, 1)
    assert ia == ib and ia[0] > 60 and ia[1] == 0          # launches; nothing of the inverse behind the factorisation at q = 8
    # (the bytes differ only where the header records the device's compute-unit count, identical here)
    assert np.array_equal(a, b)
    _, i1 = build(4096, 1, 1)
    assert i1[1] == 1                                      # one component per rank: L^-1 behind the chain
    _, i2 = build(4096, 8, 1, fill_leaf=0, fill_step=0)
    assert i2[0] != ia[0]                                  # the schedule is part of the plan
    nb = C.c_size_t(0)
    assert lib.lcgp_plan_build(0, 4096, 8, 1, None, C.c_void_p(a.ctypes.data), C.c_size_t(16)) < 0
    assert b'too small' in lib.lcgp_last_error()
    junk = np.zeros(512, np.uint8)
    assert lib.lcgp_plan_info(C.c_void_p(junk.ctypes.data), None, None) < 0


def test_native_library_is_the_one_built_from_these_sources():
    """The shared object carries the sha256 of the sources it was compiled from (lcgp_source_hash()); build() and the
    loader compare it with the tree, so a stale prebuilt binary cannot pass for the current kernels (neither here nor
    on the GPU box, where the prebuilt .so travels with the snapshot)."""
    from lcgp_amd import _hip
    _hip.build_library()
    assert _hip.binary_hash() == _hip.source_hash()
    assert _hip.loaded_hash() == _hip.source_hash()
    assert not _hip.needs_build()


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU behaviour")
def test_hot_path_fails_loudly_without_gpu():
    x, y = synth.make_full(9, 20, 2, 3, 2)
    m = LCGP(y=y, x=x)
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        m.loss()
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        m.fit()
    from lcgp_amd import Matern32
    with pytest.raises(RuntimeError):
        Matern32(x, x, np.ones(2), 1.0, 1e-3)
    with pytest.raises(AssertionError):
        Matern32(np.linspace(0, 1, 5), np.linspace(0, 1, 5), 1.0, 1.0, -12.0)


def test_drop_in_import_shim_and_metrics():
    import lcgp
    from lcgp.covmat import Matern32 as M32
    from lcgp.lcgp import LCGP as L2
    from lcgp import evaluation
    from lcgp_amd import Matern32
    assert lcgp.__all__ == ['LCGP', 'Matern32', 'test'] and lcgp.LCGP is LCGP and L2 is LCGP and M32 is Matern32
    rng = np.random.default_rng(0)
    y = rng.standard_normal((3, 50))
    m = y + 0.1 * rng.standard_normal((3, 50))
    v = np.full((3, 50), 0.01)
    assert abs(evaluation.rmse(y, y)) == 0.0
    assert abs(evaluation.rmse(y, m) - orc.rmse(y, m)) < 1e-15
    assert abs(evaluation.normalized_rmse(y, m) - orc.normalized_rmse(y, m)) < 1e-15
    c1, w1 = evaluation.intervalstats(y, m, v)
    c2, w2 = orc.intervalstats(y, m, v)
    assert abs(c1 - c2) < 1e-15 and abs(w1 - w2) < 1e-15 and 0 <= c1 <= 1
    assert abs(evaluation.dss(y, m, v, use_diag=True) - orc.dss_diag(y, m, v)) < 1e-12
    cov = np.stack([np.diag(v[:, i]) for i in range(50)], axis=2)
    assert abs(evaluation.dss(y, m, cov, use_diag=False) - evaluation.dss(y, m, v, use_diag=True)) < 1e-10


def test_fit_backs_off_from_a_non_positive_definite_trial_point():
    """A trial point at which some I + D_k C_k is not numerically positive definite (float32, SoftClip edges) must not
    abort fit(): the closure reports a huge finite value with a zero gradient and L-BFGS-B's line search backs off (the
    reference's eigendecomposition form yields a non-finite / huge value there).  loss() called directly still raises."""
    x, y = synth.make_full(41, 40, 2, 3, 2)
    m = patch_engine(LCGP(y=y, x=x, q=2))
    eng = m._get_engine()
    real = eng.evaluate_partial
    calls = {'n': 0, 'bad': 0}

    def flaky(theta_rows, guard=0.0):
        calls['n'] += 1
        part = real(theta_rows, guard)
        if calls['n'] in (3, 4):                 # two consecutive trial points "fail" (info = 7 on component 0)
            calls['bad'] += 1
            part = part.clone()
            part[1] = 7.0
        return part
    eng.evaluate_partial = flaky
    before = None
    eng.evaluate_partial = real
    before = float(m.loss())
    eng.evaluate_partial = flaky
    calls['n'] = 0
    m.fit()
    assert calls['bad'] == 2 and m.opt_result.nfev > 5
    eng.evaluate_partial = real
    assert float(m.loss()) < before
    # outside fit() the failure is an exception
    eng.evaluate_partial = lambda th, guard=0.0: (lambda p: (p.__setitem__(1, 3.0), p)[1])(real(th, guard).clone())
    with pytest.raises(np.linalg.LinAlgError):
        m.loss()
    # ... and a failure at the very first evaluation of fit() has no value to back off to: it raises as well
    with pytest.raises(np.linalg.LinAlgError):
        m.fit()
