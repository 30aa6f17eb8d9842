"""Edge cases of the HIP path against the oracle: tiny and ragged sizes, maximum input dimension, duplicated
inputs (kernel collisions), parameters at the SoftClip bounds, predict with one and with many new inputs,
float32 on the replicated path."""
import ctypes as C

import numpy as np
import pytest
import torch

from lcgp_amd import LCGP, synth, _hip
from oracle import lcgp_oracle as orc

pytestmark = pytest.mark.gpu


def _same(m, o, u, nll_tol=1e-6, grad_tol=1e-5):
    o.phi = m.phi.numpy().copy()
    v1, g1 = m.loss_and_grad(u)
    v2, g2 = o.loss_and_grad_unconstrained(u)
    assert np.isfinite(v1) and np.all(np.isfinite(g1))
    assert abs(v1 - v2) <= nll_tol * max(abs(v2), 1e-12), (v1, v2)
    assert np.max(np.abs(g1 - g2)) <= grad_tol * max(np.max(np.abs(g2)), 1e-12)


@pytest.mark.parametrize('n,d,p,q', [(3, 1, 1, 1), (2, 2, 2, 1), (127, 1, 2, 2), (128, 2, 3, 3), (129, 3, 2, 2)])
def test_tiny_and_tile_boundary_sizes(n, d, p, q):
    x, y = synth.make_full(300 + n, n, d, p, q)
    kw = dict(q=q, robust_mean=False)     # robust spread of a 2-3 point row can be 0 -> the reference gives NaN too
    m = LCGP(y=y, x=x, **kw)
    o = orc.OracleLCGP(y=y, x=x, **kw)
    _same(m, o, o.get_unconstrained())
    out = m.predict(x[:1])
    assert out[0].shape == (p, 1) and np.all(np.isfinite(out[0].numpy()))


@pytest.mark.parametrize('seed,d', [(316, 16), (317, 17), (318, 32)])
def test_wide_input_dimensions(seed, d):
    """the fused kernels are instantiated for d <= 2, 4, 6, 10, 16, 32: the last two against the oracle"""
    x, y = synth.make_full(seed, 90, d, 3, 2)
    m = LCGP(y=y, x=x, q=2)
    o = orc.OracleLCGP(y=y, x=x, q=2)
    _same(m, o, synth.param_points(seed, o.get_unconstrained())[1])
    np.testing.assert_allclose(m.predict(x[:5])[0].numpy(), o.predict(x[:5])[0], rtol=1e-7, atol=1e-9)


@pytest.mark.parametrize('seed,n,d,kw', [(319, 90, 33, {}), (3191, 150, 40, {}), (3192, 70, 64, {}), (3193, 200, 75, dict(q=1)),
                                         (3194, 80, 126, {})])
def test_more_than_32_input_dimensions(seed, n, d, kw):
    """The reference's kernel loops over any number of input dimensions (covmat.py:35-42).  Beyond 32 the kernels stage the
    dimensions 32 at a time (kernel build, cross covariance, grad_kernel_wide): NLL, every gradient entry (d lengthscales
    per component) and predictions against the oracle, and Matern32 itself against the restatement."""
    x, y = synth.make_full(seed, n, d, 3, 2)
    q = kw.get('q', 2)
    m = LCGP(y=y, x=x, q=q)
    o = orc.OracleLCGP(y=y, x=x, q=q)
    for u in synth.param_points(seed, o.get_unconstrained())[:2]:
        _same(m, o, u)
    for a, b in zip(m.predict(x[:7] * 0.98 + 0.01), o.predict(x[:7] * 0.98 + 0.01)):
        np.testing.assert_allclose(a.numpy(), b, rtol=1e-7, atol=1e-9)
    from lcgp_amd import Matern32
    ell = np.exp(np.random.default_rng(seed).uniform(0.5, 2.0, d))
    got = Matern32(x[:70], x[:50], ell, 1.7, 0.01).numpy()
    want = orc.matern32(x[:70], x[:50], ell, 1.7, 0.01)
    np.testing.assert_allclose(got, want, rtol=1e-13, atol=1e-300)


def test_refusal_above_the_maximum_input_dimension():
    x127, y127 = synth.make_full(319, 40, 127, 2, 2)
    m127 = LCGP(y=y127, x=x127, q=2)
    with pytest.raises(RuntimeError, match='d must be'):
        m127.loss()


def test_float32_with_more_than_32_input_dimensions():
    """the float32 variant of the wide kernels (incl. the cancellation-free quadratic form) against this build's float64"""
    x, y = synth.make_full(3195, 300, 45, 4, 2)
    m64 = LCGP(y=y, x=x, q=2)
    m32 = LCGP(y=y, x=x, q=2, dtype='float32')
    u = synth.param_points(3195, m64._get_flat())[1]
    v64, g64 = m64.loss_and_grad(u)
    v32, g32 = m32.loss_and_grad(u)
    assert abs(v32 - v64) <= 1e-5 * abs(v64)
    assert np.max(np.abs(g32 - g64)) <= 1e-3 * np.max(np.abs(g64))


def test_duplicated_inputs_in_full_mode():
    """identical rows of x give identical rows of C (the nugget only sits on the diagonal)."""
    x, y = synth.make_full(320, 100, 2, 3, 3)
    x[10] = x[3]
    x[50:60] = x[20:30]
    m = LCGP(y=y, x=x, q=3)
    o = orc.OracleLCGP(y=y, x=x, q=3)
    for u in synth.param_points(320, o.get_unconstrained()):
        _same(m, o, u)


def test_parameters_near_the_softclip_bounds():
    x, y = synth.make_full(321, 150, 2, 3, 2)
    m = LCGP(y=y, x=x, q=2)
    o = orc.OracleLCGP(y=y, x=x, q=2)
    u0 = o.get_unconstrained()
    for shift in (-25.0, +12.0):          # lengthscales ~1e-6 (C -> nugget-free identity) / large (C -> all ones)
        u = u0.copy()
        u[:4] = u0[:4] + shift
        _same(m, o, u, grad_tol=1e-4)
    u = u0.copy()
    u[6:8] = 30.0                         # nugget at its upper bound e^-2
    _same(m, o, u)
    u = u0.copy()
    u[6:8] = -40.0                        # nugget at its lower bound e^-16
    _same(m, o, u)


def test_float32_survives_collapsed_lengthscales():
    """Ten input dimensions with every lengthscale at its lower SoftClip bound (1e-6): the polynomial factor of the Matern
    kernel is ~(1e6)^10, beyond float32's range, and its exponential factor is zero -- inf x 0 used to put NaNs into the
    float32 matrix (status word 2 at a perfectly conditioned matrix; this is what sent float32 fits of configs[3] to
    float64: profiles/r06_fp32_breakdown.txt).  Both precisions must evaluate such a point, and agree."""
    x, y = synth.make_full(323, 300, 10, 4, 2)
    m64 = LCGP(y=y, x=x, q=2)
    m32 = LCGP(y=y, x=x, q=2, dtype='float32')
    m32.float32_fallback = False                     # no float64 repeat: the float32 engine itself has to cope
    u = m64._get_flat().copy()
    u[:20] = -40.0                                   # all 2 x 10 lengthscales at the bound
    v64, g64 = m64.loss_and_grad(u)
    v32, g32 = m32.loss_and_grad(u)
    assert np.isfinite(v32) and np.all(np.isfinite(g32))
    assert abs(v32 - v64) <= 2e-4 * abs(v64)
    assert np.max(np.abs(g32 - g64)) <= 2e-2 * max(np.max(np.abs(g64)), 1e-300)
    o = orc.OracleLCGP(y=y, x=x, q=2)
    o.phi = m64.phi.numpy().copy()
    vo, go = o.loss_and_grad_unconstrained(u)
    assert abs(v64 - vo) <= 1e-6 * abs(vo) and np.max(np.abs(g64 - go)) <= 1e-5 * max(np.max(np.abs(go)), 1e-300)


@pytest.mark.parametrize('n0', [1, 63, 64, 65, 700])
def test_predict_sizes(n0):
    x, y = synth.make_full(322, 200, 2, 3, 3)
    m = LCGP(y=y, x=x, q=3)
    o = orc.OracleLCGP(y=y, x=x, q=3)
    o.phi = m.phi.numpy().copy()
    u = synth.param_points(322, o.get_unconstrained())[1]
    m._set_flat(u)
    o.set_unconstrained(u)
    x0 = np.random.default_rng(n0).uniform(-0.2, 1.2, (n0, 2))     # also outside the training box
    got = m.predict(x0)
    want = o.predict(x0)
    for a, b in zip(got, want):
        assert a.shape == (3, n0)
        np.testing.assert_allclose(a.numpy(), b, rtol=1e-6, atol=1e-9)


def test_var_threshold_and_single_component():
    x, y = synth.make_full(323, 120, 2, 6, 2)
    m = LCGP(y=y, x=x, var_threshold=0.6)
    o = orc.OracleLCGP(y=y, x=x, var_threshold=0.6)
    assert m.q == o.q
    _same(m, o, o.get_unconstrained())
    m1 = LCGP(y=y, x=x, q=1)
    o1 = orc.OracleLCGP(y=y, x=x, q=1)
    _same(m1, o1, o1.get_unconstrained())


def test_many_outputs():
    x, y = synth.make_full(324, 96, 2, 150, 3)
    m = LCGP(y=y, x=x, q=3)
    o = orc.OracleLCGP(y=y, x=x, q=3)
    _same(m, o, synth.param_points(324, o.get_unconstrained())[1])


def test_float32_replicated_path():
    x, y = synth.make_rep(325, 300, 3, 2, 4, 4)
    m64 = LCGP(y=y, x=x, submethod='rep')
    m32 = LCGP(y=y, x=x, submethod='rep', dtype='float32')
    u = m64._get_flat()
    v64, g64 = m64.loss_and_grad(u)
    v32, g32 = m32.loss_and_grad(u)
    assert abs(v32 - v64) <= 1e-3 * abs(v64)
    assert np.max(np.abs(g32 - g64)) <= 2e-2 * np.max(np.abs(g64))
    p64 = m64.predict(x[:20])[0].numpy()
    p32 = m32.predict(x[:20])[0].numpy()
    np.testing.assert_allclose(p32, p64, rtol=5e-3, atol=5e-3)


def test_c_abi_argument_checks():
    lib = _hip.load()
    nbytes = C.c_size_t(0)
    assert lib.lcgp_workspace_bytes(2, 10, 2, 2, 1, C.byref(nbytes)) < 0          # bad dtype
    assert lib.lcgp_workspace_bytes(0, 0, 2, 2, 1, C.byref(nbytes)) < 0           # n < 1
    assert lib.lcgp_workspace_bytes(0, 10, 2, 2, 0, C.byref(nbytes)) < 0          # q_local < 1
    assert lib.lcgp_nll_grad(None, 0, 0, 10, 2, 2, 1, None, None, None, None, None, None, None, None) < 0
    assert b'NULL' in lib.lcgp_last_error()
    bad = _hip.default_sched()
    bad.outer_blocks = 65
    assert lib.lcgp_potri(None, 0, 10, 2, 2, 1, C.c_void_p(8), C.byref(bad)) < 0     # rejected before any launch
    assert b'outer_blocks' in lib.lcgp_last_error()


def _sched(**fields):
    """The default launch schedule with some fields replaced (lcgp_sched travels with every call: no global state)."""
    sc = _hip.default_sched()
    for k, v in fields.items():
        assert hasattr(sc, k), k
        setattr(sc, k, v)
    return sc


def test_schedule_parameters_do_not_change_results():
    """Every launch schedule selectable through lcgp_sched computes the same numbers: panel widths, no filler /
    different filler sizes, with and without the trailing update that factors the next diagonal block, tile sizes of
    the inverse."""
    x, y = synth.make_full(326, 700, 3, 4, 4)
    m = LCGP(y=y, x=x, q=4)
    u = synth.param_points(326, m._get_flat())[1]
    ref_v, ref_g = m.loss_and_grad(u)
    eng = m._get_engine()
    try:
        for fields in (dict(outer_blocks=2), dict(outer_blocks=8), dict(outer_blocks=3), dict(outer_blocks=1),
                       dict(fill_leaf=0, fill_step=0), dict(fill_leaf=16, fill_step=24), dict(leaf_in_wide=0),
                       dict(leaf_in_wide=100000), dict(leaf_in_wide=100000, outer_blocks=2), dict(trtri_level_small=0),
                       dict(trtri_level_small=100000, trtri_small_tiles=0), dict(lauum_small_tiles=0),
                       dict(lauum_small_tiles=100000, trtri_small_tiles=100000),
                       # paired panels (one K = 2-panel update of the middle columns): off, and forced at this size
                       dict(pair_tiles=0), dict(pair_tiles=1, progressive_tiles=0), dict(pair_tiles=1, progressive_tiles=0, leaf_in_wide=0),
                       dict(pair_tiles=1, progressive_tiles=0, leaf_in_wide=0, outer_blocks=2, fill_leaf=4, fill_step=4),
                       # the inverse formed behind the chain (fill_sched.h) and after it
                       dict(progressive_tiles=0), dict(progressive_tiles=1 << 30), dict(progressive_tiles=1 << 30, outer_blocks=2),
                       dict(progressive_tiles=1 << 30, outer_blocks=8), dict(progressive_tiles=1 << 30, fill_leaf=0, fill_step=0),
                       dict(progressive_tiles=1 << 30, fill_leaf=12, fill_step=20), dict(progressive_tiles=1 << 30, progressive_far=0),
                       dict(progressive_tiles=1 << 30, leaf_in_wide=0), dict(progressive_tiles=1 << 30, outer_blocks=3),
                       dict(progressive_tiles=1 << 30, progressive_lauum=0), dict(progressive_tiles=1 << 30, progressive_lauum=0, outer_blocks=2)):
            eng.sched = _sched(**fields)
            v, g = m.loss_and_grad(u)
            assert abs(v - ref_v) <= 1e-11 * abs(ref_v), fields
            assert np.max(np.abs(g - ref_g)) <= 1e-10 * np.max(np.abs(ref_g)), fields
    finally:
        eng.sched = None


def test_wide_tile_schedules_agree_at_medium_size():
    """Forces the 128x128-tile trailing update with filler tiles on a problem small enough for a unit test (the default
    thresholds only use it from ~n = 3000 on) and compares every chain variant with the default schedule: the
    transitions 128-tile update -> 64-tile update that also factors the next diagonal block -> no update are all hit."""
    x, y = synth.make_full(327, 1500, 3, 5, 4)
    m = LCGP(y=y, x=x, q=4)
    u = synth.param_points(327, m._get_flat())[1]
    ref_v, ref_g = m.loss_and_grad(u)
    eng = m._get_engine()
    try:
        for fields in (dict(syrk_small_tiles=16), dict(syrk_small_tiles=16, leaf_in_wide=0),
                       dict(syrk_small_tiles=16, leaf_in_wide=100000), dict(syrk_small_tiles=16, fill_leaf=40, fill_step=56),
                       dict(syrk_small_tiles=16, outer_blocks=2), dict(syrk_small_tiles=16, outer_blocks=8),
                       dict(syrk_small_tiles=200, leaf_in_wide=300), dict(syrk_small_tiles=1, fill_leaf=8, fill_step=8),
                       dict(pair_tiles=1, progressive_tiles=0, leaf_in_wide=0), dict(pair_tiles=1, progressive_tiles=0, leaf_in_wide=0, syrk_small_tiles=16),
                       dict(pair_tiles=1, progressive_tiles=0, leaf_in_wide=0, fill_leaf=8, fill_step=8, outer_blocks=2), dict(pair_tiles=0),
                       dict(pair_tiles=1, progressive_tiles=0, leaf_in_wide=0, fill_leaf=8, fill_step=8),
                       dict(pair_tiles=1, progressive_tiles=0, leaf_in_wide=0, fill_leaf=8, fill_step=8, syrk_small_tiles=16),
                       dict(progressive_tiles=0), dict(progressive_tiles=1 << 30), dict(progressive_tiles=1 << 30, syrk_small_tiles=16),
                       dict(progressive_tiles=1 << 30, progressive_far=0, syrk_small_tiles=16),
                       dict(progressive_tiles=1 << 30, fill_leaf=40, fill_step=56), dict(progressive_tiles=1 << 30, outer_blocks=8)):
            eng.sched = _sched(**fields)
            v, g = m.loss_and_grad(u)
            assert abs(v - ref_v) <= 1e-11 * abs(ref_v), fields
            assert np.max(np.abs(g - ref_g)) <= 1e-10 * np.max(np.abs(ref_g)), fields
    finally:
        eng.sched = None


def test_chain_variants_match_oracle_at_several_sizes():
    """The fused chain step and the in-wave diagonal block against the oracle where the panel structure differs:
    a single block, a partial panel, several panels with a ragged last one."""
    for seed, n in ((331, 64), (332, 200), (333, 330), (334, 900)):
        x, y = synth.make_full(seed, n, 2, 6, 2)
        o = orc.OracleLCGP(y=y, x=x, q=2)
        u = synth.param_points(seed, o.get_unconstrained())[1]
        for fields in ({}, dict(leaf_in_wide=0), dict(outer_blocks=3)):
            m = LCGP(y=y, x=x, q=2)
            m._get_engine().sched = _sched(**fields)
            _same(m, o, u)


def test_progressive_inverse_with_jobs_split_over_launches():
    """The inverse formed behind the factorisation (fill_sched.h) where the filler capacity cuts jobs at every kind of
    boundary: n = 2048 with 6 components splits a level of a block inverse over two launches (the shape that exposed a
    job-offset bug during development), tiny capacities split every job many times.  L^-1, A^-1, z, NLL and gradient
    against the schedule that inverts after the factorisation; predictions (which read L^-1) as well."""
    x, y = synth.make_full(351, 2048, 3, 12, 6)
    m = LCGP(y=y, x=x, q=6)
    u = synth.param_points(351, m._get_flat())[1]
    eng = m._get_engine()
    try:
        eng.sched = _sched(progressive_tiles=0)
        ref_v, ref_g = m.loss_and_grad(u)
        ref_p = [t.numpy() for t in m.predict(x[:50] + 0.003)]
        ref_w, ref_a, ref_z = np.tril(eng.fetch_matrix(1, 5)), np.tril(eng.fetch_matrix(2, 5)), eng.fetch_vector(1, 5)
        for fields in (dict(progressive_tiles=1 << 30), dict(progressive_tiles=1 << 30, fill_leaf=30, fill_step=18),
                       dict(progressive_tiles=1 << 30, fill_leaf=6, fill_step=6), dict(progressive_tiles=1 << 30, progressive_far=0),
                       dict(progressive_tiles=1 << 30, progressive_lauum=0), dict(progressive_tiles=1 << 30, progressive_lauum=0, fill_leaf=30, fill_step=18)):
            eng.sched = _sched(**fields)
            v, g = m.loss_and_grad(u)
            assert abs(v - ref_v) <= 1e-11 * abs(ref_v), fields
            assert np.max(np.abs(g - ref_g)) <= 1e-10 * np.max(np.abs(ref_g)), fields
            assert np.max(np.abs(np.tril(eng.fetch_matrix(1, 5)) - ref_w)) <= 1e-11 * np.max(np.abs(ref_w)), fields
            assert np.max(np.abs(np.tril(eng.fetch_matrix(2, 5)) - ref_a)) <= 1e-10 * np.max(np.abs(ref_a)), fields
            assert np.max(np.abs(eng.fetch_vector(1, 5) - ref_z)) <= 1e-10 * np.max(np.abs(ref_z)), fields
            for a, b in zip(ref_p, (t.numpy() for t in m.predict(x[:50] + 0.003))):
                assert np.max(np.abs(a - b)) <= 1e-10 * np.max(np.abs(a)), fields
    finally:
        eng.sched = None


def _first_bad_pivot(a):
    """1-based index of the first non-positive pivot of an unblocked Cholesky of `a` (LAPACK's info), 0 if none."""
    a = a.copy()
    n = a.shape[0]
    for j in range(n):
        if not a[j, j] > 0.0:
            return j + 1
        a[j + 1:, j] /= a[j, j]
        a[j + 1:, j + 1:] -= np.outer(a[j + 1:, j], a[j + 1:, j]) * a[j, j]
    return 0


@pytest.mark.parametrize('variant', [{}, dict(leaf_in_wide=0), dict(outer_blocks=2), dict(fill_leaf=0, fill_step=0),
                                     dict(progressive_tiles=0), dict(progressive_tiles=1 << 30)])
def test_info_is_the_first_bad_pivot(variant):
    """info of the output block = position of the first non-positive pivot (as LAPACK dpotrf reports it), wherever
    it falls: first block, inside a later 16-column panel of a diagonal block, in a later block or panel."""
    x, y = synth.make_full(340, 330, 2, 4, 3)
    m = LCGP(y=y, x=x, q=3)
    ell, scale, nug = m.lLmb.numpy(), m.lLmb0.numpy(), m.lnugGPs.numpy()
    xs = m.x.numpy()
    want, dvals = [], []
    for k, target in enumerate((1, 40, 300)):
        c = orc.matern32(xs, xs, ell[k], scale[k], nug[k])
        # A = I + D C is indefinite for D < -1 / lambda_max(C); pick D so that the leading `target`-minor fails first
        w = np.linalg.eigvalsh(c[:target, :target])
        dk = -1.0 / w[-1] * 1.05
        a = np.eye(330) + dk * c
        want.append(_first_bad_pivot(a))
        dvals.append(dk)
    assert want[0] >= 1 and len(set(want)) > 1
    m.diag_D = torch.as_tensor(np.array(dvals))
    eng = m._get_engine()
    sig_eff = np.exp(0.5 * np.repeat(m.lsigma2s.numpy(), np.asarray(m.diag_error_structure, int))) / m._std
    eng.sched = _sched(**variant)
    out = eng.evaluate(m._theta_rows(sig_eff))
    assert [int(r[2]) for r in out] == want


def test_random_shapes_and_schedules_against_oracle():
    """Seeded sweep: random (n, d, p, q) with a random schedule variant each, NLL and gradient against the oracle.
    (sizes up to a few panels; the tolerances are the parity bar of the path: 1e-6 / 1e-5 relative)"""
    rng = np.random.default_rng(20260401)
    variants = [{}, dict(leaf_in_wide=0), dict(leaf_in_wide=100000), dict(outer_blocks=2), dict(outer_blocks=3),
                dict(outer_blocks=8), dict(fill_leaf=8, fill_step=16), dict(syrk_small_tiles=1),
                dict(syrk_small_tiles=1, leaf_in_wide=100000), dict(syrk_small_tiles=1, fill_leaf=24, fill_step=40),
                dict(trtri_small_tiles=0, trtri_level_small=0, lauum_small_tiles=0),
                dict(progressive_tiles=0), dict(progressive_tiles=1 << 30, fill_leaf=16, fill_step=8),
                dict(progressive_tiles=1 << 30, outer_blocks=2), dict(progressive_tiles=1 << 30, progressive_far=0)]
    for case in range(18):
        n = int(rng.integers(1, 3)) if case == 0 else int(rng.integers(2, 1100))
        d = int(rng.integers(1, 5))
        p = int(rng.integers(1, 7))
        q = int(rng.integers(1, min(p, 3) + 1))
        seed = 5000 + case
        x, y = synth.make_full(seed, max(n, 2), d, p, q)
        kw = dict(q=q, robust_mean=n > 8)
        o = orc.OracleLCGP(y=y, x=x, **kw)
        u = synth.param_points(seed, o.get_unconstrained())[1 if n > 8 else 0]
        fields = variants[int(rng.integers(0, len(variants)))]
        m = LCGP(y=y, x=x, **kw)
        m._get_engine().sched = _sched(**fields)
        try:
            _same(m, o, u)
        except AssertionError as e:
            raise AssertionError('case %d: n=%d d=%d p=%d q=%d sched=%r: %s' % (case, n, d, p, q, fields, e))
