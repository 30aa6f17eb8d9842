"""The factorisation as ONE persistent launch with dependencies inside it (lcgp_sched.dag = 1; lcgp_hip.hip: dag_kernel,
fill_sched.h: DagBuilder) against the launch-by-launch executor: the same tile bodies run on the same data, so the
results must be IDENTICAL, whatever order the workgroups happen to take.  Replaces tf.linalg.eigh / cholesky of the
reference (lcgp.py:652, 617) like the launch-by-launch form does."""
import numpy as np
import pytest

from lcgp_amd import LCGP, synth, _hip
from oracle import lcgp_oracle as orc

pytestmark = pytest.mark.gpu


def _sched(**fields):
    sc = _hip.default_sched()
    for k, v in fields.items():
        assert hasattr(sc, k), k
        setattr(sc, k, v)
    return sc


CASES = [
    # seed, n, d, p, q, schedule variants
    (401, 64, 2, 3, 1, [{}]),
    (402, 200, 2, 4, 2, [{}, dict(outer_blocks=2)]),
    (403, 700, 3, 4, 4, [{}, dict(outer_blocks=2), dict(outer_blocks=8), dict(fill_leaf=16, fill_step=24), dict(leaf_in_wide=0),
                         dict(leaf_in_wide=100000), dict(progressive_tiles=0), dict(progressive_tiles=1 << 30),
                         dict(progressive_tiles=1 << 30, fill_leaf=12, fill_step=20),
                         dict(progressive_tiles=1 << 30, progressive_lauum=0)]),
    (404, 1500, 3, 5, 4, [dict(syrk_small_tiles=16), dict(syrk_small_tiles=16, leaf_in_wide=100000),
                          dict(syrk_small_tiles=16, fill_leaf=40, fill_step=56), dict(syrk_small_tiles=1, fill_leaf=8, fill_step=8),
                          dict(progressive_tiles=1 << 30, syrk_small_tiles=16)]),
    (405, 2048, 3, 12, 6, [{}, dict(progressive_tiles=1 << 30, fill_leaf=30, fill_step=18)]),
    (406, 1000, 3, 9, 8, [{}]),
    (407, 1100, 2, 3, 1, [{}, dict(progressive_lauum=0)]),
]


@pytest.mark.parametrize('seed,n,d,p,q,variants', CASES)
def test_persistent_launch_equals_launch_by_launch(seed, n, d, p, q, variants):
    x, y = synth.make_full(seed, n, d, p, q)
    m = LCGP(y=y, x=x, q=q)
    u = synth.param_points(seed, m._get_flat())[1]
    eng = m._get_engine()
    try:
        for fields in variants:
            eng.sched = _sched(dag=0, **fields)
            ref_v, ref_g = m.loss_and_grad(u)
            ref_out = eng.out_dev.cpu().numpy().copy()
            eng.sched = _sched(dag=1, **fields)
            info = eng.plan_info()
            assert info['segments'] > 0 and info['tasks'] > 0, 'the persistent form must be available: %r' % (info,)
            for rep in range(2):            # twice: a second run starts from the control words the first one left behind
                v, g = m.loss_and_grad(u)
                out = eng.out_dev.cpu().numpy()
                assert np.array_equal(out, ref_out), (fields, rep, float(np.max(np.abs(out - ref_out))))
                assert v == ref_v and np.array_equal(g, ref_g), fields
            # the persistent launch with its own task order (near / far updates alternating with the chain): other tile
            # shapes in places, so equal to rounding, and bitwise repeatable
            eng.sched = _sched(dag=2, **fields)
            assert eng.plan_info()['segments'] > 0
            v2, g2 = m.loss_and_grad(u)
            out2 = eng.out_dev.cpu().numpy().copy()
            assert abs(v2 - ref_v) <= 1e-11 * abs(ref_v), fields
            assert np.max(np.abs(g2 - ref_g)) <= 1e-10 * np.max(np.abs(ref_g)), fields
            m.loss_and_grad(u)
            assert np.array_equal(eng.out_dev.cpu().numpy(), out2), fields
            # ... and the same list executed launch by launch (no device copy of the plan)
            eng.use_plan = False
            eng.sched = _sched(dag=2, **fields)
            v3, g3 = m.loss_and_grad(u)
            eng.use_plan = True
            assert abs(v3 - ref_v) <= 1e-11 * abs(ref_v) and np.max(np.abs(g3 - ref_g)) <= 1e-10 * np.max(np.abs(ref_g)), fields
    finally:
        eng.sched = None


def test_persistent_launch_matches_the_oracle():
    x, y = synth.make_full(411, 900, 2, 6, 3)
    o = orc.OracleLCGP(y=y, x=x, q=3)
    m = LCGP(y=y, x=x, q=3)
    o.phi = m.phi.numpy().copy()
    m._get_engine().sched = _sched(dag=1)
    for u in synth.param_points(411, o.get_unconstrained()):
        v1, g1 = m.loss_and_grad(u)
        v2, g2 = o.loss_and_grad_unconstrained(u)
        assert abs(v1 - v2) <= 1e-10 * abs(v2)
        assert np.max(np.abs(g1 - g2)) <= 1e-9 * np.max(np.abs(g2))


def test_a_wait_that_expires_is_reported_not_spun_on():
    """dag_spin_limit = 1: (almost) every wait inside the launch gives up at once.  The launch must still END (every
    later wait returns immediately, nothing spins), and every component must report info = -1 instead of numbers."""
    x, y = synth.make_full(412, 1200, 2, 4, 4)
    m = LCGP(y=y, x=x, q=4)
    eng = m._get_engine()
    sig_eff = np.exp(0.5 * np.repeat(m.lsigma2s.numpy(), np.asarray(m.diag_error_structure, int))) / m._std
    theta = m._theta_rows(sig_eff)
    try:
        eng.sched = _sched(dag=1, dag_spin_limit=1)
        out = eng.evaluate(theta)
        assert [int(r[2]) for r in out] == [-1] * 4
        with pytest.raises(Exception):
            m.loss_and_grad(m._get_flat())
        eng.sched = _sched(dag=1)
        out = eng.evaluate(theta)
        assert [int(r[2]) for r in out] == [0] * 4          # the same workspace recovers: control words are zeroed per call
    finally:
        eng.sched = None


def test_first_bad_pivot_is_reported_by_the_persistent_launch():
    from tests.test_gpu_edge_cases import _first_bad_pivot
    import torch
    x, y = synth.make_full(340, 330, 2, 4, 3)
    m = LCGP(y=y, x=x, q=3)
    ell, scale, nug = m.lLmb.numpy(), m.lLmb0.numpy(), m.lnugGPs.numpy()
    xs = m.x.numpy()
    want, dvals = [], []
    for k, target in enumerate((1, 40, 300)):
        c = orc.matern32(xs, xs, ell[k], scale[k], nug[k])
        w = np.linalg.eigvalsh(c[:target, :target])
        dk = -1.0 / w[-1] * 1.05
        want.append(_first_bad_pivot(np.eye(330) + dk * c))
        dvals.append(dk)
    m.diag_D = torch.as_tensor(np.array(dvals))
    eng = m._get_engine()
    sig_eff = np.exp(0.5 * np.repeat(m.lsigma2s.numpy(), np.asarray(m.diag_error_structure, int))) / m._std
    eng.sched = _sched(dag=1)
    out = eng.evaluate(m._theta_rows(sig_eff))
    assert [int(r[2]) for r in out] == want
