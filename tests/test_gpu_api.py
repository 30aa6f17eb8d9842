"""The reference's own behavioural tests (test_training.py, test_rep.py sections 3-4, test_coverage_gaps.py
prediction parts) run against the HIP path."""
import copy

import numpy as np
import pytest

from lcgp_amd import LCGP

pytestmark = pytest.mark.gpu


def test_native_library_is_the_one_running():
    """The library loaded in THIS process (on the GPU box: the prebuilt .so that travelled with the snapshot) was
    compiled from exactly the lcgp_hip.hip / lcgp_hip.h in the tree."""
    from lcgp_amd import _hip
    assert _hip.loaded_hash() == _hip.source_hash()
    assert _hip.load().lcgp_version() >= 200


def test_c_abi_from_plain_cpp_without_torch():
    """tests/native/test_kernels (built by `make` / __graft_entry__.build()) links liblcgp_hip.so from plain C++ -- no
    Python, no torch -- and checks every entry point against host formulas: the C ABI is usable on its own."""
    import os
    import subprocess
    drv = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'native', 'test_kernels')
    if not os.path.exists(drv):
        pytest.skip('native driver not built')
    res = subprocess.run([drv, '64', '200', '700'], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0 and 'ALL OK' in res.stdout, res.stdout[-2000:] + res.stderr[-1000:]


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


def test_production_path_over_a_one_rank_rccl_group():
    """The N > 1 bench runs over RCCL (backend "nccl"), which this one-GPU box cannot form with several ranks.  A 1-rank
    RCCL group passed explicitly (`process_group=`) FORCES every collective of the production path -- the broadcast of
    the SVD basis in the constructor, the in-place all-reduce of the device-resident partial vector in every
    evaluation, the (2, q, n0) gather in predict(), the cache gathers -- through RCCL, and the results must equal the
    collective-free run and the oracle."""
    import os
    import tempfile
    import torch
    import torch.distributed as dist
    from lcgp_amd import dist as ldist
    from lcgp_amd import synth
    from oracle import lcgp_oracle as orc
    if dist.is_initialized():
        pytest.skip('a process group already exists in this process')
    store = tempfile.NamedTemporaryFile(delete=False)
    store.close()
    calls = {'all_reduce': 0, 'broadcast': 0}
    real_ar, real_bc = dist.all_reduce, dist.broadcast

    def spy_ar(t, *a, **k):
        assert t.is_cuda and t.dtype == torch.float64        # reduced where it lives: no host round trip
        calls['all_reduce'] += 1
        return real_ar(t, *a, **k)

    def spy_bc(t, *a, **k):
        calls['broadcast'] += 1
        return real_bc(t, *a, **k)

    try:
        dist.init_process_group('nccl', init_method='file://' + store.name, rank=0, world_size=1,
                                device_id=torch.device('cuda', torch.cuda.current_device()))
        grp = dist.group.WORLD
        assert ldist.use_collectives(grp) and ldist.backend_is_nccl(grp) and not ldist.is_distributed(grp)
        dist.all_reduce, dist.broadcast = spy_ar, spy_bc
        for mode, q, maker in (('full', 3, lambda: synth.make_full(901, 200, 2, 4, 3)),
                               ('rep', 4, lambda: synth.make_rep(902, 60, 3, 2, 4, 4))):
            x, y = maker()
            m0 = LCGP(y=y, x=x, q=q, submethod=mode)
            m = LCGP(y=y, x=x, q=q, submethod=mode, process_group=grp)
            assert calls['broadcast'] >= 3                    # phi, g, diag_D
            o = orc.OracleLCGP(y=y, x=x, q=q, submethod=mode)
            o.phi = m.phi.numpy().copy()
            u = synth.param_points(901, o.get_unconstrained())[1]
            n_before = calls['all_reduce']
            v, g = m.loss_and_grad(u)
            assert calls['all_reduce'] == n_before + 1        # ONE collective per evaluation
            v0, g0 = m0.loss_and_grad(u)
            v2, g2 = o.loss_and_grad_unconstrained(u)
            assert v == v0 and np.array_equal(g, g0)
            assert abs(v - v2) <= 1e-6 * abs(v2) and np.max(np.abs(g - g2)) <= 1e-5 * np.max(np.abs(g2))
            m.fit()
            m0.fit()
            assert np.array_equal(m._get_flat(), m0._get_flat())
            n_before = calls['all_reduce']
            x0 = np.random.default_rng(3).uniform(0, 1, (9, 2))
            got = m.predict(x0)
            assert calls['all_reduce'] - n_before <= 2        # (the factorisation is still valid after fit) + ONE gather
            want = m0.predict(x0)
            for a, b in zip(got, want):
                if a is not None:
                    np.testing.assert_array_equal(a.numpy(), b.numpy())
            np.testing.assert_array_equal(m.CinvMs.numpy(), m0.CinvMs.numpy())
        dist.barrier()
    finally:
        dist.all_reduce, dist.broadcast = real_ar, real_bc
        if dist.is_initialized():
            dist.destroy_process_group()
        if os.path.exists(store.name):
            os.unlink(store.name)


def test_predict_after_fit_reuses_the_factorisation():
    """fit() ends on the point L-BFGS-B evaluated last; predict() must not pay another evaluation for it."""
    from lcgp_amd import synth
    x, y = synth.make_full(903, 150, 2, 4, 3)
    m = LCGP(y=y, x=x, q=3)
    m.fit()
    if not np.array_equal(m._u_last, m._get_flat()):
        pytest.skip('the optimiser returned an earlier iterate than the last one evaluated')
    eng = m._get_engine()
    count = {'n': 0}
    real = eng.evaluate_partial

    def spy(theta, guard=0.0):
        count['n'] += 1
        return real(theta, guard)
    eng.evaluate_partial = spy
    m.predict(x[:7])
    assert count['n'] == 0
    m.lsigma2s.assign(m.lsigma2s.numpy() + 0.1)              # any parameter change invalidates it
    m.predict(x[:7])
    assert count['n'] == 1


def test_predict_chunks_agree_with_one_call():
    """x0 is processed in chunks of engine.PREDICT_CHUNK rows (bounded scratch); the nugget term of
    predict(training inputs) must land on the right rows of every chunk."""
    import lcgp_amd.engine as eng_mod
    from lcgp_amd import synth
    x, y = synth.make_full(904, 330, 2, 3, 2)
    m = LCGP(y=y, x=x, q=2)
    m.lnugGPs.assign(np.full(2, 0.05))                        # a visible nugget
    x0 = np.random.default_rng(9).uniform(0, 1, (301, 2))
    ref_new = [t.numpy() for t in m.predict(x0)]
    ref_train = [t.numpy() for t in m.predict(x)]
    old = eng_mod.PREDICT_CHUNK
    try:
        eng_mod.PREDICT_CHUNK = 128
        for a, b in zip(m.predict(x0), ref_new):
            np.testing.assert_allclose(a.numpy(), b, rtol=1e-12, atol=1e-13)
        for a, b in zip(m.predict(x), ref_train):
            np.testing.assert_allclose(a.numpy(), b, rtol=1e-12, atol=1e-13)
    finally:
        eng_mod.PREDICT_CHUNK = old


def test_the_inverse_launch_reports_the_clock_it_ran_at():
    """lcgp_lauum_clock: shader-clock cycles and 10 ns ticks of the longest K loop of the last A^-1 = W^T W launch (the same
    launch lcgp_nll_grad enqueues at this size): a plausible MI355X clock, measured in the un-profiled path."""
    import ctypes as C
    import torch
    from lcgp_amd import _hip, synth
    x, y = synth.make_full(77, 2048, 3, 6, 4)
    m = LCGP(y=y, x=x, q=4)
    m.loss()
    eng = m._engine
    sc = _hip.default_sched()
    sc.lauum_small_tiles = 0            # the 128x128-tile launch (the default at the headline size), which carries the stamps
    _hip.check(eng.lib.lcgp_lauum(eng._stream(), eng.dtype, eng.n, eng.d, eng.p, eng.q_local, eng._p(eng.workspace), C.byref(sc)),
               'lcgp_lauum')
    clk = torch.zeros(2, dtype=torch.int64, device=eng.device)
    _hip.check(eng.lib.lcgp_lauum_clock(eng._stream(), eng.dtype, eng.n, eng.d, eng.p, eng.q_local, eng._p(eng.workspace),
                                        C.c_void_p(clk.data_ptr())), 'lcgp_lauum_clock')
    cyc, ticks = (int(v) for v in clk.cpu())
    assert ticks > 0 and cyc > 0
    mhz = 100.0 * cyc / ticks
    assert 500.0 < mhz < 3000.0, mhz
    # read-and-clear: without another stamped launch the words are zero
    _hip.check(eng.lib.lcgp_lauum_clock(eng._stream(), eng.dtype, eng.n, eng.d, eng.p, eng.q_local, eng._p(eng.workspace),
                                        C.c_void_p(clk.data_ptr())), 'lcgp_lauum_clock')
    assert [int(v) for v in clk.cpu()] == [0, 0]
    # the 64x64-tile form of the launch (small problems, single components) stamps its first block too, and so does the first
    # trailing update of a factorisation (what is left to read where A^-1 is accumulated behind the chain)
    big = 10 ** 9
    for which in ('lauum64', 'potrf'):
        if which == 'lauum64':
            sc.lauum_small_tiles = big
            _hip.check(eng.lib.lcgp_lauum(eng._stream(), eng.dtype, eng.n, eng.d, eng.p, eng.q_local, eng._p(eng.workspace),
                                          C.byref(sc)), 'lcgp_lauum')
        else:
            m.loss()                            # kernel build + factorisation + ... (the last stamped launch: A^-1)
            _hip.check(eng.lib.lcgp_lauum_clock(eng._stream(), eng.dtype, eng.n, eng.d, eng.p, eng.q_local, eng._p(eng.workspace),
                                                C.c_void_p(clk.data_ptr())), 'lcgp_lauum_clock')
            _hip.check(eng.lib.lcgp_kernel_build(eng._stream(), eng.dtype, eng.kernel_id, eng.n, eng.d, eng.p, eng.q_local,
                                                 eng._p(eng.x), eng._p(eng.sr), eng._p(eng.theta_dev), eng._p(eng.workspace)),
                       'lcgp_kernel_build')
            _hip.check(eng.lib.lcgp_potrf_logdet(eng._stream(), eng.dtype, eng.n, eng.d, eng.p, eng.q_local, eng._p(eng.workspace),
                                                 None, None, None, None), 'lcgp_potrf_logdet')
        _hip.check(eng.lib.lcgp_lauum_clock(eng._stream(), eng.dtype, eng.n, eng.d, eng.p, eng.q_local, eng._p(eng.workspace),
                                            C.c_void_p(clk.data_ptr())), 'lcgp_lauum_clock')
        cyc, ticks = (int(v) for v in clk.cpu())
        assert ticks > 0 and 500.0 < 100.0 * cyc / ticks < 3000.0, (which, cyc, ticks)


def test_bench_line_keeps_its_contract():
    """`python bench.py` prints ONE JSON line with the fields the driver reads (metric / value / unit / n_gpus / steps / warmup /
    ms_per_step / higher_is_better / scaling / vs_baseline / dtype / data / config.workload) plus the `roofline` block of the
    dominant kernel, measured live (HIP events, in-kernel clock).  Run as the driver runs it, in a process of its own."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '1', '--steps', '3', '--warmup', '1',
                          '--no-cpu-baseline', '--no-fit', '--predict', '0', '--blocks', '1'],
                         capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d['metric'] == 'NLL+grad evals/sec' and d['unit'] == 'evals/s' and d['higher_is_better'] is True
    assert d['n_gpus'] == 1 and d['steps'] == 3 and d['warmup'] == 1 and d['dtype'] == 'f64' and d['data'] == 'synthetic'
    assert d['vs_baseline'] is None and d['scaling'] in ('strong', 'weak') and 'n=4096' in d['config']['workload']
    assert abs(d['value'] - 1e3 / d['ms_per_step']) < 1e-6 * d['value'] and d['value'] > 0.0
    r = d['roofline']
    assert r['bound'] == 'mfma' and r['unit'] == 'TFLOP/s' and r['peak'] > 0.0
    assert abs(r['frac'] - r['achieved'] / r['peak']) < 1e-12 and 0.0 < r['frac'] < 1.0
    assert abs(r['achieved'] - r['flops_per_launch'] / (r['launch_ms'] * 1e-3) / 1e12) < 1e-9 * r['achieved']
    assert r['clock_mhz'] is not None and r['clock_mhz'] > 0.0 and r['clock_window_us'] > 0.0
    assert r['traffic'] is None or r['traffic'] > 1e9
    assert set(d['roofline_stages']) == {'potrf', 'trtri', 'lauum'}
