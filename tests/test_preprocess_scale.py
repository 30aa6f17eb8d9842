"""the host preprocessing at the sizes the hot path is benchmarked on.

The reference builds an (n, n) temporary per input dimension for `xnorm` (lcgp.py:304-309: 2 GB per dimension at
n=16384) and loops over groups in Python for `ybar` (lcgp.py:358-367).  This build uses a sort-based O(n log n)
formula with the SAME value and vectorised group means; here they are checked against the literal O(n^2) / looped
restatements of the oracle at n = 2048, and the constructor is timed at the full sizes of BASELINE.json's
configs[3] (n=16384, d=10) and configs[4] (N=10240 rows -> n_unique=2048).  The device is not touched: the engine is
only created at the first loss()/fit()/predict()."""
import time

import numpy as np
import torch

from lcgp_amd import LCGP, synth
from oracle import lcgp_oracle as orc


def test_xnorm_sort_formula_equals_the_pairwise_definition_at_n2048():
    rng = np.random.default_rng(5)
    x = rng.uniform(-3.0, 7.0, (2048, 4))
    x[100:140] = x[500:540]                      # tied rows: zero distances are excluded from the mean
    x[:, 3] = np.round(x[:, 3], 1)               # a heavily tied column
    t0 = time.perf_counter()
    xs, xmin, xmax, _, xnorm = LCGP.init_standard_x(torch.as_tensor(x))
    t_fast = time.perf_counter() - t0
    t0 = time.perf_counter()
    want = orc.xnorm_pairs(x)                    # literal lcgp.py:304-309
    t_ref = time.perf_counter() - t0
    np.testing.assert_allclose(xnorm.numpy(), want, rtol=1e-11)
    np.testing.assert_array_equal(xs.numpy(), orc.standardize_x(x)[0])
    print('xnorm at n=2048, d=4: sort-based %.4f s, pairwise definition %.3f s' % (t_fast, t_ref))


def test_grouping_and_ybar_equal_the_looped_definition_at_n2048x5():
    x, y, cfg = synth.make_config(5)             # N = 10240 rows, 2048 unique
    m = LCGP(y=y, x=x, q=cfg['q'], submethod='rep')
    xu, inv, r, ybar = orc.group_replicates(x, y)        # literal lcgp.py:349-367
    np.testing.assert_array_equal(m.x_unique.numpy(), xu)
    np.testing.assert_array_equal(m.group_ids.numpy(), inv)
    np.testing.assert_array_equal(m.r.numpy(), r)
    np.testing.assert_allclose(m.ybar.numpy(), ybar, rtol=1e-13, atol=1e-15)
    c, s = orc.center_spread(ybar, True, guard_zero=True)
    np.testing.assert_allclose(m.ybar_mean.numpy(), c, rtol=1e-13)
    np.testing.assert_allclose(m.ybar_std.numpy(), s, rtol=1e-13)
    np.testing.assert_allclose(m.ybar_s.numpy(), (ybar - c) / s, rtol=1e-12, atol=1e-14)


def test_constructor_wall_clock_at_full_sizes():
    """LCGP(...) at configs[3] and configs[4] must stay a small fraction of a fit (a fit is ~50-200 evaluations of
    ~0.3 s / ~2 ms each on the GPU).  Bounds are generous (shared CI cores); the measured times are printed and
    recorded in DESIGN.md."""
    x, y, cfg = synth.make_config(4)
    t0 = time.perf_counter()
    m = LCGP(y=y, x=x, q=cfg['q'], dtype='float32')
    t4 = time.perf_counter() - t0
    assert int(m.n) == 16384 and m.q == 8 and m._engine is None
    assert np.all(np.isfinite(m.xnorm.numpy())) and np.all(m.xnorm.numpy() > 0)
    x, y, cfg = synth.make_config(5)
    t0 = time.perf_counter()
    m = LCGP(y=y, x=x, q=cfg['q'], submethod='rep')
    t5 = time.perf_counter() - t0
    assert int(m.n) == 2048
    print('constructor: configs[3] n=16384 d=10 p=32 -> %.2f s;  configs[4] N=10240 -> n_unique=2048 -> %.2f s' % (t4, t5))
    assert t4 < 30.0 and t5 < 30.0
