"""Deterministic synthetic workloads of SURVEY.md 8(d) / BASELINE.json `configs`.

Shared by `bench.py`, the parity tests and the golden-fixture generator so that every
leg (HIP path, CPU oracle, fixtures) sees bit-identical inputs.
"""
import numpy as np

# config id -> (n, d, p, q, submethod)
CONFIGS = {
    2: dict(n=1024, d=3, p=16, q=4, submethod='full', dtype='f64'),
    3: dict(n=4096, d=6, p=64, q=8, submethod='full', dtype='f64'),
    4: dict(n=16384, d=10, p=32, q=8, submethod='full', dtype='f32'),
    5: dict(n=2048, d=3, p=12, q=6, submethod='rep', reps=5, dtype='f64'),
}


def make_full(c, n, d, p, q, noise=0.1):
    """x ~ U(0,1)^(n x d);  y = W sin(2 pi x w_k + phi_k) + noise * eps, layout (p, n)."""
    rng = np.random.default_rng(20260000 + c)
    x = rng.uniform(0.0, 1.0, (n, d))
    w = rng.standard_normal((d, q))
    ph = rng.uniform(0.0, 2.0 * np.pi, q)
    g = np.sin(2.0 * np.pi * (x @ w) + ph[None, :]).T          # (q, n)
    load = rng.standard_normal((p, q))
    y = load @ g + noise * rng.standard_normal((p, n))
    return x, y


def make_rep(c, n_unique, reps, d, p, q):
    """Replicated inputs (np.tile pattern of the reference's test_rep.py:16-20), heteroskedastic noise."""
    rng = np.random.default_rng(20260000 + c)
    xu = rng.uniform(0.0, 1.0, (n_unique, d))
    x = np.tile(xu, (reps, 1))
    w = rng.standard_normal((d, q))
    ph = rng.uniform(0.0, 2.0 * np.pi, q)
    g = np.sin(2.0 * np.pi * (x @ w) + ph[None, :]).T
    load = rng.standard_normal((p, q))
    sd = 0.05 + 0.2 * x[:, 0]
    y = load @ g + sd[None, :] * rng.standard_normal((p, x.shape[0]))
    return x, y


def make_config(c, **override):
    cfg = dict(CONFIGS[c])
    cfg.update(override)
    if cfg['submethod'] == 'rep':
        x, y = make_rep(c, cfg['n'], cfg['reps'], cfg['d'], cfg['p'], cfg['q'])
    else:
        x, y = make_full(c, cfg['n'], cfg['d'], cfg['p'], cfg['q'])
    return x, y, cfg


def param_points(c, u0, count=3, step=0.3):
    """theta_0 = init values; theta_i = u0 + step * N(0,1), seed c*10+i (unconstrained space)."""
    pts = [np.array(u0, dtype=np.float64)]
    for i in range(1, count):
        rng = np.random.default_rng(c * 10 + i)
        pts.append(u0 + step * rng.standard_normal(u0.shape))
    return pts
