"""Metric fixtures from the REFERENCE's own evaluation.py (src/lcgp/evaluation.py:5-63).

That file needs only numpy and scipy, so -- unlike lcgp.py / covmat.py, which need TensorFlow -- it can be
loaded in the build container, by file path, without importing the `lcgp` package:

    python tests/golden/make_eval_golden.py      # needs /root/reference; writes tests/golden/evaluation_golden.npz

Only numbers are stored (seeds of the inputs + the reference's outputs); no reference file travels anywhere.
tests/test_evaluation_golden.py regenerates the inputs from the seeds and checks lcgp_amd/evaluation.py.
"""
import importlib.util
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get('LCGP_REFERENCE', '/root/reference')

# (seed, p, n): shapes cover p = 1, n = 1, and the KAT's (3, 400)
CASES = [(1, 3, 400), (2, 1, 17), (3, 5, 1), (4, 8, 64), (5, 2, 1000)]


def case_inputs(seed, p, n):
    """y, predictive mean, predictive variances (p, n) and full covariances (p, p, n), all from one seed."""
    rng = np.random.default_rng(7700 + seed)
    y = rng.standard_normal((p, n)) * rng.uniform(0.5, 3.0, (p, 1)) + rng.uniform(-2, 2, (p, 1))
    mean = y + 0.3 * rng.standard_normal((p, n))
    var = rng.uniform(0.01, 0.5, (p, n))
    a = rng.standard_normal((p, p, n))
    cov = np.einsum('ijn,kjn->ikn', a, a) + 0.1 * np.eye(p)[:, :, None]
    return y, mean, var, cov


def main():
    spec = importlib.util.spec_from_file_location('ref_evaluation', os.path.join(REF, 'src', 'lcgp', 'evaluation.py'))
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    out = {'cases': np.array(CASES)}
    for seed, p, n in CASES:
        y, mean, var, cov = case_inputs(seed, p, n)
        key = 'case%d/' % seed
        out[key + 'rmse'] = np.float64(ref.rmse(y, mean))
        if n > 1:                        # (a single point has zero range: the reference divides by it)
            out[key + 'nrmse'] = np.float64(ref.normalized_rmse(y, mean))
        cover, length = ref.intervalstats(y, mean, var)
        out[key + 'cover'], out[key + 'length'] = np.float64(cover), np.float64(length)
        out[key + 'dss_diag'] = np.float64(ref.dss(y, mean, var, use_diag=True))
        out[key + 'dss_full'] = np.float64(ref.dss(y, mean, cov, use_diag=False))
    np.savez_compressed(os.path.join(HERE, 'evaluation_golden.npz'), **out)
    print('wrote', os.path.join(HERE, 'evaluation_golden.npz'), {k: float(v) for k, v in out.items() if k != 'cases'})


if __name__ == '__main__':
    sys.exit(main())
