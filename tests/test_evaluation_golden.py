"""lcgp_amd/evaluation.py against numbers produced by the REFERENCE's evaluation.py (fixtures written by
tests/golden/make_eval_golden.py in the build container; the reference file itself does not travel)."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, 'golden'))
import make_eval_golden as gen  # noqa: E402

from lcgp_amd import evaluation  # noqa: E402

GOLD = np.load(os.path.join(HERE, 'golden', 'evaluation_golden.npz'))


@pytest.mark.parametrize('seed,p,n', gen.CASES)
def test_metrics_equal_the_reference_outputs(seed, p, n):
    assert [seed, p, n] in GOLD['cases'].tolist()
    y, mean, var, cov = gen.case_inputs(seed, p, n)
    key = 'case%d/' % seed
    tol = dict(rtol=1e-12, atol=1e-14)
    np.testing.assert_allclose(evaluation.rmse(y, mean), GOLD[key + 'rmse'], **tol)
    if n > 1:
        np.testing.assert_allclose(evaluation.normalized_rmse(y, mean), GOLD[key + 'nrmse'], **tol)
    cover, length = evaluation.intervalstats(y, mean, var)
    np.testing.assert_allclose(cover, GOLD[key + 'cover'], **tol)
    np.testing.assert_allclose(length, GOLD[key + 'length'], **tol)
    np.testing.assert_allclose(evaluation.dss(y, mean, var, use_diag=True), GOLD[key + 'dss_diag'], **tol)
    np.testing.assert_allclose(evaluation.dss(y, mean, cov, use_diag=False), GOLD[key + 'dss_full'], rtol=1e-10, atol=1e-12)


def test_generator_is_not_needed_at_test_time():
    """The fixture file is self-contained: nothing here reads /root/reference (absent on the GPU box)."""
    src = open(os.path.join(HERE, 'test_evaluation_golden.py')).read()
    assert 'spec_from_file_location' not in src.split('def test_generator_is_not_needed_at_test_time')[0]
