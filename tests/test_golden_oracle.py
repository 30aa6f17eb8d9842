"""The committed golden vectors are reproducible from the oracle (guards against drift in either)."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, 'golden'))
import make_golden  # noqa: E402

GOLD = np.load(os.path.join(HERE, 'golden', 'lcgp_golden.npz'))


@pytest.mark.parametrize("want", ['kat_rep', 'full_n64', 'full_n150_grouped', 'rep_n70', 'rep_n33_raw'])
def test_oracle_reproduces_golden(want):
    for name, c, m, x0 in make_golden.case_models():
        if name != want:
            if name == 'cfg2_n1024':
                break
            continue
        for i, u in enumerate(GOLD[name + '/u']):
            v, g = m.loss_and_grad_unconstrained(u)
            assert abs(v - GOLD[name + '/nll'][i]) <= 1e-10 * max(1.0, abs(v))
            np.testing.assert_allclose(g, GOLD[name + '/grad'][i], rtol=1e-7, atol=1e-9 * np.max(np.abs(g)))
        m.set_unconstrained(GOLD[name + '/u'][1])
        pred = m.predict(GOLD[name + '/x0'])
        np.testing.assert_allclose(pred[0], GOLD[name + '/ypred'], rtol=1e-8, atol=1e-10)
        np.testing.assert_allclose(pred[1], GOLD[name + '/ypredvar'], rtol=1e-7, atol=1e-10)
        return
    raise AssertionError(want)
