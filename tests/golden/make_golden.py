"""Generates the committed golden vectors from the CPU oracle (run AFTER tests/test_oracle_kat.py passes).

The reference itself cannot be imported in the build container (tensorflow / tfp / gpflow absent),
so these vectors come from the oracle, which is pinned by the reference's stored notebook outputs
(KAT-1 / KAT-2) and by the cross-identities in tests/test_oracle_identities.py.

    python tests/golden/make_golden.py            # small cases  -> lcgp_golden.npz        (seconds)
    python tests/golden/make_golden.py --ragged   # n = 4000 (not a multiple of any tile size), q = 2 -> lcgp_golden_ragged.npz
    python tests/golden/make_golden.py --large    # BASELINE.json's full-size configurations -> lcgp_golden_large.npz
                                                  # (n=4096 q=8 fp64; rep n_unique=2048 x 5; the 4096-point prefix of
                                                  #  the n=16384 fp32 configuration in fp64) -- about 5 minutes of CPU

    python tests/golden/make_golden.py --huge     # ONE component of configs[3] at its FULL size n = 16384 in float64 (the size
                                                  # the other fixtures only reach through properties): NLL_k and its 12 kernel
                                                  # gradient entries at one parameter point -> lcgp_golden_huge.npz
                                                  # (about 20 GB of host memory, a few CPU-minutes)

Inputs are regenerated from seeds (lcgp_amd/synth.py, tests/kat_data.py); only expected outputs are
stored: NLL, gradient w.r.t. the unconstrained vector, per-component pieces, and a few predictions.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import lcgp_oracle as orc  # noqa: E402
from lcgp_amd import synth  # noqa: E402
from tests import kat_data  # noqa: E402


def case_models():
    xtr, ytr, xte, _ = kat_data.kat_dataset()
    yield 'kat_rep', 1, orc.OracleLCGP(y=ytr, x=xtr, q=3, diag_error_structure=[1, 1, 1], submethod='rep'), xte[::40]
    x, y = synth.make_full(64, 64, 2, 4, 3)
    yield 'full_n64', 64, orc.OracleLCGP(y=y, x=x, q=3, submethod='full'), \
        np.random.default_rng(640).uniform(0, 1, (7, 2))
    x, y = synth.make_full(65, 150, 3, 5, 2)
    yield 'full_n150_grouped', 65, orc.OracleLCGP(y=y, x=x, q=2, submethod='full', diag_error_structure=[2, 3],
                                                   robust_mean=False), \
        np.random.default_rng(650).uniform(0, 1, (5, 3))
    x, y = synth.make_rep(66, 70, 3, 2, 4, 4)
    yield 'rep_n70', 66, orc.OracleLCGP(y=y, x=x, submethod='rep'), np.random.default_rng(660).uniform(0, 1, (6, 2))
    x, y = synth.make_rep(67, 33, 4, 1, 3, 3)
    yield 'rep_n33_raw', 67, orc.OracleLCGP(y=y, x=x, submethod='rep', rep_standardize_ybar=False), \
        np.random.default_rng(670).uniform(0, 1, (6, 1))
    x, y, cfg = synth.make_config(2)
    yield 'cfg2_n1024', 2, orc.OracleLCGP(y=y, x=x, q=cfg['q'], submethod='full'), None


def large_case_models():
    """BASELINE.json configs[2], [4] and (its 4096-point prefix, fp64 oracle values for the fp32 run) [3]."""
    x, y, cfg = synth.make_config(3)
    yield 'cfg3_n4096', 3, orc.OracleLCGP(y=y, x=x, q=cfg['q'], submethod='full'), \
        np.random.default_rng(30).uniform(0, 1, (10, cfg['d']))
    x, y, cfg = synth.make_config(5)
    yield 'cfg5_rep_n2048x5', 5, orc.OracleLCGP(y=y, x=x, q=cfg['q'], submethod='rep'), \
        np.random.default_rng(50).uniform(0, 1, (10, cfg['d']))
    x, y, cfg = synth.make_config(4)
    yield 'cfg4_prefix4096', 4, orc.OracleLCGP(y=y[:, :4096], x=x[:4096], q=cfg['q'], submethod='full'), \
        np.random.default_rng(40).uniform(0, 1, (10, cfg['d']))


def ragged_case_models():
    """A headline-scale size that is NOT a multiple of the tile sizes: n = 4000 pads to 4096 (identity padding inside the
    last 128-tile, filler tiles, the epilogue of the one-launch A^-1 = W^T W); two components = the 4-GPU share."""
    x, y = synth.make_full(4000, 4000, 6, 8, 2)
    yield 'ragged_n4000', 7, orc.OracleLCGP(y=y, x=x, q=2, submethod='full'), \
        np.random.default_rng(70).uniform(0, 1, (10, 6))


def generate(cases, fname):
    out = {}
    for name, c, m, x0 in cases:
        pts = synth.param_points(c, m.get_unconstrained())
        vals, grads = [], []
        for u in pts:
            v, g = m.loss_and_grad_unconstrained(u)
            vals.append(v)
            grads.append(g)
        out[name + '/u'] = np.stack(pts)
        out[name + '/nll'] = np.array(vals)
        out[name + '/grad'] = np.stack(grads)
        out[name + '/diag_D'] = np.asarray(m.diag_D)
        if x0 is not None:
            m.set_unconstrained(pts[1])
            pred = m.predict(x0, return_fullcov=(m.submethod == 'full'))
            out[name + '/x0'] = x0
            out[name + '/ypred'] = pred[0]
            out[name + '/ypredvar'] = pred[1]
            out[name + '/yconfvar'] = pred[2]
            if m.submethod == 'full':
                out[name + '/fullcov'] = pred[3]
        print(name, 'n=%d q=%d' % (m.n, m.q), 'nll', vals, flush=True)
    np.savez_compressed(os.path.join(HERE, fname), **out)
    print('wrote', os.path.join(HERE, fname))


def generate_huge():
    """configs[3] (n = 16384, d = 10, p = 32, q = 8) at full size, component 0 only, parameter point theta_1: the Cholesky form
    of lcgp.py:635-666 with the closed-form gradient (oracle/cpu_baseline.py::chol_form_component, the same restatement the
    bench's CPU leg times, pinned by tests/test_oracle_identities.py against the literal eigh form and autodiff).  Stored:
    the theta row of the component (so that the GPU test feeds the identical numbers), half log-determinant, quadratic
    form, NLL_k, d/d(ell_1..10, scale, nug), and Y (b - z)."""
    from oracle import cpu_baseline
    x, y, cfg = synth.make_config(4)
    m = orc.OracleLCGP(y=y, x=x, q=cfg['q'], submethod='full')
    u = synth.param_points(4, m.get_unconstrained())[1]
    m.set_unconstrained(u)
    lLmb, lLmb0, ls2b, lnug = m.get_param()
    k = 0
    dt, pc = cpu_baseline.chol_form_component(m.x, m.y, m.phi[:, k], float(m.diag_D[k]), lLmb[k], float(lLmb0[k]),
                                              float(lnug[k]), ls2b)
    theta = np.concatenate([lLmb[k], [lLmb0[k], lnug[k], m.diag_D[k]], m.phi[:, k] / np.exp(0.5 * ls2b)])
    out = {'cfg4_full_k0/u': u, 'cfg4_full_k0/theta': theta, 'cfg4_full_k0/half_logdet': pc['half_logdet'],
           'cfg4_full_k0/quad': pc['quad'], 'cfg4_full_k0/nll_k': pc['value'],
           'cfg4_full_k0/g_kernel': np.concatenate([pc['g_ell'], [pc['g_scale'], pc['g_nug']]]),
           'cfg4_full_k0/gsig': pc['gsig'], 'cfg4_full_k0/diag_D': np.asarray(m.diag_D)}
    print('cfg4 full size, component 0: %.1f s on the host; NLL_k %.10g, half logdet %.10g, quad %.10g' % (
        dt, pc['value'], pc['half_logdet'], pc['quad']), flush=True)
    np.savez_compressed(os.path.join(HERE, 'lcgp_golden_huge.npz'), **out)
    print('wrote', os.path.join(HERE, 'lcgp_golden_huge.npz'))


def main():
    if '--huge' in sys.argv[1:]:
        generate_huge()
        return
    if '--ragged' in sys.argv[1:]:
        generate(ragged_case_models(), 'lcgp_golden_ragged.npz')
    elif '--large' in sys.argv[1:]:
        generate(large_case_models(), 'lcgp_golden_large.npz')
    else:
        generate(case_models(), 'lcgp_golden.npz')


if __name__ == '__main__':
    main()
