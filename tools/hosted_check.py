"""Hosted panels (lcgp_sched.hosted) against the default schedule on the same inputs: L, L^-1, the output block.

    python tools/hosted_check.py [config id] [q or 0] [n or 0] [sched field=value ...]
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from lcgp_amd import LCGP, synth, _hip  # noqa: E402

cfgid = int(sys.argv[1]) if len(sys.argv) > 1 else 3
q = int(sys.argv[2]) if len(sys.argv) > 2 else 0
n = int(sys.argv[3]) if len(sys.argv) > 3 else 0
over = {'q': q} if q else {}
x, y, cfg = synth.make_config(cfgid, **over)
if n:
    x, y = x[:n], y[:, :n]
m = LCGP(y=y, x=x, q=cfg['q'], submethod=cfg['submethod'], dtype='float64' if cfg['dtype'] == 'f64' else 'float32')
eng = m._get_engine()
sig_eff = np.exp(0.5 * np.repeat(m.lsigma2s.numpy(), np.asarray(m.diag_error_structure, int))) / m._std
theta = m._theta_rows(sig_eff)


def run(pairs):
    sc = _hip.default_sched()
    for k, v in pairs:
        assert hasattr(sc, k), k
        setattr(sc, k, v)
    eng.sched = sc
    out = eng.evaluate(theta).copy()
    torch.cuda.synchronize()
    ks = sorted({0, eng.q_local - 1})
    return out, [eng.fetch_matrix(0, k) for k in ks], [eng.fetch_matrix(1, k) for k in ks]


extra = [(a.split('=')[0], int(a.split('=')[1])) for a in sys.argv[4:]]
o0, L0, W0 = run([])
o1, L1, W1 = run([('hosted', 1)] + extra)
o2, _, _ = run([('hosted', 1)] + extra)
scale = np.abs(o0).max(axis=0) + 1e-300
print('n = %d, q_local = %d, %s' % (x.shape[0], eng.q_local, cfg['dtype']))
print('output block, hosted vs default: max rel diff %.3e (info %s / %s)' % (np.max(np.abs(o1 - o0) / scale), o0[:, 2], o1[:, 2]))
print('output block, hosted run twice: bitwise equal = %s' % np.array_equal(o1, o2))
for i, (a, b) in enumerate(zip(L0, L1)):
    print('L   component %d: max |diff| %.3e (max |L| %.3e)' % (i, np.max(np.abs(np.tril(a) - np.tril(b))), np.max(np.abs(np.tril(a)))))
for i, (a, b) in enumerate(zip(W0, W1)):
    print('L^-1 component %d: max |diff| %.3e (max %.3e)' % (i, np.max(np.abs(np.tril(a) - np.tril(b))), np.max(np.abs(np.tril(a)))))
ok = np.max(np.abs(o1 - o0) / scale) < (1e-9 if cfg['dtype'] == 'f64' else 1e-2) and np.array_equal(o1, o2)
print('OK' if ok else 'MISMATCH')
sys.exit(0 if ok else 1)
