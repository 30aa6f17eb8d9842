"""Where and why does the float32 factorisation of configs[3] (n = 16384, d = 10, p = 32, q = 8) break down during fit()?

Runs LCGP(dtype='float32').fit() with the float64 repeat enabled and the switch to float64-only DISABLED (so every
evaluation is tried in float32 first) and logs every evaluation whose float32 factorisation failed: the evaluation number,
per failing component the status word (1 + index of the first non-positive pivot, LAPACK style), the constrained
parameters that set the scale of A = I + D (C o s s^T), and lambda_max(A) by power iteration on the float64 matrix
(lambda_min(A) >= 1 analytically, so lambda_max IS the condition number up to that bound).

    python tools/fp32_breakdown.py [config id = 4] [max logged failures = 12]
"""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lcgp_amd import LCGP, synth, _hip  # noqa: E402

cfgid = int(sys.argv[1]) if len(sys.argv) > 1 else 4
maxlog = int(sys.argv[2]) if len(sys.argv) > 2 else 12
x, y, cfg = synth.make_config(cfgid)
m = LCGP(y=y, x=x, q=cfg['q'], dtype='float32')
m.float32_switch_after = 0             # keep trying float32 at every point
q, d, n = int(m.q), int(m.d), int(m.n)
print('configs[%d]: n = %d, d = %d, p = %d, q = %d; diag_D = %s' % (cfgid - 1, n, d, int(m.p), q, np.array2string(m.diag_D.numpy(), precision=3)))


def lam_max(eng, k, iters=40):
    """power iteration on A_k = I + D_k (C_k o s s^T) built by the float64 kernel-build launch (lower tiles, mirrored)"""
    lib, st = eng.lib, eng._stream()
    _hip.check(lib.lcgp_kernel_build(st, eng.dtype, eng.kernel_id, eng.n, eng.d, eng.p, eng.q_local, eng._p(eng.x), eng._p(eng.sr),
                                     eng._p(eng.theta_dev), eng._p(eng.workspace)), 'build')
    a = torch.empty((n, n), dtype=torch.float64, device=eng.device)
    _hip.check(lib.lcgp_fetch_matrix(st, eng.dtype, eng.n, eng.d, eng.p, eng.q_local, eng._p(eng.workspace), 0, k,
                                     C.c_void_p(a.data_ptr())), 'fetch')
    v = torch.ones(n, dtype=torch.float64, device=eng.device) / np.sqrt(n)
    lam = 0.0
    for _ in range(iters):
        w = a @ v
        lam = float(torch.linalg.vector_norm(w))
        v = w / lam
    del a
    return lam


count, logged = [0], [0]
orig = m.loss_and_grad


def wrapped(u):
    count[0] += 1
    before = m.float32_fallbacks
    try:
        out = orig(u)
        ok64 = True
    except np.linalg.LinAlgError:
        out, ok64 = None, False
    if m.float32_fallbacks > before and logged[0] < maxlog:
        logged[0] += 1
        info32 = m._engine.out_dev.cpu().numpy().reshape(q, -1)[:, 2].copy()
        lLmb, lLmb0, ls2, lnug = (t.numpy() for t in m.get_param())
        print('evaluation %d: float32 factorisation failed (float64 repeat %s)' % (count[0], 'succeeded, NLL %.6e' % out[0] if ok64 else 'failed too'))
        for k in range(q):
            if info32[k] != 0:
                lm = lam_max(m._engine64, k) if ok64 else float('nan')
                if ok64:
                    m._engine64.enqueue()          # restore the factorisation the caches expect in the float64 workspace
                print('   k=%d  info32 = %6d  D_k = %9.3e  scale = %9.3e  nugget = %8.2e  min ell = %8.4f  max ell = %8.2f  lambda_max(A) = %9.3e  eps32 x lambda_max = %.2f'
                      % (k, int(info32[k]), m.diag_D.numpy()[k], lLmb0[k], lnug[k], lLmb[k].min(), lLmb[k].max(), lm, 1.19e-7 * lm))
    if out is None:
        raise np.linalg.LinAlgError('not positive definite')
    return out


m.loss_and_grad = wrapped
t0 = time.perf_counter()
m.fit()
t1 = time.perf_counter()
r = m.opt_result
print('fit: %.1f s, %d iterations, %d evaluations (%d restarts), final loss %.6e, %d evaluations repeated in float64'
      % (t1 - t0, r.nit, r.nfev, len(r.restarts) - 1, r.fun, m.float32_fallbacks))
