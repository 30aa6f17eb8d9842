"""Wall-clock of LCGP.fit() + predict() at a synthetic configuration (default: the headline n=4096, d=6, q=8)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from lcgp_amd import LCGP, synth
cfgid = int(sys.argv[1]) if len(sys.argv) > 1 else 3
force = sys.argv[2] if len(sys.argv) > 2 else None          # 'float64': the float32 configuration fitted in float64 (comparison)
x, y, cfg = synth.make_config(cfgid)
t0 = time.perf_counter()
dtype = force or ('float64' if cfg['dtype'] == 'f64' else 'float32')
m = LCGP(y=y, x=x, q=cfg['q'], submethod=cfg['submethod'], dtype=dtype)
t1 = time.perf_counter()
l0 = float(m.loss())
t2 = time.perf_counter()
m.fit()
t3 = time.perf_counter()
res = m.opt_result
x0 = np.random.default_rng(1).uniform(0, 1, (2000, x.shape[1]))
out = m.predict(x0)
t4 = time.perf_counter()
print('config %d: construct %.2f s | fit %.2f s (%d iterations, %d evaluations, %.1f ms/eval, loss %.4f -> %.4f, %s) | '
      'predict(2000) %.3f s' % (cfgid, t1 - t0, t3 - t2, res.nit, res.nfev, 1e3 * (t3 - t2) / res.nfev, l0, res.fun,
                                res.message if isinstance(res.message, str) else res.message.decode(), t4 - t3)
      + (' | dtype %s' % dtype)
      + (' | float32: %d evaluations repeated in float64, %s, %d restarts' % (
          m.float32_fallbacks, 'continued in float64 only' if res.float64_only else 'stayed in float32', len(res.restarts) - 1)
         if dtype == 'float32' else ''))
