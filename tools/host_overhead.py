"""Host enqueue time vs GPU time of one lcgp_nll_grad call, for several group counts."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from lcgp_amd import LCGP, synth, _hip
cfgid = int(sys.argv[1]) if len(sys.argv) > 1 else 3
x, y, cfg = synth.make_config(cfgid)
m = LCGP(y=y, x=x, q=cfg['q'])
u = m._get_flat()
m.loss_and_grad(u)
eng = m._engine
lib = _hip.load()
for g in (1,):
    for rep in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.enqueue()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
    print('cfg %d groups %d: enqueue %.3f ms, total %.3f ms' % (cfgid, g, 1e3 * (t1 - t0), 1e3 * (t2 - t0)))

import time as _t
t0=_t.perf_counter()
for i in range(20): m.loss_and_grad(u)
print('cfg %d: loss_and_grad %.3f ms per call' % (cfgid, 1e3*(_t.perf_counter()-t0)/20))
