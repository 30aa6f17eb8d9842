"""Host enqueue time vs GPU time of one lcgp_nll_grad call, for several group counts."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from lcgp_amd import LCGP, synth, _hip
x, y, cfg = synth.make_config(3)
m = LCGP(y=y, x=x, q=cfg['q'])
u = m._get_flat()
m.loss_and_grad(u)
eng = m._engine
lib = _hip.load()
for g in (1, 2, 4):
    lib.lcgp_set_tuning(1, g)
    for rep in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.enqueue()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
    print('groups %d: enqueue %.3f ms, total %.3f ms' % (g, 1e3 * (t1 - t0), 1e3 * (t2 - t0)))
