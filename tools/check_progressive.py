"""Progressive inverse (sched.progressive_tiles) against the classic schedule on the GPU: NLL / gradient / predictions of a
few shapes through both, largest relative difference (the two differ by rounding only).

    python tools/check_progressive.py
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lcgp_amd import LCGP, synth, _hip  # noqa: E402


def run(tag, x, y, q, submethod='full', dtype='float64', seed=1, ob=0):
    out = []
    for prog in (0, 1 << 30):
        m = LCGP(y=y, x=x, q=q, submethod=submethod, dtype=dtype)
        sc = _hip.default_sched()
        sc.progressive_tiles = prog
        sc.progressive_lauum = 1 << 30
        sc.outer_blocks = ob
        m._get_engine().sched = sc
        u = synth.param_points(seed, m._get_flat())[1]
        v, g = m.loss_and_grad(u)
        pr = m.predict(x[:37] + 0.01)
        w = m._engine.fetch_matrix(1, 0)
        ai = m._engine.fetch_matrix(2, 0)
        out.append((v, g, pr[0].numpy(), pr[1].numpy(), np.tril(w), np.tril(ai)))
    a, b = out
    ev = abs(a[0] - b[0]) / abs(a[0])
    eg = np.max(np.abs(a[1] - b[1])) / np.max(np.abs(a[1]))
    ep = max(np.max(np.abs(a[2] - b[2])) / np.max(np.abs(a[2])), np.max(np.abs(a[3] - b[3])) / np.max(np.abs(a[3])))
    ew = np.max(np.abs(a[4] - b[4])) / np.max(np.abs(a[4]))
    ea = np.max(np.abs(a[5] - b[5])) / np.max(np.abs(a[5]))
    print('%-34s nll %.2e  grad %.2e  predict %.2e  L^-1 %.2e  A^-1 %.2e' % (tag, ev, eg, ep, ew, ea), flush=True)
    return max(ev, eg, ep, ew, ea)


worst = 0.0
for (n, d, p, q) in ((100, 2, 4, 2), (300, 3, 5, 3), (700, 2, 6, 2), (1100, 4, 6, 1), (1500, 3, 4, 2)):
    x, y = synth.make_full(n + q, n, d, p, q)
    worst = max(worst, run('full n=%d d=%d q=%d' % (n, d, q), x, y, q))
x, y = synth.make_full(5, 900, 3, 4, 2)
worst = max(worst, run('full n=900 outer_blocks=2', x, y, 2, ob=2))
worst = max(worst, run('full n=900 outer_blocks=8', x, y, 2, ob=8))
x, y, cfg = synth.make_config(2)
worst = max(worst, run('configs[1] n=1024 q=4', x, y, cfg['q'], seed=2))
x, y, cfg = synth.make_config(3, q=1)
worst = max(worst, run('configs[2] n=4096 q=1', x, y, 1, seed=3))
x, y, cfg = synth.make_config(3, q=2)
worst = max(worst, run('configs[2] n=4096 q=2', x, y, 2, seed=3))
x, y, cfg = synth.make_config(5)
worst = max(worst, run('configs[4] rep 2048x5 q=6', x, y, cfg['q'], submethod='rep', seed=5))
print('worst %.2e' % worst)
x, y, cfg = synth.make_config(4, n=4096, q=2)
run('configs[3] prefix n=4096 q=2 float32', x, y, 2, dtype='float32', seed=4)
sys.exit(0 if worst < 1e-9 else 1)
