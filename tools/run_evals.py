"""Runs a few evaluations of one configuration (for rocprofv3 traces).

    python tools/run_evals.py <config id> <q or 0> <evaluations> [sched field=value ...]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from lcgp_amd import LCGP, synth, _hip  # noqa: E402

cfgid, q, nev = (int(a) for a in sys.argv[1:4])
over = {'q': q} if q else {}
x, y, cfg = synth.make_config(cfgid, **over)
m = LCGP(y=y, x=x, q=cfg['q'], submethod=cfg['submethod'], dtype='float64' if cfg['dtype'] == 'f64' else 'float32')
pts = synth.param_points(cfgid, m._get_flat())
if len(sys.argv) > 4:
    sc = _hip.default_sched()
    for kv in sys.argv[4:]:
        k, v = kv.split('=')
        assert hasattr(sc, k), k
        setattr(sc, k, int(v))
    m._get_engine().sched = sc
for i in range(nev):
    m.loss_and_grad(pts[i % len(pts)])
torch.cuda.synchronize()
print('done')
