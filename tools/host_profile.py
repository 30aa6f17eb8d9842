"""cProfile of the host side of loss_and_grad (where the time outside the GPU goes, function by function).

    python tools/host_profile.py <config id> <q or 0> <evaluations>
"""
import cProfile
import os
import pstats
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from lcgp_amd import LCGP, synth  # noqa: E402

cfgid, q, nev = (int(a) for a in sys.argv[1:4])
over = {'q': q} if q else {}
x, y, cfg = synth.make_config(cfgid, **over)
m = LCGP(y=y, x=x, q=cfg['q'], submethod=cfg['submethod'], dtype='float64' if cfg['dtype'] == 'f64' else 'float32')
pts = synth.param_points(cfgid, m._get_flat())
for i in range(5):
    m.loss_and_grad(pts[i % len(pts)])
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for i in range(nev):
    m.loss_and_grad(pts[i % len(pts)])
pr.disable()
st = pstats.Stats(pr)
st.sort_stats('tottime').print_stats(28)
