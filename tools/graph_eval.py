"""Does replaying ONE captured HIP graph per evaluation beat ~120 plain launches?  (wall time per synchronised evaluation)

    python tools/graph_eval.py <config id> <q or 0> [evaluations]
The graph is captured by the CALLER around the library's enqueue call (torch.cuda.CUDAGraph on the current stream): the
library itself creates no stream, event or graph.
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from lcgp_amd import LCGP, synth, _hip  # noqa: E402
from lcgp_amd import dist as _dist  # noqa: E402

cfgid, q = int(sys.argv[1]), int(sys.argv[2])
nev = int(sys.argv[3]) if len(sys.argv) > 3 else 50
over = {'q': q} if q else {}
x, y, cfg = synth.make_config(cfgid, **over)
m = LCGP(y=y, x=x, q=cfg['q'], submethod=cfg['submethod'], dtype='float64' if cfg['dtype'] == 'f64' else 'float32')
eng = m._get_engine()
sig_eff = np.exp(0.5 * np.repeat(m.lsigma2s.numpy(), np.asarray(m.diag_error_structure, int))) / m._std
theta = m._theta_rows(sig_eff)
ref = eng.evaluate_partial(theta).cpu().numpy().copy()

side = torch.cuda.Stream()


def plain():
    part = eng.evaluate_partial(theta)
    return _dist.reduce_to_host(part, None)


g = torch.cuda.CUDAGraph()
with torch.cuda.stream(side):
    eng.evaluate_partial(theta)          # warm-up on the capture stream
    side.synchronize()
    with torch.cuda.graph(g, stream=side):
        st = torch.cuda.current_stream()
        eng.enqueue(st)
        import ctypes as C
        _hip.check(eng.lib.lcgp_pack_partial(C.c_void_p(st.cuda_stream), eng.d, eng.p, eng.q_local, eng.q_total, *eng._pack_ptrs),
                   'lcgp_pack_partial')


def graphed():
    with torch.cuda.stream(side):
        eng.upload_theta(theta, 0.0, torch.cuda.current_stream())
        g.replay()
        return _dist.reduce_to_host(eng.partial_dev, None)


out = graphed()
assert np.array_equal(out, ref), 'graph replay differs from plain launches'
for name, fn in (('plain launches', plain), ('graph replay', graphed), ('plain launches', plain), ('graph replay', graphed)):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(nev):
        fn()
    torch.cuda.synchronize()
    print('%-16s %8.3f ms per synchronised evaluation (config %d, q_local %d)' % (name, (time.perf_counter() - t0) / nev * 1e3, cfgid, eng.q_local))
