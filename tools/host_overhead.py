"""Host time per evaluation: wall-clock of LCGP.loss_and_grad() minus the GPU time of the same evaluation.

    python tools/host_overhead.py [config id, default 3]

GPU time = HIP events around everything one evaluation enqueues (H2D of the theta block, lcgp_nll_grad,
lcgp_pack_partial); wall time = loss_and_grad() with the device idle before and after.  The difference is what the
host adds: packing theta, launching ~200 kernels ahead of the GPU, the D2H copy of the reduced vector, the chain rule.
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from lcgp_amd import LCGP, synth  # noqa: E402

cfgid = int(sys.argv[1]) if len(sys.argv) > 1 else 3
over = {'q': int(sys.argv[2])} if len(sys.argv) > 2 else {}          # e.g. `3 1`: one rank's share of the headline configuration
x, y, cfg = synth.make_config(cfgid, **over)
m = LCGP(y=y, x=x, q=cfg['q'], submethod=cfg['submethod'], dtype='float64' if cfg['dtype'] == 'f64' else 'float32')
pts = synth.param_points(cfgid, m._get_flat())
for u in pts:
    m.loss_and_grad(u)
eng = m._engine
st = torch.cuda.current_stream(eng.device)
sig = np.exp(0.5 * np.repeat(m.lsigma2s.numpy(), np.asarray(m.diag_error_structure, int))) / m._std
theta = m._theta_rows(sig)
def measure():
    wall, gpu, enq = [], [], []
    for rep in range(12):
        u = pts[rep % len(pts)]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        m.loss_and_grad(u)
        wall.append(1e3 * (time.perf_counter() - t0))
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record(st)
        eng.evaluate_partial(theta)
        e1.record(st)
        enq.append(1e3 * (time.perf_counter() - t0))
        torch.cuda.synchronize()
        gpu.append(e0.elapsed_time(e1))
    return np.median(wall), np.median(gpu), np.median(enq)


# the launch plan built ONCE by the caller (lcgp_plan_build, the engine's default) against planning inside every call
for label, use_plan in (('caller-owned plan', True), ('plan per call (plan_host = NULL)', False)):
    eng.use_plan = use_plan
    m.loss_and_grad(pts[0])
    w, g, e = measure()
    print('cfg %d (n=%d q=%d %s), %s: loss_and_grad %.3f ms wall, GPU %.3f ms, host overhead %.3f ms per evaluation '
          '(enqueue of one evaluation returns after %.3f ms)' % (cfgid, int(m.n), int(m.q), cfg['dtype'], label, w, g, w - g, e))
eng.use_plan = True
