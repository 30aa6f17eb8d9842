"""A/B timing of launch schedules inside ONE process on ONE box (box-to-box noise is +-3 %).

usage: python tools/ab.py [--q Q] [--n N] [--reps R] [--steps K] "name:field=val,field=val" "name2:..." ...
Each setting is a comma-separated list of lcgp_sched field=value pairs (include/lcgp_hip.h; empty = defaults), passed
per call through the engine -- the library has no global tuning state.  The settings are timed round-robin R times,
K evaluations each; prints the min and mean ms per evaluation.
"""
import argparse
import sys
import time

import numpy as np
import torch

sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from lcgp_amd import LCGP, synth, _hip  # noqa: E402



def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--q', type=int, default=8)
    ap.add_argument('--n', type=int, default=None)
    ap.add_argument('--config', type=int, default=3, help='BASELINE.json configuration (3 = headline fp64, 4 = fp32 n=16384)')
    ap.add_argument('--reps', type=int, default=4)
    ap.add_argument('--steps', type=int, default=6)
    ap.add_argument('--stages', action='store_true', help='also print the HIP-event times of the stages (bench.py: stage_times)')
    ap.add_argument('settings', nargs='+')
    a = ap.parse_args()
    x, y, cfg = synth.make_config(a.config)
    if cfg['submethod'] != 'full':
        raise SystemExit('ab.py times the full path only')
    if a.n:
        x, y = x[:a.n], y[:, :a.n]
    a.n = x.shape[0]
    dtype = 'float64' if cfg['dtype'] == 'f64' else 'float32'
    m = LCGP(y=y, x=x, q=a.q, dtype=dtype)          # q components = one rank's share of the configuration
    u = m._get_flat()
    eng = None
    res = {}
    names = []
    for s in a.settings:
        name, _, kv = s.partition(':')
        pairs = [(p.split('=')[0], int(p.split('=')[1])) for p in kv.split(',') if p]
        names.append((name, pairs))
        res[name] = []

    eng = m._get_engine()

    def apply(pairs):
        sc = _hip.default_sched()
        for k, v in pairs:
            assert hasattr(sc, k), k
            setattr(sc, k, v)
        eng.sched = sc

    sig_eff = np.exp(0.5 * np.repeat(m.lsigma2s.numpy(), np.asarray(m.diag_error_structure, int))) / m._std
    theta = m._theta_rows(sig_eff)
    eng.evaluate(theta)
    for r in range(a.reps):
        for name, pairs in names:
            apply(pairs)
            eng.evaluate(theta)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(a.steps):
                eng.enqueue()
            torch.cuda.synchronize()
            res[name].append((time.perf_counter() - t0) / a.steps * 1e3)
    apply(())
    if a.stages:
        sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
        import bench
        m._engine = eng
        st, clk = bench.stage_times(m, reps=5)
        print('   stages (ms): ' + '  '.join('%s %.3f' % (k, v) for k, v in st.items()) + '   clock %s MHz' % (clk['clock_mhz'] and round(clk['clock_mhz'])))
    for name, _ in names:
        v = np.array(res[name])
        print(f"{name:28s} min {v.min():8.3f} ms   mean {v.mean():8.3f} ms   ({a.reps} x {a.steps} evaluations, q_local={a.q}, n={a.n})")


if __name__ == '__main__':
    main()
