"""How much do a chain-bound factorisation and a wide triangular product disturb each other on two HIP streams?

No dependency between the two streams (two engines, two workspaces): this measures interference only --
engine A: kernel build + factorisation (n_a, q_a) on stream 1; engine B: one stage of the inverse (n_b, q_b) on
stream 2, repeated `--reps-b` times so that it covers A.  Prints each alone and both together.

usage: python tools/stream_overlap.py [--na 2048] [--qa 8] [--nb 4096] [--qb 8] [--stage trtri|lauum] [--prio]
"""
import argparse
import ctypes as C
import sys
import time

import numpy as np
import torch

sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from lcgp_amd import LCGP, synth, _hip  # noqa: E402


def model(n, q):
    x, y, cfg = synth.make_config(3)
    x, y = x[:n], y[:, :n]
    m = LCGP(y=y, x=x, q=q, dtype='float64')
    eng = m._get_engine()
    sig_eff = np.exp(0.5 * np.repeat(m.lsigma2s.numpy(), np.asarray(m.diag_error_structure, int))) / m._std
    theta = m._theta_rows(sig_eff)
    eng.evaluate(theta)
    return m, eng


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--na', type=int, default=2048)
    ap.add_argument('--qa', type=int, default=8)
    ap.add_argument('--nb', type=int, default=4096)
    ap.add_argument('--qb', type=int, default=8)
    ap.add_argument('--stage', default='trtri')
    ap.add_argument('--reps-b', type=int, default=1)
    ap.add_argument('--prio', action='store_true', help='stream 1 (the chain) with high priority')
    ap.add_argument('--iters', type=int, default=10)
    ap.add_argument('--sched', default='', help='lcgp_sched fields of A: field=val,field=val')
    ap.add_argument('--a-first', action='store_true', help='enqueue A before B')
    a = ap.parse_args()
    ma, ea = model(a.na, a.qa)
    mb, eb = model(a.nb, a.qb)
    lib = ea.lib
    if a.sched:
        sc = _hip.default_sched()
        for kv in a.sched.split(','):
            k, v = kv.split('=')
            assert hasattr(sc, k), k
            setattr(sc, k, int(v))
        ea.sched = sc
    dev = ea.device
    s1 = torch.cuda.Stream(device=dev, priority=-1 if a.prio else 0)
    s2 = torch.cuda.Stream(device=dev, priority=0)
    ld = torch.zeros(a.qa, dtype=torch.float64, device=dev)
    info = torch.zeros(a.qa, dtype=torch.int32, device=dev)

    def run_a():
        st = C.c_void_p(s1.cuda_stream)
        _hip.check(lib.lcgp_kernel_build(st, ea.dtype, ea.kernel_id, ea.n, ea.d, ea.p, ea.q_local, ea._p(ea.x), ea._p(ea.sr),
                                         ea._p(ea.theta_dev), ea._p(ea.workspace)), 'build')
        _hip.check(lib.lcgp_potrf_logdet(st, ea.dtype, ea.n, ea.d, ea.p, ea.q_local, ea._p(ea.workspace),
                                         C.c_void_p(ld.data_ptr()), C.c_void_p(info.data_ptr()), ea._sched(), ea.plan(False)), 'potrf')

    fn_b = getattr(lib, 'lcgp_' + a.stage)

    def run_b():
        st = C.c_void_p(s2.cuda_stream)
        for _ in range(a.reps_b):
            _hip.check(fn_b(st, eb.dtype, eb.n, eb.d, eb.p, eb.q_local, eb._p(eb.workspace), None), a.stage)

    def timed(fa, fb):
        best = []
        for _ in range(a.iters):
            torch.cuda.synchronize()
            e0a, e1a = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0b, e1b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0 = time.perf_counter()
            if fa and a.a_first:
                e0a.record(s1)
                fa()
                e1a.record(s1)
            if fb:
                e0b.record(s2)
                fb()
                e1b.record(s2)
            if fa and not a.a_first:
                e0a.record(s1)
                fa()
                e1a.record(s1)
            torch.cuda.synchronize()
            wall = (time.perf_counter() - t0) * 1e3
            best.append((wall, e0a.elapsed_time(e1a) if fa else 0.0, e0b.elapsed_time(e1b) if fb else 0.0))
        v = np.array(best)
        return v.min(axis=0), np.median(v, axis=0)

    for _ in range(2):
        run_a(); run_b()
    torch.cuda.synchronize()
    for name, fa, fb in (('A alone (build + potrf)', run_a, None), ('B alone (%s x %d)' % (a.stage, a.reps_b), None, run_b),
                         ('A and B together', run_a, run_b)):
        mn, md = timed(fa, fb)
        print(f"{name:28s} wall min {mn[0]:7.3f} med {md[0]:7.3f} ms | A events min {mn[1]:7.3f} med {md[1]:7.3f} | "
              f"B events min {mn[2]:7.3f} med {md[2]:7.3f}   (A: n={a.na} q={a.qa}; B: n={a.nb} q={a.qb})")
    print('info', info.cpu().numpy())


if __name__ == '__main__':
    main()
