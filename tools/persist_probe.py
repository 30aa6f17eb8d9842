"""Feasibility probe for a resident chain kernel on a second stream (tools/persist_probe.hip).

    python tools/persist_probe.py  [--q 8] [--evals 4]

Prints: ms per evaluation of the main stream alone / beside the resident kernel (by LDS size of the resident workgroups), the
resident kernel's time per 64^3 product round and per dependent global round trip alone / beside the evaluation, and the cost
of a flag hand-off between one-thread kernels of the main stream and the resident kernel."""
import argparse
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from lcgp_amd import LCGP, synth  # noqa: E402


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
    ap.add_argument('--q', type=int, default=8)
    ap.add_argument('--n', type=int, default=4096)
    ap.add_argument('--evals', type=int, default=4)
    a = ap.parse_args()
    so = os.path.join(ROOT, 'tools', 'libpersist_probe.so')
    if not os.path.exists(so):
        subprocess.run(['hipcc', '-O3', '--offload-arch=gfx950', '-fPIC', '-shared', '-o', so,
                        os.path.join(ROOT, 'tools', 'persist_probe.hip')], check=True)
    lib = C.CDLL(so)
    for f in ('probe_resident', 'probe_pong', 'probe_signal', 'probe_wait'):
        getattr(lib, f).restype = C.c_int
    m, eng = model(a.n, a.q)
    dev = eng.device
    main_s = torch.cuda.current_stream(dev)
    aux = torch.cuda.Stream(device=dev)
    p = lambda t: C.c_void_p(t.data_ptr())
    chase = torch.zeros(1024 * 16, dtype=torch.int32, device=dev)
    sink = torch.zeros(4, dtype=torch.float64, device=dev)

    def evals(k):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(main_s)
        for _ in range(k):
            eng.enqueue()
        e1.record(main_s)
        return e0, e1

    def resident(nwg, lds, iters, chase_every):
        st = torch.zeros(nwg * 6, dtype=torch.int64, device=dev)
        rc = lib.probe_resident(C.c_void_p(aux.cuda_stream), nwg, lds, iters, chase_every, p(chase), p(st), p(sink))
        assert rc == 0, rc
        return st

    def report(tag, st, nwg, iters, chase_every):
        s = st.cpu().numpy().reshape(nwg, 6)
        wall = (s[:, 1] - s[:, 0]) / 100.0
        prod = s[:, 2] / 100.0 / iters
        nch = np.maximum(s[:, 4], 1)
        ch = s[:, 3] / 100.0 / nch
        print('%-42s resident wall %8.1f us (max)  product round %.3f us (median, max %.3f)  round trip %.3f us (median, max %.3f)  xcc %s'
              % (tag, wall.max(), np.median(prod), prod.max(), np.median(ch), ch.max(), sorted(set(int(v) & 15 for v in s[:, 5]))))

    torch.cuda.synchronize()
    # ---- main stream alone
    e0, e1 = evals(2)
    torch.cuda.synchronize()
    res = []
    for _ in range(3):
        e0, e1 = evals(a.evals)
        torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / a.evals)
    base = min(res)
    print('main stream alone: %.3f ms per evaluation (min of 3 x %d)' % (base, a.evals))
    # ---- resident kernel alone (calibration)
    nwg = a.q
    for lds in (40 * 1024, 100 * 1024):
        st = resident(nwg, lds, 2000, 8)
        torch.cuda.synchronize()
        report('resident alone, lds %d KB' % (lds // 1024), st, nwg, 2000, 8)
    s = st.cpu().numpy().reshape(nwg, 6)
    per_iter_us = float(np.max(s[:, 1] - s[:, 0])) / 100.0 / 2000
    iters = int(1.15 * a.evals * base * 1000.0 / per_iter_us)
    # ---- both
    for lds in (40 * 1024, 100 * 1024, 100 * 1024):
        for order in ('resident first',):
            torch.cuda.synchronize()
            st = resident(nwg, lds, iters, 8)
            e0, e1 = evals(a.evals)
            torch.cuda.synchronize()
            print('main stream beside the resident kernel (lds %d KB): %.3f ms per evaluation (%+.1f %%)'
                  % (lds // 1024, e0.elapsed_time(e1) / a.evals, 100.0 * (e0.elapsed_time(e1) / a.evals / base - 1.0)))
            report('resident beside the evaluation, lds %d KB' % (lds // 1024), st, nwg, iters, 8)
    # ---- 16 and 32 resident workgroups (a chain team per component)
    for nw in (16, 32):
        torch.cuda.synchronize()
        st = resident(nw, 100 * 1024, iters, 8)
        e0, e1 = evals(a.evals)
        torch.cuda.synchronize()
        print('main stream beside %d resident workgroups (lds 100 KB): %.3f ms per evaluation (%+.1f %%)'
              % (nw, e0.elapsed_time(e1) / a.evals, 100.0 * (e0.elapsed_time(e1) / a.evals / base - 1.0)))
        report('resident x %d beside the evaluation' % nw, st, nw, iters, 8)
    # ---- flag ping-pong
    n = 200
    flags = torch.zeros(64, dtype=torch.int32, device=dev)       # ping at 0, pong at 16, fail at 32
    stamps = torch.zeros(2 * n, dtype=torch.int64, device=dev)
    hs = torch.zeros(2 * n, dtype=torch.int64, device=dev)
    fp = flags.data_ptr()
    for load in (False, True):
        flags.zero_()
        torch.cuda.synchronize()
        rc = lib.probe_pong(C.c_void_p(aux.cuda_stream), 100 * 1024, n, C.c_void_p(fp), C.c_void_p(fp + 64), C.c_void_p(fp + 128), p(stamps))
        assert rc == 0
        ms = C.c_void_p(main_s.cuda_stream)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(main_s)
        for i in range(1, n + 1):
            if load and i % 20 == 0:
                lib_l = eng.lib
                from lcgp_amd import _hip
                _hip.check(lib_l.lcgp_lauum(ms, eng.dtype, eng.n, eng.d, eng.p, eng.q_local, eng._p(eng.workspace), None), 'lauum')
            lib.probe_signal(ms, C.c_void_p(fp), i, C.c_void_p(hs.data_ptr() + 16 * (i - 1)))
            lib.probe_wait(ms, C.c_void_p(fp + 64), i, C.c_void_p(fp + 128), C.c_void_p(hs.data_ptr() + 16 * (i - 1) + 8))
        e1.record(main_s)
        torch.cuda.synchronize()
        st = stamps.cpu().numpy().reshape(n, 2)
        h = hs.cpu().numpy().reshape(n, 2)
        seen = (st[:, 0] - h[:, 0]) / 100.0          # signal stored -> resident saw it
        back = (h[:, 1] - st[:, 1]) / 100.0          # resident stored pong -> wait kernel saw it (incl. its launch)
        print('ping-pong (%s): %.2f us per round trip on the main stream (2 one-thread kernels + resident answer), fail word %d'
              % ('with a wide launch every 20' if load else 'idle chip', e0.elapsed_time(e1) * 1000.0 / n, int(flags[32])))
        print('   signal -> seen by the resident kernel: median %.2f us (p90 %.2f);  pong -> end of the wait kernel: median %.2f us (p90 %.2f)'
              % (np.median(seen), np.percentile(seen, 90), np.median(back), np.percentile(back, 90)))


if __name__ == '__main__':
    main()
