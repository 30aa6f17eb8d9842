"""Where the time of the persistent factorisation launch goes, task by task (tool build: `make trace`).

    LCGP_HIP_LIB=lcgp_amd/liblcgp_hip_trace.so python tools/dag_trace.py [--q Q] [--dag 1|2] [--config C] [field=value ...]

The stamped build (-DLCGP_DAG_TRACE) keeps four s_memrealtime stamps (100 MHz) per task at the END of the workspace:
taken from the queue, dependencies satisfied, body done, published.  Prints per segment kind the number of tasks and the
mean / total time waiting, computing and publishing, the busy fraction of the workgroup slots over the launch, and a
coarse timeline (per 100 us: slots computing / waiting).  The plan's segment table is read back from the engine's
plan block, so no second copy of the decoding lives here.
"""
import argparse
import ctypes as C
import sys

import numpy as np
import torch

sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from lcgp_amd import LCGP, synth, _hip  # noqa: E402

CAP = 1 << 18


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--q', type=int, default=8)
    ap.add_argument('--dag', type=int, default=2)
    ap.add_argument('--config', type=int, default=3)
    ap.add_argument('--bucket', type=float, default=100.0)
    ap.add_argument('--chain', type=int, default=0, help='print the first N tasks of the diagonal-block chain of component 0')
    ap.add_argument('fields', nargs='*')
    a = ap.parse_args()
    x, y, cfg = synth.make_config(a.config)
    m = LCGP(y=y, x=x, q=a.q, dtype='float64' if cfg['dtype'] == 'f64' else 'float32')
    eng = m._get_engine()
    sc = _hip.default_sched()
    sc.dag = a.dag
    for kv in a.fields:
        k, v = kv.split('=')
        setattr(sc, k, int(v))
    eng.sched = sc
    sig_eff = np.exp(0.5 * np.repeat(m.lsigma2s.numpy(), np.asarray(m.diag_error_structure, int))) / m._std
    theta = m._theta_rows(sig_eff)
    for _ in range(3):
        eng.evaluate(theta)
    torch.cuda.synchronize()
    info = eng.plan_info()
    nt = info['tasks']
    assert 0 < nt <= CAP, info
    tr = eng.workspace[-CAP * 32:].view(torch.int64).cpu().numpy().reshape(CAP, 4)[:nt].astype(np.uint64)
    t_take, t_ready, t_body = (tr[:, i].astype(np.float64) / 100.0 for i in range(3))     # us
    t_pub = (tr[:, 3] & np.uint64((1 << 44) - 1)).astype(np.float64) / 100.0
    seg = ((tr[:, 3] >> np.uint64(44)) & np.uint64(0xffff)).astype(int)
    xcc = (tr[:, 3] >> np.uint64(60)).astype(int)
    t0 = t_take.min()
    # the low 44 bits of the last stamp wrapped relative to the others: rebuild from the body stamp
    hi = np.floor(t_body * 100.0 / float(1 << 44))
    t_pub = t_pub + hi * float(1 << 44) / 100.0
    t_pub = np.where(t_pub < t_body, t_pub + float(1 << 44) / 100.0, t_pub)
    span = t_pub.max() - t0
    # segment kinds from the plan block (header + DagSeg table): kind is the first int of a DagSeg
    host = eng._plan_cache[True][0]
    hdr = np.frombuffer(host[:256].tobytes(), dtype=np.uint8)
    # PlanHeader: magic u32, version, dtype, n, nb, q, with_inverse, nlaunch, nseg, ntasks, inverse_done, num_cu (12 ints), sched, offsets
    ints = np.frombuffer(host[:48].tobytes(), dtype=np.int32)
    nseg = int(ints[8])
    sched_ints = C.sizeof(_hip.Sched) // 4
    off = 48 + 4 * sched_ints + 4
    off = (off + 7) & ~7
    sim_us = float(np.frombuffer(host[off:off + 8].tobytes(), dtype=np.float64)[0])
    off += 8
    off_launch, off_seg, off_run, nbytes = (int(v) for v in np.frombuffer(host[off:off + 32].tobytes(), dtype=np.uint64))
    print('list-schedule estimate of the launch: %.0f us' % sim_us)
    # sizeof(DagSeg): derive from the table span (padded to 256): use the known layout instead
    import struct
    seg_ints = 6 + 16 + 16 + 7 + 4 + 2 + 4 + 4 + 10    # kind,t0,ntasks,per_comp,k_off,ndeps | dep | need | J.. | c_lo.. | t_first,t_count | r_lo,r_hi,trmm_r0,upd_r0 | tri_* | job (FillJob: 10 ints)
    kinds = []
    for i in range(nseg):
        b = host[off_seg + i * seg_ints * 4: off_seg + (i + 1) * seg_ints * 4].tobytes()
        v = struct.unpack('%di' % seg_ints, b)
        assert 1 <= v[0] <= 6, v[:8]
        kinds.append((v[0], v[38 + 7], v[38 + 8], v[38 + 9], v[38 + 4], v[53]))     # kind, c_lo, c_hi, tiles128, has_special, trmm_r0
    names = {1: 'leaf', 2: 'step', 3: 'trail', 4: 'fill', 5: 'tri', 6: 'psolve'}
    cs = []
    for i in range(nseg):
        b = host[off_seg + i * seg_ints * 4: off_seg + (i + 1) * seg_ints * 4].tobytes()
        cs.append(struct.unpack('%di' % seg_ints, b)[38 + 2])

    def seg_c(si):
        return cs[si]
    print('tasks %d, segments %d, span %.1f us, workgroup slots %d' % (nt, nseg, span, len(np.unique(np.round(t_take, 2)))))
    rows = {}
    for i in range(nt):
        kd = kinds[seg[i]]
        key = names[kd[0]] + ('128' if kd[0] in (3, 5) and kd[3] else '') + ('-chain' if kd[0] == 2 and kd[5] <= seg_c(seg[i]) + 1 else '')
        r = rows.setdefault(key, [0, 0.0, 0.0, 0.0])
        r[0] += 1
        r[1] += t_ready[i] - t_take[i]
        r[2] += t_body[i] - t_ready[i]
        r[3] += t_pub[i] - t_body[i]
    print('%-10s %8s %12s %12s %12s   (mean us per task; total slot-ms)' % ('kind', 'tasks', 'wait', 'body', 'publish'))
    tot = [0.0, 0.0, 0.0]
    for key, r in sorted(rows.items()):
        print('%-10s %8d %6.2f %6.1f %6.2f %6.1f %6.2f %6.1f' % (key, r[0], r[1] / r[0], r[1] / 1e3, r[2] / r[0], r[2] / 1e3,
                                                                  r[3] / r[0], r[3] / 1e3))
        for j in range(3):
            tot[j] += r[1 + j]
    slots = 512
    print('slot-time: wait %.1f %%, body %.1f %%, publish+queue %.1f %%, idle/other %.1f %% of %d slots x %.1f us' % (
        100 * tot[0] / (slots * span), 100 * tot[1] / (slots * span), 100 * tot[2] / (slots * span),
        100 * (1 - sum(tot) / (slots * span)), slots, span))
    if a.chain:
        # the chain of component 0: leaf tasks and the special task (task 0 of a step segment that has one)
        rows = []
        for i in range(nt):
            kd = kinds[seg[i]]
            if kd[0] == 1 or (kd[0] == 2 and kd[4]):
                rows.append((t_take[i], i))
        rows.sort()
        seen = set()
        prev_pub = None
        print('chain of component 0: segment kind | taken | waited | body | publish | gap since the previous chain task was published')
        n = 0
        for _, i in rows:
            if seg[i] in seen:
                continue                       # (first task of the segment in sequence order = component 0)
            first = min(j for j in range(max(0, i - 16), min(nt, i + 16)) if seg[j] == seg[i])
            if first != i:
                continue
            seen.add(seg[i])
            gap = (t_ready[i] - prev_pub) if prev_pub is not None else 0.0
            print('  %-5s taken %9.1f waited %6.1f body %6.1f publish %5.1f   gap %6.1f' % (
                names[kinds[seg[i]][0]], t_take[i] - t0, t_ready[i] - t_take[i], t_body[i] - t_ready[i], t_pub[i] - t_body[i], gap))
            prev_pub = t_pub[i]
            n += 1
            if n >= a.chain:
                break
    nbk = int(span / a.bucket) + 1
    comp = np.zeros(nbk)
    wait = np.zeros(nbk)
    for i in range(nt):
        for arr, lo, hi_ in ((wait, t_take[i], t_ready[i]), (comp, t_ready[i], t_body[i])):
            lo -= t0
            hi_ -= t0
            b0, b1 = int(lo / a.bucket), int(hi_ / a.bucket)
            for b in range(b0, min(b1, nbk - 1) + 1):
                arr[b] += max(0.0, min(hi_, (b + 1) * a.bucket) - max(lo, b * a.bucket))
    print('timeline (bucket %.0f us): mean slots computing / waiting' % a.bucket)
    for b in range(nbk):
        print('%8.0f  %6.1f  %6.1f' % (b * a.bucket, comp[b] / a.bucket, wait[b] / a.bucket))


if __name__ == '__main__':
    main()
