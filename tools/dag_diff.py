"""Which blocks of L, L^-1, A^-1 differ between the launch-by-launch executor and the persistent launch?
    python tools/dag_diff.py n q reps [dag] [flags]"""
import sys
import numpy as np
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from lcgp_amd import LCGP, synth, _hip  # noqa: E402

n, q, reps = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
dag = int(sys.argv[4]) if len(sys.argv) > 4 else 1
flags = int(sys.argv[5]) if len(sys.argv) > 5 else 0
extra = dict((kv.split('=')[0], int(kv.split('=')[1])) for kv in sys.argv[6:])
x, y = synth.make_full(77, n, 2, max(q, 4), q)
m = LCGP(y=y, x=x, q=q)
eng = m._get_engine()
sig_eff = np.exp(0.5 * np.repeat(m.lsigma2s.numpy(), np.asarray(m.diag_error_structure, int))) / m._std
theta = m._theta_rows(sig_eff)


def sched(**kw):
    sc = _hip.default_sched()
    for k, v in kw.items():
        setattr(sc, k, v)
    return sc


eng.sched = sched(dag=0, **extra)
eng.evaluate(theta)
ref = [[np.tril(eng.fetch_matrix(w, k)) for w in range(3)] for k in range(q)]
eng.sched = sched(dag=dag, dag_flags=flags, **extra)
print(eng.plan_info())
nb = (n + 63) // 64
bad = 0
for r in range(reps):
    eng.workspace.fill_(255 if r % 2 == 0 else 0)
    out = eng.evaluate(theta)
    line = []
    for k in range(q):
        for w, name in enumerate('LWV'):
            a = np.tril(eng.fetch_matrix(w, k))
            with np.errstate(invalid='ignore'):
                d = a != ref[k][w]
            d |= np.isnan(a) != np.isnan(ref[k][w])
            if d.any():
                blocks = sorted({(int(i) // 64, int(j) // 64) for i, j in zip(*np.nonzero(d))})
                rel = np.nanmax(np.abs(a - ref[k][w])) / np.nanmax(np.abs(ref[k][w]))
                nn = int(np.isnan(a).sum())
                line.append('k%d%s:%d blk rel %.1e nan %d first %s' % (k, name, len(blocks), rel, nn, blocks[:3]))
    real = [x for x in line if 'rel' in x and (float(x.split('rel ')[1].split()[0]) > 1e-9 or ' nan 0 ' not in x)]
    if real:
        bad += 1
        print('rep %d: %s' % (r, ' | '.join(real[:6])))
print('reps with real differences: %d of %d' % (bad, reps))
print('done', sys.argv[1:])
