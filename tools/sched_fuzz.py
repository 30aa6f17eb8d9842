"""Randomised check that every launch schedule computes the same numbers (GPU): random sizes, component counts and lcgp_sched
fields -- paired panels, filler capacities, panel widths, tile-size thresholds, the inverse behind the chain -- against the
plainest schedule (no filler, no pairs, inverse after the factorisation) of the same problem.

    python tools/sched_fuzz.py [cases] [seed] [big]
"""
import sys
import numpy as np

sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from lcgp_amd import LCGP, synth, _hip  # noqa: E402


def sched(**kw):
    sc = _hip.default_sched()
    for k, v in kw.items():
        assert hasattr(sc, k), k
        setattr(sc, k, v)
    return sc


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    big = len(sys.argv) > 3 and sys.argv[3] == 'big'          # sizes around the headline configuration
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    worst = 0.0
    for case in range(cases):
        n = int(rng.integers(2600, 4300)) if big else int(rng.choice([rng.integers(65, 400), rng.integers(400, 1700), rng.integers(1700, 2600)]))
        q = int(rng.integers(1, 9))
        d = int(rng.integers(1, 7))
        p = int(rng.integers(q, q + 6))
        x, y = synth.make_full(1000 + case, n, d, p, q)
        m = LCGP(y=y, x=x, q=q)
        u = synth.param_points(1000 + case, m._get_flat())[1]
        eng = m._get_engine()
        eng.sched = sched(fill_leaf=0, fill_step=0, pair_tiles=0, progressive_tiles=0, leaf_in_wide=0)
        v0, g0 = m.loss_and_grad(u)
        for rep in range(4):
            f = dict(pair_tiles=int(rng.choice([0, 1, 50, 4000])), fill_leaf=int(rng.choice([0, 3, 17, 64, 248])),
                     fill_step=int(rng.choice([0, 5, 24, 248])), leaf_in_wide=int(rng.choice([0, 200, 2048, 100000])),
                     outer_blocks=int(rng.choice([0, 0, 2, 3, 4, 6, 8])), syrk_small_tiles=int(rng.choice([1, 16, 300, 3000])),
                     progressive_tiles=int(rng.choice([0, 0, 600, 1 << 30])), trtri_level_small=int(rng.choice([0, 600, 100000])),
                     lauum_small_tiles=int(rng.choice([0, 2048, 100000])), trtri_small_tiles=int(rng.choice([0, 4200, 100000])))
            eng.sched = sched(**f)
            v, g = m.loss_and_grad(u)
            ev = abs(v - v0) / abs(v0)
            eg = float(np.max(np.abs(g - g0)) / np.max(np.abs(g0)))
            worst = max(worst, ev, eg)
            assert np.isfinite(v) and ev <= 1e-10 and eg <= 1e-9, (n, q, d, p, f, ev, eg)
        print('case %2d  n=%4d q=%d d=%d p=%2d  ok' % (case, n, q, d, p), flush=True)
    print('all schedules agree; worst relative difference %.2e' % worst)


if __name__ == '__main__':
    main()
