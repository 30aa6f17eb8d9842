"""Worker of tests/test_dist_gloo.py: one rank of a world_size-2 gloo job (CPU).  The hot path itself is
answered by the test-only OracleEngine; what is under test is the component sharding (k -> rank k mod G), the
single all-reduce of the (P+1)-vector, the lock-step L-BFGS-B and the gather in predict()."""
import os
import sys

import numpy as np
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from lcgp_amd import LCGP, synth  # noqa: E402
from lcgp_amd import dist as ldist  # noqa: E402
from oracle import lcgp_oracle as orc  # noqa: E402
from tests.helpers import patch_engine  # noqa: E402


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    assert world == 2
    assert ldist.local_components(5, rank, world) == ([0, 2, 4] if rank == 0 else [1, 3])
    for mode, q in (("full", 3), ("rep", 4)):
        if mode == "full":
            x, y = synth.make_full(21, 40, 2, 4, 3)
        else:
            x, y = synth.make_rep(22, 15, 3, 2, 4, 4)
        m = patch_engine(LCGP(y=y, x=x, q=q, submethod=mode))
        o = orc.OracleLCGP(y=y, x=x, q=q, submethod=mode)
        o.phi = m.phi.numpy().copy()
        for u in synth.param_points(21, o.get_unconstrained()):
            v1, g1 = m.loss_and_grad(u)
            v2, g2 = o.loss_and_grad_unconstrained(u)
            assert len(m._local_ks) == len(ldist.local_components(q, rank, world))
            assert abs(v1 - v2) <= 1e-12 * max(1.0, abs(v2)), (rank, v1, v2)
            np.testing.assert_allclose(g1, g2, rtol=1e-10, atol=1e-12 * np.max(np.abs(g2)))
        m.fit()
        # lock-step: both ranks must hold bit-identical parameters without any broadcast
        flat = m._get_flat()
        both = [None, None]
        dist.all_gather_object(both, flat.tobytes())
        assert both[0] == both[1]
        o.set_unconstrained(flat)
        x0 = np.random.default_rng(5).uniform(0, 1, (7, 2))
        got = m.predict(x0)
        want = o.predict(x0)
        for a, b in zip(got, want):
            np.testing.assert_allclose(a.numpy(), b, rtol=1e-7, atol=1e-9)
        # the cache views (lcgp.py:709-715 / 783-788) with the components spread over the ranks (3 on 2 ranks: 2 + 1; 4 on 2:
        # 2 + 2): ONE all_gather of the rows each rank owns -- for Ths every rank takes the matrix square root of its OWN
        # components only -- and every rank ends with the identical, complete (q, ...) array
        views = dict(CinvMs=m.CinvMs.numpy(), mks=m.mks.numpy())
        if mode == 'full':
            views['Ths'] = m.Ths.numpy()
            assert views['Ths'].shape == (q, int(m.n), int(m.n))
            # Th_k Th_k = D_k A_k^-1 (symmetric square root): check against the oracle's A_k^-1 through CinvM = A^-1 B
            aux = o._aux_full()
            np.testing.assert_allclose(views['Ths'], np.asarray(aux['Ths']), rtol=1e-6, atol=1e-9)
        else:
            views['Tks'] = m.Tks.numpy()
            assert views['Tks'].shape == (q, int(m.n), int(m.n))
            aux = o._aux_rep()
            # (the oracle follows the reference literally here: explicit inverses of C_k and of C_k^-1 + D_k R, lcgp.py:783-788,
            # which lose digits at the fitted nugget of 1e-7; the path under test forms D R^1/2 A^-1 R^1/2 from the well
            # conditioned A -- what this job checks is the GATHER: complete, finite, identical on both ranks)
            want = np.asarray(aux['Tks'])
            np.testing.assert_allclose(views['Tks'], want, rtol=2e-2, atol=2e-3 * np.max(np.abs(want)))
            np.testing.assert_allclose(views['mks'], np.asarray(aux['mks']), rtol=1e-6, atol=1e-9)
        np.testing.assert_allclose(views['CinvMs'], np.asarray(aux['CinvMs']), rtol=1e-6, atol=1e-9)
        for name, v in views.items():
            assert np.all(np.isfinite(v)), name
            both = [None, None]
            dist.all_gather_object(both, v.tobytes())
            assert both[0] == both[1], name
    # lock-step guard: one rank evaluates a parameter vector that differs by ONE ULP in one entry -- every rank must
    # raise at that evaluation (after the same collective), not hang in a later one; both can carry on afterwards
    x, y = synth.make_full(24, 30, 2, 3, 2)
    m = patch_engine(LCGP(y=y, x=x, q=2))
    u = m._get_flat().copy()
    m.loss_and_grad(u)
    ub = u.copy()
    if rank == 1:
        ub[3] = np.nextafter(ub[3], np.inf)
    try:
        m.loss_and_grad(ub)
        raise AssertionError('rank %d did not notice the drift' % rank)
    except RuntimeError as e:
        assert 'lock-step' in str(e)
    v_again, _ = m.loss_and_grad(u)          # still usable, still in step
    both = [None, None]
    dist.all_gather_object(both, float(v_again))
    assert both[0] == both[1]
    # q < world: rank 1 holds no component and still takes part in the collectives
    x, y = synth.make_full(23, 30, 2, 3, 1)
    m = patch_engine(LCGP(y=y, x=x, q=1))
    o = orc.OracleLCGP(y=y, x=x, q=1)
    o.phi = m.phi.numpy().copy()
    v1, g1 = m.loss_and_grad(o.get_unconstrained())
    v2, g2 = o.loss_and_grad_unconstrained(o.get_unconstrained())
    assert abs(v1 - v2) <= 1e-12 * abs(v2)
    np.testing.assert_allclose(g1, g2, rtol=1e-10, atol=1e-13)
    np.testing.assert_allclose(m.predict(x[:5])[0].numpy(), o.predict(x[:5])[0], rtol=1e-7, atol=1e-9)
    # the cache views are gathers too: the rank WITHOUT a component (and without an engine) must enter them as well,
    # or the other rank would wait in the collective forever
    cinv = m.CinvMs.numpy()
    ths = m.Ths.numpy()
    assert cinv.shape == (1, 30) and ths.shape == (1, 30, 30) and np.all(np.isfinite(cinv)) and np.all(np.isfinite(ths))
    both = [None, None]
    dist.all_gather_object(both, cinv.tobytes())
    assert both[0] == both[1]
    dist.barrier()
    dist.destroy_process_group()
    print("RANK %d OK" % rank)


if __name__ == "__main__":
    main()
