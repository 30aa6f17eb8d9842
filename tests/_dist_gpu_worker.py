"""Worker of tests/test_gpu_dist.py: one rank of a 2-rank job in which BOTH ranks drive the same GPU through the real
HIP engine (component k -> rank k mod 2), with gloo as the process-group backend (RCCL needs one GPU per rank; the
collective then runs on host memory, everything else -- sharding, the device-side partial vector, lock-step L-BFGS-B,
the predict gather, a rank without components -- is the production path)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from lcgp_amd import LCGP, synth  # noqa: E402
from oracle import lcgp_oracle as orc  # noqa: E402


def main():
    torch.cuda.set_device(0)
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    assert world == 2
    for mode, q, maker in (("full", 3, lambda: synth.make_full(31, 300, 2, 4, 3)),
                           ("rep", 4, lambda: synth.make_rep(32, 70, 3, 2, 4, 4))):
        x, y = maker()
        m = LCGP(y=y, x=x, q=q, submethod=mode, device="cuda:0")
        o = orc.OracleLCGP(y=y, x=x, q=q, submethod=mode)
        o.phi = m.phi.numpy().copy()                       # rank 0's basis was broadcast
        for u in synth.param_points(31, o.get_unconstrained()):
            v1, g1 = m.loss_and_grad(u)
            v2, g2 = o.loss_and_grad_unconstrained(u)
            assert len(m._local_ks) == len(range(rank, q, world))
            assert m._engine.q_local == len(m._local_ks) and m._engine.q_total == q
            assert abs(v1 - v2) <= 1e-6 * abs(v2), (rank, v1, v2)
            assert np.max(np.abs(g1 - g2)) <= 1e-5 * np.max(np.abs(g2))
        m.fit()
        flat = m._get_flat()
        both = [None, None]
        dist.all_gather_object(both, flat.tobytes())
        assert both[0] == both[1]                          # lock-step without any broadcast of the iterate
        o.set_unconstrained(flat)
        x0 = np.random.default_rng(5).uniform(0, 1, (9, 2))
        for a, b in zip(m.predict(x0), o.predict(x0)):
            np.testing.assert_allclose(a.numpy(), b, rtol=1e-6, atol=1e-8)
        assert np.all(np.isfinite(m.CinvMs.numpy()))
    # lock-step guard through the real engines: a one-ulp difference of ONE entry on ONE rank raises on both
    u = m._get_flat().copy()
    ub = u.copy()
    if rank == 0:
        ub[1] = np.nextafter(ub[1], -np.inf)
    try:
        m.loss_and_grad(ub)
        raise AssertionError('rank %d did not notice the drift' % rank)
    except RuntimeError as e:
        assert 'lock-step' in str(e)
    m.loss_and_grad(u)
    # the headline configuration as a 2-GPU job would run it (q_local = 4 per rank: that share's tile-size thresholds)
    # against the committed golden: NLL and all 128 gradient entries at one point
    gold = np.load(os.path.join(ROOT, 'tests', 'golden', 'lcgp_golden_large.npz'))
    x, y, cfg = synth.make_config(3)
    m = LCGP(y=y, x=x, q=cfg['q'], device="cuda:0")
    v, g = m.loss_and_grad(gold['cfg3_n4096/u'][1])
    assert m._engine.q_local == 4 and m._local_ks == list(range(rank, 8, 2))
    want_v, want_g = gold['cfg3_n4096/nll'][1], gold['cfg3_n4096/grad'][1]
    assert abs(v - want_v) <= 1e-6 * abs(want_v), (rank, v, want_v)
    assert np.max(np.abs(g - want_g)) <= 1e-5 * np.max(np.abs(want_g))
    del m
    torch.cuda.empty_cache()
    # q < world: rank 1 holds no component, has no engine, and still takes part in every collective
    x, y = synth.make_full(33, 200, 2, 3, 1)
    m = LCGP(y=y, x=x, q=1, device="cuda:0")
    o = orc.OracleLCGP(y=y, x=x, q=1)
    o.phi = m.phi.numpy().copy()
    v1, g1 = m.loss_and_grad(o.get_unconstrained())
    v2, g2 = o.loss_and_grad_unconstrained(o.get_unconstrained())
    assert (m._engine is None) == (rank == 1)
    assert abs(v1 - v2) <= 1e-6 * abs(v2) and np.max(np.abs(g1 - g2)) <= 1e-5 * np.max(np.abs(g2))
    np.testing.assert_allclose(m.predict(x[:5])[0].numpy(), o.predict(x[:5])[0], rtol=1e-6, atol=1e-8)
    assert m.CinvMs.shape == (1, 200)
    # ... and the guard works when the drifting rank is the one WITHOUT an engine
    u = m._get_flat().copy()
    ub = u.copy()
    if rank == 1:
        ub[0] = np.nextafter(ub[0], np.inf)
    try:
        m.loss_and_grad(ub)
        raise AssertionError('rank %d did not notice the drift' % rank)
    except RuntimeError as e:
        assert 'lock-step' in str(e)
    dist.barrier()
    dist.destroy_process_group()
    print("RANK %d OK" % rank)


if __name__ == "__main__":
    main()
