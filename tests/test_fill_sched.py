"""CPU replay of the launch plan of the factorisation + progressive inverse (lcgp_amd/csrc/fill_sched.h).

The planner is host-only C++ shared with the HIP library; `tests/native/dump_plan.cpp` prints its launch list and this
test executes that list on numpy matrices with the semantics of the kernels (a small block stands for a 64x64 block):
every launch reads a SNAPSHOT of the state before it (nothing inside a launch may depend on anything else inside it)
and the replay checks, launch by launch, that no block has two writers and that no work item reads a block another
item of the same launch writes.  Buffers start as NaN, so a read of something not yet produced poisons the result.
At the end L, L^-1 and A^-1 must equal numpy's (the reference computes these through tf.linalg.eigh / cholesky,
lcgp.py:652 / 617 / 785)."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TS = 2          # emulated block size (stands for 64)


@pytest.fixture(scope='module')
def dump_plan(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp('plan') / 'dump_plan')
    subprocess.run(['g++', '-std=c++17', '-O1', '-I', os.path.join(ROOT, 'lcgp_amd', 'csrc'), '-o', exe,
                    os.path.join(ROOT, 'tests', 'native', 'dump_plan.cpp')], check=True)

    def run(nb, q, ob, syrk_small=3000, fill_leaf=248, fill_step=248, leaf_in_wide=2048, progressive=1, far_rides=1, with_dupd=1,
            pair_tiles=0):
        out = subprocess.run([exe] + [str(v) for v in (nb, q, ob, syrk_small, fill_leaf, fill_step, leaf_in_wide, progressive,
                                                       far_rides, with_dupd, pair_tiles)],
                             check=True, capture_output=True, text=True).stdout
        assert 'FAILED' not in out
        launches = []
        for line in out.splitlines():
            f = line.split()
            d = {k: int(v) for k, _, v in (kv.partition('=') for kv in f[1:])}
            if f[0] == 'L':
                d['jobs'] = []
                launches.append(d)
            else:
                assert f[0] == 'J'
                d['t0'] = d.pop('jt0')
                launches[-1]['jobs'].append(d)
        return launches
    return run


class Replay:
    def __init__(self, nb, q, seed=0):
        self.nb, self.q = nb, q
        n = nb * TS
        rng = np.random.default_rng(seed)
        x = rng.standard_normal((n, n))
        self.A = x @ x.T / n + 2.0 * np.eye(n)
        self.M = np.tril(self.A) + np.triu(np.full((n, n), np.nan), 1)
        for b in range(nb):                      # diagonal tiles are written symmetric by the kernel build
            s = slice(b * TS, (b + 1) * TS)
            self.M[s, s] = self.A[s, s]
        self.W = np.full((n, n), np.nan)
        self.V = np.full((n, n), np.nan)
        self.St = np.zeros((TS, TS))        # stands for the running log-determinant / status words of the component
        self.log = []

    # ---- block access with read / write logging (per work item) ----
    def begin_launch(self):
        self.S = dict(M=self.M.copy(), W=self.W.copy(), V=self.V.copy(), St=self.St.copy())
        self.writes = {}          # (buf, r, c) -> item
        self.reads = []           # (item, buf, r, c)
        self.item = 0

    def new_item(self):
        self.item += 1

    def rd(self, buf, r0, r1, c0, c1):
        """blocks [r0, r1) x [c0, c1) of the snapshot"""
        for r in range(r0, r1):
            for c in range(c0, c1):
                self.reads.append((self.item, buf, r, c))
        return self.S[buf][r0 * TS:r1 * TS, c0 * TS:c1 * TS]

    def wr(self, buf, r0, c0, val):
        live = getattr(self, buf)
        nr, nc = val.shape[0] // TS, val.shape[1] // TS
        for r in range(r0, r0 + nr):
            for c in range(c0, c0 + nc):
                key = (buf, r, c)
                assert key not in self.writes or self.writes[key] == self.item, 'two writers of %s in one launch' % (key,)
                self.writes[key] = self.item
        live[r0 * TS:(r0 + nr) * TS, c0 * TS:(c0 + nc) * TS] = val

    def end_launch(self):
        for item, buf, r, c in self.reads:
            w = self.writes.get((buf, r, c))
            assert w is None or w == item, 'block %s read by item %d and written by item %d in the same launch' % ((buf, r, c), item, w)

    # ---- kernels ----
    def leaf(self, blk_val, j):
        L = np.linalg.cholesky(blk_val)
        self.wr('St', 0, 0, self.rd('St', 0, 1, 0, 1) + 1.0)
        self.wr('M', j, j, L)
        self.wr('W', j, j, np.tril(np.linalg.inv(L)))
        if j % 2 == 0 and j + 1 < self.nb:
            self.wr('W', j, j + 1, np.zeros((TS, TS)))

    def run_leaf(self, l):
        self.new_item()
        self.leaf(self.rd('M', l['J'], l['J'] + 1, l['J'], l['J'] + 1), l['J'])

    def run_step(self, l):
        nb, c, J = self.nb, l['c'], l['J']
        t0, u0 = c + 1, c + 2                   # first block row of the solve tiles / of the delayed-update tiles
        assert l['n_trmm'] == nb - 1 - c
        assert t0 + l['n_trmm'] <= nb and u0 + l['n_upd'] <= nb
        if l['has_special']:
            assert t0 == c + 1
        for r in range(t0, t0 + l['n_trmm']):
            self.new_item()
            tile = self.rd('M', r, r + 1, c, c + 1).copy()
            if c > J:
                tile -= self.rd('M', r, r + 1, c - 1, c) @ self.rd('M', c, c + 1, c - 1, c).T
            Lrc = tile @ self.rd('W', c, c + 1, c, c + 1).T
            self.wr('M', r, c, Lrc)
            if r < l['diag_end']:
                D = self.rd('M', r, r + 1, r, r + 1) - Lrc @ Lrc.T
                if l['has_special'] and r == c + 1:
                    self.leaf(D, r)
                else:
                    self.wr('M', r, r, D)
        if l['n_upd']:
            assert l['n_upd'] == nb - (c + 1) - 1
            for r in range(u0, u0 + l['n_upd']):
                self.new_item()
                t = self.rd('M', r, r + 1, c + 1, c + 2).copy()
                for j in range(J, c):
                    t -= self.rd('M', r, r + 1, j, j + 1) @ self.rd('M', c + 1, c + 2, j, j + 1).T
                self.wr('M', r, c + 1, t)

    def run_trail(self, l):
        J, pe = l['J'], l['pe']
        for C in range(l['c_lo'], l['c_hi']):
            for r in range(C, self.nb):
                self.new_item()
                if l['with_leaf'] and r == C == l['c_lo']:
                    self.leaf(self.rd('M', r, r + 1, C, C + 1), C)
                    continue
                t = self.rd('M', r, r + 1, C, C + 1) - self.rd('M', r, r + 1, J, pe) @ self.rd('M', C, C + 1, J, pe).T
                self.wr('M', r, C, t)

    def run_jobs(self, l):
        q, nb = self.q, self.nb
        for jb in l['jobs']:
            assert jb['nblk'] % q == 0
            ty, kb0, kb1 = jb['type'], jb['kb0'], jb['kb1']
            for t in range(jb['t0'], jb['t0'] + jb['nblk'] // q):
                self.new_item()
                if ty == 1:         # SYRK
                    j, tt = jb['j0'], t
                    while tt >= jb['R1'] - (j >> 1):
                        tt -= jb['R1'] - (j >> 1)
                        j += 1
                    R = (j >> 1) + tt
                    assert j < jb['j1'] and R < nb // 2
                    v = self.rd('M', 2 * R, 2 * R + 2, j, j + 1) - self.rd('M', 2 * R, 2 * R + 2, kb0, kb1) @ self.rd('M', j, j + 1, kb0, kb1).T
                    self.wr('M', 2 * R, j, v)
                elif ty == 2:     # BROW
                    nc = jb['j1'] - jb['j0']
                    R, j = jb['R0'] + t // nc, jb['j0'] + t % nc
                    assert R < jb['R1']
                    ke = min(kb1, 2 * R + 2)
                    self.wr('W', 2 * R, j, -(self.rd('W', 2 * R, 2 * R + 2, kb0, ke) @ self.rd('V', kb0, ke, j, j + 1)))
                elif ty == 3:     # CUPD
                    nc = jb['j1'] - jb['j0']
                    R, j = jb['R0'] + t // nc, jb['j0'] + t % nc
                    assert R < jb['R1']
                    own = j >= kb0
                    ks = kb0 + (((j - kb0) & ~1) if own else 0)
                    v = self.rd('M', 2 * R, 2 * R + 2, ks, kb1) @ self.rd('W', ks, kb1, j, j + 1)
                    if not own:
                        v = v + self.rd('V', 2 * R, 2 * R + 2, j, j + 1)
                    self.wr('V', 2 * R, j, v)
                elif ty == 4:     # DUPD
                    R = int((np.sqrt(4.0 * t + 1.0) - 1.0) * 0.5)
                    while (R + 1) * (R + 2) <= t:
                        R += 1
                    while R * (R + 1) > t:
                        R -= 1
                    j = t - R * (R + 1)
                    assert 2 * R + 1 < nb and j < 2 * R + 2
                    own = 2 * R >= kb0
                    ks = 2 * R if own else kb0
                    v = self.rd('W', ks, kb1, 2 * R, 2 * R + 2).T @ self.rd('W', ks, kb1, j, j + 1)
                    if not own:
                        v = v + self.rd('V', 2 * R, 2 * R + 2, j, j + 1)
                    self.wr('V', 2 * R, j, v)
                else:             # TRI_T (5) / TRI_W (6): tile `t` of one level of the block inverse
                    mb, npair, pair0 = jb['R0'], jb['R1'], jb['j0']
                    pr, rem = pair0 + t % npair, t // npair
                    if ty == 5:
                        cl, rl = rem // mb, rem % mb
                    else:
                        rl, cl = mb - 1 - rem // mb, rem % mb
                    C0 = 2 * pr * mb
                    R0 = C0 + mb
                    if R0 + rl >= nb:
                        continue
                    acc = np.zeros((TS, TS))
                    if ty == 5:
                        for kt in range(cl, mb):
                            acc += self.rd('M', R0 + rl, R0 + rl + 1, C0 + kt, C0 + kt + 1) @ self.rd('W', C0 + kt, C0 + kt + 1, C0 + cl, C0 + cl + 1)
                        self.wr('V', R0 + rl, C0 + cl, acc)
                    else:
                        for kt in range(0, rl + 1):
                            acc += self.rd('W', R0 + rl, R0 + rl + 1, R0 + kt, R0 + kt + 1) @ self.rd('V', R0 + kt, R0 + kt + 1, C0 + cl, C0 + cl + 1)
                        self.wr('W', R0 + rl, C0 + cl, -acc)

    def run(self, launches):
        for l in launches:
            self.begin_launch()
            if l['kind'] == 1:
                self.run_leaf(l)
            elif l['kind'] == 2:
                self.run_step(l)
            elif l['kind'] == 3:
                self.run_trail(l)
            self.run_jobs(l)
            self.end_launch()

    def check(self, inverse, ainv=True):
        L = np.linalg.cholesky(self.A)
        n = L.shape[0]
        low = np.tril(np.ones((n, n), bool))
        assert np.allclose(self.M[low], L[low], rtol=0, atol=1e-10), 'L'
        if inverse:
            Wt = np.linalg.inv(L)
            Ai = np.linalg.inv(self.A)
            assert np.allclose(self.W[low], Wt[low], rtol=0, atol=1e-9), 'L^-1'
            if not ainv:
                return
            assert np.allclose(self.V[low], Ai[low], rtol=0, atol=1e-9), 'A^-1'
            for b in range(self.nb):     # whole diagonal tiles of A^-1 (the symv pass reads them whole)
                s = slice(b * TS, (b + 1) * TS)
                assert np.allclose(self.V[s, s], Ai[s, s], rtol=0, atol=1e-9)


CASES = [
    # nb, q, ob, kwargs
    (64, 1, 4, {}),                               # the headline size, one component per GPU
    (64, 2, 4, {}),
    (64, 8, 4, {}),
    (16, 4, 4, {}),                               # configs[1]
    (32, 6, 4, {}),                               # configs[4]
    (18, 1, 4, {}),                               # a short last panel
    (10, 3, 4, {}),
    (6, 1, 4, {}),
    (4, 1, 4, {}),
    (2, 1, 4, {}),
    (24, 2, 8, {}),                               # float32 panels
    (20, 1, 8, {}),
    (12, 1, 2, {}),
    (64, 1, 4, dict(fill_leaf=0, fill_step=0)),   # nothing rides: everything runs in the tail
    (32, 1, 4, dict(fill_leaf=40, fill_step=24)),
    (32, 2, 4, dict(leaf_in_wide=0)),
    (32, 1, 4, dict(syrk_small=0)),               # wide updates on 128-tiles
    (32, 2, 4, dict(far_rides=0)),                # every trailing update is one wide launch
]


@pytest.mark.parametrize('nb,q,ob,kw', CASES)
def test_plan_with_progressive_inverse_replays_to_the_inverse(dump_plan, nb, q, ob, kw):
    launches = dump_plan(nb, q, ob, progressive=1, **kw)
    r = Replay(nb, q, seed=nb + q)
    r.run(launches)
    r.check(inverse=True)


@pytest.mark.parametrize('nb,q,ob,kw', [(64, 1, 4, {}), (64, 2, 4, dict(far_rides=0)), (18, 1, 4, {}), (24, 2, 8, {}), (6, 1, 4, {}),
                                        (32, 1, 4, dict(fill_leaf=40, fill_step=24))])
def test_plan_with_the_triangular_inverse_only(dump_plan, nb, q, ob, kw):
    """large matrices: only L^-1 is formed behind the chain (A^-1 = W^T W takes its one launch afterwards)"""
    launches = dump_plan(nb, q, ob, progressive=1, with_dupd=0, **kw)
    assert not any(jb['type'] == 4 for l in launches for jb in l['jobs'])
    r = Replay(nb, q, seed=3 * nb + q)
    r.run(launches)
    r.check(inverse=True, ainv=False)


@pytest.mark.parametrize('nb,q,ob,kw', CASES[:8] + [(14, 2, 6, {}), (16, 1, 3, {})])
def test_plan_without_inverse_replays_to_the_factor(dump_plan, nb, q, ob, kw):
    launches = dump_plan(nb, q, ob, progressive=0, **kw)
    assert all(jb['type'] == 1 for l in launches for jb in l['jobs'])       # only trailing-update filler
    r = Replay(nb, q, seed=nb)
    r.run(launches)
    r.check(inverse=False)


@pytest.mark.parametrize('nb,q,ob,kw', [(64, 8, 4, dict(pair_tiles=4000)), (64, 8, 4, dict(pair_tiles=1)), (32, 6, 4, dict(pair_tiles=1)),
                                        (18, 1, 4, dict(pair_tiles=1)), (10, 3, 4, dict(pair_tiles=1)), (24, 2, 8, dict(pair_tiles=1)),
                                        (32, 2, 4, dict(pair_tiles=1, far_rides=0)), (32, 1, 4, dict(pair_tiles=1, fill_leaf=40, fill_step=24)),
                                        (40, 2, 2, dict(pair_tiles=1, leaf_in_wide=0)), (24, 4, 4, dict(pair_tiles=1, leaf_in_wide=0, fill_leaf=8, fill_step=8)),
                                        (24, 4, 4, dict(pair_tiles=1, leaf_in_wide=0, fill_leaf=8, fill_step=8, syrk_small=16)),
                                        (26, 1, 4, dict(pair_tiles=1, leaf_in_wide=0, fill_leaf=0, fill_step=0)), (22, 3, 6, dict(pair_tiles=1, leaf_in_wide=0, fill_leaf=6, fill_step=6))])
def test_plan_with_paired_panels_replays_to_the_factor(dump_plan, nb, q, ob, kw):
    """two consecutive panels share one trailing update with K = 2 ob on the columns between the second panel and the far
    columns (PlanParams::pair_tiles): same factor, every hazard check of the replay"""
    launches = dump_plan(nb, q, ob, progressive=0, **kw)
    wide = [l for l in launches if l['kind'] == 3]
    if nb == 64 or (kw.get('leaf_in_wide') == 0 and ob < 6):
        assert any(l['pe'] - l['J'] == 2 * ob for l in wide), 'no paired update in this plan'
    for l in wide:
        assert l['pe'] - l['J'] in (ob, 2 * ob) or l['pe'] == nb
        assert not (l['pe'] - l['J'] == 2 * ob and l['with_leaf'])      # the diagonal block would lack the first panel
    r = Replay(nb, q, seed=5 * nb + q)
    r.run(launches)
    r.check(inverse=False)


def test_filler_capacity_is_respected(dump_plan):
    for q in (1, 2, 8):
        for l in dump_plan(64, q, 4, progressive=1):
            if l['kind'] in (1, 2):
                assert l['nblk'] <= 248
                assert len(l['jobs']) <= 6

