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

    def run(nb, q, ob, syrk_small=2000, fill_leaf=248, fill_step=248, leaf_in_wide=1024, progressive=1, far_rides=1, with_dupd=1,
            dag=0, interleaved=0, with_trtri=0, trtri_all_small=0, fill_wide=0):
        out = subprocess.run([exe] + [str(v) for v in (nb, q, ob, syrk_small, fill_leaf, fill_step, leaf_in_wide, progressive,
                                                       far_rides, with_dupd, dag, interleaved, with_trtri, trtri_all_small, fill_wide)],
                             check=True, capture_output=True, text=True).stdout
        assert 'FAILED' not in out
        launches = []
        segs = []
        runs = []
        for line in out.splitlines():
            f = line.split()
            d = {}
            deps = []
            for kv in f[1:]:
                k, _, v = kv.partition('=')
                if k == 'deps':
                    deps = [tuple(int(y) for y in x.split(':')) for x in v.split(',') if x]
                else:
                    d[k] = int(v)
            if f[0] == 'R':
                runs.append(d)
            elif f[0] == 'M':
                continue
            elif f[0] == 'L':
                d['jobs'] = []
                launches.append(d)
            elif f[0] == 'J':
                d['t0'] = d.pop('jt0')
                launches[-1]['jobs'].append(d)
            else:
                assert len(deps) == d['ndeps']
                d['deps'] = deps
                d['jobs'] = [dict(type=d['type'], nblk=d['nblk'], t0=d['jt0'], R0=d['R0'], R1=d['R1'], j0=d['j0'], j1=d['j1'],
                                  kb0=d['kb0'], kb1=d['kb1'], wide=d.get('wide', 0))] if d['kind'] == 4 else []
                segs.append(d)
        run.last_runs = runs
        return (launches, segs) if dag else launches
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
        t0 = l.get('trmm_r0') or c + 1          # first block row of the solve tiles / of the delayed-update tiles
        u0 = l.get('upd_r0') or c + 2
        if not l.get('trmm_r0'):
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
            if not l.get('upd_r0'):
                assert l['n_upd'] == nb - (c + 1) - 1
            for r in range(u0, u0 + l['n_upd']):
                self.new_item()
                t = self.rd('M', r, r + 1, c + 1, c + 2).copy()
                for j in range(J, c):
                    t -= self.rd('M', r, r + 1, j, j + 1) @ self.rd('M', c + 1, c + 2, j, j + 1).T
                self.wr('M', r, c + 1, t)

    def run_trail(self, l):
        J, pe = l['J'], l['pe']
        if l.get('t_count', 0) or l.get('r_lo', 0) or l.get('r_hi', 0):
            # tiles [t_first, t_first + t_count) of a region: column-major over the tile columns [c_lo, c_hi), in column C the
            # tile rows [max(C, r_lo), r_hi) (gemm_body, OP_SYRK); on 128x128 tiles a tile is 2 x 2 blocks (3 on the diagonal)
            u = 2 if l['tiles128'] else 1
            nbt = self.nb // u
            rlo, rhi = l.get('r_lo', 0) // u, (l.get('r_hi', 0) // u) or nbt
            tiles = []
            for C in range(l['c_lo'] // u, l['c_hi'] // u):
                for r in range(max(C, rlo), rhi):
                    tiles.append((r, C))
            cnt = l.get('t_count', 0) or len(tiles)
            first = l.get('t_first', 0)
            assert first + cnt <= len(tiles)
            for r, C in tiles[first:first + cnt]:
                self.new_item()
                for cc in range(C * u, (C + 1) * u):
                    for rr in range(max(r * u, cc), (r + 1) * u):
                        t = self.rd('M', rr, rr + 1, cc, cc + 1) - self.rd('M', rr, rr + 1, J, pe) @ self.rd('M', cc, cc + 1, J, pe).T
                        self.wr('M', rr, cc, t)
            return
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
                if ty == 1 and jb.get('wide'):       # SYRK on 128 x 128 tiles: column pairs
                    j, tt = jb['j0'], t
                    assert j % 2 == 0 and jb['j1'] % 2 == 0
                    while tt >= jb['R1'] - (j >> 1):
                        tt -= jb['R1'] - (j >> 1)
                        j += 2
                    R = (j >> 1) + tt
                    assert j < jb['j1'] and R < nb // 2
                    v = self.rd('M', 2 * R, 2 * R + 2, j, j + 2) - self.rd('M', 2 * R, 2 * R + 2, kb0, kb1) @ self.rd('M', j, j + 2, kb0, kb1).T
                    if R == j >> 1:     # the diagonal tile: only its lower blocks are data (the kernel rewrites the upper one too)
                        self.wr('M', 2 * R, j, v[:, :TS])
                        self.wr('M', 2 * R + 1, j + 1, v[TS:, TS:])
                    else:
                        self.wr('M', 2 * R, j, v)
                elif ty == 3 and jb.get('wide'):     # CUPD on 128 x 128 tiles
                    assert jb['j0'] % 2 == 0 and jb['j1'] % 2 == 0 and kb0 % 2 == 0
                    nc = (jb['j1'] - jb['j0']) // 2
                    R, j = jb['R0'] + t // nc, jb['j0'] + 2 * (t % nc)
                    assert R < jb['R1']
                    own = j >= kb0
                    ks = j if own else kb0
                    v = self.rd('M', 2 * R, 2 * R + 2, ks, kb1) @ self.rd('W', ks, kb1, j, j + 2)
                    if not own:
                        v = v + self.rd('V', 2 * R, 2 * R + 2, j, j + 2)
                    self.wr('V', 2 * R, j, v)
                elif ty == 1:       # SYRK
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

    def run_tri(self, l):
        """one step of one level of the triangular inverse for the pairs [tri_p0, tri_p0 + tri_np): tiles of `u` blocks"""
        u = 2 if l['tiles128'] else 1
        mb, nbt = l['tri_mb'], self.nb // u
        for pr in range(l['tri_p0'], l['tri_p0'] + l['tri_np']):
            C0 = 2 * pr * mb
            R0 = C0 + mb
            for rl in range(mb):
                for cl in range(mb):
                    self.new_item()
                    if R0 + rl >= nbt:
                        continue
                    r, c = (R0 + rl) * u, (C0 + cl) * u
                    acc = np.zeros((TS * u, TS * u))
                    if l['tri_w'] == 0:
                        for kt in range(cl, mb):
                            k = (C0 + kt) * u
                            acc += self.rd('M', r, r + u, k, k + u) @ self.rd('W', k, k + u, c, c + u)
                        self.wr('V', r, c, acc)
                    else:
                        for kt in range(0, rl + 1):
                            k = (R0 + kt) * u
                            acc += self.rd('W', r, r + u, k, k + u) @ self.rd('V', k, k + u, c, c + u)
                        self.wr('W', r, c, -acc)

    def run_psolve(self, l):
        """rows from block row r_lo on: L[R, jt] = sum_{kt <= jt} A[R, kt] W_PP[jt, kt]^T on 128 x 128 tiles (2 x 2 blocks)"""
        J, jt = l['J'], l['c_lo']
        c0 = J + 2 * jt
        for R in range(l['r_lo'] // 2, self.nb // 2):
            self.new_item()
            v = self.rd('M', 2 * R, 2 * R + 2, J, c0 + 2) @ self.rd('W', c0, c0 + 2, J, c0 + 2).T
            self.wr('M', 2 * R, c0, v)

    def run(self, launches):
        for l in launches:
            self.begin_launch()
            if l['kind'] == 1:
                self.run_leaf(l)
            elif l['kind'] == 2:
                self.run_step(l)
            elif l['kind'] == 3:
                self.run_trail(l)
            elif l['kind'] == 5:
                self.run_tri(l)
            elif l['kind'] == 6:
                self.run_psolve(l)
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


class DagReplay(Replay):
    """The same plan as the task graph of the persistent launch (fill_sched.h: DagBuilder).  Segments are executed in
    sequence order, each on a snapshot of the state before it; every block a segment reads must have been last written
    by a segment in the transitive closure of its declared dependencies (or be initial data), and every block it writes
    must have had all its earlier readers and its last writer in that closure: with these two properties ANY execution
    that respects the per-(segment, component) counters gives the result of the sequential one."""

    def run_dag(self, segs):
        nseg = len(segs)
        clo = []
        for i, sg in enumerate(segs):
            c = set()
            for d, need in sg['deps']:
                assert 0 <= d < i, 'dependencies point backwards in the sequence'
                assert need == segs[d]['per_comp']
                c.add(d)
                c |= clo[d]
            clo.append(c)
        last_w = {}          # block -> segment
        readers = {}         # block -> segments that read it since its last write
        t_expect = 0
        for i, sg in enumerate(segs):
            assert sg['t0'] == t_expect and sg['ntasks'] == sg['per_comp'] * self.q
            t_expect += sg['ntasks']
            self.begin_launch()
            if sg['kind'] == 1:
                self.run_leaf(sg)
            elif sg['kind'] == 2:
                self.run_step(sg)
            elif sg['kind'] == 3:
                self.run_trail(sg)
            elif sg['kind'] == 5:
                self.run_tri(sg)
            elif sg['kind'] == 6:
                self.run_psolve(sg)
            else:
                self.run_jobs(sg)
            self.end_launch()
            nitems = self.item
            if sg['kind'] == 3 and sg['tiles128'] and not (sg['t_count'] or sg['r_lo'] or sg['r_hi']):
                # the kernel works on 128x128 tiles there (4 blocks, 3 on the diagonal, per task): same blocks, fewer tasks
                nitems = sg['per_comp']
            if sg['kind'] not in (4, 5) or (sg['kind'] == 4 and sg['type'] < 5):     # (a task of a block-inverse level may fall outside the matrix)
                assert nitems == sg['per_comp'], (i, sg, nitems)
            rset = {(b, r, c) for _, b, r, c in self.reads}
            for key in rset:
                w = last_w.get(key)
                assert w is None or w == i or w in clo[i], 'segment %d reads %s written by %d: not a dependency' % (i, key, w)
            for key in self.writes:
                for o in [last_w.get(key)] + sorted(readers.get(key, ())):
                    assert o is None or o == i or o in clo[i], 'segment %d writes %s last touched by %d: not a dependency' % (i, key, o)
            for key in self.writes:
                last_w[key] = i
                readers[key] = set()
            for key in rset:
                if key not in self.writes:
                    readers.setdefault(key, set()).add(i)
        return max(sg['ndeps'] for sg in segs)


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


WIDE_CASES = [
    (64, 8, 4, dict(fill_wide=1, fill_leaf=496, fill_step=496)),      # the headline size, eight components: one wide tile per CU
    (64, 8, 4, dict(fill_wide=1, fill_leaf=992, fill_step=992)),
    (64, 1, 4, dict(fill_wide=1)),
    (64, 2, 4, dict(fill_wide=1, fill_leaf=496, fill_step=496, far_rides=0)),
    (32, 6, 4, dict(fill_wide=1, fill_leaf=496, fill_step=496)),
    (18, 1, 4, dict(fill_wide=1)),                                    # a short last panel
    (10, 3, 4, dict(fill_wide=1, fill_leaf=40, fill_step=24)),
    (24, 2, 8, dict(fill_wide=1)),
    (12, 1, 2, dict(fill_wide=1)),
    (14, 2, 6, dict(fill_wide=1)),
    (16, 1, 3, dict(fill_wide=1)),                                    # odd panels: the jobs stay 128 x 64
]


@pytest.mark.parametrize('nb,q,ob,kw', WIDE_CASES)
@pytest.mark.parametrize('mode', ['factor', 'inverse', 'inverse_and_ainv'])
def test_plans_with_128x128_filler_tiles(dump_plan, nb, q, ob, kw, mode):
    """FillJob::wide: the far columns of the trailing update and the rank-(64 ob) updates of the inverse as column pairs"""
    prog = 0 if mode == 'factor' else 1
    if prog and ob & (ob - 1):
        pytest.skip('the progressive inverse needs a power-of-two panel (plan_params)')
    launches = dump_plan(nb, q, ob, progressive=prog, with_dupd=1 if mode == 'inverse_and_ainv' else 0, **kw)
    big = [jb for l in launches for jb in l['jobs'] if jb['type'] in (1, 3)]
    assert all(bool(jb.get('wide')) == (ob % 2 == 0) for jb in big)
    assert not any(jb.get('wide') for l in launches for jb in l['jobs'] if jb['type'] not in (1, 3))
    cap = max(kw.get('fill_leaf', 248), kw.get('fill_step', 248))
    for l in launches:
        if l['kind'] in (1, 2):
            assert sum(jb['nblk'] * (2 if jb.get('wide') else 1) for jb in l['jobs']) <= cap
    r = Replay(nb, q, seed=5 * nb + q)
    r.run(launches)
    r.check(inverse=prog == 1, ainv=mode == 'inverse_and_ainv')


def test_filler_capacity_is_respected(dump_plan):
    for q in (1, 2, 8):
        for l in dump_plan(64, q, 4, progressive=1):
            if l['kind'] in (1, 2):
                assert l['nblk'] <= 248
                assert len(l['jobs']) <= 6


@pytest.mark.parametrize('nb,q,ob,kw', CASES + [(14, 2, 6, {}), (16, 1, 3, {})])
@pytest.mark.parametrize('progressive', [0, 1])
def test_task_graph_dependencies_cover_every_block_hazard(dump_plan, nb, q, ob, kw, progressive):
    if progressive and (ob & (ob - 1)):
        pytest.skip('the progressive inverse needs a power-of-two panel')
    launches, segs = dump_plan(nb, q, ob, progressive=progressive, dag=1, **kw)
    r = DagReplay(nb, q, seed=5 * nb + q)
    maxdep = r.run_dag(segs)
    assert maxdep <= 16
    r.check(inverse=bool(progressive))
    # the graph holds exactly the work of the launch list
    assert sum(sg['ntasks'] for sg in segs if sg['kind'] == 4) == sum(jb['nblk'] for l in launches for jb in l['jobs'])


def test_task_graph_mutations_are_caught(dump_plan):
    """dropping any single declared dependency of the headline graph must trip the hazard check (the lists are minimal)"""
    launches, segs = dump_plan(16, 2, 4, progressive=1, dag=1)
    caught = total = 0
    for i, sg in enumerate(segs):
        for j in range(sg['ndeps']):
            import copy
            mut = copy.deepcopy(segs)
            del mut[i]['deps'][j]
            mut[i]['ndeps'] -= 1
            total += 1
            try:
                DagReplay(16, 2, seed=1).run_dag(mut)
            except AssertionError:
                caught += 1
    assert total > 20 and caught == total, (caught, total)


@pytest.mark.parametrize('nb,q,ob,kw', [(64, 8, 4, {}), (64, 1, 4, {}), (64, 2, 4, {}), (16, 4, 4, {}), (32, 6, 4, {}), (18, 1, 4, {}),
                                        (10, 3, 4, {}), (6, 1, 4, {}), (4, 1, 4, {}), (2, 1, 4, {}), (24, 2, 8, {}), (20, 1, 8, {}),
                                        (12, 1, 2, {}), (14, 2, 6, {}), (16, 1, 3, {}), (32, 1, 4, dict(syrk_small=0)),
                                        (32, 3, 2, dict(syrk_small=0)), (64, 4, 4, dict(syrk_small=100))])
def test_interleaved_order_for_the_persistent_launch(dump_plan, nb, q, ob, kw):
    """near / far trailing updates, the far part in chunks that alternate with the next panel's chain (Planner::
    run_interleaved): valid as a launch list, and as a task graph with the derived dependencies"""
    launches, segs = dump_plan(nb, q, ob, progressive=0, dag=1, interleaved=1, **kw)
    r = Replay(nb, q, seed=7 * nb + q)
    r.run(launches)
    r.check(inverse=False)
    r = DagReplay(nb, q, seed=7 * nb + q)
    assert r.run_dag(segs) <= 16
    r.check(inverse=False)
    # a chain launch never waits for a far chunk of the update before it: that is the point of the order
    kinds = {i: sg for i, sg in enumerate(segs)}
    for i, sg in enumerate(segs):
        if sg['kind'] in (1, 2):
            for d, _ in sg['deps']:
                assert not (kinds[d]['kind'] == 3 and kinds[d]['c_lo'] > sg['pe']), (i, sg, kinds[d])
    # ... and the chain part of a step never waits for the bulk part of another
    for i, sg in enumerate(segs):
        if sg['kind'] == 2 and sg['has_special']:
            for d, _ in sg['deps']:
                assert not (kinds[d]['kind'] == 2 and kinds[d]['trmm_r0'] > kinds[d]['c'] + 1), (i, sg, kinds[d])


@pytest.mark.parametrize('nb,q,ob,kw', [(64, 8, 4, {}), (64, 1, 4, {}), (64, 2, 4, {}), (16, 4, 4, {}), (32, 6, 4, {}), (18, 1, 4, {}),
                                        (10, 3, 4, {}), (6, 1, 4, {}), (4, 1, 4, {}), (2, 1, 4, {}), (24, 2, 8, {}), (20, 1, 8, {}),
                                        (12, 1, 2, {}), (14, 2, 6, {}), (16, 1, 3, {}), (32, 2, 4, dict(trtri_all_small=1)),
                                        (22, 3, 4, dict(trtri_all_small=1)), (64, 4, 4, dict(syrk_small=100))])
def test_interleaved_order_with_the_triangular_inverse(dump_plan, nb, q, ob, kw):
    """the factorisation AND W = L^-1 as one sequence: the early levels of the inverse ride on the chain-bound end of the
    factorisation, the rest follows; replayed as a launch list and as a task graph"""
    launches, segs = dump_plan(nb, q, ob, progressive=0, dag=1, interleaved=1, with_trtri=1, **kw)
    assert any(l['kind'] == 5 for l in launches) or nb < 2
    r = Replay(nb, q, seed=11 * nb + q)
    r.run(launches)
    r.check(inverse=True, ainv=False)
    r = DagReplay(nb, q, seed=11 * nb + q)
    assert r.run_dag(segs) <= 16
    r.check(inverse=True, ainv=False)
    # no step of the factorisation waits for a unit of the inverse
    for i, sg in enumerate(segs):
        if sg['kind'] in (1, 2, 3):
            assert all(segs[d]['kind'] != 5 for d, _ in sg['deps']), (i, sg)


def check_runs(segs, runs, q):
    """the sequence the persistent kernel takes its tasks in: every segment's tasks exactly once, in order, and every run
    behind ALL runs of everything its segment depends on (so that a task that has been taken only ever waits for tasks
    before it in the sequence: no deadlock, whatever the number of resident workgroups)"""
    pos_last = {}
    nxt = [0] * len(segs)
    t0 = 0
    for i, r in enumerate(runs):
        assert r['t0'] == t0 and r['n'] > 0
        t0 += r['n']
        sg = segs[r['seg']]
        assert r['b0'] == nxt[r['seg']], 'runs of a segment in task order'
        nxt[r['seg']] += r['n']
        for d, _ in sg['deps']:
            assert nxt[d] == segs[d]['ntasks'], 'run %d of segment %d starts before segment %d has been taken whole' % (i, r['seg'], d)
    assert all(nxt[i] == sg['ntasks'] for i, sg in enumerate(segs))
    assert t0 == sum(sg['ntasks'] for sg in segs)


@pytest.mark.parametrize('nb,q,ob,kw', [(64, 8, 4, {}), (64, 1, 4, {}), (64, 2, 4, {}), (16, 4, 4, {}), (32, 6, 4, {}), (18, 1, 4, {}),
                                        (10, 3, 4, {}), (2, 1, 4, {}), (24, 2, 8, {}), (12, 1, 2, {}), (14, 2, 6, {}),
                                        (32, 2, 4, dict(trtri_all_small=1))])
@pytest.mark.parametrize('mode', ['launch order', 'interleaved', 'interleaved with inverse'])
def test_scheduled_sequence_is_a_valid_order_of_the_graph(dump_plan, nb, q, ob, kw, mode):
    extra = dict(interleaved=0) if mode == 'launch order' else dict(interleaved=1, with_trtri=int(mode.endswith('inverse')))
    if mode == 'launch order':
        kw = {k: v for k, v in kw.items() if k != 'trtri_all_small'}
    launches, segs = dump_plan(nb, q, ob, progressive=int(mode == 'launch order' and (ob & (ob - 1)) == 0), dag=1, **extra, **kw)
    check_runs(segs, dump_plan.last_runs, q)
