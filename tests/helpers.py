"""Test-only stand-ins.  `OracleEngine` answers the engine interface of lcgp_amd.engine.HotPathEngine with the
CPU oracle's arithmetic, so the HOST logic around the hot path (theta packing, assembly of the reduced
(P+1)-vector, chain rule, component sharding, cache handling) can be exercised without a GPU.  It is never
used by the product and never by the `-m gpu` parity tests (those go through liblcgp_hip.so)."""
import numpy as np
import scipy.linalg as sla

from oracle import lcgp_oracle as orc


class OracleEngine:
    device = None

    def __init__(self, x, Y, sr=None, q_local=1, dtype='float64', device=None, comp_ids=None, q_total=None, kernel='matern32'):
        self.kernel = kernel
        self.comp_ids = list(range(q_local)) if comp_ids is None else list(comp_ids)
        self.q_total = q_local if q_total is None else q_total
        self.x = np.asarray(x, np.float64)
        self.Y = np.asarray(Y, np.float64)
        self.sr = None if sr is None else np.asarray(sr, np.float64)
        self.n, self.d = self.x.shape
        self.p = self.Y.shape[0]
        self.q_local = q_local
        self._state = None

    def evaluate(self, theta_rows):
        d, p = self.d, self.p
        out = np.zeros((self.q_local, d + 5 + p))
        state = []
        for i, th in enumerate(np.asarray(theta_rows, np.float64)):
            ell, scale, nug, D, psi = th[:d], th[d], th[d + 1], th[d + 2], th[d + 3:]
            b = self.Y.T @ psi
            low, c0, s_all, hl, z = orc._chol_component(self.x, ell, scale, nug, D, b, self.sr, kernel=self.kernel)
            ge, gs, gn = orc._kernel_param_grads(low, c0, s_all, z, D, ell, scale, nug, self.sr, kernel=self.kernel)
            out[i, 0], out[i, 1], out[i, 2] = hl, b @ (b - z), 0.0
            out[i, 3:3 + d], out[i, 3 + d], out[i, 4 + d] = ge, gs, gn
            out[i, 5 + d:] = self.Y @ (b - z)
            state.append((th.copy(), low, z, b))
        self._state = state
        return out

    def evaluate_partial(self, theta_rows, guard=0.0):
        """What lcgp_pack_partial assembles on the device (include/lcgp_hip.h), as a CPU torch tensor."""
        import torch
        theta_rows = np.asarray(theta_rows, np.float64)
        out = self.evaluate(theta_rows)
        d, p, q = self.d, self.p, self.q_total
        vec = np.zeros(3 + q * d + 2 * q + p)
        vec[-1] = guard
        for row, th, k in zip(out, theta_rows, self.comp_ids):
            D, psi = th[d + 2], th[d + 3:]
            vec[0] += row[0] - row[1] / (2.0 * D)
            vec[1] += row[2]
            vec[2 + k * d:2 + (k + 1) * d] = row[3:3 + d]
            vec[2 + q * d + k] = row[3 + d]
            vec[2 + q * d + q + k] = row[4 + d]
            vec[2 + q * d + 2 * q:-1] += 0.5 * psi * row[5 + d:5 + d + p] / D
        return torch.as_tensor(vec)

    def predict_device(self, x0s, same=False):
        import torch
        gh, gv = self.predict(x0s, same)
        return torch.as_tensor(gh), torch.as_tensor(gv)

    def predict_block(self, x0s, same=False):
        import torch
        return torch.stack(self.predict_device(x0s, same))

    def predict(self, x0s, same=False):
        n0 = x0s.shape[0]
        gh = np.zeros((self.q_local, n0))
        gv = np.zeros((self.q_local, n0))
        sr = np.ones(self.n) if self.sr is None else self.sr
        for i, (th, low, z, b) in enumerate(self._state):
            d = self.d
            ell, scale, nug, D = th[:d], th[d], th[d + 1], th[d + 2]
            a = x0s / ell
            bb = self.x / ell
            S = np.abs(a[:, None, :] - bb[None, :, :])
            c0 = np.exp(-0.5 * (S * S).sum(axis=2)) if self.kernel == 'se' else np.prod(1 + S, axis=2) * np.exp(-S.sum(axis=2))
            nt = nug / (1 + nug)
            c = scale * ((1 - nt) * c0 + (nt * np.eye(n0) if same else 0.0)) * sr[None, :]
            gh[i] = c @ z
            u = sla.solve_triangular(low, c.T, lower=True)
            gv[i] = scale - D * np.sum(u * u, axis=0)
        return gh, gv

    def fetch_vector(self, which, k):
        return self._state[k][3 if which == 0 else 2].copy()

    def fetch_matrix(self, which, k):
        low = self._state[k][1]
        n = self.n
        if which == 0:
            return low + np.tril(low, -1).T
        w = sla.solve_triangular(low, np.eye(n), lower=True)
        if which == 1:
            return w + np.tril(w, -1).T
        return w.T @ w


def patch_engine(model):
    """Make `model` (lcgp_amd.LCGP) use the OracleEngine stand-in."""
    import lcgp_amd.engine as eng_mod
    from lcgp_amd import dist as _dist

    def _make():
        rank, world = _dist.rank_world(model._group)
        model._local_ks = _dist.local_components(model.q, rank, world)
        if not model._local_ks:
            return None
        import numpy as _np
        if model.submethod == 'rep':
            sr = _np.sqrt(model.r.numpy().astype(float))
            yb = (model.ybar_s if model.rep_standardize_ybar else model.ybar).numpy()
            return OracleEngine(model.x_unique_s.numpy(), yb * sr[None, :], sr, len(model._local_ks),
                                comp_ids=model._local_ks, q_total=model.q, kernel=model.kernel)
        return OracleEngine(model.x.numpy(), model.y.numpy(), None, len(model._local_ks),
                            comp_ids=model._local_ks, q_total=model.q, kernel=model.kernel)

    model._make_engine = _make
    return model
