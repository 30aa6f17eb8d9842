"""Cross-identities that pin the oracle's `full` path and its closed-form gradients."""
import numpy as np
import pytest

from oracle import lcgp_oracle as orc
from lcgp_amd import synth


def _full_model(n=48, d=2, p=4, q=3, seed=7, **kw):
    x, y = synth.make_full(seed, n, d, p, q)
    return orc.OracleLCGP(y=y, x=x, q=q, submethod='full', **kw)


def _rep_model(n_unique=20, reps=3, d=2, p=4, q=4, seed=9, **kw):
    x, y = synth.make_rep(seed, n_unique, reps, d, p, q)
    return orc.OracleLCGP(y=y, x=x, q=q, submethod='rep', **kw)


def _fd_grad(model, u, h=1e-6):
    g = np.zeros_like(u)
    for i in range(len(u)):
        e = np.zeros_like(u)
        e[i] = h
        model.set_unconstrained(u + e)
        fp = model.loss()
        model.set_unconstrained(u - e)
        fm = model.loss()
        g[i] = (fp - fm) / (2 * h)
    model.set_unconstrained(u)
    return g


def test_softclip_roundtrip_and_derivative():
    for lo, hi in (orc.LLMB_BOUNDS, orc.LLMB0_BOUNDS, orc.LNUG_BOUNDS):
        v = np.exp(np.linspace(np.log(lo * 1.5), np.log(hi * 0.2), 9))
        u = orc.softclip_inverse(v, lo, hi)
        # TFP's formula subtracts numbers of size `hi`: absolute accuracy ~ hi * eps, same as the reference
        np.testing.assert_allclose(orc.softclip_forward(u, lo, hi), v, rtol=1e-9, atol=hi * 1e-15)
    lo, hi = orc.LNUG_BOUNDS
    u = np.linspace(-3.0, 3.0, 13)
    fd = (orc.softclip_forward(u + 1e-6, lo, hi) - orc.softclip_forward(u - 1e-6, lo, hi)) / 2e-6
    np.testing.assert_allclose(orc.softclip_grad(u, lo, hi), fd, rtol=1e-6, atol=1e-12)


def test_percentile_nearest_indices():
    for m, idx in ((40, 20), (50, 24), (1024, 512), (4096, 2048), (5, 2)):
        row = np.random.default_rng(m).permutation(m).astype(float)[None, :]
        assert orc.percentile50_nearest(row, axis=1)[0, 0] == float(idx)


@pytest.mark.parametrize("kernel", ["matern32", "se"])
def test_eigh_form_equals_cholesky_form(kernel):
    m = _full_model(diag_error_structure=[2, 1, 1], kernel=kernel)
    for u in synth.param_points(11, m.get_unconstrained()):
        m.set_unconstrained(u)
        a, b = m.loss_reference_form(), m.loss()
        assert abs(a - b) <= 1e-11 * max(1.0, abs(a))


def test_rep_literal_equals_cholesky_form():
    for use_std in (True, False):
        m = _rep_model(rep_standardize_ybar=use_std)
        for u in synth.param_points(12, m.get_unconstrained()):
            m.set_unconstrained(u)
            a, b = m.loss_reference_form(), m.loss()
            assert abs(a - b) <= 1e-10 * max(1.0, abs(a))


def test_full_equals_n_times_rep_when_no_replicates():
    """r_i = 1, same standardisation, lsigma2s_rep = lsigma2s_full + 2 log(std)."""
    x, y = synth.make_full(5, 30, 2, 3, 3)
    mf = orc.OracleLCGP(y=y, x=x, q=3, submethod='full')
    mr = orc.OracleLCGP(y=y, x=x, q=3, submethod='rep')
    # rep mode sorts the unique rows; the full objective is permutation invariant
    order = np.lexsort(x.T[::-1])
    np.testing.assert_allclose(mr.x_unique, x[order])
    np.testing.assert_allclose(np.abs(mr.diag_D), np.abs(mf.diag_D), rtol=1e-10)
    mr.phi = mf.phi.copy()  # SVD sign freedom: use one basis for both
    for i, u in enumerate(synth.param_points(13, mf.get_unconstrained())):
        mf.set_unconstrained(u)
        mr.set_unconstrained(u)
        mr.lsigma2s = mf.lsigma2s + 2.0 * np.log(mf.ystd[:, 0])
        a = mf.loss()
        b = mr.n * mr.loss()
        assert abs(a - b) <= 1e-9 * max(1.0, abs(a)), (i, a, b)


@pytest.mark.parametrize("kind", ["full", "full_grouped", "rep_std", "rep_raw", "full_se", "rep_se"])
def test_closed_form_gradient_matches_finite_differences(kind):
    if kind == "full":
        m = _full_model()
    elif kind == "full_se":         # the squared-exponential extension: no reference, identities are its only check
        m = _full_model(kernel='se')
    elif kind == "rep_se":
        m = _rep_model(kernel='se')
    elif kind == "full_grouped":
        m = _full_model(diag_error_structure=[1, 3], robust_mean=False)
    elif kind == "rep_std":
        m = _rep_model()
    else:
        m = _rep_model(rep_standardize_ybar=False)
    for u in synth.param_points(14, m.get_unconstrained(), count=2):
        val, g = m.loss_and_grad_unconstrained(u)
        assert abs(val - m.loss()) < 1e-12 * max(1, abs(val))
        fd = _fd_grad(m, u)
        np.testing.assert_allclose(g, fd, rtol=2e-5, atol=2e-6 * max(1.0, np.max(np.abs(fd))))


@pytest.mark.parametrize("kernel", ["matern32", "se"])
def test_gradient_matches_torch_autograd_of_literal_form(kernel):
    """Independent check: autograd through the eigh-form objective (what TF's tape does)."""
    torch = pytest.importorskip("torch")
    m = _full_model(n=40, kernel=kernel)
    u = synth.param_points(15, m.get_unconstrained())[1]
    _, g = m.loss_and_grad_unconstrained(u)
    lLmb, lLmb0, ls2b, lnug = m.get_param()
    t = dict(l=torch.tensor(lLmb, requires_grad=True), s=torch.tensor(lLmb0, requires_grad=True),
             v=torch.tensor(lnug, requires_grad=True), e=torch.tensor(np.asarray(m.lsigma2s), requires_grad=True))
    x = torch.tensor(m.x)
    y = torch.tensor(m.y)
    phi = torch.tensor(m.phi)
    n = m.n
    psi_c = phi.T / torch.sqrt(torch.exp(t['e']))
    nlp = 0.0
    for k in range(m.q):
        a = x / t['l'][k]
        S = (a[:, None, :] - a[None, :, :]).abs()
        c0 = torch.exp(-0.5 * (S * S).sum(dim=2)) if kernel == 'se' else torch.prod(1 + S, dim=2) * torch.exp(-S.sum(dim=2))
        nt = t['v'][k] / (1 + t['v'][k])
        ck = t['s'][k] * ((1 - nt) * c0 + nt * torch.eye(n, dtype=torch.float64))
        wk, uk = torch.linalg.eigh(ck)
        qk = uk @ torch.diag(1 / (m.diag_D[k] + 1 / wk)) @ uk.T
        nlp = nlp + 0.5 * torch.sum(torch.log(1 + m.diag_D[k] * wk))
        nlp = nlp - 0.5 * torch.sum((y @ qk) * (torch.outer(psi_c[k], psi_c[k]) @ y))
    nlp = nlp + n / 2 * t['e'].sum() + 0.5 * torch.sum((y.T / torch.sqrt(torch.exp(t['e']))) ** 2)
    nlp.backward()
    u1, u2, u3, _ = m._split(u)
    ref = np.concatenate([
        (t['l'].grad.numpy() * orc.softclip_grad(u1, *orc.LLMB_BOUNDS)).reshape(-1),
        t['s'].grad.numpy() * orc.softclip_grad(u2, *orc.LLMB0_BOUNDS),
        t['v'].grad.numpy() * orc.softclip_grad(u3, *orc.LNUG_BOUNDS),
        t['e'].grad.numpy()])
    assert abs(float(nlp) - m.loss()) < 1e-10 * abs(m.loss())
    np.testing.assert_allclose(g, ref, rtol=1e-8, atol=1e-9 * np.max(np.abs(ref)))


def test_predict_rep_simplified_identities():
    """Identities the HIP predict path relies on: CinvM = sqrt(r) o z, T = D R^1/2 A^-1 R^1/2."""
    m = _rep_model()
    m.set_unconstrained(synth.param_points(16, m.get_unconstrained())[1])
    aux = m._aux_rep()
    lLmb, lLmb0, ls2b, lnug = m.get_param()
    r = m.r.astype(float)
    sr = np.sqrt(r)
    sis = np.exp(-0.5 * ls2b) * m.ybar_std[:, 0]
    for k in range(m.q):
        ck = orc.matern32(m.x_unique_s, m.x_unique_s, lLmb[k], lLmb0[k], lnug[k])
        a = np.eye(m.n) + m.diag_D[k] * ck * sr[:, None] * sr[None, :]
        ainv = np.linalg.inv(a)
        beta = sr * (m.ybar_s.T @ (sis * m.phi[:, k]))
        z = ainv @ beta
        np.testing.assert_allclose(aux['CinvMs'][k], sr * z, rtol=1e-7, atol=1e-9)
        np.testing.assert_allclose(aux['mks'][k], (beta - z) / (m.diag_D[k] * sr), rtol=1e-7, atol=1e-9)
        np.testing.assert_allclose(aux['Tks'][k], m.diag_D[k] * ainv * sr[:, None] * sr[None, :], rtol=1e-5, atol=1e-7)


def test_se_kernel_definition_and_rep_literal_form():
    """the extension's kernel: exp(-1/2 sum ((x - x') / ell)^2) with Matern32's nugget / scale structure; its replicated
    objective equals the step-by-step form as the Matern one does"""
    rng = np.random.default_rng(3)
    a, b = rng.uniform(size=(7, 3)), rng.uniform(size=(5, 3))
    ell = np.array([0.6, 1.1, 0.3])
    want = 1.7 * (1 - 0.02 / 1.02) * np.exp(-0.5 * (((a[:, None, :] - b[None, :, :]) / ell) ** 2).sum(axis=2))
    np.testing.assert_allclose(orc.matern32(a, b, ell, 1.7, 0.02, kernel='se'), want, rtol=1e-14)
    same = orc.matern32(a, a, ell, 1.7, 0.02, kernel='se')
    np.testing.assert_allclose(np.diag(same), 1.7, rtol=1e-14)          # (1 - nt) + nt on the diagonal
    m = _rep_model(kernel='se')
    for u in synth.param_points(12, m.get_unconstrained()):
        m.set_unconstrained(u)
        assert abs(m.loss_reference_form() - m.loss()) <= 1e-10 * max(1.0, abs(m.loss()))
