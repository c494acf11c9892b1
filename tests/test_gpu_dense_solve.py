"""The reduced system's solver on its own (calico_rp_dense_solve: blocked Cholesky, inverse of
the factor formed beside the panel updates, x = Z^T (Z b)) against numpy at sizes around the
64-column panel boundaries, a full last panel, config D's 945, config B's refined 1141 and 33 panels (2100)."""
import numpy as np
import pytest

from calico_amd import capi, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def handle():
    cfg = synth.custom(2, 1, 2)
    return capi.Problem(cfg.cn, cfg.pn, cfg.tn, cfg.p_star_var, cfg.t_star_var, device=0)


def _spd(n, cond, seed):
    rng = np.random.default_rng(seed)
    q, _ = np.linalg.qr(rng.standard_normal((n, n)))
    ev = np.logspace(0.0, np.log10(cond), n)
    return (q * ev) @ q.T


@pytest.mark.parametrize("n", [1, 7, 63, 64, 65, 127, 128, 129, 192, 200, 483, 945, 1141, 2100])
def test_against_numpy(handle, n):
    S = _spd(n, 1e4, n)
    S = 0.5 * (S + S.T)
    x_true = np.random.default_rng(1000 + n).standard_normal(n)
    b = S @ x_true
    x = handle.dense_solve(S, b)
    ref = np.linalg.solve(S, b)
    scale = np.abs(ref).max()
    assert np.abs(x - ref).max() <= 1e-10 * scale, np.abs(x - ref).max() / scale
    # backward error of the solve itself
    assert np.abs(S @ x - b).max() <= 1e-12 * np.abs(S).max() * scale * n


def test_ill_conditioned_matches_a_substitution_solve(handle):
    """cond 1e10: the product with the explicit inverse of the factor loses no more than the
    substitution does (both are bounded by cond * eps relative to the solution)."""
    n = 300
    S = _spd(n, 1e10, 5)
    S = 0.5 * (S + S.T)
    x_true = np.random.default_rng(6).standard_normal(n)
    b = S @ x_true
    x = handle.dense_solve(S, b)
    L = np.linalg.cholesky(S)
    from scipy.linalg import solve_triangular
    ref = solve_triangular(L.T, solve_triangular(L, b, lower=True), lower=False)
    scale = np.abs(x_true).max()
    err_ref = np.abs(ref - x_true).max() / scale
    err = np.abs(x - x_true).max() / scale
    assert err <= max(10.0 * err_ref, 1e-12), (err, err_ref)


def test_not_positive_definite_is_reported(handle):
    n = 100
    S = _spd(n, 10.0, 2)
    S[70, 70] = -1.0
    with pytest.raises(capi.RpError):
        handle.dense_solve(S, np.ones(n))
