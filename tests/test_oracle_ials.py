"""Pins the iALS oracle (oracle/ials_oracle.cpp) against the closed-form float64 checks
the reference's own tests hold for this path (the reference cannot be compiled or
imported offline, so these checks are what anchors parity).

Each test cites the reference test it restates (/root/reference/tests/recommenders/test_ials.py).
"""
import math

import numpy as np
import pytest
import scipy.sparse as sps

import oracle as O


def train(X, K, epochs, kind, steps=3, **kw):
    mc = O.model_config(K, **kw)
    sc = O.solver_config(1, kind, steps)
    t = O.IALSTrainer(mc, X)
    for _ in range(epochs):
        t.step(sc)
    return t, mc, sc


def binarised(X):
    Xd = X.toarray()
    Xd[Xd.nonzero()] = 1.0
    return Xd


def test_overfit_cholesky(X_small):
    # test_ials.py:54-76
    t, _, _ = train(X_small, 4, 100, "CHOLESKY", alpha0=100, reg=1e-1, nu=0, loss_type="ORIGINAL")
    sp = O.solver_config(1, "CHOLESKY", 5)
    u, i = t.transform_user(X_small, sp), t.transform_item(X_small, sp)
    np.testing.assert_allclose(u @ i.T, binarised(X_small), rtol=1e-2, atol=1e-2)


def test_overfit_cg_and_shape_mismatch(X_small):
    # test_ials.py:516-548
    t, _, _ = train(X_small, 3, 100, "CG", alpha0=100, reg=1e-1, nu=0, loss_type="ORIGINAL")
    sp = O.solver_config(1, "CG", 5)
    u, i = t.transform_user(X_small, sp), t.transform_item(X_small, sp)
    np.testing.assert_allclose(u @ i.T, binarised(X_small), rtol=1e-2, atol=1e-2)
    with pytest.raises(ValueError, match="Shape mismatch"):
        t.transform_item(X_small.T.tocsr(), sp)


@pytest.mark.parametrize("sub", [1, 2, 3, 4])
def test_overfit_ialspp(X_small, sub):
    # test_ials.py:573-599
    mc = O.model_config(4, alpha0=100, reg=1.0, nu=0, loss_type="ORIGINAL")
    sc = O.solver_config(1, "IALSPP", 3, ialspp_subspace_dimension=sub)
    t = O.IALSTrainer(mc, X_small)
    for _ in range(300):
        t.step(sc)
    np.testing.assert_allclose(t.user @ t.item.T, binarised(X_small), rtol=1e-2, atol=1e-2)


def test_loss_original(X_small):
    # test_ials.py:456-483
    alpha0, reg = 0.1, 0.1
    t, _, sc = train(X_small, 2, 2, "CHOLESKY", alpha0=alpha0, reg=reg, nu=0, loss_type="ORIGINAL")
    u, v = t.user.astype(np.float64), t.item.astype(np.float64)
    ui = u @ v.T
    row, col = X_small.nonzero()
    manual = (X_small.data + alpha0) @ ((ui[row, col] - 1) ** 2)
    ui[row, col] = 0.0
    manual += alpha0 * (ui.ravel() @ ui.ravel())
    manual += reg * ((u ** 2).sum() + (v ** 2).sum())
    assert t.compute_loss(sc) == pytest.approx(manual / 2, rel=1e-5)


@pytest.mark.parametrize("alpha0", [0.0, 0.1])
def test_loss_ialspp(X_small, alpha0):
    # test_ials.py:486-513
    reg = 0.1
    t, _, sc = train(X_small, 2, 2, "CHOLESKY", alpha0=alpha0, reg=reg, nu=0, loss_type="IALSPP")
    u, v = t.user.astype(np.float64), t.item.astype(np.float64)
    ui = u @ v.T
    row, col = X_small.nonzero()
    manual = X_small.data @ ((ui[row, col] - 1) ** 2)
    manual += alpha0 * (ui.ravel() @ ui.ravel())
    manual += reg * ((u ** 2).sum() + (v ** 2).sum())
    assert t.compute_loss(sc) == pytest.approx(manual / 2, rel=1e-5)


def test_cg_matches_cholesky(X_small):
    # test_ials.py:627-661
    a, _, _ = train(X_small, 4, 5, "CHOLESKY", alpha0=1 / 4.5, reg=3)
    b, _, _ = train(X_small, 4, 5, "CG", steps=5, alpha0=1 / 4.5, reg=3)
    sa, sb = O.solver_config(1, "CHOLESKY", 5), O.solver_config(1, "CG", 5)
    np.testing.assert_allclose(a.transform_user(X_small, sa), b.transform_user(X_small, sb),
                               atol=1e-3, rtol=1e-4)
    np.testing.assert_allclose(a.transform_item(X_small, sa), b.transform_item(X_small, sb),
                               atol=1e-3, rtol=1e-4)


def ials_grad(X, u, v, reg, alpha0, epsilon):
    # test_ials.py:19-51
    weight = (lambda x: x) if epsilon is None else (lambda x: math.log(1 + x / epsilon))
    uv = u @ v.T
    gu, gv = np.zeros_like(u), np.zeros_like(v)
    for a in range(u.shape[0]):
        for b in range(v.shape[0]):
            x = X[a, b]
            sc = alpha0 * uv[a, b] if x == 0 else (alpha0 + weight(x)) * (uv[a, b] - 1)
            gu[a] += v[b] * sc
            gv[b] += u[a] * sc
    return gu + reg * u, gv + reg * v


def test_gradient_vanishes_at_cholesky_optimum_logscale(X_small):
    # test_ials.py:664-697 (the recommender's log scaling applied to X up front, ials.py:437-446)
    ALPHA0, REG, EPS, K = 2.4, 1.1, 3.0, 5
    Xs = X_small.copy()
    Xs.data = np.log(1 + Xs.data / EPS)
    t, _, _ = train(Xs, K, 200, "CHOLESKY", alpha0=ALPHA0, reg=REG, nu=0, loss_type="ORIGINAL")
    u = t.user.astype(np.float64)
    v = t.transform_item(Xs, O.solver_config(1, "CHOLESKY", 5)).astype(np.float64)
    gu, gv = ials_grad(X_small, u, v, REG, ALPHA0, EPS)
    np.testing.assert_allclose(gv, 0, atol=1e-5)
    np.testing.assert_allclose(gu, 0, atol=1e-5)


def test_half_step_vs_normal_equations():
    # per-row np.linalg.solve of the normal equations (test_ials.py:185-227, 431-449 style)
    rng = np.random.default_rng(0)
    X = sps.random(60, 40, density=0.2, format="csr", random_state=1, dtype=np.float64)
    X.data = rng.uniform(0.5, 3, X.nnz)
    K, alpha0, reg, nu = 8, 0.3, 0.05, 0.5
    for loss in ("IALSPP", "ORIGINAL"):
        mc = O.model_config(K, alpha0=alpha0, reg=reg, nu=nu, loss_type=loss)
        t = O.IALSTrainer(mc, X)
        item0 = t.item.astype(np.float64)
        t.step(O.solver_config(2, "CHOLESKY"))
        P = alpha0 * item0.T @ item0
        bias = 0.0 if loss == "IALSPP" else alpha0
        for r in range(X.shape[0]):
            sl = slice(X.indptr[r], X.indptr[r + 1])
            V, c = item0[X.indices[sl]], X.data[sl]
            regr = reg * (alpha0 * X.shape[1] + (sl.stop - sl.start)) ** nu
            A = P + (V * c[:, None]).T @ V + regr * np.eye(K)
            b = ((c + bias)[:, None] * V).sum(axis=0)
            np.testing.assert_allclose(t.user[r], np.linalg.solve(A, b), rtol=2e-4, atol=2e-6)


@pytest.mark.parametrize("n_threads", [1, 4, 64])
def test_user_scores(n_threads):
    # test_ials.py:551-570
    rng = np.random.default_rng(0)
    n_users, n_items, K = 513, 257, 31
    t = O.IALSTrainer(O.model_config(K), sps.csr_matrix((n_users, n_items), dtype=np.float32))
    user = rng.standard_normal((n_users, K)).astype(np.float32)
    item = rng.standard_normal((n_items, K)).astype(np.float32)
    t.user, t.item = user, item
    sc = O.solver_config(n_threads)
    for b, e in [(0, n_users), (17, 193), (n_users, n_users)]:
        np.testing.assert_allclose(t.user_scores(b, e, sc), user[b:e] @ item.T, rtol=2e-5, atol=2e-5)


def test_errors(X_small):
    # empty row with alpha0 = 0: Cholesky throws (hpp:316-318), CG zeroes the row (hpp:207-210)
    mc = O.model_config(4, alpha0=0.0, reg=1e-3)
    with pytest.raises(RuntimeError, match="Cholesky"):
        O.IALSTrainer(mc, X_small).step(O.solver_config(1, "CHOLESKY"))
    t = O.IALSTrainer(mc, X_small)
    t.step(O.solver_config(1, "CG"))
    assert np.all(t.user[3] == 0)
    with pytest.raises(ValueError):
        O.IALSTrainer(mc, X_small).step(O.solver_config(0, "CG"))


def test_init_golden_stream():
    # golden vector: libstdc++ mt19937(42) + normal_distribution<float>(0, 0.1/sqrt(K)), hpp:64-76
    import os

    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "ials_init_seed42.npz"))
    for K in (10, 16, 40, 64):  # 10, 40: float vs double stddev quotient differ by an ulp
        np.testing.assert_array_equal(O.ials_init(8, K, 0.1, 42), g[f"K{K}"])


def ialspp_half_step_float64(X, target, other, alpha0, reg, nu, bias, sub, iterations):
    """float64 restatement of Solver::step_ialspp / _step_dimrange / _prediction
    (hpp:387-535): block-coordinate Newton steps on the row objective, with the prediction
    cache corrected after every block."""
    X = sps.csr_matrix(X).astype(np.float64)
    x_all = target.astype(np.float64).copy()
    V = other.astype(np.float64)
    K = V.shape[1]
    P = alpha0 * V.T @ V
    for _ in range(iterations):
        for r in range(X.shape[0]):
            sl = slice(X.indptr[r], X.indptr[r + 1])
            Vr, c = V[X.indices[sl]], X.data[sl]
            reg_r = float(np.float32(reg) * np.float32(np.float32(alpha0) * V.shape[0] + (sl.stop - sl.start)) ** np.float32(nu))
            x = x_all[r]
            pred = Vr @ x  # hpp:410-413
            for ds in range(0, K, sub):
                b = slice(ds, min(ds + sub, K))
                grad = P[b] @ x + reg_r * x[b] + Vr[:, b].T @ (c * (pred - 1.0) - bias)  # hpp:468-482
                A = P[b, b] + (Vr[:, b] * c[:, None]).T @ Vr[:, b] + reg_r * np.eye(b.stop - b.start)
                delta = np.linalg.solve(A, grad)  # hpp:495-497
                x[b] -= delta
                pred -= Vr[:, b] @ delta  # hpp:500-506
    return x_all


@pytest.mark.parametrize("K,sub,loss", [(8, 3, "IALSPP"), (16, 16, "ORIGINAL"), (24, 8, "IALSPP"),
                                        (20, 64, "ORIGINAL"), (12, 1, "IALSPP")])
def test_ialspp_half_step_matches_float64_block_newton(K, sub, loss):
    """Pins the iALS++ / iCD oracle to 2e-5 per row (the over-fit check above only sees 1e-2).
    sub = 1 is the iCD branch (hpp:673-677, _step_icd :555-630: the same update per dimension)."""
    rng = np.random.default_rng(3)
    X = sps.random(60, 45, density=0.2, format="csr", random_state=5, dtype=np.float64)
    X.data = rng.uniform(0.5, 3.0, X.nnz)
    alpha0, reg, nu = 0.2, 0.05, 0.5
    mc = O.model_config(K, alpha0=alpha0, reg=reg, nu=nu, loss_type=loss)
    sc = O.solver_config(2, "IALSPP", 3, ialspp_subspace_dimension=sub, ialspp_iteration=2)
    user0 = (rng.standard_normal((60, K)) * 0.3).astype(np.float32)
    item0 = (rng.standard_normal((45, K)) * 0.3).astype(np.float32)
    P = O.ials_gramian(item0, alpha0, 1)
    got = O.ials_solver_step(user0, X, item0, P, mc, sc)
    bias = 0.0 if loss == "IALSPP" else alpha0
    want = ialspp_half_step_float64(X.astype(np.float32), user0, item0, alpha0, reg, nu, bias, sub, 2)
    err = np.linalg.norm(got - want, axis=1) / np.linalg.norm(want, axis=1)
    assert err.max() < 2e-5, err.max()
