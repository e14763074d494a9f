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


def cg_half_step_float64(X, target, other, alpha0, reg, nu, bias, steps, warm=True):
    """float64 restatement of Solver::step_cg (hpp:199-264), matrix free like the reference:
    warm start from the row's current value (:199), `nnz == 0` zeroes the row (:207-210),
    b (:212-221), r = b - P x - reg x - sum c (v.x) v (:222-228), at most `steps` iterations
    (0 means K, :232-234) with the two absolute exits at ||r||^2 <= 1e-20 (:238, :258) and
    beta = new ||r||^2 / old (:261).  Returns the rows and the number of iterations each ran."""
    X = sps.csr_matrix(X).astype(np.float64)
    V = other.astype(np.float64)
    K = V.shape[1]
    P = alpha0 * V.T @ V
    out = np.zeros((X.shape[0], K))
    iters = np.zeros(X.shape[0], dtype=np.int64)
    for r_ in range(X.shape[0]):
        sl = slice(X.indptr[r_], X.indptr[r_ + 1])
        nnz = sl.stop - sl.start
        if nnz == 0:
            continue
        Vr, c = V[X.indices[sl]], X.data[sl]
        reg_r = float(np.float32(reg) * np.float32(np.float32(alpha0) * V.shape[0] + nnz) ** np.float32(nu))
        x = target[r_].astype(np.float64).copy() if warm else np.zeros(K)
        b = ((c + bias)[:, None] * Vr).sum(axis=0)
        r = b - P @ x - reg_r * x - Vr.T @ (c * (Vr @ x))
        p = r.copy()
        for _ in range(K if steps == 0 else steps):
            r2 = r @ r
            if r2 <= 1e-20:
                break
            Ap = P @ p + reg_r * p + Vr.T @ (c * (Vr @ p))
            alpha = r2 / (p @ Ap)
            x += alpha * p
            r -= alpha * Ap
            iters[r_] += 1
            if r @ r <= 1e-20:
                break
            p = r + ((r @ r) / r2) * p
        out[r_] = x
    return out, iters


@pytest.mark.parametrize("steps", [1, 3, 5])
@pytest.mark.parametrize("loss", ["IALSPP", "ORIGINAL"])
def test_truncated_cg_half_step_matches_float64(steps, loss):
    """Pins the oracle's TRUNCATED CG - the reference default (max_cg_steps = 3,
    IALSLearningConfig.hpp:117) and the CG legs of bench.py - to 2e-5 per row; the
    converged-CG == Cholesky check above cannot see a wrong beta, a wrong warm start or a
    wrong step count.  Rows with stored entries start from a random row (the warm start,
    hpp:199); one row is empty (zeroed, hpp:207-210)."""
    rng = np.random.default_rng(11 + steps)
    X = sps.random(70, 50, density=0.15, format="csr", random_state=8, dtype=np.float64)
    X.data = rng.uniform(0.5, 3.0, X.nnz)
    X = X.tolil()
    X.rows[5], X.data[5] = [], []
    X = X.tocsr()
    K, alpha0, reg, nu = 12, 0.25, 0.05, 0.5
    mc = O.model_config(K, alpha0=alpha0, reg=reg, nu=nu, loss_type=loss)
    sc = O.solver_config(2, "CG", steps)
    user0 = (rng.standard_normal((70, K)) * 0.3).astype(np.float32)
    item0 = (rng.standard_normal((50, K)) * 0.3).astype(np.float32)
    P = O.ials_gramian(item0, alpha0, 1)
    got = O.ials_solver_step(user0, X, item0, P, mc, sc)
    bias = 0.0 if loss == "IALSPP" else alpha0
    want, iters = cg_half_step_float64(X.astype(np.float32), user0, item0, alpha0, reg, nu, bias, steps)
    assert np.all(got[5] == 0) and np.all(want[5] == 0)
    live = np.flatnonzero(np.diff(X.indptr) > 0)
    assert iters[live].min() == steps  # truncated: no row converged early
    err = np.linalg.norm(got[live] - want[live], axis=1) / np.linalg.norm(want[live], axis=1)
    assert err.max() < 2e-5, err.max()
    # a different step count must be visibly different at this bar (the pin has teeth)
    other_steps, _ = cg_half_step_float64(X.astype(np.float32), user0, item0, alpha0, reg, nu, bias, steps + 1)
    gap = np.linalg.norm(other_steps[live] - want[live], axis=1) / np.linalg.norm(want[live], axis=1)
    assert np.median(gap) > 1e-3


def test_cg_early_exit_and_zero_means_K_steps():
    """`max_cg_steps = 0` runs up to K iterations (hpp:232-234) and the absolute
    ||r||^2 <= 1e-20 exits stop a row that has converged (hpp:238, 258): a warm start AT the
    solution must come back unchanged to float rounding, and the K-step run must equal the
    float64 normal-equation solve."""
    rng = np.random.default_rng(4)
    X = sps.random(40, 30, density=0.2, format="csr", random_state=2, dtype=np.float64)
    X.data = rng.uniform(0.5, 2.0, X.nnz)
    K, alpha0, reg = 6, 0.2, 0.5
    mc = O.model_config(K, alpha0=alpha0, reg=reg, nu=0.0, loss_type="IALSPP")
    user0 = (rng.standard_normal((40, K)) * 0.3).astype(np.float32)
    item0 = (rng.standard_normal((30, K)) * 0.3).astype(np.float32)
    P = O.ials_gramian(item0, alpha0, 1)
    full = O.ials_solver_step(user0, X, item0, P, mc, O.solver_config(1, "CG", 0))
    want, iters = cg_half_step_float64(X.astype(np.float32), user0, item0, alpha0, reg, 0.0, 0.0, 0)
    live = np.flatnonzero(np.diff(X.indptr) > 0)
    assert iters[live].max() <= K
    err = np.linalg.norm(full[live] - want[live], axis=1) / np.linalg.norm(want[live], axis=1)
    assert err.max() < 2e-5, err.max()
    chol = O.ials_solver_step(user0, X, item0, P, mc, O.solver_config(1, "CHOLESKY", 0))
    np.testing.assert_allclose(full[live], chol[live], rtol=2e-4, atol=2e-6)
    # warm start at the converged point: further steps stay there (per row, float32 rounding)
    again = O.ials_solver_step(full, X, item0, P, mc, O.solver_config(1, "CG", 3))
    err = np.linalg.norm(again[live] - full[live], axis=1) / np.linalg.norm(full[live], axis=1)
    assert err.max() < 2e-5, err.max()
    # the exit at ||r||^2 <= 1e-20 BEFORE the first step (hpp:238): a row whose only item has a
    # zero factor and that starts at zero has b = 0, r = 0; without the exit the step would
    # divide 0 / 0 and the denominator test (hpp:250-254) would throw
    item1 = item0.copy()
    item1[7] = 0.0
    X1 = sps.csr_matrix((np.ones(1), ([0], [7])), shape=(3, 30))
    P1 = O.ials_gramian(item1, alpha0, 1)
    z = O.ials_solver_step(np.zeros((3, K), dtype=np.float32), X1, item1, P1, mc, O.solver_config(1, "CG", 3))
    assert np.all(z == 0)


def test_cg_fold_in_starts_from_zero(X_small):
    """transform_user (X_to_vector, hpp:122-141) starts CG from zeros, not from the trained row
    (hpp:132): equal to the float64 CG from a zero start."""
    t, mc, _ = train(X_small, 4, 3, "CG", alpha0=0.3, reg=0.2, nu=0.0)
    got = t.transform_user(X_small, O.solver_config(1, "CG", 2))
    want, _ = cg_half_step_float64(X_small.astype(np.float32), np.zeros_like(t.user), t.item,
                                   0.3, 0.2, 0.0, 0.0, 2, warm=False)
    live = np.flatnonzero(np.diff(X_small.indptr) > 0)
    err = np.linalg.norm(got[live] - want[live], axis=1) / np.linalg.norm(want[live], axis=1)
    assert err.max() < 2e-5, err.max()
    assert np.all(got[3] == 0)


@pytest.mark.parametrize("kind,steps", [("CG", 1), ("CG", 3), ("CHOLESKY", 0)])
@pytest.mark.parametrize("loss", ["IALSPP", "ORIGINAL"])
def test_float64_arbiter_build_is_pinned(kind, steps, loss):
    """``liboracle_f64.so`` (the oracle's sources with Real = double: the arbiter of the GPU
    parity tests) against the independent numpy float64 restatements above, at 1e-10 (CG) / 1e-7
    (Cholesky) per row - orders below the distances it arbitrates - and the float32 oracle against it at the
    2e-5 / 2e-4 of its own pins: one algorithm, three arithmetic widths."""
    rng = np.random.default_rng(5 + steps)
    X = sps.random(70, 50, density=0.15, format="csr", random_state=8, dtype=np.float64)
    X.data = rng.uniform(0.5, 3.0, X.nnz)
    X = X.tolil()
    X.rows[5], X.data[5] = [], []
    X = X.tocsr()
    K, alpha0, reg, nu = 12, 0.25, 0.05, 0.5
    mc = O.model_config(K, alpha0=alpha0, reg=reg, nu=nu, loss_type=loss)
    sc = O.solver_config(2, kind, max(steps, 1))
    user0 = (rng.standard_normal((70, K)) * 0.3).astype(np.float32)
    item0 = (rng.standard_normal((50, K)) * 0.3).astype(np.float32)
    bias = 0.0 if loss == "IALSPP" else alpha0
    Xf = X.astype(np.float32)
    live = np.flatnonzero(np.diff(X.indptr) > 0)
    got64 = O.ials_solver_step_f64(user0, Xf, item0, None, mc, sc, 2)
    assert got64.dtype == np.float64
    if kind == "CG":
        want, _ = cg_half_step_float64(Xf, user0, item0, alpha0, reg, nu, bias, steps)
    else:  # the normal equations of test_half_step_vs_normal_equations, float regulariser
        i64 = item0.astype(np.float64)
        P = np.float64(np.float32(alpha0)) * i64.T @ i64
        want = np.zeros((70, K))
        for r in live:
            sl = slice(X.indptr[r], X.indptr[r + 1])
            V, c = i64[X.indices[sl]], Xf.data[sl].astype(np.float64)
            regr = np.float32(reg) * np.power(np.float32(alpha0) * np.float32(50) + np.float32(sl.stop - sl.start), np.float32(nu))
            A = P + (V * c[:, None]).T @ V + float(regr) * np.eye(K)
            want[r] = np.linalg.solve(A, ((c + np.float64(np.float32(bias)))[:, None] * V).sum(axis=0))
    err = np.linalg.norm(got64[live] - want[live], axis=1) / np.linalg.norm(want[live], axis=1)
    # (Cholesky: numpy's float32 power and libm's powf may differ in the last bit of the
    # regulariser of hpp:117-120, 6e-8 of it: 1e-7 there, 1e-10 where the restatement takes the float)
    assert err.max() < (1e-10 if kind == "CG" else 1e-7), err.max()
    if kind == "CG":
        assert np.all(got64[5] == 0)
    got32 = O.ials_solver_step(user0, Xf, item0, O.ials_gramian(item0, alpha0, 1), mc, sc)
    e32 = np.linalg.norm(got32[live] - got64[live], axis=1) / np.linalg.norm(got64[live], axis=1)
    assert e32.max() < (2e-5 if kind == "CG" else 2e-4), e32.max()
