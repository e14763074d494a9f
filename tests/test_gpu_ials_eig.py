"""The eigenbasis short-row path (irspack_amd/csrc/ials_eig_kernels.hpp): the float64 Jacobi
eigen-decomposition kernel against numpy, and the Cholesky (low-rank form) / CG (diagonal P) solves
of rows with at most 32 stored entries against the CPU oracle - the path configs[3] (10 M users
with ~10 entries each, K = 128) takes; the full-size checks are in test_gpu_fullsize.py.
Reference: IALSTrainer.hpp:273-331 (step_cholesky), :170-271 (step_cg)."""
import ctypes as C

import numpy as np
import pytest
import scipy.sparse as sps

import oracle as O
from conftest import row_rel_err
from irspack_amd import _lib
from irspack_amd.recommenders._ials_core import (IALSModelConfigBuilder, IALSSolverConfigBuilder,
                                                  IALSTrainer, LossType, SolverType)

pytestmark = pytest.mark.gpu
RTOL = 1e-4


@pytest.mark.parametrize("K", [5, 33, 64, 65, 100, 128])
def test_jacobi_eigen_kernel_matches_numpy(K):
    rng = np.random.default_rng(K)
    F = rng.standard_normal((500, K)) * (0.5 + rng.random(K))
    if K > 40:
        F[:, 3] = F[:, 7]  # a rank deficiency (a zero eigenvalue)
    P = (0.1 * F.T @ F).astype(np.float32)
    P = ((P + P.T) / 2).astype(np.float32)
    Q = np.zeros((K, K), dtype=np.float32)
    lam = np.zeros(K, dtype=np.float32)
    st = np.zeros(3, dtype=np.float32)
    f = _lib.lib().irs_ials_eigen_debug
    f.restype = C.c_int32
    # cold start, then warm-started from the eigenvectors of a nearby matrix (the Gramian of the
    # previous epoch): same answer, fewer sweeps
    P_prev = (P + 0.02 * np.abs(P).max() * np.diag(rng.random(K))).astype(np.float32)
    sweeps = []
    for prev in (None, P_prev):
        rc = f(P.ctypes.data_as(C.c_void_p), C.c_int64(K), C.c_int32(0), Q.ctypes.data_as(C.c_void_p),
               lam.ctypes.data_as(C.c_void_p), st.ctypes.data_as(C.c_void_p),
               prev.ctypes.data_as(C.c_void_p) if prev is not None else None)
        assert rc == 0
        sweeps.append(int(st[2]))
        check_decomposition(P, Q, lam, st, K)
    assert sweeps[1] <= sweeps[0]


def check_decomposition(P, Q, lam, st, K):
    Q64, P64 = Q.astype(np.float64), P.astype(np.float64)
    scale = np.abs(np.linalg.eigvalsh(P64)).max()
    np.testing.assert_allclose(np.sort(lam), np.linalg.eigvalsh(P64), atol=2e-6 * scale)
    np.testing.assert_allclose(Q64 @ Q64.T, np.eye(K), atol=5e-6)                 # orthonormal rows
    np.testing.assert_allclose(Q64.T @ np.diag(lam.astype(np.float64)) @ Q64, P64, atol=5e-6 * scale)
    assert st[0] == lam.max() and st[1] == lam.min() and 1 <= st[2] <= 16


def short_row_matrix(n_users, n_items, seed, weights):
    rng = np.random.default_rng(seed)
    deg = rng.integers(0, 33, size=n_users)
    deg[:10] = [0, 1, 16, 17, 32, 2, 8, 9, 7, 31]  # (the class boundaries of the short-row kernels: 8 | 9, 16 | 17, 32)
    rows = np.repeat(np.arange(n_users), deg)
    cols = np.concatenate([rng.choice(n_items, size=d, replace=False) for d in deg])
    vals = (rng.uniform(0.25, 3.0, size=rows.shape[0]) if weights else np.ones(rows.shape[0])).astype(np.float32)
    return sps.csr_matrix((vals, (rows, cols)), shape=(n_users, n_items), dtype=np.float32)


@pytest.mark.parametrize("kind", ["CHOLESKY", "CG"])
@pytest.mark.parametrize("K,weights,loss", [(128, False, "IALSPP"), (128, True, "ORIGINAL"), (100, True, "IALSPP"),
                                            (64, False, "ORIGINAL"), (40, True, "IALSPP")])
def test_short_rows_in_the_eigenbasis_match_oracle(K, weights, loss, kind, monkeypatch):
    """Enough short rows that the eigenbasis path is taken on the user side (the switch-off
    environment variable gives the same rows through the dense kernels for comparison): every row
    within 1e-4 of the oracle, Cholesky and three-step CG, unit and weighted confidences, both
    losses; the second half-step starts from the oracle's factors (warm start, hpp:199)."""
    n_users = 160_000 if K > 64 else 1_200_000  # (the path needs n_short * KP^3 >= 3e11)
    X = short_row_matrix(n_users, 300, 3, weights)
    mc = (IALSModelConfigBuilder().set_K(K).set_alpha0(0.1).set_reg(2e-2).set_nu(1.0).set_init_stdev(0.1)
          .set_random_seed(42).set_loss_type(LossType[loss]).build())
    omc = O.model_config(K, alpha0=0.1, reg=2e-2, nu=1.0, init_stdev=0.1, random_seed=42, loss_type=loss)
    sc = (IALSSolverConfigBuilder().set_n_threads(1).set_solver_type(SolverType[kind]).set_max_cg_steps(3).build())
    osc = O.solver_config(8, kind, 3)
    t = IALSTrainer(mc, X)
    user0, item0 = t.user, t.item
    P = O.ials_gramian(item0, 0.1, 8)
    want = O.ials_solver_step(user0, X, item0, P, omc, osc)
    t.partial_gramian_async(0)
    t.finish_gramian_async(0)
    t.half_step_async(0, sc)
    t.synchronize()
    got = t.user
    assert t.last_half_step_used_eigenbasis()
    assert row_rel_err(got, want) < RTOL
    assert np.all(got[0] == 0)  # the empty row
    # a second, warm-started half-step from the oracle's result
    t.user = want
    want2 = O.ials_solver_step(want, X, item0, P, omc, osc)
    t.half_step_async(0, sc)
    t.synchronize()
    assert row_rel_err(t.user, want2) < RTOL
    # the same rows through the dense kernels
    monkeypatch.setenv("IRSPACK_AMD_IALS_EIG", "0")
    d = IALSTrainer(mc, X)
    d.partial_gramian_async(0)
    d.finish_gramian_async(0)
    d.half_step_async(0, sc)
    d.synchronize()
    assert not d.last_half_step_used_eigenbasis()
    assert row_rel_err(d.user, want) < RTOL


@pytest.mark.parametrize("kind", ["CHOLESKY", "CG"])
def test_eigenbasis_passes_alternate_between_two_streams(kind, monkeypatch):
    """The short rows are taken in passes (2^20 rows by default, 30 000 here: six passes of mixed
    17..32-entry and <= 16-entry rows) that alternate between two streams with their own scratch."""
    monkeypatch.setenv("IRSPACK_AMD_IALS_EIG_PASS_ROWS", "30000")
    K = 128
    X = short_row_matrix(160_000, 300, 5, True)
    mc = (IALSModelConfigBuilder().set_K(K).set_alpha0(0.1).set_reg(2e-2).set_nu(1.0).set_init_stdev(0.1)
          .set_random_seed(7).set_loss_type(LossType.IALSPP).build())
    omc = O.model_config(K, alpha0=0.1, reg=2e-2, nu=1.0, init_stdev=0.1, random_seed=7, loss_type="IALSPP")
    sc = (IALSSolverConfigBuilder().set_n_threads(1).set_solver_type(SolverType[kind]).set_max_cg_steps(3).build())
    osc = O.solver_config(8, kind, 3)
    t = IALSTrainer(mc, X)
    user0, item0 = t.user, t.item
    want = O.ials_solver_step(user0, X, item0, O.ials_gramian(item0, 0.1, 8), omc, osc)
    for _ in range(2):  # (the second call reuses streams and scratch)
        t.user = user0
        t.partial_gramian_async(0)
        t.finish_gramian_async(0)
        t.half_step_async(0, sc)
        t.synchronize()
        assert t.last_half_step_used_eigenbasis()
        assert row_rel_err(t.user, want) < RTOL


def test_ill_conditioned_gramian_falls_back():
    """lambda_max + reg > 1e4 (lambda_min + reg) - here 100 items at K = 128: P has 28 zero
    eigenvalues and reg_r = 1e-7 - the eigenbasis path declines, the dense kernels run"""
    X = short_row_matrix(160_000, 100, 5, False)
    mc = (IALSModelConfigBuilder().set_K(128).set_alpha0(10.0).set_reg(1e-7).set_nu(0.0).set_init_stdev(0.1)
          .set_random_seed(1).build())
    sc = IALSSolverConfigBuilder().set_n_threads(1).set_solver_type(SolverType.CG).set_max_cg_steps(3).build()
    t = IALSTrainer(mc, X)
    t.partial_gramian_async(0)
    t.finish_gramian_async(0)
    t.half_step_async(0, sc)
    t.synchronize()
    assert not t.last_half_step_used_eigenbasis()


@pytest.mark.parametrize("K", [128])
def test_ialspp_with_one_block_keeps_the_references_form(K):
    """iALS++ whose one block covers every dimension.  Rounds 2-4 computed it as the direct solve
    (hpp:436-502 is a Newton step of a quadratic) through the Cholesky kernels, the eigenbasis path
    included; round 5 found that form kappa 2^-24 away from the reference's own arithmetic where the
    reference's has no such error, so it is gone: K <= 64 runs the reference's
    gradient form on the tuned kernel, a block wider than 64 dims (this case: K = 128 with
    ialspp_subspace_dimension = 128) the general block kernels - never the eigenbasis Cholesky.
    Equal to the oracle's block form within 1e-4, empty rows included."""
    X = short_row_matrix(160_000 if K > 64 else 1_200_000, 300, 9, True)
    mc = (IALSModelConfigBuilder().set_K(K).set_alpha0(0.1).set_reg(2e-2).set_nu(1.0).set_init_stdev(0.1)
          .set_random_seed(42).build())
    omc = O.model_config(K, alpha0=0.1, reg=2e-2, nu=1.0, init_stdev=0.1, random_seed=42)
    sc = (IALSSolverConfigBuilder().set_n_threads(1).set_solver_type(SolverType.IALSPP)
          .set_ialspp_subspace_dimension(128).set_ialspp_iteration(1).build())
    osc = O.solver_config(8, "IALSPP", 3, ialspp_subspace_dimension=128, ialspp_iteration=1)
    t = IALSTrainer(mc, X)
    user0, item0 = t.user, t.item
    P = O.ials_gramian(item0, 0.1, 8)
    want = O.ials_solver_step(user0, X, item0, P, omc, osc)
    t.partial_gramian_async(0)
    t.finish_gramian_async(0)
    t.half_step_async(0, sc)
    t.synchronize()
    assert not t.last_half_step_used_eigenbasis()
    # (an empty row: both forms leave rounding noise of ~1e-9 where the exact answer is 0 - compared in
    # absolute terms)
    live = np.diff(X.indptr) > 0
    got = t.user
    assert row_rel_err(got[live], want[live]) < RTOL
    assert np.abs(got[~live]).max() < 1e-6 and np.abs(want[~live]).max() < 1e-6
