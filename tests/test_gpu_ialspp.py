"""GPU parity tests of the iALS++ / iCD subspace solver (SolverType.IALSPP) against the CPU
oracle (oracle/ials_oracle.cpp: step_ialspp / step_dimrange / prediction) and the
reference's own convergence check (tests/recommenders/test_ials.py:573-599).
Tolerance: 1e-4 relative on the factor matrices for the same factors in.
"""
import numpy as np
import pytest
import scipy.sparse as sps

import oracle as O
from conftest import random_csr, row_rel_err
from irspack_amd.recommenders._ials_core import (IALSModelConfigBuilder, IALSSolverConfigBuilder,
                                                  IALSTrainer, LossType, SolverType)

pytestmark = pytest.mark.gpu

RTOL = 1e-4


rel_err = row_rel_err  # per row: ||a_r - b_r|| / ||b_r||, worst row


def build(K, alpha0=0.1, reg=1e-2, nu=1.0, loss="IALSPP"):
    mc = (IALSModelConfigBuilder().set_K(K).set_alpha0(alpha0).set_reg(reg).set_nu(nu)
          .set_init_stdev(0.1).set_random_seed(42).set_loss_type(LossType[loss]).build())
    omc = O.model_config(K, alpha0=alpha0, reg=reg, nu=nu, init_stdev=0.1, random_seed=42,
                         loss_type=loss)
    return mc, omc


def solvers(sub, iters):
    sc = (IALSSolverConfigBuilder().set_n_threads(1).set_solver_type(SolverType.IALSPP)
          .set_ialspp_subspace_dimension(sub).set_ialspp_iteration(iters).build())
    return sc, O.solver_config(1, "IALSPP", 3, ialspp_subspace_dimension=sub,
                               ialspp_iteration=iters)


_KSUB = [(4, 1), (4, 3), (16, 16), (16, 5), (20, 8), (32, 32), (48, 16),
         (64, 64), (64, 16), (64, 24), (64, 33), (40, 64), (128, 64), (130, 32),
         # 64-dim blocks of longer rows: the chained passes (2, 3 and 4 blocks, a 2-dim and
         # a 36-dim last block)
         (130, 64), (192, 64), (228, 64), (256, 64)]
# IRSPACK_AMD_IALSPP_DIRECT only matters when one block covers the row (sub >= K): the
# "0" form is generated for those shapes only (no skipped combinations)
_KSUB_DIRECT = [(K, sub, "1") for K, sub in _KSUB] + [(K, sub, "0") for K, sub in _KSUB if sub >= K]


@pytest.mark.parametrize("K,sub,direct", _KSUB_DIRECT)
@pytest.mark.parametrize("loss", ["IALSPP", "ORIGINAL"])
def test_epoch_matches_oracle(K, sub, loss, direct, monkeypatch):
    """Every epoch starts from the oracle's factors (teacher forcing), so the comparison is one
    iALS++ epoch (user half then item half) for the same input.  ``direct``: when one block
    covers every dimension (sub >= K) the library computes the block Newton step as the exact
    solve it is (``IRSPACK_AMD_IALSPP_DIRECT``, default on); both forms must match the oracle's
    two-step restatement of hpp:436-502."""
    monkeypatch.setenv("IRSPACK_AMD_IALSPP_DIRECT", direct)
    X = random_csr(150, 110, 0.1, 3, empty_rows=(7, 40))
    mc, omc = build(K, loss=loss)
    sc, osc = solvers(sub, 2)
    t = IALSTrainer(mc, X)
    o = O.IALSTrainer(omc, X)
    np.testing.assert_array_equal(t.user, o.user)
    for _ in range(2):
        t.user, t.item = o.user, o.item
        t.step(sc)
        o.step(osc)
        assert rel_err(t.user, o.user) < RTOL
        assert rel_err(t.item, o.item) < 10 * RTOL  # the item half sees the user half's rounding


def test_long_rows_and_explicit_weights():
    """Rows far longer than one 16-entry pipeline group, non-binary confidences."""
    rng = np.random.default_rng(4)
    X = random_csr(40, 900, 0.6, 9)
    X.data[:] = rng.uniform(0.5, 3.0, size=X.nnz).astype(np.float32)
    mc, omc = build(64, alpha0=0.05, reg=1e-2)
    sc, osc = solvers(32, 1)
    t = IALSTrainer(mc, X)
    o = O.IALSTrainer(omc, X)
    t.step(sc)
    o.step(osc)
    assert rel_err(t.user, o.user) < RTOL
    assert rel_err(t.item, o.item) < 10 * RTOL


@pytest.mark.parametrize("sub", [1, 2, 3, 4])
def test_overfit_ialspp(X_small, sub):
    # tests/recommenders/test_ials.py:573-599: with a tiny regulariser the model reproduces X
    mc = (IALSModelConfigBuilder().set_K(4).set_alpha0(100).set_reg(1.0).set_nu(0)
          .set_loss_type(LossType.ORIGINAL).build())
    sc = (IALSSolverConfigBuilder().set_solver_type(SolverType.IALSPP)
          .set_ialspp_subspace_dimension(sub).build())
    t = IALSTrainer(mc, X_small)
    for _ in range(300):
        t.step(sc)
    B = sps.csr_matrix(X_small).toarray()
    B[B > 0] = 1
    np.testing.assert_allclose(t.user @ t.item.T, B, rtol=1e-2, atol=1e-2)


def test_transform_with_ialspp(X_small):
    """Fold-in with the subspace solver starts from zero (hpp:132) and matches the oracle."""
    mc, omc = build(16, alpha0=0.1, reg=1e-1)
    t = IALSTrainer(mc, X_small)
    o = O.IALSTrainer(omc, X_small)
    sc, osc = solvers(8, 3)
    np.testing.assert_array_equal(t.item, o.item)
    u, ou = t.transform_user(X_small, sc), o.transform_user(X_small, osc)
    assert rel_err(u, ou) < RTOL


@pytest.mark.parametrize("K,sub", [(128, 100), (130, 65), (200, 128), (256, 128), (256, 200), (256, 255),
                                   (300, 64), (300, 128), (320, 150), (300, 7), (270, 1), (300, 300), (300, 1000)])
@pytest.mark.parametrize("loss", ["IALSPP", "ORIGINAL"])
def test_wide_blocks_and_large_k_match_oracle(K, sub, loss):
    """ialspp_subspace_dimension has no limit in the reference (IALSLearningConfig.hpp:119,
    139-141; _step_dimrange loops over any block width, hpp:516-535): blocks wider than 64 dims
    and every block width at K > 256 run on the general-size kernels (ials_gk_kernels.hpp:
    scratch systems, prediction cache corrected after each block); sub >= K is the direct
    solve; sub = 1 the iCD branch.  Two sweeps per half-step, teacher-forced epochs."""
    X = random_csr(150, 110, 0.1, 3, empty_rows=(7, 40))
    mc, omc = build(K, loss=loss)
    sc, osc = solvers(sub, 2)
    t = IALSTrainer(mc, X)
    o = O.IALSTrainer(omc, X)
    np.testing.assert_array_equal(t.user, o.user)
    for _ in range(2):
        t.user, t.item = o.user, o.item
        t.step(sc)
        o.step(osc)
        assert rel_err(t.user, o.user) < RTOL
        assert rel_err(t.item, o.item) < 10 * RTOL  # the item half sees the user half's rounding


def test_wide_blocks_long_rows_and_weights(monkeypatch):
    """128-dim blocks at K = 256 over rows of thousands of weighted entries, in several scratch
    batches."""
    monkeypatch.setenv("IRSPACK_AMD_GK_SCRATCH_MB", "1")
    rng = np.random.default_rng(8)
    X = random_csr(24, 6000, 0.5, 13)
    X.data[:] = rng.uniform(0.5, 2.0, size=X.nnz).astype(np.float32)
    mc, omc = build(256, alpha0=0.02, reg=1e-2)
    sc, osc = solvers(128, 1)
    t = IALSTrainer(mc, X)
    o = O.IALSTrainer(omc, X)
    t.partial_gramian_async(0)
    t.finish_gramian_async(0)
    t.half_step_async(0, sc)
    t.synchronize()
    o.step(osc)
    assert rel_err(t.user, o.user) < RTOL


@pytest.mark.parametrize("K,sub", [(64, 64), (32, 16), (20, 7)])
def test_rows_above_the_workgroup_threshold(K, sub):
    """Rows with more than 2048 stored entries run on ialspp_long_kernel (8 waves per row)."""
    rng = np.random.default_rng(8)
    X = random_csr(24, 6000, 0.5, 13)
    X.data[:] = rng.uniform(0.5, 2.0, size=X.nnz).astype(np.float32)
    assert np.diff(X.indptr).max() > 2048
    mc, omc = build(K, alpha0=0.02, reg=1e-2)
    sc, osc = solvers(sub, 2)
    t = IALSTrainer(mc, X)
    o = O.IALSTrainer(omc, X)
    t.step(sc)
    o.step(osc)
    assert rel_err(t.user, o.user) < RTOL
    assert rel_err(t.item, o.item) < 10 * RTOL


@pytest.mark.parametrize("K", [128, 200])
def test_chained_passes_match_the_three_pass_form(K, monkeypatch):
    """64-dim blocks: the merged passes (coalesced prediction pass, cache correction inside the
    next block's rank update, packed P blocks; ``IRSPACK_AMD_IALSPP_CHAIN``) against the plain
    three-pass form of the same kernel, rows above the workgroup threshold included."""
    rng = np.random.default_rng(5)
    X = random_csr(300, 260, 0.08, 11, empty_rows=(3,))
    X = sps.vstack([X, sps.csr_matrix((rng.random((2, 260)) < 0.95).astype(np.float32))]).tocsr()
    Xl = sps.hstack([X, sps.csr_matrix((rng.random((302, 2400)) < 0.9).astype(np.float32))]).tocsr()
    mc, _ = build(K)
    sc, _ = solvers(64, 2)
    out = {}
    for flag in ("1", "0"):
        monkeypatch.setenv("IRSPACK_AMD_IALSPP_CHAIN", flag)
        t = IALSTrainer(mc, Xl)
        for _ in range(2):
            t.step(sc)
        out[flag] = (t.user, t.item)
    assert rel_err(out["1"][0], out["0"][0]) < RTOL
    assert rel_err(out["1"][1], out["0"][1]) < RTOL
