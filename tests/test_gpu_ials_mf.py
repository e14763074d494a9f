"""GPU parity tests of the matrix-free CG kernels at 128 < K <= 256
(irspack_amd/csrc/ials_mf_kernels.hpp; step_cg, hpp:199-264) against the CPU oracle: every row
class - empty rows, the resident classes of <= 40 / 80 / 160 / 320 stored entries (four, two
or one row per workgroup) at their boundaries, level-synchronous rows of one chunk, two chunks and many - both padded widths
(KP = 192, 256), both losses, weighted and unit confidences, warm start over two epochs,
``max_cg_steps`` of 1, 3, 7 and 0 (= K steps, hpp:232-234), fold-in from zero, the singular-system
error, and the explicit-matrix kernels the path replaced (``IRSPACK_AMD_IALS_MF=0``) as a second
reference.  Bar: 1e-4 per row."""
import numpy as np
import pytest
import scipy.sparse as sps

import oracle as O
from conftest import assert_float64_bar, row_rel_err
from irspack_amd.recommenders._ials_core import IALSTrainer
from test_gpu_ials import build, solver

pytestmark = pytest.mark.gpu
RTOL = 1e-4

# every class boundary of mf_cg_rows_kernel (40 | 80 | 160 | 320), the chunk boundary of the
# level-synchronous rows (1024), empty rows, and row counts that leave the last workgroup of the
# two- and four-row classes partly empty
LENGTHS = [0, 1, 2, 31, 39, 40, 41, 79, 80, 81, 159, 160, 161, 319, 320, 321, 500, 1023, 1024,
           1025, 2047, 2050, 3333, 0, 17, 40, 64, 64, 100, 150, 260, 5000]


def matrix(n_other, lengths, seed, unit=False):
    rng = np.random.default_rng(seed)
    rows = [np.sort(rng.choice(n_other, size=d, replace=False)) for d in lengths]
    indptr = np.concatenate([[0], np.cumsum([len(r) for r in rows])])
    vals = (np.ones(indptr[-1]) if unit else rng.uniform(0.5, 2.0, size=indptr[-1])).astype(np.float32)
    return sps.csr_matrix((vals, np.concatenate(rows).astype(np.int32), indptr),
                          shape=(len(lengths), n_other))


def half_step(t, side, sc):
    t.partial_gramian_async(side)
    t.finish_gramian_async(side)
    t.half_step_async(side, sc)
    t.synchronize()


def check_side(t, side, X, tgt0, oth0, omc, osc, sc, what):
    """One half-step of `side` from (tgt0, oth0) on the GPU, by the float32 oracle and by the
    float64 arbiter; every row is measured against float64.  Returns the oracle's rows."""
    half_step(t, side, sc)
    got = t.user if side == 0 else t.item
    want32 = O.ials_solver_step(tgt0, X, oth0, O.ials_gramian(oth0, omc.alpha0, 2), omc, osc)
    want64 = O.ials_solver_step_f64(tgt0, X, oth0, None, omc, osc, 2)
    assert_float64_bar(got, want32, want64, what, test="mf_cg")
    return want32


@pytest.mark.parametrize("K", [130, 192, 200, 256])
@pytest.mark.parametrize("loss,unit", [("IALSPP", False), ("ORIGINAL", True)])
def test_every_row_class_matches_oracle(K, loss, unit):
    X = matrix(6000, LENGTHS, 7, unit)
    Xt = sps.csr_matrix(X.T)
    mc, omc = build(K, alpha0=0.05, reg=1e-2, loss=loss)
    sc, osc = solver("CG", steps=3)
    t, o = IALSTrainer(mc, X), O.IALSTrainer(omc, X)
    user, item = o.user, o.item
    for ep in range(2):  # the second epoch starts from the first one's rows (warm start, hpp:199)
        t.user, t.item = user, item
        user = check_side(t, 0, X, user, item, omc, osc, sc, f"K={K} {loss} unit={unit} users ep{ep}")
        t.user = user
        # the item side: 6000 short rows (0 .. ~15 entries)
        item = check_side(t, 1, Xt, item, user, omc, osc, sc, f"K={K} {loss} unit={unit} items ep{ep}")
    # fold-in: zero start (hpp:132), rows of every class again
    t.user, t.item = user, item
    o.user, o.item = user, item
    got, want = t.transform_user(X, sc), o.transform_user(X, osc)
    want64 = O.ials_solver_step_f64(np.zeros_like(user), X, item, None, omc, osc, 2)
    assert_float64_bar(got, want, want64, f"K={K} {loss} unit={unit} fold-in", test="mf_cg")
    assert t.last_half_step_used_eigenbasis() is False


def fresh_user_half(K, X, steps, what, alpha0=0.1, reg=1e-2):
    """User half-step from the shared initial factors: GPU vs float32 oracle vs float64."""
    mc, omc = build(K, alpha0=alpha0, reg=reg)
    sc, osc = solver("CG", steps=steps)
    t, o = IALSTrainer(mc, X), O.IALSTrainer(omc, X)
    user0, item0 = o.user, o.item
    half_step(t, 0, sc)
    want32 = O.ials_solver_step(user0, X, item0, O.ials_gramian(item0, omc.alpha0, 2), omc, osc)
    want64 = O.ials_solver_step_f64(user0, X, item0, None, omc, osc, 2)
    assert_float64_bar(t.user, want32, want64, what, test="mf_cg")
    return t, mc


@pytest.mark.parametrize("steps", [1, 7, 0])
def test_step_counts_and_k_steps(steps):
    # max_cg_steps == 0 means K steps (hpp:232-234): the level-synchronous rows take 2 K + 3
    # launches, rows leave the loop through the 1e-20 exits on the way (hpp:238, 258)
    X = matrix(3000, [0, 5, 40, 100, 200, 330, 1500, 2500], 11)
    # (converged CG: the float32 implementations leave the loop on different steps; both are at
    # rounding distance from the float64 iterate - achieved 3e-7 against the bar of 1e-4)
    t, mc = fresh_user_half(140, X, steps, f"K=140 steps={steps}", reg=5e-2)
    if steps == 0:  # and K steps reach the Cholesky solution (test_ials.py:627-661)
        sc2, _ = solver("CHOLESKY")
        t2 = IALSTrainer(mc, X)
        half_step(t2, 0, sc2)
        assert row_rel_err(t.user, t2.user) < 1e-3


def test_matches_the_explicit_matrix_kernels(monkeypatch):
    X = matrix(4000, [0, 3, 33, 97, 200, 321, 1100, 2600, 64, 250], 5)
    mc, _ = build(256, alpha0=0.1, reg=1e-2)
    sc, _ = solver("CG", steps=3)
    t = IALSTrainer(mc, X)
    half_step(t, 0, sc)
    monkeypatch.setenv("IRSPACK_AMD_IALS_MF", "0")
    t2 = IALSTrainer(mc, X)
    half_step(t2, 0, sc)
    assert row_rel_err(t.user, t2.user) < RTOL


def test_many_level_synchronous_rows_and_chunks():
    """A side whose rows are ALL level-synchronous (no resident launch: one stream) and a row of
    20 chunks next to 300 one-chunk rows; K = 256."""
    rng = np.random.default_rng(1)
    lengths = [20000] + list(rng.integers(321, 1200, size=300))
    X = matrix(30000, lengths, 2)
    fresh_user_half(256, X, 3, "K=256 all rows level-synchronous", alpha0=0.02)


def test_singular_system_is_reported():
    # negative confidences with alpha0 = reg = 0 make A = sum c v v^T negative semi-definite:
    # p . A p <= 0 on the first step, which the reference throws on (hpp:250-254); one resident
    # and one level-synchronous row
    for lengths in ([40, 7], [400, 7]):
        X = matrix(500, lengths, 3)
        X.data[: lengths[0]] = -1.0
        mc, _ = build(160, alpha0=0.0, reg=0.0, loss="ORIGINAL")
        sc, _ = solver("CG", steps=3)
        t = IALSTrainer(mc, X)
        with pytest.raises(RuntimeError, match="singular"):
            half_step(t, 0, sc)
            t.user
