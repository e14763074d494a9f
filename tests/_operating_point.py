"""Shared machinery of the operating-point parity tests (tests/test_gpu_operating_point.py) and of
scripts/operating_point_probe.py: one half-step per side of the HIP path, of the float32 oracle and of
the float64 arbiter from IDENTICAL factors, at the hyper-parameters the reference's users actually run
(`IALSRecommender.__init__` defaults and the corners of `default_tune_range`,
/root/reference/src/irspack/recommenders/ials.py:357-379).

What is measured per row r (float64 arithmetic, A_r = P + sum_j c_j v_j v_j^T + reg_r I, the system
of IALSTrainer.hpp:273-331 / :170-271):
  * factor distance  ||x_gpu - x_64|| / ||x_64||                       (`fac`)
  * score distance on the row's OWN items  ||V_r x_gpu - V_r x_64|| / ||V_r x_64||  (`sco`) - what a
    recommender built on the factors observes; insensitive to the directions a rank-deficient A_r
    (fewer entries than K, alpha0 = 0) leaves to the ridge alone
  * Cholesky only: the normwise backward error ||A_r x - b_r|| / (||A_r|| ||x|| + ||b_r||) of the GPU's
    row and of the oracle's row (`res`), which a backward-stable factorisation keeps near K 2^-24
    whatever the conditioning
and kappa_r, the 2-norm condition number of A_r (dense eigvalsh; only where asked for).
"""
import os

import numpy as np
import scipy.sparse as sps

import oracle as O
from irspack_amd.recommenders._ials_core import (IALSModelConfigBuilder, IALSSolverConfigBuilder,
                                                  IALSTrainer, LossType, SolverType)

CORES = os.cpu_count() or 1


def configs(K, kind, alpha0, reg, nu=1.0, loss="IALSPP", steps=3, subspace=64, init=0.1, seed=42):
    mc = (IALSModelConfigBuilder().set_K(K).set_alpha0(alpha0).set_reg(reg).set_nu(nu)
          .set_init_stdev(init).set_random_seed(seed).set_loss_type(LossType[loss]).build())
    sc = (IALSSolverConfigBuilder().set_n_threads(1).set_solver_type(SolverType[kind])
          .set_max_cg_steps(steps).set_ialspp_subspace_dimension(subspace).build())
    omc = O.model_config(K, alpha0=alpha0, reg=reg, nu=nu, init_stdev=init, random_seed=seed,
                         loss_type=loss)
    osc = O.solver_config(CORES, kind, steps, subspace, 1)
    return mc, sc, omc, osc


def raises(fn):
    """(exception type name, message) of `fn()`, or None."""
    try:
        fn()
    except (RuntimeError, ValueError) as exc:
        return type(exc).__name__, str(exc)
    return None


def row_reg(Xs, n_other, alpha0, reg, nu):
    """compute_reg (hpp:117-120) evaluated the way the reference does: float arithmetic."""
    nnz = np.diff(Xs.indptr).astype(np.float32)
    base = np.float32(alpha0) * np.float32(n_other) + nnz
    return (np.float32(reg) * np.power(base, np.float32(nu), dtype=np.float32)).astype(np.float64)


def apply_A(Xs, other64, P64, regs, x):
    """A_r x_r for every row, without forming A_r: P x + sum_j c_j v_j (v_j . x) + reg_r x."""
    x = np.asarray(x, np.float64)
    rows = np.repeat(np.arange(Xs.shape[0]), np.diff(Xs.indptr))
    V = other64[Xs.indices]
    dots = np.einsum("ij,ij->i", V, x[rows]) * Xs.data.astype(np.float64)
    out = x @ P64 + regs[:, None] * x
    _segment_add(out, Xs, V * dots[:, None])
    return out


def _segment_add(out, Xs, contrib):
    """out[r] += sum of contrib over the row's entries (np.add.reduceat over non-empty rows)."""
    nnz = np.diff(Xs.indptr)
    ne = np.flatnonzero(nnz > 0)
    out[ne] += np.add.reduceat(contrib, Xs.indptr[ne], axis=0)


def rhs(Xs, other64, bias):
    rows_ne = np.flatnonzero(np.diff(Xs.indptr) > 0)
    b = np.zeros((Xs.shape[0], other64.shape[1]))
    if rows_ne.size:
        w = (Xs.data.astype(np.float64) + bias)[:, None] * other64[Xs.indices]
        b[rows_ne] = np.add.reduceat(w, Xs.indptr[rows_ne], axis=0)
    return b


def own_item_score_err(Xs, other64, x, x64):
    """per row: ||V_r (x - x64)|| / ||V_r x64|| over the row's own stored items (0 for empty rows)"""
    nnz = np.diff(Xs.indptr)
    rows = np.repeat(np.arange(Xs.shape[0]), nnz)
    V = other64[Xs.indices]
    d = np.einsum("ij,ij->i", V, (np.asarray(x, np.float64) - x64)[rows]) ** 2
    s = np.einsum("ij,ij->i", V, x64[rows]) ** 2
    ne = np.flatnonzero(nnz > 0)
    num, den = np.zeros(Xs.shape[0]), np.zeros(Xs.shape[0])
    if ne.size:
        num[ne] = np.sqrt(np.add.reduceat(d, Xs.indptr[ne]))
        den[ne] = np.sqrt(np.add.reduceat(s, Xs.indptr[ne]))
    floor = 1e-6 * (den.max() if den.size else 0.0)
    return np.where(den > 0, num / np.maximum(den, max(floor, 1e-300)), num)


def row_dist(a, ref, floor=1e-6):
    ref = np.asarray(ref, np.float64)
    den = np.linalg.norm(ref, axis=1)
    den = np.maximum(den, floor * max(float(den.max()) if den.size else 0.0, 1e-300))
    return np.linalg.norm(np.asarray(a, np.float64) - ref, axis=1) / den


def condition_numbers(Xs, other64, P64, regs, rows):
    """2-norm condition number of A_r for `rows` (dense, float64)."""
    out = np.empty(len(rows))
    for n, r in enumerate(rows):
        sl = slice(Xs.indptr[r], Xs.indptr[r + 1])
        V = other64[Xs.indices[sl]]
        A = P64 + (V * Xs.data[sl].astype(np.float64)[:, None]).T @ V + regs[r] * np.eye(P64.shape[0])
        w = np.linalg.eigvalsh(A)
        out[n] = w[-1] / w[0] if w[0] > 0 else np.inf
    return out


def gpu_half_step(t, side, sc):
    t.partial_gramian_async(side)
    t.finish_gramian_async(side)
    t.half_step_async(side, sc)
    t.synchronize()


def q(e, p):
    return float(np.quantile(e, p)) if e.size else 0.0


def measure_side(t, side, Xs, tgt0, oth0, kind, alpha0, reg, nu, loss, cfg, kappa_rows=0, seed=0):
    """One half-step of `side` from (tgt0, oth0) on the GPU trainer `t`, the float32 oracle and the
    float64 arbiter.  Returns a dict of per-row error arrays + the error-parity pair."""
    mc, sc, omc, osc = cfg
    user0, item0 = (tgt0, oth0) if side == 0 else (oth0, tgt0)
    t.user, t.item = user0, item0
    gpu_exc = raises(lambda: gpu_half_step(t, side, sc))
    P32 = O.ials_gramian(oth0, omc.alpha0, CORES)
    box = {}

    def run_oracle():
        box["want"] = O.ials_solver_step(tgt0, Xs, oth0, P32, omc, osc)

    orc_exc = raises(run_oracle)
    out = dict(gpu_exc=gpu_exc, orc_exc=orc_exc)
    if gpu_exc or orc_exc:
        return out
    got = t.user if side == 0 else t.item
    want = box["want"]
    x64 = O.ials_solver_step_f64(tgt0, Xs, oth0, None, omc, osc, CORES)
    other64 = oth0.astype(np.float64)
    out["fac_gpu"], out["fac_orc"] = row_dist(got, x64), row_dist(want, x64)
    out["sco_gpu"] = own_item_score_err(Xs, other64, got, x64)
    out["sco_orc"] = own_item_score_err(Xs, other64, want, x64)
    out["finite"] = bool(np.isfinite(got).all())
    out["orc_finite"] = bool(np.isfinite(want).all() and np.isfinite(x64).all())
    out["nnz"] = np.diff(Xs.indptr)
    out["K"] = int(oth0.shape[1])
    if kind == "CHOLESKY":
        P64 = O.ials_gramian_f64(other64, omc.alpha0, CORES)
        regs = row_reg(Xs, oth0.shape[0], alpha0, reg, nu)
        b = rhs(Xs, other64, 0.0 if loss == "IALSPP" else alpha0)
        bn = np.linalg.norm(b, axis=1)
        ne = np.diff(Xs.indptr) > 0
        # an upper bound of kappa(A_r) for EVERY row: (lmax(P) + sum_j c_j |v_j|^2 + reg_r) / (lmin(P) + reg_r)
        w = np.linalg.eigvalsh(P64)
        s_r = np.zeros(Xs.shape[0])
        nz = np.flatnonzero(ne)
        if nz.size:
            s_r[nz] = np.add.reduceat(Xs.data.astype(np.float64) * np.einsum("ij,ij->i", other64[Xs.indices],
                                                                              other64[Xs.indices]), Xs.indptr[nz])
        # `res`: the normwise BACKWARD error ||A x - b|| / (||A|| ||x|| + ||b||) with ||A|| bounded from above
        # by lmax(P) + sum_j c_j |v_j|^2 + reg_r - what a backward-stable solve keeps at ~K 2^-24 whatever
        # the conditioning (||A x - b|| / ||b|| alone grows with ||A|| ||x|| / ||b||)
        a_norm = max(float(w[-1]), 0.0) + s_r + regs
        for name, x in (("gpu", got), ("orc", want)):
            xn = np.linalg.norm(np.asarray(x, np.float64), axis=1)
            r = np.linalg.norm(apply_A(Xs, other64, P64, regs, x) - b, axis=1) / np.maximum(a_norm * xn + bn, 1e-300)
            out["res_" + name] = np.where(ne, r, 0.0)
        den = max(float(w[0]), 0.0) + regs
        out["kappa_bound"] = np.where(den > 0, (float(w[-1]) + s_r + regs) / np.where(den > 0, den, 1.0), np.inf)
        if kappa_rows:
            rows = np.random.default_rng(seed).choice(Xs.shape[0], size=min(kappa_rows, Xs.shape[0]),
                                                      replace=False)
            out["kappa_rows"], out["kappa"] = rows, condition_numbers(Xs, other64, P64, regs, rows)
    return out


def summary(m):
    """the scalar digest of measure_side's arrays that goes to the parity log"""
    s = {}
    for k in ("fac_gpu", "fac_orc", "sco_gpu", "sco_orc", "res_gpu", "res_orc"):
        if k in m:
            e = m[k]
            s[k + "_worst"], s[k + "_p999"], s[k + "_median"] = float(e.max()) if e.size else 0.0, q(e, 0.999), q(e, 0.5)
    if "kappa" in m:
        s["kappa_max"], s["kappa_median"] = float(np.max(m["kappa"])), float(np.median(m["kappa"]))
    if "kappa_bound" in m:
        kb = m["kappa_bound"]
        s["kappa_bound_median"] = float(np.median(kb)) if kb.size else 0.0
        s["n_rows_well_conditioned"] = int((kb * 2.0 ** -24 < 1e-4).sum())
    if "fac_gpu" in m:
        s["n_rows"] = int(m["fac_gpu"].size)
        s["n_fac_gpu_over_1e_4"] = int((m["fac_gpu"] >= 1e-4).sum())
        s["n_fac_orc_over_1e_4"] = int((m["fac_orc"] >= 1e-4).sum())
        s["n_sco_gpu_over_1e_4"] = int((m["sco_gpu"] >= 1e-4).sum())
        s["n_sco_orc_over_1e_4"] = int((m["sco_orc"] >= 1e-4).sum())
        for k in ("fac_gpu", "fac_orc", "sco_gpu", "sco_orc"):  # (the second threshold of the count bars)
            s["n_" + k + "_over_1e_3"] = int((m[k] >= 1e-3).sum())
    return s


def run_point(X, Xt, K, kind, alpha0, reg, nu=1.0, loss="IALSPP", steps=3, subspace=64,
              epochs_before=1, kappa_rows=0):
    """Error parity over `epochs_before` full epochs from the seeded init, then one measured
    half-step per side from the GPU's factors.  Returns {'train_exc': (gpu, oracle), 'sides': [...]}"""
    cfg = configs(K, kind, alpha0, reg, nu, loss, steps, subspace)
    mc, sc, omc, osc = cfg
    t = IALSTrainer(mc, X)
    o = O.IALSTrainer(omc, X)

    def steps_of(tr, conf):
        def go():
            for _ in range(epochs_before):
                tr.step(conf)
        return go

    gpu_exc, orc_exc = raises(steps_of(t, sc)), raises(steps_of(o, osc))
    res = dict(train_exc=(gpu_exc, orc_exc), sides=[])
    if gpu_exc or orc_exc:
        return res
    user0, item0 = t.user, t.item
    res["train_fac_user"] = float(row_dist(user0, o.user).max())
    res["train_fac_item"] = float(row_dist(item0, o.item).max())
    for side, (Xs, tgt0, oth0) in enumerate(((X, user0, item0), (Xt, item0, user0))):
        res["sides"].append(measure_side(t, side, Xs, tgt0, oth0, kind, alpha0, reg, nu, loss, cfg,
                                         kappa_rows=kappa_rows))
    return res


def transpose_csr(X):
    Xt = sps.csr_matrix(X.T)
    Xt.sort_indices()
    return Xt
