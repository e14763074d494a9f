"""Development probe: worst rows of the K = 64 CG half-step at the ML-20M shape, GPU and oracle
against a float64 CG (3 steps, warm start) on the same inputs."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle as O  # noqa: E402
from test_gpu_fullsize import configs, half_step, oracle_rows, row_sample  # noqa: E402
from irspack_amd.recommenders._ials_core import IALSTrainer  # noqa: E402
from irspack_amd.synthetic import make_interactions  # noqa: E402


def cg64(A, b, x0, steps):
    x = x0.copy()
    r = b - A @ x
    p = r.copy()
    for _ in range(steps):
        r2 = r @ r
        if r2 <= 1e-20:
            break
        Ap = A @ p
        a = r2 / (p @ Ap)
        x += a * p
        r -= a * Ap
        r2n = r @ r
        if r2n <= 1e-20:
            break
        p = r + (r2n / r2) * p
    return x


X = make_interactions("ml20m")
Xt = X.T.tocsr()
Xt.sort_indices()
mc, sc, omc, osc = configs(64, "CG")
t = IALSTrainer(mc, X)
t.step(sc)
user0, item0 = t.user, t.item
for side, (Xs, tgt0, oth0) in enumerate(((X, user0, item0), (Xt, item0, user0))):
    t.user, t.item = user0, item0
    half_step(t, side, sc)
    got = t.user if side == 0 else t.item
    rows, split = row_sample(Xs, 2000, seed=side)
    want = oracle_rows(tgt0, Xs, rows, oth0, omc, osc)
    num = np.linalg.norm(got[rows].astype(np.float64) - want, axis=1)
    den = np.linalg.norm(want.astype(np.float64), axis=1)
    err = num / den
    order = np.argsort(-err)[:8]
    nnz = np.diff(Xs.indptr)
    P = 0.1 * oth0.astype(np.float64).T @ oth0.astype(np.float64)
    print(f"side {side}: worst {err[order[0]]:.3e}; rows > 1e-4: {(err > 1e-4).sum()} of {rows.size}")
    for j in order:
        r = rows[j]
        sl = slice(Xs.indptr[r], Xs.indptr[r + 1])
        V = oth0[Xs.indices[sl]].astype(np.float64)
        reg = np.float32(1e-3) * (np.float32(0.1) * np.float32(Xs.shape[1]) + np.float32(nnz[r]))
        A = P + V.T @ V + float(reg) * np.eye(64)
        b = V.sum(axis=0)
        ref = cg64(A, b, tgt0[r].astype(np.float64), 3)
        full = np.linalg.solve(A, b)
        print(f"  row {r} nnz {nnz[r]} |x| {den[j]:.3e} gpu-orc {err[j]:.2e} gpu-f64 "
              f"{np.linalg.norm(got[r] - ref) / np.linalg.norm(ref):.2e} orc-f64 "
              f"{np.linalg.norm(want[j] - ref) / np.linalg.norm(ref):.2e} |cg3 - exact| "
              f"{np.linalg.norm(ref - full) / np.linalg.norm(full):.2e} cond {np.linalg.cond(A):.1f}")
