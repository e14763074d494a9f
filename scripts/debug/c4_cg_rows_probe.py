"""Where do GPU and oracle differ under truncated CG on the full configs[3] matrix?
Prints, for the worst user rows of one CG half-step: nnz, norms, distance of each side from
the float64 iteration, and the float64 residual history of the row."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle as O
from irspack_amd.recommenders._ials_core import IALSTrainer
from irspack_amd.synthetic import make_interactions
from test_gpu_fullsize import configs, half_step, oracle_rows, row_sample

name = sys.argv[1] if len(sys.argv) > 1 else "c4"
X = make_interactions(name)
K = 128
mc, sc, omc, osc = configs(K, "CG")
t = IALSTrainer(mc, X)
t.step(sc)
user0, item0 = t.user, t.item
half_step(t, 0, sc)
got = t.user
rows, _ = row_sample(X, 20000, seed=40)
want = oracle_rows(user0, X, rows, item0, omc, osc)
g = got[rows].astype(np.float64)
num = np.linalg.norm(g - want, axis=1)
den = np.linalg.norm(want.astype(np.float64), axis=1)
err = num / np.maximum(den, 1e-6 * den.max())
order = np.argsort(-err)[:12]
O64 = item0.astype(np.float64)
P = 0.1 * O64.T @ O64
P32g = None
print("den.max", den.max(), "median den", np.median(den))
for j in order:
    r = rows[j]
    sl = slice(X.indptr[r], X.indptr[r + 1])
    V = O64[X.indices[sl]]
    reg = float(np.float32(1e-3) * (np.float32(0.1) * np.float32(X.shape[1]) + np.float32(sl.stop - sl.start)))
    A = P + V.T @ V + reg * np.eye(K)
    b = V.sum(axis=0)
    x = user0[r].astype(np.float64).copy()
    rr = b - A @ x
    p = rr.copy()
    hist = [float(rr @ rr)]
    for _ in range(3):
        r2 = rr @ rr
        if r2 <= 1e-20:
            break
        Ap = A @ p
        al = r2 / (p @ Ap)
        x += al * p
        rr -= al * Ap
        hist.append(float(rr @ rr))
        if rr @ rr <= 1e-20:
            break
        p = rr + ((rr @ rr) / r2) * p
    xs = np.linalg.solve(A, b)
    nx = np.linalg.norm(x)
    print(f"row {r} nnz {sl.stop - sl.start} err {err[j]:.2e} |x64| {nx:.3e} |x0| {np.linalg.norm(user0[r]):.3e} "
          f"|b| {np.linalg.norm(b):.3e} gpu-f64 {np.linalg.norm(g[j] - x) / nx:.2e} orc-f64 {np.linalg.norm(want[j] - x) / nx:.2e} "
          f"cg3-exact {np.linalg.norm(x - xs) / np.linalg.norm(xs):.2e} r2 hist {['%.1e' % h for h in hist]}")
