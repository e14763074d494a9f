"""Why is one ML-20M user row (38077) the worst row of the K = 64 CG half-step every time?
Prints the float64 CG history of the worst rows (residual norms, step sizes, conditioning)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle as O
from irspack_amd.recommenders._ials_core import IALSTrainer
from irspack_amd.synthetic import make_interactions
from test_gpu_fullsize import CORES, configs, half_step

X = make_interactions("ml20m")
K = 64
mc, sc, omc, osc = configs(K, "CG")
t = IALSTrainer(mc, X)
t.step(sc)
user0, item0 = t.user, t.item
half_step(t, 0, sc)
got = t.user.astype(np.float64)
P32 = O.ials_gramian(item0, omc.alpha0, CORES)
want = O.ials_solver_step(user0, X, item0, P32, omc, osc).astype(np.float64)
err = np.linalg.norm(got - want, axis=1) / np.linalg.norm(want, axis=1)
O64 = item0.astype(np.float64)
P = 0.1 * O64.T @ O64
for r in np.argsort(-err)[:6]:
    sl = slice(X.indptr[r], X.indptr[r + 1])
    V = O64[X.indices[sl]]
    reg = float(np.float32(1e-3) * (np.float32(0.1) * np.float32(X.shape[1]) + np.float32(sl.stop - sl.start)))
    A = P + V.T @ V + reg * np.eye(K)
    b = V.sum(axis=0)
    x = user0[r].astype(np.float64).copy()
    rr = b - A @ x
    p = rr.copy()
    hist, alphas = [float(rr @ rr)], []
    for _ in range(3):
        r2 = rr @ rr
        Ap = A @ p
        al = r2 / (p @ Ap)
        alphas.append(al)
        x += al * p
        rr -= al * Ap
        hist.append(float(rr @ rr))
        p = rr + ((rr @ rr) / r2) * p
    xs = np.linalg.solve(A, b)
    ev = np.linalg.eigvalsh(A)
    print(f"row {r} nnz {sl.stop - sl.start} err {err[r]:.2e} gpu-f64 {np.linalg.norm(got[r] - x) / np.linalg.norm(x):.2e} "
          f"orc-f64 {np.linalg.norm(want[r] - x) / np.linalg.norm(x):.2e} |x0| {np.linalg.norm(user0[r]):.3f} |x3| {np.linalg.norm(x):.3f} "
          f"|x*| {np.linalg.norm(xs):.3f} cg3-x* {np.linalg.norm(x - xs) / np.linalg.norm(xs):.2e} cond {ev[-1] / ev[0]:.1e} "
          f"r2 {['%.1e' % h for h in hist]} alpha {['%.2e' % a for a in alphas]}")
