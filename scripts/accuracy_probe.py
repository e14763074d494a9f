"""Development probe: per-half-step accuracy of GPU vs CPU oracle vs float64 closed form."""
import os
import sys

import numpy as np
import scipy.sparse as sps

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle as O  # noqa: E402
from irspack_amd.recommenders._ials_core import (  # noqa: E402
    IALSModelConfigBuilder, IALSSolverConfigBuilder, IALSTrainer, SolverType)
from irspack_amd.synthetic import make_interactions  # noqa: E402


def closed_form(X, other, alpha0, reg_rows, bias, x0=None, cg_steps=None):
    X = sps.csr_matrix(X).astype(np.float64)
    other = other.astype(np.float64)
    K = other.shape[1]
    P = alpha0 * other.T @ other
    out = np.zeros((X.shape[0], K))
    for r in range(X.shape[0]):
        sl = slice(X.indptr[r], X.indptr[r + 1])
        V = other[X.indices[sl]]
        c = X.data[sl]
        A = P + (V * c[:, None]).T @ V + reg_rows[r] * np.eye(K)
        b = ((c + bias)[:, None] * V).sum(axis=0)
        if cg_steps is None:
            out[r] = np.linalg.solve(A, b)
        else:
            x = x0[r].astype(np.float64).copy()
            if sl.stop == sl.start:
                out[r] = 0
                continue
            rr = b - A @ x
            p = rr.copy()
            for _ in range(cg_steps):
                r2 = rr @ rr
                if r2 <= 1e-20:
                    break
                Ap = A @ p
                al = r2 / (p @ Ap)
                x += al * p
                rr -= al * Ap
                r2n = rr @ rr
                if r2n <= 1e-20:
                    break
                p = rr + (r2n / r2) * p
            out[r] = x
    return out


def rel(a, b):
    return float(np.abs(a - b).max() / np.abs(b).max())


def main():
    shape = sys.argv[1] if len(sys.argv) > 1 else "ml100k"
    K = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    X = make_interactions(shape)
    alpha0, reg = 0.1, 1e-3
    omc = O.model_config(K, alpha0=alpha0, reg=reg, nu=1.0)
    mc = IALSModelConfigBuilder().set_K(K).set_alpha0(alpha0).set_reg(reg).build()
    for kind in ["CHOLESKY", "CG"]:
        osc = O.solver_config(8, kind, 3)
        sc = (IALSSolverConfigBuilder().set_n_threads(1).set_solver_type(SolverType[kind])
              .set_max_cg_steps(3).build())
        o = O.IALSTrainer(omc, X)
        for _ in range(3):
            o.step(osc)
        U0, V0 = o.user, o.item
        t = IALSTrainer(mc, X)
        t.user, t.item = U0, V0
        t.step(sc)
        o.step(osc)
        nnz = np.diff(X.indptr)
        reg_rows = reg * (alpha0 * X.shape[1] + nnz)
        exact = closed_form(X, V0, alpha0, reg_rows, 0.0, x0=U0,
                            cg_steps=None if kind == "CHOLESKY" else 3)
        print(kind, "user half: gpu-vs-f64", rel(t.user, exact), "cpu-vs-f64", rel(o.user, exact),
              "gpu-vs-cpu", rel(t.user, o.user), flush=True)
        print(kind, "item half (after user): gpu-vs-cpu", rel(t.item, o.item), flush=True)


if __name__ == "__main__":
    main()
