"""Which term of the one-block iALS++ sweep (RESID kernels) is off?  GPU row vs float64 variants."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, scipy.sparse as sps
from conftest import random_csr
from irspack_amd.recommenders._ials_core import (IALSModelConfigBuilder, IALSSolverConfigBuilder, IALSTrainer, SolverType, LossType)

for K in (16, 32, 64):
  for binary in (False, True):
    X = random_csr(150, 110, 0.1, 3, empty_rows=(7, 40), binary=binary)
    alpha0, reg = 0.1, 1e-2
    mc = IALSModelConfigBuilder().set_K(K).set_alpha0(alpha0).set_reg(reg).build()
    sc = (IALSSolverConfigBuilder().set_solver_type(SolverType.IALSPP).set_ialspp_subspace_dimension(64).set_ialspp_iteration(1).build())
    t = IALSTrainer(mc, X)
    U0, V0 = t.user.astype(np.float64), t.item.astype(np.float64)
    t.partial_gramian_async(0); t.finish_gramian_async(0); t.half_step_async(0, sc); t.synchronize()
    got = t.user.astype(np.float64)
    P = alpha0 * V0.T @ V0
    def variant(with_px=True, with_reg=True, add_x=True):
        out = np.zeros_like(U0)
        for r in range(X.shape[0]):
            sl = slice(X.indptr[r], X.indptr[r + 1]); Vr = V0[X.indices[sl]]; c = X.data[sl].astype(np.float64)
            regr = reg * (alpha0 * X.shape[1] + len(c))
            A = P + (Vr * c[:, None]).T @ Vr + regr * np.eye(K)
            x = U0[r]
            rhs = ((c - c * (Vr @ x))[:, None] * Vr).sum(0)
            if with_px: rhs = rhs - P @ x
            if with_reg: rhs = rhs - regr * x
            d = np.linalg.solve(A, rhs)
            out[r] = (x if add_x else 0) + d
        return out
    def err(a):
        n = np.linalg.norm(a, axis=1); return float((np.linalg.norm(got - a, axis=1) / np.maximum(n, 1e-6 * n.max())).max())
    print(K, "binary" if binary else "general", "full", err(variant()), "no_px", err(variant(with_px=False)), "no_reg", err(variant(with_reg=False)),
          "no_add", err(variant(add_x=False)))
