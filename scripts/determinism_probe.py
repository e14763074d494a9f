import sys, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import oracle as O
from irspack_amd.recommenders._ials_core import *
from irspack_amd.synthetic import make_interactions
X = make_interactions("ml100k")
mc = IALSModelConfigBuilder().set_K(16).set_alpha0(0.1).set_reg(1e-3).set_init_stdev(0.1).build()
omc = O.model_config(16, alpha0=0.1, reg=1e-3, nu=1.0, init_stdev=0.1, random_seed=42, loss_type="IALSPP")
for kind in ("CHOLESKY","CG"):
    sc = IALSSolverConfigBuilder().set_n_threads(4).set_solver_type(SolverType[kind]).set_max_cg_steps(3).build()
    outs=[]; oouts=[]
    for rep in range(6):
        t = IALSTrainer(mc, X); t.step(sc); t.step(sc); outs.append((t.user.copy(), t.item.copy()))
        o = O.IALSTrainer(omc, X); osc=O.solver_config(4, kind, 3); o.step(osc); o.step(osc); oouts.append((o.user.copy(), o.item.copy()))
    print(kind, "gpu identical:", all(np.array_equal(outs[0][0], x[0]) and np.array_equal(outs[0][1], x[1]) for x in outs),
          "oracle identical:", all(np.array_equal(oouts[0][0], x[0]) and np.array_equal(oouts[0][1], x[1]) for x in oouts),
          "max oracle diff", max(np.abs(oouts[0][0]-x[0]).max() for x in oouts), "max gpu diff", max(np.abs(outs[0][0]-x[0]).max() for x in outs))
