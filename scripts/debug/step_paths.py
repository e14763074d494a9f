import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from bench import model_config, solver_config
from irspack_amd.sharding import HipLocalSolver, ShardedIALSTrainer, equal_shard_bounds
from irspack_amd.synthetic import make_interactions
X = make_interactions("ml20m")
ub, ib = equal_shard_bounds(X, 1)
local = HipLocalSolver(model_config(64), X, (ub[0], ub[1], ib[0], ib[1]), 0)
tr = ShardedIALSTrainer(local, ub, ib)
sc = solver_config("CHOLESKY")
for name, stepper in (("sharded loop", tr.step), ("C step", local.trainer.step), ("sharded loop", tr.step), ("C step", local.trainer.step)):
    for prof in (True, False):
        for _ in range(3): stepper(sc)
        local.trainer.synchronize(); local.trainer.profile(prof)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): stepper(sc)
        local.trainer.synchronize(); torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 20 * 1e3
        if prof: local.trainer.profile_read()
        local.trainer.profile(False)
        print(name, "profile", prof, "%.3f ms/epoch" % dt, flush=True)
