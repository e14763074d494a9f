"""Epoch time of irs_ials_sharded_step on ONE GPU (world size 1: the collectives are identities) with
the shard's rows solved and exchanged in 1, 2, 4 chunks: what chunking costs the compute side.
    python scripts/shard_chunks_probe.py [K] [solver]"""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(K, kind):
    import numpy as np  # noqa: F401

    from irspack_amd.recommenders._ials_core import IALSModelConfigBuilder, IALSSolverConfigBuilder, SolverType
    from irspack_amd.sharding import HipLocalSolver, ShardedIALSTrainer
    from irspack_amd.synthetic import make_interactions

    X = make_interactions("ml20m")
    U, I = X.shape
    mc = IALSModelConfigBuilder().set_K(K).set_alpha0(0.1).set_reg(1e-3).build()
    sc = IALSSolverConfigBuilder().set_solver_type(SolverType[kind]).set_max_cg_steps(3).build()
    local = HipLocalSolver(mc, X, (0, U, 0, I), 0)
    tr = ShardedIALSTrainer(local, [0, U], [0, I], native=True)
    for _ in range(3):
        tr.step(sc)
    ts = []
    for _ in range(10):
        t0 = time.perf_counter()
        tr.step(sc)
        ts.append(time.perf_counter() - t0)
    print(f"chunks={os.environ.get('IRSPACK_AMD_SHARD_CHUNKS', '1')} K={K} {kind}: median {sorted(ts)[5] * 1e3:.3f} ms / epoch")


if __name__ == "__main__":
    K = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    kind = sys.argv[2] if len(sys.argv) > 2 else "CHOLESKY"
    if os.environ.get("_PROBE_CHILD"):
        run(K, kind)
    else:
        for c in ("1", "2", "4"):
            env = dict(os.environ, IRSPACK_AMD_SHARD_CHUNKS=c, _PROBE_CHILD="1")
            subprocess.run([sys.executable, __file__, str(K), kind], env=env, check=False)
