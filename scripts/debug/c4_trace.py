"""Development probe: three CG epochs on the configs[3] shape (10 M x 1 M, K = 128) for a kernel trace
(rocprofv3 --kernel-trace -- python3 scripts/debug/c4_trace.py); prints the epoch times."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))

from irspack_amd.recommenders._ials_core import (IALSModelConfigBuilder, IALSSolverConfigBuilder,  # noqa: E402
                                                  IALSTrainer, SolverType)
from irspack_amd.synthetic import make_interactions  # noqa: E402


def main() -> None:
    kind = sys.argv[1] if len(sys.argv) > 1 else "CG"
    X = make_interactions(sys.argv[2] if len(sys.argv) > 2 else "c4")
    mc = IALSModelConfigBuilder().set_K(128).set_alpha0(0.1).set_reg(1e-3).build()
    sc = (IALSSolverConfigBuilder().set_n_threads(1).set_solver_type(SolverType[kind]).set_max_cg_steps(3).build())
    t = IALSTrainer(mc, X)
    for ep in range(4):
        t0 = time.perf_counter()
        t.step(sc)
        t.synchronize()
        print(f"epoch {ep}: {(time.perf_counter() - t0) * 1e3:.2f} ms", flush=True)


if __name__ == "__main__":
    main()
