"""Phase times of irs_ials_create (IRSPACK_AMD_IALS_TIMING=1 prints them to stderr) and of a second
construction, for one shape: python scripts/create_timing.py [shape] [K]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["IRSPACK_AMD_IALS_TIMING"] = "1"

from irspack_amd.recommenders._ials_core import IALSModelConfigBuilder, IALSTrainer  # noqa: E402
from irspack_amd.synthetic import make_interactions  # noqa: E402

shape = sys.argv[1] if len(sys.argv) > 1 else "ml20m"
K = int(sys.argv[2]) if len(sys.argv) > 2 else 64
X = make_interactions(shape)
mc = IALSModelConfigBuilder().set_K(K).set_alpha0(0.1).set_reg(1e-3).build()
print(f"host threads: {os.cpu_count()}", file=sys.stderr)
from irspack_amd import _lib  # noqa: E402
import numpy as np  # noqa: E402

for i in range(3):
    t0 = time.perf_counter()
    _lib.csr_arrays(X, np.float32)
    print(f"csr_arrays: {(time.perf_counter() - t0) * 1e3:.1f} ms", file=sys.stderr)
    t0 = time.perf_counter()
    t = IALSTrainer(mc, X)
    print(f"create #{i}: {(time.perf_counter() - t0) * 1e3:.1f} ms", file=sys.stderr, flush=True)
    del t
