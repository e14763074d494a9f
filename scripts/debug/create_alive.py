import os, sys, time
sys.path.insert(0, "/root/repo")
os.environ["IRSPACK_AMD_IALS_TIMING"] = "1"
from irspack_amd.recommenders._ials_core import IALSModelConfigBuilder, IALSTrainer
from irspack_amd.synthetic import make_interactions
X = make_interactions("ml20m")
mc = IALSModelConfigBuilder().set_K(64).set_alpha0(0.1).set_reg(1e-3).build()
keep = []
for i in range(4):
    t0 = time.perf_counter()
    t = IALSTrainer(mc, X)
    print(f"create #{i} (previous trainers alive: {len(keep)}): {(time.perf_counter() - t0) * 1e3:.1f} ms", file=sys.stderr, flush=True)
    keep.append(t)
