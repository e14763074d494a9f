import time, os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
from irspack_amd.synthetic import make_interactions
from irspack_amd.recommenders._ials_core import IALSModelConfigBuilder, IALSTrainer
t=time.time(); X=make_interactions("c4"); print("gen", time.time()-t, flush=True)
mc = IALSModelConfigBuilder().set_K(128).set_alpha0(0.1).set_reg(1e-3).build()
for i in range(2):
    t=time.time(); tr=IALSTrainer(mc, X); print("create", time.time()-t, flush=True); del tr
