import numpy as np, torch, sys, os
sys.path.insert(0, os.getcwd())
from irspack_amd.recommenders._ials_core import IALSModelConfigBuilder, IALSTrainer
from irspack_amd.synthetic import make_interactions
X = make_interactions("tiny")
t = IALSTrainer(IALSModelConfigBuilder().set_K(20).build(), X, device=0)
ptr, rows, ld = t.device_buffer(0)
class Wrap:
    def __init__(s, ptr, shape): s.__cuda_array_interface__ = {"shape": shape, "typestr": "<f4", "data": (ptr, False), "version": 2, "strides": None}
w = torch.as_tensor(Wrap(ptr, (rows, ld)), device="cuda:0")
print(w.shape, w.dtype, w.data_ptr() == ptr)
print(np.abs(w[:, :20].cpu().numpy() - t.user).max())
w[3, :20] = 7.0
torch.cuda.synchronize()
print(t.user[3, :3])
