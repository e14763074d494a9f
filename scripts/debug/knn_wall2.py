import os, sys, time, ctypes as C
import numpy as np, scipy.sparse as sps
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from irspack_amd import _lib
from irspack_amd._lib import lib, ptr, check
from irspack_amd.recommenders._knn import CosineSimilarityComputer
from irspack_amd.synthetic import make_interactions
X = make_interactions("ml20m")
Xt = sps.csr_matrix(X.T, dtype=np.float64); Xt.data[:] = 1.0
comp = CosineSimilarityComputer(Xt, 0.0, True)
comp.compute_similarity(Xt, 100, rows=(0, 64))
for _ in range(3):
    t0 = time.perf_counter()
    Xc, indptr, indices, data = _lib.csr_arrays(Xt, np.float64)
    t1 = time.perf_counter()
    nnz = C.c_int64(0)
    check(lib().irs_knn_compute(comp._h, C.c_int64(Xc.shape[0]), C.c_int64(Xc.shape[1]), ptr(indptr, C.c_int64),
                                ptr(indices, C.c_int32), ptr(data, C.c_double), C.c_int64(100), C.c_int32(0),
                                C.c_int64(0), C.c_int64(Xc.shape[0]), C.byref(nnz)))
    t2 = time.perf_counter()
    o_ptr = np.empty(Xc.shape[0] + 1, dtype=np.int64); o_idx = np.empty(nnz.value, dtype=np.int32); o_val = np.empty(nnz.value, dtype=np.float64)
    t3 = time.perf_counter()
    check(lib().irs_knn_fetch(comp._h, ptr(o_ptr, C.c_int64), ptr(o_idx, C.c_int32), ptr(o_val, C.c_double)))
    t4 = time.perf_counter()
    res = sps.csr_matrix((o_val, o_idx, o_ptr), shape=(Xc.shape[0], Xc.shape[0]))
    t5 = time.perf_counter()
    print("csr_arrays %.2f compute %.2f empty %.2f fetch %.2f csr %.2f total %.2f ms" % tuple(1e3 * x for x in (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4, t5 - t0)), flush=True)
