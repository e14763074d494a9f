"""Development probe: where the wall time of one compute_similarity call goes (ML-20M shape, cosine,
top_k = 100) for several row-chunk counts (IRSPACK_AMD_KNN_CHUNKS): library compute, fetch, scipy."""
import ctypes as C
import os
import sys
import time

import numpy as np
import scipy.sparse as sps

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))

from irspack_amd import _lib  # noqa: E402
from irspack_amd._lib import check, lib, ptr  # noqa: E402
from irspack_amd.recommenders._knn import CosineSimilarityComputer  # noqa: E402
from irspack_amd.synthetic import make_interactions  # noqa: E402


def main() -> None:
    X = make_interactions("ml20m")
    Xt = sps.csr_matrix(X.T, dtype=np.float64)
    Xt.data[:] = 1.0
    comp = CosineSimilarityComputer(Xt, 0.0, True)
    Xc, indptr, indices, data = _lib.csr_arrays(Xt, np.float64)
    ref = None
    for chunks in sys.argv[1:] or ["1", "4", "6", "8", "default"]:
        if chunks == "default":
            os.environ.pop("IRSPACK_AMD_KNN_CHUNKS", None)
        else:
            os.environ["IRSPACK_AMD_KNN_CHUNKS"] = chunks
        best = None
        for rep in range(6):
            if rep == 5:
                os.environ["IRSPACK_AMD_KNN_TIMING"] = "1"
            t0 = time.perf_counter()
            nnz = C.c_int64(0)
            check(lib().irs_knn_compute(comp._h, C.c_int64(Xc.shape[0]), C.c_int64(Xc.shape[1]),
                                        ptr(indptr, C.c_int64), ptr(indices, C.c_int32), ptr(data, C.c_double),
                                        C.c_int64(100), C.c_int32(0), C.c_int64(0), C.c_int64(Xc.shape[0]),
                                        C.byref(nnz)))
            t1 = time.perf_counter()
            o_ptr = np.empty(Xc.shape[0] + 1, dtype=np.int64)
            o_idx = np.empty(max(nnz.value, 1), dtype=np.int32)
            o_val = np.empty(max(nnz.value, 1), dtype=np.float64)
            check(lib().irs_knn_fetch(comp._h, ptr(o_ptr, C.c_int64), ptr(o_idx, C.c_int32), ptr(o_val, C.c_double)))
            t2 = time.perf_counter()
            ms = C.c_double(0)
            macs = C.c_int64(0)
            check(lib().irs_knn_last_stats(comp._h, C.byref(ms), C.byref(macs)))
            os.environ.pop("IRSPACK_AMD_KNN_TIMING", None)
            row = (t2 - t0, t1 - t0, t2 - t1, ms.value)
            if rep < 5 and (best is None or row[0] < best[0]):
                best = row
        S = sps.csr_matrix((o_val, o_idx, o_ptr), shape=(Xc.shape[0], Xc.shape[0]))
        if ref is None:
            ref = S
        same = (np.array_equal(ref.indptr, S.indptr) and np.array_equal(ref.indices, S.indices)
                and np.array_equal(ref.data, S.data))
        t0 = time.perf_counter()
        comp.compute_similarity(Xt, 100)
        full = time.perf_counter() - t0
        print(f"chunks {chunks:>7}: compute+fetch {best[0]*1e3:6.2f} ms = compute {best[1]*1e3:6.2f} + fetch {best[2]*1e3:5.2f}; "
              f"kernel events {best[3]:5.2f} ms; compute_similarity() {full*1e3:6.2f} ms; same bits as the first: {same}", flush=True)


if __name__ == "__main__":
    main()
