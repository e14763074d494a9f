"""ctypes loader for ``libirspack_amd.so`` (the C ABI of include/irspack_amd.h).

Loading fails loudly when the shared library has not been built; compute calls
fail loudly (``RuntimeError``) when no HIP device is visible.  There is no
fallback implementation.
"""

import ctypes as C
import os
from typing import Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# IRSPACK_AMD_LIB selects another build of the same ABI (kernel tuning experiments)
LIB_PATH = os.environ.get("IRSPACK_AMD_LIB") or os.path.join(_HERE, "libirspack_amd.so")


class ModelConfigStruct(C.Structure):
    _fields_ = [
        ("K", C.c_uint64),
        ("alpha0", C.c_float),
        ("reg", C.c_float),
        ("nu", C.c_float),
        ("init_stdev", C.c_float),
        ("random_seed", C.c_int32),
        ("loss_type", C.c_int32),
        ("lambda_user_feature", C.c_float),
        ("lambda_item_feature", C.c_float),
        ("feature_warmup_epochs", C.c_uint64),
    ]


class SolverConfigStruct(C.Structure):
    _fields_ = [
        ("n_threads", C.c_uint64),
        ("solver_type", C.c_int32),
        ("max_cg_steps", C.c_uint64),
        ("ialspp_subspace_dimension", C.c_uint64),
        ("ialspp_iteration", C.c_uint64),
    ]


class ShardStruct(C.Structure):
    _fields_ = [
        ("user_begin", C.c_int64),
        ("user_end", C.c_int64),
        ("item_begin", C.c_int64),
        ("item_end", C.c_int64),
    ]


class MetricsStruct(C.Structure):
    _fields_ = [
        ("valid_user", C.c_uint64),
        ("total_user", C.c_uint64),
        ("hit", C.c_double),
        ("recall", C.c_double),
        ("ndcg", C.c_double),
        ("precision", C.c_double),
        ("map", C.c_double),
    ]


class EvalStatsStruct(C.Structure):  # irs_eval_stats
    _fields_ = [
        ("path", C.c_int32),
        ("hard_rows", C.c_int32),
        ("tiles_total", C.c_int64),
        ("tiles_scored", C.c_int64),
        ("sample_items", C.c_int64),
        ("call_ms", C.c_double),
        ("device_span_ms", C.c_double),
    ]


class CeilingsStruct(C.Structure):  # irs_ceilings
    _fields_ = [
        ("copy_gbs", C.c_double),
        ("triad_gbs", C.c_double),
        ("mfma_f32_tflops", C.c_double),
        ("lds_atomic_u32_gops", C.c_double),
        ("clock_mhz", C.c_double),
        ("n_cu", C.c_int32),
        ("gather256_gbs", C.c_double),
        ("gather512_gbs", C.c_double),
    ]


ABI_VERSION = 4  # IRS_ABI_VERSION of include/irspack_amd.h
# IRS_EXCHANGE_* of include/irspack_amd.h: how irs_ials_sharded_step moves the solved rows
EXCHANGE_MODES = {"auto": 0, "broadcast": 1, "mesh": 2, "peer": 3}
COMM_HANDLE_BYTES = 256

# every symbol include/irspack_amd.h declares
EXPORTED_SYMBOLS = [
    "irs_last_error",
    "irs_abi_version",
    "irs_device_count",
    "irs_ials_create",
    "irs_ials_create_from_factors",
    "irs_ials_destroy",
    "irs_ials_step",
    "irs_ials_get_factor",
    "irs_ials_set_factor",
    "irs_ials_user_scores",
    "irs_ials_transform",
    "irs_ials_transform_with_prior",
    "irs_ials_set_prior",
    "irs_ials_set_features",
    "irs_ials_apply_feature_prior",
    "irs_ials_feature_rhs",
    "irs_ials_compute_loss",
    "irs_ials_set_stream",
    "irs_ials_device_buffer",
    "irs_ials_copy_rows_async",
    "irs_ials_partial_gramian_async",
    "irs_ials_finish_gramian_async",
    "irs_ials_gramian_async",
    "irs_ials_half_step_async",
    "irs_ials_synchronize",
    "irs_comm_unique_id",
    "irs_comm_create",
    "irs_comm_create_local",
    "irs_comm_destroy",
    "irs_comm_export",
    "irs_comm_attach",
    "irs_comm_set_exchange",
    "irs_comm_get_exchange",
    "irs_ials_sharded_step",
    "irs_ials_last_eigenbasis",
    "irs_ials_eigen_debug",
    "irs_ials_profile",
    "irs_ials_profile_read",
    "irs_knn_create",
    "irs_knn_weight",
    "irs_knn_destroy",
    "irs_knn_compute",
    "irs_knn_fetch",
    "irs_knn_fetch_csc",
    "irs_knn_last_stats",
    "irs_knn_last_walked",
    "irs_remove_diagonal",
    "irs_retrieve_recommend",
    "irs_eval_create",
    "irs_eval_destroy",
    "irs_eval_get_metrics",
    "irs_eval_get_metrics_masked",
    "irs_eval_get_metrics_similarity",
    "irs_eval_get_metrics_ials",
    "irs_eval_cache_mask",
    "irs_eval_last_stats",
    "irs_fingerprint",
    "irs_measure_ceilings",
]

_lib: Optional[C.CDLL] = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `make -C irspack_amd/csrc` "
                "(or `python -c 'import __graft_entry__ as g; g.build()'`). "
                "irspack_amd has no fallback implementation."
            )
        _lib = C.CDLL(LIB_PATH)
        _lib.irs_last_error.restype = C.c_char_p
        _lib.irs_abi_version.restype = C.c_int32
        _lib.irs_device_count.restype = C.c_int32
        if _lib.irs_abi_version() != ABI_VERSION:
            found = _lib.irs_abi_version()
            _lib = None
            raise RuntimeError(
                f"{LIB_PATH} has struct layout version {found}, this package was written against "
                f"{ABI_VERSION} (include/irspack_amd.h): rebuild it with `make -C irspack_amd/csrc`.")
    return _lib


def check(status: int) -> None:
    """Map an irs_status to the exception the reference raises."""
    if status == 0:
        return
    msg = lib().irs_last_error().decode("utf-8", "replace")
    if status == 1:
        raise ValueError(msg)
    raise RuntimeError(msg)


def ptr(a: np.ndarray, ctype):
    return a.ctypes.data_as(C.POINTER(ctype))


def device_count() -> int:
    return int(lib().irs_device_count())


def measure_ceilings(device: Optional[int] = None) -> dict:
    """Measured device ceilings (irs_measure_ceilings) as a dict."""
    st = CeilingsStruct()
    check(lib().irs_measure_ceilings(C.c_int32(default_device() if device is None else device),
                                     C.byref(st)))
    return {name: getattr(st, name) for name, _ in CeilingsStruct._fields_}


def default_device() -> int:
    """LOCAL_RANK when launched one process per GPU, else IRSPACK_AMD_DEVICE or 0."""
    for key in ("IRSPACK_AMD_DEVICE", "LOCAL_RANK"):
        v = os.environ.get(key)
        if v is not None:
            try:
                return int(v)
            except ValueError:
                pass
    return 0


LAYOUT_CSR, LAYOUT_CSC = 0, 1  # IRS_LAYOUT_* of include/irspack_amd.h
WEIGHT_NONE, WEIGHT_TF_IDF, WEIGHT_BM25 = 0, 1, 2  # IRS_WEIGHT_*


class KnnInputStruct(C.Structure):  # irs_knn_input
    _fields_ = [
        ("layout", C.c_int32),
        ("weighting", C.c_int32),
        ("smooth", C.c_int32),
        ("reserved", C.c_int32),
        ("k1", C.c_double),
        ("b", C.c_double),
    ]


def sparse_arrays(X, dtype):
    """scipy sparse -> (matrix, layout, indptr int64, indices int32, data dtype) WITHOUT changing the
    layout: a CSC matrix (what ``X.T`` of a CSR is - a view of the same arrays) is handed to the library
    as it is (IRS_LAYOUT_CSC), like nanobind's Eigen caster takes either; anything else becomes CSR.
    Indices sorted (a transposed view inherits the flag of its base)."""
    import scipy.sparse as sps

    if sps.isspmatrix_csc(X):
        layout = LAYOUT_CSC
    else:
        layout = LAYOUT_CSR
        if not sps.isspmatrix_csr(X):
            X = sps.csr_matrix(X)
    if not X.has_sorted_indices:
        X = X.sorted_indices()
    indptr = np.ascontiguousarray(X.indptr, dtype=np.int64)
    indices = np.ascontiguousarray(X.indices, dtype=np.int32)
    data = np.ascontiguousarray(X.data, dtype=dtype)
    return X, layout, indptr, indices, data


def csr_arrays(X, dtype):
    """scipy sparse -> (csr, indptr int64, indices int32, data dtype) with sorted indices."""
    import scipy.sparse as sps

    # (an existing CSR matrix is used as it is: scipy caches its "sorted" flag on the object, a
    # fresh wrapper would rescan the indices - 4 ms for 20 M entries - on every call)
    if not sps.isspmatrix_csr(X):
        X = sps.csr_matrix(X)
    if not X.has_sorted_indices:
        X = X.sorted_indices()
    indptr = np.ascontiguousarray(X.indptr, dtype=np.int64)
    indices = np.ascontiguousarray(X.indices, dtype=np.int32)
    data = np.ascontiguousarray(X.data, dtype=dtype)
    return X, indptr, indices, data
