"""ORACLE — TEST INFRASTRUCTURE ONLY.

ctypes binding of ``oracle/liboracle.so`` (the CPU restatement of the
reference's iALS / kNN / evaluator path; see the headers of ``oracle/*.cpp``).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this package.  Nothing under ``irspack_amd/`` imports it.

Parity pinning: the reference can be neither compiled nor imported in the build
container, so this oracle is pinned by the closed-form float64 checks the
reference's own tests hold (tests/test_oracle_*.py restate them).
"""

import ctypes as C
import os
import subprocess
from typing import List, Optional, Sequence, Tuple

import numpy as np
import scipy.sparse as sps

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liboracle.so")


def build(force: bool = False) -> str:
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith(".cpp")]
    stale = force or not os.path.exists(_SO) or any(
        os.path.getmtime(s) > os.path.getmtime(_SO) for s in srcs
    )
    if stale:
        subprocess.check_call(["make", "-s", "-C", _HERE, "-B"])
    return _SO


_lib: Optional[C.CDLL] = None


def use_fast_build() -> str:
    """bench.py's cpu_baseline only: (re)builds ``oracle/_fast/liboracle_fast.so`` ON THIS BOX
    (``make fast``: -O3 -march=native -ffp-contract=fast, register-blocked rank update - the
    same algorithm tuned like a CPU BLAS would be, NOT the parity oracle) and makes this
    process's oracle calls go to it.  Never call it from a parity test."""
    global _lib
    subprocess.check_call(["make", "-s", "-C", _HERE, "fast"])
    so = os.path.join(_HERE, "_fast", "liboracle_fast.so")
    _lib = None
    _load(so)
    return so


def lib() -> C.CDLL:
    if _lib is None:
        build()
        _load(_SO)
    return _lib


def _load(so: str) -> None:
    global _lib
    _lib = C.CDLL(so)
    _lib.orc_last_error.restype = C.c_char_p
    _lib.orc_knn_last_error.restype = C.c_char_p
    _lib.orc_eval_last_error.restype = C.c_char_p
    _lib.orc_ials_create.restype = C.c_void_p
    _lib.orc_ials_user_ptr.restype = C.POINTER(C.c_float)
    _lib.orc_ials_item_ptr.restype = C.POINTER(C.c_float)
    _lib.orc_metrics_create.restype = C.c_void_p


def use_parity_build() -> None:
    """back to ``liboracle.so`` (the parity oracle) after ``use_fast_build``"""
    global _lib
    _lib = None
    lib()


class ModelConfig(C.Structure):
    _fields_ = [
        ("K", C.c_uint64),
        ("alpha0", C.c_float),
        ("reg", C.c_float),
        ("nu", C.c_float),
        ("init_stdev", C.c_float),
        ("random_seed", C.c_int32),
        ("loss_type", C.c_int32),
    ]


class SolverConfig(C.Structure):
    _fields_ = [
        ("n_threads", C.c_uint64),
        ("solver_type", C.c_int32),
        ("max_cg_steps", C.c_uint64),
        ("ialspp_subspace_dimension", C.c_uint64),
        ("ialspp_iteration", C.c_uint64),
    ]


LOSS = {"ORIGINAL": 0, "IALSPP": 1}
SOLVER = {"CHOLESKY": 0, "CG": 1, "IALSPP": 2}


def _check(rc: int, errfn) -> None:
    if rc == 0:
        return
    msg = errfn().decode()
    if rc == 1:
        raise ValueError(msg)
    raise RuntimeError(msg)


def _csr_args(X: sps.csr_matrix, dtype):
    X = sps.csr_matrix(X)
    X.sort_indices()
    indptr = np.ascontiguousarray(X.indptr, dtype=np.int64)
    indices = np.ascontiguousarray(X.indices, dtype=np.int32)
    data = np.ascontiguousarray(X.data, dtype=dtype)
    return X, indptr, indices, data


def _p(a: np.ndarray, t):
    return a.ctypes.data_as(C.POINTER(t))


def model_config(K, alpha0=0.1, reg=0.1, nu=1.0, init_stdev=0.1, random_seed=42,
                 loss_type="IALSPP") -> ModelConfig:
    return ModelConfig(K, alpha0, reg, nu, init_stdev, random_seed, LOSS[loss_type])


def solver_config(n_threads=1, solver_type="CG", max_cg_steps=3,
                  ialspp_subspace_dimension=64, ialspp_iteration=1) -> SolverConfig:
    return SolverConfig(n_threads, SOLVER[solver_type], max_cg_steps,
                        ialspp_subspace_dimension, ialspp_iteration)


def ials_init(rows: int, K: int, init_stdev: float, seed: int) -> np.ndarray:
    out = np.empty((rows, K), dtype=np.float32)
    _check(lib().orc_ials_init(_p(out, C.c_float), C.c_int64(rows), C.c_int64(K),
                               C.c_float(init_stdev), C.c_int32(seed)),
           lib().orc_last_error)
    return out


def ials_gramian(F: np.ndarray, alpha0: float, n_threads: int = 1) -> np.ndarray:
    F = np.ascontiguousarray(F, dtype=np.float32)
    n, K = F.shape
    P = np.empty((K, K), dtype=np.float32)
    _check(lib().orc_ials_gramian(_p(F, C.c_float), C.c_int64(n), C.c_int64(K),
                                  C.c_float(alpha0), C.c_uint64(n_threads),
                                  _p(P, C.c_float)), lib().orc_last_error)
    return P


def ials_solver_step(target: np.ndarray, X: sps.csr_matrix, other: np.ndarray,
                     P: np.ndarray, mc: ModelConfig, sc: SolverConfig,
                     row_begin: int = 0, row_end: Optional[int] = None) -> np.ndarray:
    """One Solver::step (hpp:664-679) on a copy of ``target``; returns the copy."""
    X, indptr, indices, data = _csr_args(X, np.float32)
    tgt = np.array(target, dtype=np.float32, order="C", copy=True)
    other = np.ascontiguousarray(other, dtype=np.float32)
    P = np.ascontiguousarray(P, dtype=np.float32)
    if row_end is None:
        row_end = X.shape[0]
    _check(lib().orc_ials_solver_step(
        _p(tgt, C.c_float), C.c_int64(X.shape[0]), C.c_int64(X.shape[1]),
        _p(indptr, C.c_int64), _p(indices, C.c_int32), _p(data, C.c_float),
        _p(other, C.c_float), _p(P, C.c_float), C.byref(mc), C.byref(sc),
        C.c_int64(row_begin), C.c_int64(row_end)), lib().orc_last_error)
    return tgt


# ---- the float64 arbiter (liboracle_f64.so: ials_oracle.cpp compiled with Real = double) ----
_SO64 = os.path.join(_HERE, "liboracle_f64.so")
_lib64: Optional[C.CDLL] = None


def lib64() -> C.CDLL:
    global _lib64
    if _lib64 is None:
        src = os.path.join(_HERE, "ials_oracle.cpp")
        if not os.path.exists(_SO64) or os.path.getmtime(src) > os.path.getmtime(_SO64):
            subprocess.check_call(["make", "-s", "-C", _HERE, "-B", "f64"])
        _lib64 = C.CDLL(_SO64)
        _lib64.orc_last_error.restype = C.c_char_p
    return _lib64


def ials_gramian_f64(F: np.ndarray, alpha0: float, n_threads: int = 1) -> np.ndarray:
    """prepare_p (hpp:78-115) of the float32 factors ``F`` in float64."""
    F = np.ascontiguousarray(F, dtype=np.float64)
    n, K = F.shape
    P = np.empty((K, K), dtype=np.float64)
    _check(lib64().orc_ials_gramian(_p(F, C.c_double), C.c_int64(n), C.c_int64(K),
                                    C.c_float(alpha0), C.c_uint64(n_threads),
                                    _p(P, C.c_double)), lib64().orc_last_error)
    return P


def ials_solver_step_f64(target: np.ndarray, X: sps.csr_matrix, other: np.ndarray,
                         P: Optional[np.ndarray], mc: ModelConfig, sc: SolverConfig,
                         n_threads: int = 1) -> np.ndarray:
    """One Solver::step (hpp:664-679) - the SAME restatement as ``ials_solver_step``, same
    iteration, exits and regulariser - with factors, Gramian and every intermediate in float64:
    what both float32 implementations (this oracle's and the GPU's) approximate.  ``P`` = None
    computes the float64 Gramian of ``other``.  Returns float64 rows."""
    X, indptr, indices, data = _csr_args(X, np.float32)
    tgt = np.array(target, dtype=np.float64, order="C", copy=True)
    other = np.ascontiguousarray(other, dtype=np.float64)
    if P is None:
        P = ials_gramian_f64(other, mc.alpha0, n_threads)
    P = np.ascontiguousarray(P, dtype=np.float64)
    _check(lib64().orc_ials_solver_step(
        _p(tgt, C.c_double), C.c_int64(X.shape[0]), C.c_int64(X.shape[1]),
        _p(indptr, C.c_int64), _p(indices, C.c_int32), _p(data, C.c_float),
        _p(other, C.c_double), _p(P, C.c_double), C.byref(mc), C.byref(sc),
        C.c_int64(0), C.c_int64(X.shape[0])), lib64().orc_last_error)
    return tgt


class IALSTrainer:
    """Restatement of irspack::ials::IALSTrainer (hpp:709-984)."""

    def __init__(self, mc: ModelConfig, X: sps.csr_matrix):
        X, indptr, indices, data = _csr_args(X, np.float32)
        self.mc = mc
        self.n_users, self.n_items = X.shape
        self.K = int(mc.K)
        self._h = C.c_void_p(lib().orc_ials_create(
            C.byref(mc), C.c_int64(X.shape[0]), C.c_int64(X.shape[1]),
            _p(indptr, C.c_int64), _p(indices, C.c_int32), _p(data, C.c_float)))

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_ials_destroy(self._h)
            self._h = None

    def _view(self, which: str) -> np.ndarray:
        f = lib().orc_ials_user_ptr if which == "user" else lib().orc_ials_item_ptr
        n = self.n_users if which == "user" else self.n_items
        ptr = f(self._h)
        if n * self.K == 0:
            return np.zeros((n, self.K), dtype=np.float32)
        return np.ctypeslib.as_array(ptr, shape=(n, self.K))

    @property
    def user(self) -> np.ndarray:
        return self._view("user").copy()

    @user.setter
    def user(self, v: np.ndarray) -> None:
        self._view("user")[...] = v

    @property
    def item(self) -> np.ndarray:
        return self._view("item").copy()

    @item.setter
    def item(self, v: np.ndarray) -> None:
        self._view("item")[...] = v

    def step(self, sc: SolverConfig) -> None:
        _check(lib().orc_ials_step(self._h, C.byref(sc)), lib().orc_last_error)

    def _transform(self, side: int, X, sc: SolverConfig) -> np.ndarray:
        X, indptr, indices, data = _csr_args(X, np.float32)
        m = X.shape[0] if side == 0 else X.shape[1]
        out = np.zeros((m, self.K), dtype=np.float32)
        _check(lib().orc_ials_transform(
            self._h, C.c_int(side), C.c_int64(X.shape[0]), C.c_int64(X.shape[1]),
            _p(indptr, C.c_int64), _p(indices, C.c_int32), _p(data, C.c_float),
            C.byref(sc), _p(out, C.c_float)), lib().orc_last_error)
        return out

    def transform_user(self, X, sc: SolverConfig) -> np.ndarray:
        return self._transform(0, X, sc)

    def transform_item(self, X, sc: SolverConfig) -> np.ndarray:
        return self._transform(1, X, sc)

    def compute_loss(self, sc: SolverConfig) -> float:
        out = C.c_float(0)
        _check(lib().orc_ials_compute_loss(self._h, C.byref(sc), C.byref(out)),
               lib().orc_last_error)
        return float(out.value)

    def user_scores(self, begin: int, end: int, sc: SolverConfig) -> np.ndarray:
        out = np.empty((max(end - begin, 0), self.n_items), dtype=np.float32)
        _check(lib().orc_ials_user_scores(self._h, C.c_int64(begin), C.c_int64(end),
                                          C.byref(sc), _p(out, C.c_float)),
               lib().orc_last_error)
        return out


# ---------------------------------------------------------------- kNN
SIM = {"cosine": 0, "asymmetric": 1, "jaccard": 2, "tversky": 3, "p3alpha": 4,
       "rp3beta": 5}


class KNNComputer:
    """Restatement of KNN::KNNComputer<double, Sim> (knn.hpp, similarities.hpp)."""

    def __init__(self, kind: str, X, shrinkage: float = 0.0, alpha: float = 0.0,
                 beta: float = 0.0, normalize: bool = False, n_threads: int = 1,
                 max_chunk_size: int = 128):
        X, indptr, indices, data = _csr_args(X, np.float64)
        self.kind = kind
        self.N = X.shape[0]
        h = C.c_void_p()
        _check(lib().orc_knn_create(
            C.c_int32(SIM[kind]), C.c_int64(X.shape[0]), C.c_int64(X.shape[1]),
            _p(indptr, C.c_int64), _p(indices, C.c_int32), _p(data, C.c_double),
            C.c_double(shrinkage), C.c_double(alpha), C.c_double(beta),
            C.c_int32(bool(normalize)), C.c_int64(n_threads), C.c_int64(max_chunk_size),
            C.byref(h)), lib().orc_knn_last_error)
        self._h = h

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_knn_destroy(self._h)
            self._h = None

    def _compute(self, X, top_k: int, as_w: bool) -> sps.csr_matrix:
        X, indptr, indices, data = _csr_args(X, np.float64)
        nnz = C.c_int64(0)
        _check(lib().orc_knn_compute(
            self._h, C.c_int64(X.shape[0]), C.c_int64(X.shape[1]),
            _p(indptr, C.c_int64), _p(indices, C.c_int32), _p(data, C.c_double),
            C.c_int64(top_k), C.c_int32(as_w), C.byref(nnz)), lib().orc_knn_last_error)
        o_indptr = np.empty(X.shape[0] + 1, dtype=np.int64)
        o_indices = np.empty(nnz.value, dtype=np.int32)
        o_data = np.empty(nnz.value, dtype=np.float64)
        lib().orc_knn_fetch(_p(o_indptr, C.c_int64), _p(o_indices, C.c_int32),
                            _p(o_data, C.c_double))
        return sps.csr_matrix((o_data, o_indices, o_indptr), shape=(X.shape[0], self.N))

    def compute_similarity(self, X, top_k: int) -> sps.csr_matrix:
        return self._compute(X, top_k, False)

    def compute_W(self, X, top_k: int) -> sps.csc_matrix:
        return self._compute(X, top_k, True).T.tocsc()


def remove_diagonal(X) -> sps.csr_matrix:
    X, indptr, indices, data = _csr_args(X, np.float64)
    data = data.copy()
    _check(lib().orc_remove_diagonal(C.c_int64(X.shape[0]), C.c_int64(X.shape[1]),
                                     _p(indptr, C.c_int64), _p(indices, C.c_int32),
                                     _p(data, C.c_double)), lib().orc_knn_last_error)
    return sps.csr_matrix((data, indices.copy(), indptr.copy()), shape=X.shape)


def tf_idf_weight(X, smooth: bool = True) -> sps.csr_matrix:
    X, indptr, indices, data = _csr_args(X, np.float64)
    data = data.copy()
    _check(lib().orc_tf_idf_weight(C.c_int64(X.shape[0]), C.c_int64(X.shape[1]),
                                   _p(indptr, C.c_int64), _p(indices, C.c_int32),
                                   _p(data, C.c_double), C.c_int32(smooth)),
           lib().orc_knn_last_error)
    return sps.csr_matrix((data, indices.copy(), indptr.copy()), shape=X.shape)


def okapi_BM_25_weight(X, k1: float = 1.2, b: float = 0.75) -> sps.csr_matrix:
    X, indptr, indices, data = _csr_args(X, np.float64)
    data = data.copy()
    _check(lib().orc_bm25_weight(C.c_int64(X.shape[0]), C.c_int64(X.shape[1]),
                                 _p(indptr, C.c_int64), _p(indices, C.c_int32),
                                 _p(data, C.c_double), C.c_double(k1), C.c_double(b)),
           lib().orc_knn_last_error)
    return sps.csr_matrix((data, indices.copy(), indptr.copy()), shape=X.shape)


# ---------------------------------------------------------------- evaluator
METRIC_KEYS = ["total_user", "valid_user", "n_items", "hit", "ndcg", "recall", "map",
               "precision", "appeared_item", "entropy", "gini_index"]


class Metrics:
    def __init__(self, n_item: int, _h=None):
        self.n_item = n_item
        self._h = C.c_void_p(lib().orc_metrics_create(C.c_int64(n_item))) if _h is None else _h

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_metrics_destroy(self._h)
            self._h = None

    def merge(self, other: "Metrics") -> None:
        lib().orc_metrics_merge(self._h, other._h)

    def as_dict(self) -> dict:
        out = np.empty(11, dtype=np.float64)
        lib().orc_metrics_as_array(self._h, _p(out, C.c_double))
        return dict(zip(METRIC_KEYS, out.tolist()))

    def item_cnt(self) -> np.ndarray:
        out = np.empty(self.n_item, dtype=np.int64)
        lib().orc_metrics_item_cnt(self._h, _p(out, C.c_int64))
        return out

    def raw(self) -> np.ndarray:
        out = np.empty(7, dtype=np.float64)
        lib().orc_metrics_raw(self._h, _p(out, C.c_double))
        return out


def _ragged(lists: Sequence[Sequence[int]]) -> Tuple[np.ndarray, np.ndarray]:
    ptr = np.zeros(len(lists) + 1, dtype=np.int64)
    for i, l in enumerate(lists):
        ptr[i + 1] = ptr[i] + len(l)
    flat = np.fromiter((x for l in lists for x in l), dtype=np.int64, count=int(ptr[-1]))
    return ptr, flat


class EvaluatorCore:
    """Restatement of irspack::evaluation::EvaluatorCore (evaluator.cpp:181-374)."""

    def __init__(self, ground_truth, recommendable: List[List[int]]):
        X, indptr, indices, _ = _csr_args(ground_truth, np.float64)
        self.n_users, self.n_items = X.shape
        ptr, flat = _ragged(recommendable)
        if flat.size == 0:
            flat = np.zeros(1, dtype=np.int64)
        h = C.c_void_p()
        _check(lib().orc_eval_create(
            C.c_int64(X.shape[0]), C.c_int64(X.shape[1]), _p(indptr, C.c_int64),
            _p(indices, C.c_int32), C.c_int64(len(recommendable)), _p(ptr, C.c_int64),
            _p(flat, C.c_int64), C.byref(h)), lib().orc_eval_last_error)
        self._h = h

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_eval_destroy(self._h)
            self._h = None

    def _get(self, scores: np.ndarray, is_f64: bool, cutoff, offset, n_threads,
             recall_with_cutoff) -> Metrics:
        scores = np.ascontiguousarray(scores, dtype=np.float64 if is_f64 else np.float32)
        assert scores.ndim == 2 and scores.shape[1] == self.n_items
        out = C.c_void_p()
        _check(lib().orc_eval_get_metrics(
            self._h, C.c_int32(is_f64), scores.ctypes.data_as(C.c_void_p),
            C.c_int64(scores.shape[0]), C.c_int64(cutoff), C.c_int64(offset),
            C.c_int64(n_threads), C.c_int32(bool(recall_with_cutoff)), C.byref(out)),
            lib().orc_eval_last_error)
        return Metrics(self.n_items, _h=out)

    def get_metrics_f64(self, scores, cutoff, offset, n_threads, recall_with_cutoff=False):
        return self._get(scores, True, cutoff, offset, n_threads, recall_with_cutoff)

    def get_metrics_f32(self, scores, cutoff, offset, n_threads, recall_with_cutoff=False):
        return self._get(scores, False, cutoff, offset, n_threads, recall_with_cutoff)


def evaluate_list_vs_list(recommendations, ground_truths, n_items: int,
                          n_threads: int = 1) -> Metrics:
    if len(recommendations) != len(ground_truths):
        raise ValueError("recommendation array and ground_truth array has different size.")
    rp, rf = _ragged(recommendations)
    gp, gf = _ragged(ground_truths)
    if rf.size == 0:
        rf = np.zeros(1, dtype=np.int64)
    if gf.size == 0:
        gf = np.zeros(1, dtype=np.int64)
    out = C.c_void_p()
    _check(lib().orc_eval_list_vs_list(
        C.c_int64(len(recommendations)), _p(rp, C.c_int64), _p(rf, C.c_int64),
        _p(gp, C.c_int64), _p(gf, C.c_int64), C.c_int64(n_items), C.c_int64(n_threads),
        C.byref(out)), lib().orc_eval_last_error)
    return Metrics(n_items, _h=out)
