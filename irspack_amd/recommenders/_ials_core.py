"""Counterpart of the reference's nanobind module ``irspack.recommenders._ials_core``
(/root/reference/cpp_source/als/wrapper.cpp:19-182, stub
src/irspack/recommenders/_ials_core.pyi), backed by ``libirspack_amd.so``.

Same class names, argument meaning, pickle tuples and error behaviour; the
arithmetic runs in the HIP kernels of ``irspack_amd/csrc``.
"""

import ctypes as C
import enum
from typing import Any, Dict, List, Optional, Tuple

import numpy as np
import scipy.sparse as sps

from .. import _lib
from .._lib import ModelConfigStruct, ShardStruct, SolverConfigStruct, check, lib, ptr


class LossType(enum.Enum):  # IALSLearningConfig.hpp:11, wrapper.cpp:25-27
    ORIGINAL = 0
    IALSPP = 1


class SolverType(enum.Enum):  # IALSLearningConfig.hpp:12, wrapper.cpp:29-32
    CHOLESKY = 0
    CG = 1
    IALSPP = 2


# legacy module-level aliases; IALSPP is the *SolverType* (wrapper.cpp:34-40)
ORIGINAL = LossType.ORIGINAL
CHOLESKY = SolverType.CHOLESKY
CG = SolverType.CG
IALSPP = SolverType.IALSPP


class IALSModelConfig:  # IALSLearningConfig.hpp:15-31, wrapper.cpp:42-71
    def __init__(
        self,
        K: int,
        alpha0: float,
        reg: float,
        nu: float,
        init_stdev: float,
        random_seed: int,
        loss_type: LossType,
        lambda_user_feature: float = 0.0,
        lambda_item_feature: float = 0.0,
        feature_warmup_epochs: int = 0,
    ) -> None:
        if int(K) < 0 or int(feature_warmup_epochs) < 0:
            raise TypeError("K and feature_warmup_epochs must be non-negative (size_t).")
        self.K = int(K)
        self.alpha0 = float(alpha0)
        self.reg = float(reg)
        self.nu = float(nu)
        self.init_stdev = float(init_stdev)
        self.random_seed = int(random_seed)
        self.loss_type = LossType(loss_type)
        self.lambda_user_feature = float(lambda_user_feature)
        self.lambda_item_feature = float(lambda_item_feature)
        self.feature_warmup_epochs = int(feature_warmup_epochs)

    def __getstate__(self) -> tuple:
        return (
            self.K,
            float(np.float32(self.alpha0)),
            float(np.float32(self.reg)),
            float(np.float32(self.nu)),
            float(np.float32(self.init_stdev)),
            self.random_seed,
            self.loss_type,
            float(np.float32(self.lambda_user_feature)),
            float(np.float32(self.lambda_item_feature)),
            self.feature_warmup_epochs,
        )

    def __setstate__(self, state: tuple) -> None:
        self.__init__(*state)  # type: ignore[misc]

    def _struct(self) -> ModelConfigStruct:
        return ModelConfigStruct(
            self.K,
            self.alpha0,
            self.reg,
            self.nu,
            self.init_stdev,
            self.random_seed,
            self.loss_type.value,
            self.lambda_user_feature,
            self.lambda_item_feature,
            self.feature_warmup_epochs,
        )


class IALSModelConfigBuilder:  # IALSLearningConfig.hpp:33-94, wrapper.cpp:72-90
    def __init__(self) -> None:
        self.reg = 0.1
        self.alpha0 = 0.1
        self.nu = 1.0
        self.init_stdev = 0.1
        self.K = 16
        self.random_seed = 42
        self.loss_type = LossType.IALSPP
        self.lambda_user_feature = 0.0
        self.lambda_item_feature = 0.0
        self.feature_warmup_epochs = 0

    def build(self) -> IALSModelConfig:
        return IALSModelConfig(
            self.K,
            self.alpha0,
            self.reg,
            self.nu,
            self.init_stdev,
            self.random_seed,
            self.loss_type,
            self.lambda_user_feature,
            self.lambda_item_feature,
            self.feature_warmup_epochs,
        )

    def set_K(self, K: int) -> "IALSModelConfigBuilder":
        self.K = K
        return self

    def set_alpha0(self, alpha0: float) -> "IALSModelConfigBuilder":
        self.alpha0 = alpha0
        return self

    def set_reg(self, reg: float) -> "IALSModelConfigBuilder":
        self.reg = reg
        return self

    def set_nu(self, nu: float) -> "IALSModelConfigBuilder":
        self.nu = nu
        return self

    def set_init_stdev(self, init_stdev: float) -> "IALSModelConfigBuilder":
        self.init_stdev = init_stdev
        return self

    def set_random_seed(self, random_seed: int) -> "IALSModelConfigBuilder":
        self.random_seed = random_seed
        return self

    def set_loss_type(self, loss_type: LossType) -> "IALSModelConfigBuilder":
        self.loss_type = loss_type
        return self

    def set_lambda_user_feature(self, value: float) -> "IALSModelConfigBuilder":
        self.lambda_user_feature = value
        return self

    def set_lambda_item_feature(self, value: float) -> "IALSModelConfigBuilder":
        self.lambda_item_feature = value
        return self

    def set_feature_warmup_epochs(self, value: int) -> "IALSModelConfigBuilder":
        self.feature_warmup_epochs = value
        return self


class IALSSolverConfig:  # IALSLearningConfig.hpp:97-112, wrapper.cpp:92-115
    def __init__(
        self,
        n_threads: int,
        solver_type: SolverType,
        max_cg_steps: int,
        ialspp_subspace_dimension: int,
        ialspp_iteration: int,
    ) -> None:
        for v in (n_threads, max_cg_steps, ialspp_subspace_dimension, ialspp_iteration):
            if int(v) < 0:
                raise TypeError("size_t arguments must be non-negative.")
        self.n_threads = int(n_threads)
        self.solver_type = SolverType(solver_type)
        self.max_cg_steps = int(max_cg_steps)
        self.ialspp_subspace_dimension = int(ialspp_subspace_dimension)
        self.ialspp_iteration = int(ialspp_iteration)

    def __getstate__(self) -> tuple:
        return (
            self.n_threads,
            self.solver_type,
            self.max_cg_steps,
            self.ialspp_subspace_dimension,
            self.ialspp_iteration,
        )

    def __setstate__(self, state: tuple) -> None:
        self.__init__(*state)  # type: ignore[misc]

    def _struct(self) -> SolverConfigStruct:
        return SolverConfigStruct(
            self.n_threads,
            self.solver_type.value,
            self.max_cg_steps,
            self.ialspp_subspace_dimension,
            self.ialspp_iteration,
        )


class IALSSolverConfigBuilder:  # IALSLearningConfig.hpp:114-147, wrapper.cpp:117-128
    def __init__(self) -> None:
        self.n_threads = 1
        self.solver_type = SolverType.CG
        self.max_cg_steps = 3
        self.ialspp_subspace_dimension = 64
        self.ialspp_iteration = 1

    def build(self) -> IALSSolverConfig:
        return IALSSolverConfig(
            self.n_threads,
            self.solver_type,
            self.max_cg_steps,
            self.ialspp_subspace_dimension,
            self.ialspp_iteration,
        )

    def set_n_threads(self, n_threads: int) -> "IALSSolverConfigBuilder":
        self.n_threads = n_threads
        return self

    def set_solver_type(self, solver_type: SolverType) -> "IALSSolverConfigBuilder":
        self.solver_type = solver_type
        return self

    def set_max_cg_steps(self, max_cg_steps: int) -> "IALSSolverConfigBuilder":
        self.max_cg_steps = max_cg_steps
        return self

    def set_ialspp_subspace_dimension(self, ialspp_subspace_dimension: int) -> "IALSSolverConfigBuilder":
        self.ialspp_subspace_dimension = ialspp_subspace_dimension
        return self

    def set_ialspp_iteration(self, ialspp_iteration: int) -> "IALSSolverConfigBuilder":
        self.ialspp_iteration = ialspp_iteration
        return self


def _as_feature(f: Any, rows: int) -> Any:
    """FeatureMatrix (dense or CSR float32, hpp:693-700); ``None`` is a zero-column matrix."""
    if f is None:
        return np.zeros((rows, 0), dtype=np.float32)
    if sps.issparse(f):
        return sps.csr_matrix(f, dtype=np.float32)
    f = np.ascontiguousarray(f, dtype=np.float32)
    if f.ndim != 2:
        raise TypeError("feature matrix must be 2-dimensional.")
    return f


_BLAS_CONTROLLER: Any = None


def _single_blas_thread() -> Any:
    """Context manager that keeps the host BLAS on one thread (no-op without threadpoolctl)."""
    global _BLAS_CONTROLLER
    if _BLAS_CONTROLLER is None:
        try:
            from threadpoolctl import ThreadpoolController

            _BLAS_CONTROLLER = ThreadpoolController()
        except Exception:  # pragma: no cover - threadpoolctl is optional
            _BLAS_CONTROLLER = False
    if _BLAS_CONTROLLER is False:
        import contextlib

        return contextlib.nullcontext()
    return _BLAS_CONTROLLER.limit(limits=1, user_api="blas")


class IALSTrainer:
    """``IALSTrainer(model_config, interaction)`` — wrapper.cpp:130-181, hpp:709-984.

    Factors live on the GPU; ``.user`` / ``.item`` copy to / from the host.
    Extra keyword-only arguments (not in the reference): ``device`` selects the
    GPU, ``shard`` = (user_begin, user_end, item_begin, item_end) restricts the
    rows this handle solves (multi-GPU host loop, see ``irspack_amd.sharding``).
    """

    def __init__(
        self,
        model_config: IALSModelConfig,
        interaction: Any,
        user_feature: Any = None,
        item_feature: Any = None,
        *,
        device: Optional[int] = None,
        shard: Optional[Tuple[int, int, int, int]] = None,
    ) -> None:
        if not sps.issparse(interaction):
            raise TypeError("interaction must be a scipy sparse matrix.")
        X, indptr, indices, data = _lib.csr_arrays(interaction, np.float32)
        self._config = model_config
        self._device = _lib.default_device() if device is None else int(device)
        self._n_users, self._n_items = int(X.shape[0]), int(X.shape[1])
        self._K = int(model_config.K)
        cfg = model_config._struct()
        h = C.c_void_p()
        sh = None if shard is None else ShardStruct(*[int(v) for v in shard])
        check(
            lib().irs_ials_create(
                C.byref(cfg),
                C.c_int64(self._n_users),
                C.c_int64(self._n_items),
                ptr(indptr, C.c_int64),
                ptr(indices, C.c_int32),
                ptr(data, C.c_float),
                C.c_int32(self._device),
                None if sh is None else C.byref(sh),
                C.byref(h),
            )
        )
        self._h: Optional[C.c_void_p] = h
        self._empty_feature_weight()
        if user_feature is not None or item_feature is not None:
            # feature-aware constructor, hpp:722-744 + initialize_feature_aware :1001-1015
            uf = _as_feature(user_feature, self._n_users)
            itf = _as_feature(item_feature, self._n_items)
            if uf.shape[0] != self._n_users or itf.shape[0] != self._n_items:
                raise ValueError("Feature matrix row count mismatch.")
            if (uf.shape[1] and model_config.lambda_user_feature <= 0) or (
                    itf.shape[1] and model_config.lambda_item_feature <= 0):
                raise ValueError("Feature weight regularization must be positive.")
            self._feature_aware = True
            self._features = [uf, itf]
            self._ufw = np.zeros((uf.shape[1], self._K), dtype=np.float32)
            self._ifw = np.zeros((itf.shape[1], self._K), dtype=np.float32)
            Xc = sps.csr_matrix(X)
            self._row_nnz = [np.diff(Xc.indptr), np.bincount(Xc.indices, minlength=self._n_items)]
            for side, F in enumerate(self._features):  # the products with F run on the device
                if F.shape[1]:
                    Fc, fp, fi, fd = _lib.csr_arrays(sps.csr_matrix(F), np.float32)
                    if fi.size == 0:
                        fi, fd = np.zeros(1, np.int32), np.zeros(1, np.float32)
                    check(lib().irs_ials_set_features(
                        self._h, C.c_int32(side), C.c_int64(Fc.shape[0]), C.c_int64(Fc.shape[1]),
                        ptr(fp, C.c_int64), ptr(fi, C.c_int32), ptr(fd, C.c_float)))

    def _empty_feature_weight(self) -> None:
        self._ufw = np.zeros((0, self._K), dtype=np.float32)
        self._ifw = np.zeros((0, self._K), dtype=np.float32)
        self._feature_aware = False
        self._features = [None, None]
        self._row_nnz = [None, None]
        self._ridge_cache = [None, None]
        self._epoch = 0

    # -- feature-aware pieces (hpp:758-789, 1052-1209).  The per-row solves run on the GPU
    #    with the prior added to the right-hand side; the F x F feature-weight ridge system
    #    is a small host solve in float32 like the reference's.
    def _weight(self, side: int) -> np.ndarray:
        return self._ufw if side == 0 else self._ifw

    def _row_reg(self, side: int, nnz: np.ndarray) -> np.ndarray:
        """Solver::compute_reg (hpp:117-120) in float32 for every row."""
        c = self._config
        n_other = self._n_items if side == 0 else self._n_users
        base = np.float32(c.alpha0) * np.float32(n_other) + nnz.astype(np.float32)
        return (np.float32(c.reg) * np.power(base, np.float32(c.nu), dtype=np.float32)).astype(np.float32)

    def _check_prior_defined(self, side: int, nnz: np.ndarray) -> None:
        # step_with_prior, hpp:639-653
        if self._config.alpha0 == 0:
            empty_reg = self._row_reg(side, np.zeros(1, dtype=np.int64))[0]
            if (not (empty_reg > 0) or not np.isfinite(empty_reg)) and bool((nnz == 0).any()):
                raise ValueError(
                    "Feature-prior embedding is not uniquely defined for an empty interaction "
                    "row when alpha0 and its regularization are zero.")

    def _set_prior(self, side: int, prior: Optional[np.ndarray]) -> None:
        if prior is None:
            check(lib().irs_ials_set_prior(self._h, C.c_int32(side), None))
        else:
            prior = np.ascontiguousarray(prior, dtype=np.float32)
            check(lib().irs_ials_set_prior(self._h, C.c_int32(side), ptr(prior, C.c_float)))

    def _stored_prior(self, side: int) -> np.ndarray:
        return np.asarray(self._features[side] @ self._weight(side), dtype=np.float32)

    def _update_feature_weight(self, side: int) -> None:
        """update_feature_weight (hpp:1069-1209): W = (F^T D F + lambda I)^-1 F^T D factor."""
        F = self._features[side]
        if F.shape[1] == 0:
            return
        lam = self._config.lambda_user_feature if side == 0 else self._config.lambda_item_feature
        import scipy.linalg as sla

        if self._ridge_cache[side] is None:  # initialize_feature_weight_cache, hpp:1085-1132
            w = self._row_reg(side, self._row_nnz[side])
            sw = np.sqrt(w).astype(np.float32)
            WF = F.multiply(sw[:, None]).tocsr() if sps.issparse(F) else F * sw[:, None]
            gram = np.asarray((WF.T @ WF).todense() if sps.issparse(WF) else WF.T @ WF,
                              dtype=np.float32)
            gram[np.diag_indices_from(gram)] += np.float32(lam)
            try:
                chol = sla.cho_factor(gram, lower=False, check_finite=True)
            except (sla.LinAlgError, ValueError):
                raise RuntimeError("Feature ridge Cholesky decomposition failed.")
            self._ridge_cache[side] = (w, chol)
        w, chol = self._ridge_cache[side]
        rhs = np.empty((F.shape[1], self._K), dtype=np.float32)  # F^T (D factor), hpp:1142-1171
        check(lib().irs_ials_feature_rhs(self._h, C.c_int32(side), ptr(rhs, C.c_float)))
        with _single_blas_thread():  # an F x F solve: a BLAS thread team costs more than it saves
            sol = sla.cho_solve(chol, rhs, check_finite=False).astype(np.float32)
        if not np.isfinite(sol).all():
            raise RuntimeError("Feature ridge solve failed.")
        if side == 0:
            self._ufw = sol
        else:
            self._ifw = sol

    @classmethod
    def _from_factors(
        cls, config: IALSModelConfig, user: np.ndarray, item: np.ndarray, device: Optional[int] = None
    ) -> "IALSTrainer":
        self = cls.__new__(cls)
        self._restore(config, user, item, device)
        return self

    def _restore(self, config, user, item, device=None) -> None:
        user = np.ascontiguousarray(user, dtype=np.float32)
        item = np.ascontiguousarray(item, dtype=np.float32)
        if user.ndim != 2 or item.ndim != 2 or user.shape[1] != item.shape[1]:
            raise RuntimeError("Invalid IALSTrainer pickle state.")
        # the deserialising ctor takes K from the matrices (hpp:748)
        cfgK = IALSModelConfig(user.shape[1], *config.__getstate__()[1:])
        self._config = config
        self._device = _lib.default_device() if device is None else int(device)
        self._n_users, self._n_items = int(user.shape[0]), int(item.shape[0])
        self._K = int(user.shape[1])
        cfg = cfgK._struct()
        h = C.c_void_p()
        check(
            lib().irs_ials_create_from_factors(
                C.byref(cfg),
                C.c_int64(self._n_users),
                C.c_int64(self._n_items),
                ptr(user, C.c_float),
                ptr(item, C.c_float),
                C.c_int32(self._device),
                C.byref(h),
            )
        )
        self._h = h
        self._empty_feature_weight()

    def __del__(self) -> None:
        h = getattr(self, "_h", None)
        if h:
            try:
                lib().irs_ials_destroy(h)
            except Exception:
                pass
            self._h = None

    # -- training / scoring -------------------------------------------------
    def step(self, solver_config: IALSSolverConfig) -> None:
        sc = solver_config._struct()
        if self._feature_aware and solver_config.solver_type == SolverType.IALSPP:
            raise ValueError("Feature-aware iALS does not support IALSPP.")  # hpp:759-761
        if self._feature_aware and self._epoch >= self._config.feature_warmup_epochs:
            for side in (0, 1):  # hpp:762-783
                self.partial_gramian_async(side)
                self.finish_gramian_async(side)
                if self._weight(side).shape[0]:
                    self._check_prior_defined(side, self._row_nnz[side])
                    W = np.ascontiguousarray(self._weight(side), dtype=np.float32)
                    check(lib().irs_ials_apply_feature_prior(self._h, C.c_int32(side),
                                                             ptr(W, C.c_float)))
                    try:
                        self.half_step_async(side, solver_config)
                        self.synchronize()
                    finally:
                        self._set_prior(side, None)
                    self._update_feature_weight(side)
                else:
                    self.half_step_async(side, solver_config)
            self.synchronize()
        else:
            check(lib().irs_ials_step(self._h, C.byref(sc)))
        self._epoch += 1

    def user_scores(self, begin: int, end: int, solver_config: IALSSolverConfig) -> np.ndarray:
        if begin < 0 or end < 0:
            raise TypeError("begin/end must be non-negative (size_t).")
        sc = solver_config._struct()
        out = np.empty((max(int(end) - int(begin), 0), self._n_items), dtype=np.float32)
        check(
            lib().irs_ials_user_scores(
                self._h, C.c_int64(begin), C.c_int64(end), C.byref(sc), ptr(out, C.c_float)
            )
        )
        return out

    def _transform(self, side: int, interaction: Any, solver_config: IALSSolverConfig) -> np.ndarray:
        X, indptr, indices, data = _lib.csr_arrays(interaction, np.float32)
        m = X.shape[0] if side == 0 else X.shape[1]
        out = np.zeros((m, self._K), dtype=np.float32)
        sc = solver_config._struct()
        check(
            lib().irs_ials_transform(
                self._h,
                C.c_int32(side),
                C.c_int64(X.shape[0]),
                C.c_int64(X.shape[1]),
                ptr(indptr, C.c_int64),
                ptr(indices, C.c_int32),
                ptr(data, C.c_float),
                C.byref(sc),
                ptr(out, C.c_float),
            )
        )
        return out

    def transform_user(self, interaction: Any, solver_config: IALSSolverConfig) -> np.ndarray:
        return self._transform(0, interaction, solver_config)

    def transform_item(self, interaction: Any, solver_config: IALSSolverConfig) -> np.ndarray:
        return self._transform(1, interaction, solver_config)

    def _transform_feature(self, side: int, feature: Any) -> np.ndarray:
        # transform_user_feature / transform_item_feature, hpp:826-836 + validate_* :1017-1043
        who = "user" if side == 0 else "item"
        W = self._weight(side)
        if W.ndim != 2 or W.shape[1] != self._K:
            raise ValueError(f"{who.capitalize()} feature weights are not initialized.")
        f = _as_feature(feature, 0)
        if f.shape[1] != W.shape[0]:
            raise ValueError(f"Shape mismatch: {who} feature matrix has {f.shape[1]} columns but "
                             f"{who}_feature_weight has {W.shape[0]} rows.")
        return np.ascontiguousarray(np.asarray(f @ W, dtype=np.float32))

    def _transform_with_feature(self, side: int, interaction: Any, feature: Any,
                                solver_config: IALSSolverConfig) -> np.ndarray:
        # transform_*_with_feature, hpp:803-824 + X_to_vector_with_prior :143-168
        prior = self._transform_feature(side, feature)
        X, indptr, indices, data = _lib.csr_arrays(interaction, np.float32)
        m = X.shape[0] if side == 0 else X.shape[1]
        if prior.shape[0] != m:
            raise ValueError("Feature prior shape does not match X.")
        nnz = np.diff(X.indptr) if side == 0 else np.bincount(X.indices, minlength=X.shape[1])
        self._check_prior_defined(side, nnz)
        out = np.zeros((m, self._K), dtype=np.float32)
        sc = solver_config._struct()
        check(
            lib().irs_ials_transform_with_prior(
                self._h, C.c_int32(side), C.c_int64(X.shape[0]), C.c_int64(X.shape[1]),
                ptr(indptr, C.c_int64), ptr(indices, C.c_int32), ptr(data, C.c_float),
                ptr(prior, C.c_float), C.byref(sc), ptr(out, C.c_float),
            )
        )
        return out

    def transform_user_with_feature(self, interaction, feature, solver_config):
        return self._transform_with_feature(0, interaction, feature, solver_config)

    def transform_item_with_feature(self, interaction, feature, solver_config):
        return self._transform_with_feature(1, interaction, feature, solver_config)

    def transform_user_feature(self, feature):
        return self._transform_feature(0, feature)

    def transform_item_feature(self, feature):
        return self._transform_feature(1, feature)

    def compute_loss(self, solver_config: IALSSolverConfig) -> float:
        sc = solver_config._struct()
        out = C.c_float(0.0)
        check(lib().irs_ials_compute_loss(self._h, C.byref(sc), C.byref(out)))
        loss = float(out.value)
        if self._feature_aware:  # hpp:880-882, 905-907, 917-938
            twice = 2.0 * loss
            for side in (0, 1):
                W = self._weight(side)
                if not W.shape[0]:
                    continue
                lam = (self._config.lambda_user_feature if side == 0
                       else self._config.lambda_item_feature)
                w = self._row_reg(side, self._row_nnz[side]).astype(np.float64)
                factor = self._get(side).astype(np.float64)
                resid = factor - self._stored_prior(side).astype(np.float64)
                twice -= float(w @ np.square(factor).sum(axis=1))
                twice += float(w @ np.square(resid).sum(axis=1))
                twice += float(lam) * float(np.square(W.astype(np.float64)).sum())
            loss = twice / 2.0
        return loss

    # -- read/write attributes (wrapper.cpp:158-161) --------------------------
    def _get(self, which: int) -> np.ndarray:
        n = self._n_users if which == 0 else self._n_items
        out = np.empty((n, self._K), dtype=np.float32)
        check(lib().irs_ials_get_factor(self._h, C.c_int32(which), ptr(out, C.c_float)))
        return out

    def _set(self, which: int, value: np.ndarray) -> None:
        v = np.ascontiguousarray(value, dtype=np.float32)
        if v.ndim != 2:
            raise TypeError("factor matrix must be 2-dimensional.")
        check(
            lib().irs_ials_set_factor(
                self._h, C.c_int32(which), ptr(v, C.c_float), C.c_int64(v.shape[0]), C.c_int64(v.shape[1])
            )
        )

    @property
    def user(self) -> np.ndarray:
        return self._get(0)

    @user.setter
    def user(self, value: np.ndarray) -> None:
        self._set(0, value)

    @property
    def item(self) -> np.ndarray:
        return self._get(1)

    @item.setter
    def item(self, value: np.ndarray) -> None:
        self._set(1, value)

    @property
    def user_feature_weight(self) -> np.ndarray:
        return self._ufw

    @user_feature_weight.setter
    def user_feature_weight(self, value: np.ndarray) -> None:
        value = np.asarray(value, dtype=np.float32)
        self._ufw = value.reshape(0, self._K) if value.ndim != 2 else np.ascontiguousarray(value)

    @property
    def item_feature_weight(self) -> np.ndarray:
        return self._ifw

    @item_feature_weight.setter
    def item_feature_weight(self, value: np.ndarray) -> None:
        value = np.asarray(value, dtype=np.float32)
        self._ifw = value.reshape(0, self._K) if value.ndim != 2 else np.ascontiguousarray(value)

    # -- pickle: (config, user, item, ufw, ifw); 3- or 5-tuples accepted
    #    (wrapper.cpp:162-181).  The restored object has no interaction matrix.
    def __getstate__(self) -> tuple:
        return (self._config, self.user, self.item, self._ufw, self._ifw)

    def __setstate__(self, state: tuple) -> None:
        if len(state) not in (3, 5):
            raise RuntimeError("Invalid IALSTrainer pickle state.")
        self._restore(state[0], state[1], state[2])
        if len(state) == 5:
            self._ufw = np.asarray(state[3], dtype=np.float32)
            self._ifw = np.asarray(state[4], dtype=np.float32)

    # -- device-level access for the multi-GPU host loop and the benchmark ----
    def set_stream(self, hip_stream: int) -> None:
        check(lib().irs_ials_set_stream(self._h, C.c_void_p(hip_stream)))

    def device_buffer(self, which: int) -> Tuple[int, int, int]:
        p = C.c_void_p()
        rows = C.c_int64()
        ld = C.c_int64()
        check(lib().irs_ials_device_buffer(self._h, C.c_int32(which), C.byref(p), C.byref(rows), C.byref(ld)))
        return int(p.value or 0), int(rows.value), int(ld.value)

    def copy_rows_async(self, which: int, row_begin: int, row_end: int, ext_ptr: int, to_ext: bool) -> None:
        check(
            lib().irs_ials_copy_rows_async(
                self._h, C.c_int32(which), C.c_int64(row_begin), C.c_int64(row_end),
                C.c_void_p(ext_ptr), C.c_int32(1 if to_ext else 0),
            )
        )

    def partial_gramian_async(self, side: int) -> None:
        check(lib().irs_ials_partial_gramian_async(self._h, C.c_int32(side)))

    def gramian_async(self, side: int) -> None:
        """partial + finish in one call (a trainer that holds every row of the other side)"""
        check(lib().irs_ials_gramian_async(self._h, C.c_int32(side)))

    def finish_gramian_async(self, side: int) -> None:
        check(lib().irs_ials_finish_gramian_async(self._h, C.c_int32(side)))

    def half_step_async(self, side: int, solver_config: IALSSolverConfig) -> None:
        sc = solver_config._struct()
        check(lib().irs_ials_half_step_async(self._h, C.c_int32(side), C.byref(sc)))

    def synchronize(self) -> None:
        check(lib().irs_ials_synchronize(self._h))

    def last_half_step_used_eigenbasis(self) -> bool:
        """diagnostics: the last half-step solved its short rows (<= 32 stored entries) in the
        eigenbasis of the Gramian (csrc/ials_eig_kernels.hpp)"""
        f = lib().irs_ials_last_eigenbasis
        f.restype = C.c_int32
        return bool(f(self._h) & 1)

    def profile(self, enable) -> None:
        """False / True: per-kernel device times off / on for every launch; 2: on for the dominant
        kernel only (the solve of the side with more rows) - what a timed run wants: event pairs on
        all ten launches of an epoch cost 0.05 ms of 2.1."""
        check(lib().irs_ials_profile(self._h, C.c_int32(2 if enable == 2 and enable is not True else (1 if enable else 0))))

    def profile_read(self) -> Dict[str, Dict[str, float]]:
        cap = 32
        names = ((C.c_char * 48) * cap)()
        ms = (C.c_double * cap)()
        launches = (C.c_int64 * cap)()
        count = C.c_int32(0)
        check(lib().irs_ials_profile_read(self._h, C.c_int32(cap), names, ms, launches, C.byref(count)))
        out: Dict[str, Dict[str, float]] = {}
        for i in range(count.value):
            out[names[i].value.decode()] = {"ms": float(ms[i]), "launches": int(launches[i])}
        return out
