"""``IALSRecommender`` counterpart: the caller contract of the reference's
``irspack/recommenders/ials.py:68-203, 245-562`` on top of the GPU ``IALSTrainer``.

Reproduces what the reference's Python does around the native calls: float32 cast
(:91), log confidence scaling (:437-446), the ``nu_star`` regulariser rescale
(:232-242, 412-418), the two solver configs (training / prediction time), the
epoch loop, scoring and fold-in.  Early stopping against a validation evaluator
follows ``base_earlystop.py:106-149`` without the Optuna / progress-bar parts.
"""

import enum
import pickle
from io import BytesIO
from typing import IO, Any, Callable, Optional

import numpy as np
import scipy.sparse as sps

from .._threading import get_n_threads
from ._ials_core import IALSModelConfigBuilder, IALSSolverConfigBuilder
from ._ials_core import IALSTrainer as CoreTrainer
from ._ials_core import LossType, SolverType
from .base import BaseRecommender


def _enum_by_name(enum_cls: Any, name: str) -> Any:
    """Case-insensitive lookup of an enum member; unknown names raise like the reference's
    ``getattr`` does (ials.py:46-55)."""
    try:
        return enum_cls[name.upper()]
    except KeyError:
        raise AttributeError(f"{enum_cls.__name__} has no member {name!r}.") from None


def str_to_solver_type(t: str) -> SolverType:
    return _enum_by_name(SolverType, t)


def str_to_loss_type(t: str) -> LossType:
    return _enum_by_name(LossType, t)


def _feature_matrix_as_float32(X: Any) -> Any:
    # ials.py:58-61
    if sps.issparse(X):
        return sps.csr_matrix(X, dtype=np.float32)
    return np.asarray(X, dtype=np.float32, order="C")


class IALSTrainer:
    """ials.py:68-203."""

    def __init__(self, X: Any, n_components: int, alpha0: float, reg: float, nu: float,
                 init_std: float, solver_type: SolverType, max_cg_steps: int,
                 ialspp_subspace_dimension: int, loss_type: LossType, random_seed: int,
                 n_threads: int, prediction_time_max_cg_steps: int,
                 prediction_time_ialspp_iteration: int, device: Optional[int] = None,
                 user_features: Any = None, item_features: Any = None,
                 lambda_user_feature: float = 0.0, lambda_item_feature: float = 0.0,
                 feature_warmup_epochs: int = 0) -> None:
        def solver(cg_steps: int, sweeps: int):
            # the two solver configs differ only in how long they iterate (ials.py:103-131): one sweep /
            # `max_cg_steps` while training, the prediction-time counts for fold-in
            return (IALSSolverConfigBuilder().set_n_threads(n_threads).set_solver_type(solver_type)
                    .set_max_cg_steps(cg_steps).set_ialspp_iteration(sweeps)
                    .set_ialspp_subspace_dimension(ialspp_subspace_dimension).build())

        model = IALSModelConfigBuilder()
        for setter, value in (("set_K", n_components), ("set_init_stdev", init_std), ("set_alpha0", alpha0),
                              ("set_reg", reg), ("set_nu", nu), ("set_loss_type", loss_type),
                              ("set_random_seed", random_seed),
                              ("set_lambda_user_feature", lambda_user_feature),
                              ("set_lambda_item_feature", lambda_item_feature),
                              ("set_feature_warmup_epochs", feature_warmup_epochs)):
            getattr(model, setter)(value)
        self.feature_aware = user_features is not None or item_features is not None
        self.solver_config = solver(max_cg_steps, 1)
        self.prediction_time_solver_config = solver(prediction_time_max_cg_steps,
                                                    prediction_time_ialspp_iteration)
        features = (user_features, item_features) if self.feature_aware else ()
        self.core_trainer = CoreTrainer(model.build(), X.astype(np.float32), *features, device=device)

    _STATE_FIELDS = ("user", "item", "user_feature_weight", "item_feature_weight")

    def load_state(self, ifs: IO) -> None:
        """Restores what ``save_state`` wrote (ials.py:140-147); states written before the
        feature-aware model hold the two factor matrices only."""
        state = pickle.load(ifs)
        for field in self._STATE_FIELDS:
            if field in state:
                setattr(self.core_trainer, field, state[field])

    def save_state(self, ofs: IO) -> None:
        pickle.dump({field: getattr(self.core_trainer, field) for field in self._STATE_FIELDS},
                    ofs, protocol=pickle.HIGHEST_PROTOCOL)

    def compute_loss(self) -> float:
        return self.core_trainer.compute_loss(self.solver_config)

    def run_epoch(self) -> None:
        self.core_trainer.step(self.solver_config)

    def user_scores(self, begin: int, end: int) -> np.ndarray:
        return self.core_trainer.user_scores(begin, end, self.solver_config)

    def transform_user(self, X: Any, user_features: Any = None) -> np.ndarray:
        # ials.py:167-179
        if user_features is None:
            return self.core_trainer.transform_user(X, self.prediction_time_solver_config)
        return self.core_trainer.transform_user_with_feature(
            X, _feature_matrix_as_float32(user_features), self.prediction_time_solver_config)

    def transform_item(self, X: Any, item_features: Any = None) -> np.ndarray:
        # ials.py:181-193
        if item_features is None:
            return self.core_trainer.transform_item(X, self.prediction_time_solver_config)
        return self.core_trainer.transform_item_with_feature(
            X, _feature_matrix_as_float32(item_features), self.prediction_time_solver_config)

    def transform_user_with_feature(self, X: Any, features: Any) -> np.ndarray:
        return self.transform_user(X, user_features=features)

    def transform_item_with_feature(self, X: Any, features: Any) -> np.ndarray:
        return self.transform_item(X, item_features=features)

    def transform_user_feature(self, user_features: Any) -> np.ndarray:
        # ials.py:195-198
        return self.core_trainer.transform_user_feature(_feature_matrix_as_float32(user_features))

    def transform_item_feature(self, item_features: Any) -> np.ndarray:
        # ials.py:200-203
        return self.core_trainer.transform_item_feature(_feature_matrix_as_float32(item_features))


# confidence scaling modes accepted by IALSRecommender (ials.py:113-115): "none" | "log"
IALSConfigScaling = enum.Enum("IALSConfigScaling", ["none", "log"])


def compute_reg_scale(X: sps.csr_matrix, alpha0: float, nu: float) -> float:
    # ials.py:232-242
    X_csr = sps.csr_matrix(X)
    U, I = X_csr.shape
    nnz_row = np.diff(X_csr.indptr)
    nnz_col = np.bincount(X_csr.indices, minlength=I)
    return float(((nnz_row + alpha0 * I) ** nu).sum()) + float(((nnz_col + alpha0 * U) ** nu).sum())


class IALSRecommender(BaseRecommender):
    """ials.py:245-562.  Same constructor arguments and defaults; ``device`` is an extra
    keyword.  ``user_features`` / ``item_features`` (dense or CSR, one row per user / item)
    switch on the feature-aware model (trainer hpp:722-789)."""

    def __init__(self, X_train_all: Any, n_components: int = 20, alpha0: float = 0.0,
                 reg: float = 1e-3, nu: float = 1.0, confidence_scaling: str = "none",
                 epsilon: float = 1.0, init_std: float = 0.1, solver_type: str = "CG",
                 max_cg_steps: int = 3, ialspp_subspace_dimension: int = 64,
                 loss_type: str = "IALSPP", nu_star: Optional[float] = None,
                 random_seed: int = 42, n_threads: Optional[int] = None, train_epochs: int = 16,
                 prediction_time_max_cg_steps: int = 5,
                 prediction_time_ialspp_iteration: int = 7, device: Optional[int] = None,
                 user_features: Any = None, item_features: Any = None,
                 lambda_user_feature: float = 0.0, lambda_item_feature: float = 0.0,
                 feature_warmup_epochs: int = 0) -> None:
        super().__init__(X_train_all)
        # ials.py:421-432
        self.user_features = None if user_features is None else _feature_matrix_as_float32(user_features)
        self.item_features = None if item_features is None else _feature_matrix_as_float32(item_features)
        if (self.user_features is not None or self.item_features is not None) and solver_type == "IALSPP":
            raise ValueError("Feature-aware iALS does not support IALSPP.")
        self.lambda_user_feature = lambda_user_feature
        self.lambda_item_feature = lambda_item_feature
        self.feature_warmup_epochs = feature_warmup_epochs
        self.train_epochs = train_epochs
        self.n_components = n_components
        self.alpha0 = alpha0
        self.reg = reg
        self.nu = nu
        self.confidence_scaling = IALSConfigScaling[confidence_scaling]
        self.epsilon = epsilon
        self.init_std = init_std
        self.solver_type = str_to_solver_type(solver_type)
        self.max_cg_steps = max_cg_steps
        self.ialspp_subspace_dimension = ialspp_subspace_dimension
        self.random_seed = random_seed
        self.n_threads = get_n_threads(n_threads)
        self.loss_type = str_to_loss_type(loss_type)
        self.nu_star = nu_star
        self.scaled_reg = self.reg
        if self.nu_star is not None:  # ials.py:412-418
            self.scaled_reg = (self.reg * compute_reg_scale(self.X_train_all, alpha0, self.nu_star)
                               / compute_reg_scale(self.X_train_all, alpha0, nu))
        self.prediction_time_max_cg_steps = prediction_time_max_cg_steps
        self.prediction_time_ialspp_iteration = prediction_time_ialspp_iteration
        self.device = device
        self.trainer: Optional[IALSTrainer] = None
        self.best_state: Optional[bytes] = None
        self.learnt_config = {}

    @classmethod
    def _scale_X(cls, X: sps.csr_matrix, scheme: IALSConfigScaling, epsilon: float) -> sps.csr_matrix:
        if scheme is IALSConfigScaling.none:
            return X
        X_ret: sps.csr_matrix = X.copy()
        X_ret.data = np.log(1 + X_ret.data / epsilon)  # ials.py:437-446
        return X_ret

    def _create_trainer(self) -> IALSTrainer:
        return IALSTrainer(
            X=self._scale_X(self.X_train_all, self.confidence_scaling, self.epsilon),
            n_components=self.n_components, alpha0=self.alpha0, reg=self.scaled_reg, nu=self.nu,
            init_std=self.init_std, solver_type=self.solver_type, max_cg_steps=self.max_cg_steps,
            ialspp_subspace_dimension=self.ialspp_subspace_dimension, loss_type=self.loss_type,
            random_seed=self.random_seed, n_threads=self.n_threads,
            prediction_time_max_cg_steps=self.prediction_time_max_cg_steps,
            prediction_time_ialspp_iteration=self.prediction_time_ialspp_iteration,
            device=self.device, user_features=self.user_features,
            item_features=self.item_features, lambda_user_feature=self.lambda_user_feature,
            lambda_item_feature=self.lambda_item_feature,
            feature_warmup_epochs=self.feature_warmup_epochs,
        )

    # -- base_earlystop.py:80-149 ------------------------------------------
    def start_learning(self) -> None:
        self.trainer = self._create_trainer()

    def run_epoch(self) -> None:
        self.trainer_as_ials.run_epoch()

    def save_state(self) -> None:
        with BytesIO() as ofs:
            self.trainer_as_ials.save_state(ofs)
            self.best_state = ofs.getvalue()

    def load_state(self) -> None:
        if self.best_state is None:
            raise RuntimeError("'load_state' called before achieving any results.")
        with BytesIO(self.best_state) as ifs:
            self.trainer_as_ials.load_state(ifs)

    def _learn(self) -> None:
        self.learn_with_evaluator(None, max_epoch=self.train_epochs)

    def learn_with_evaluator(self, evaluator: Any, max_epoch: int = 128, validate_epoch: int = 5,
                             score_degradation_max: int = 5) -> None:
        """base_earlystop.py:106-149 without the progress bar / Optuna hooks: every
        ``validate_epoch`` epochs the evaluator scores the model; the best state is kept and
        restored at the end, and ``score_degradation_max`` validations in a row without a new best
        end the fit.  ``evaluator=None``: exactly ``max_epoch`` epochs."""
        self.start_learning()
        best, misses = -float("inf"), 0
        for done in range(1, max_epoch + 1):
            self.run_epoch()
            if evaluator is None or done % validate_epoch:
                continue
            score = evaluator.get_target_score(self)
            if score > best:
                best, misses = score, 0
                self.save_state()
                self.learnt_config["train_epochs"] = done
                continue
            misses += 1
            if misses >= score_degradation_max:
                break
        if evaluator is not None and self.best_state is not None:
            self.load_state()

    @property
    def trainer_as_ials(self) -> IALSTrainer:
        trainer = getattr(self, "trainer", None)
        if trainer is None:  # ials.py:472-476
            raise RuntimeError("tried to fetch trainer before the training.")
        return trainer

    # -- scoring (ials.py:476-562) ---------------------------------------------
    def get_score(self, user_indices: np.ndarray) -> np.ndarray:
        return self.trainer_as_ials.core_trainer.user[user_indices].dot(self.get_item_embedding().T)

    def get_score_block(self, begin: int, end: int) -> np.ndarray:
        return self.trainer_as_ials.user_scores(begin, end)

    def get_score_cold_user(self, X: Any, user_features: Any = None) -> np.ndarray:
        # ials.py:487-491
        return self.get_score_from_user_embedding(
            self.compute_user_embedding(X, user_features=user_features))

    def _create_cold_user_with_item_features_scorer(self, item_features: Any) -> Callable[[Any], np.ndarray]:
        # ials.py:493-517: the feature-only items' embeddings are computed once, every user block
        # is then one fold-in and two products
        if self.item_features is None:
            raise NotImplementedError("IALSRecommender must be trained with item_features before it "
                                      "can score additional items from features.")
        if item_features.shape[0]:
            item_embedding = self.compute_item_embedding_from_features(item_features)
        else:
            item_embedding = np.empty((0, self.n_components), dtype=self.get_item_embedding().dtype)

        def scorer(X: Any) -> np.ndarray:
            user_embedding = self.compute_user_embedding(X)
            known_scores = self.get_score_from_user_embedding(user_embedding)
            additional_scores = user_embedding.dot(item_embedding.T)
            return np.concatenate([known_scores, additional_scores], axis=1)

        return scorer

    def get_score_cold_user_with_item_features(self, X: Any, item_features: Any) -> np.ndarray:
        # ials.py:519-525
        return self._create_cold_user_with_item_features_scorer(item_features)(X)

    def get_user_embedding(self) -> np.ndarray:
        return self.trainer_as_ials.core_trainer.user

    def get_item_embedding(self) -> np.ndarray:
        return self.trainer_as_ials.core_trainer.item

    def get_score_from_user_embedding(self, user_embedding: np.ndarray) -> np.ndarray:
        return user_embedding.dot(self.get_item_embedding().T)

    def get_score_from_item_embedding(self, user_indices: np.ndarray,
                                      item_embedding: np.ndarray) -> np.ndarray:
        return self.get_user_embedding()[user_indices].dot(item_embedding.T)

    def _scaled_f32(self, X: Any) -> sps.csr_matrix:
        return self._scale_X(sps.csr_matrix(X).astype(np.float32), self.confidence_scaling,
                             self.epsilon)

    def compute_user_embedding(self, X: Any, user_features: Any = None) -> np.ndarray:
        # ials.py:538-563
        return self.trainer_as_ials.transform_user(self._scaled_f32(X), user_features=user_features)

    def compute_user_embedding_from_features(self, user_features: Any) -> np.ndarray:
        """ials.py:565-576: the feature-aware system with an EMPTY interaction history (it keeps
        the loss on the unobserved items, so it is not ``user_features @ user_feature_weight``)."""
        X = sps.csr_matrix((user_features.shape[0], self.n_items), dtype=np.float32)
        return self.compute_user_embedding(X, user_features=user_features)

    def get_score_cold_user_from_features(self, user_features: Any) -> np.ndarray:
        # ials.py:578-582
        return self.get_score_from_user_embedding(
            self.compute_user_embedding_from_features(user_features))

    def compute_item_embedding(self, X: Any, item_features: Any = None) -> np.ndarray:
        # ials.py:584-609
        return self.trainer_as_ials.transform_item(self._scaled_f32(X), item_features=item_features)

    def compute_item_embedding_from_features(self, item_features: Any) -> np.ndarray:
        # ials.py:611-622
        X = sps.csr_matrix((self.n_users, item_features.shape[0]), dtype=np.float32)
        return self.compute_item_embedding(X, item_features=item_features)

    def get_score_from_item_features(self, user_indices: np.ndarray, item_features: Any) -> np.ndarray:
        # ials.py:624-628
        return self.get_score_from_item_embedding(
            user_indices, self.compute_item_embedding_from_features(item_features))
