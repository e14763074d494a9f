/*
 * irspack_amd.h — C ABI of libirspack_amd.so, the MI355X (gfx950) native
 * implementation of irspack's compiled hot path.
 *
 * Every entry point below is what the reference's FFI for this path binds
 * (nanobind modules `irspack.recommenders._ials_core`, `._knn`,
 * `irspack.evaluation._core_evaluator`, and `remove_diagonal` of
 * `irspack.utils._util_cpp`); the reference interface each one replaces is
 * cited as file:line relative to /root/reference.
 *
 * Conventions
 *  - plain C: opaque handles, caller-owned host pointers + sizes, no torch or
 *    HIP types in any signature (`void *stream` is a hipStream_t passed through
 *    as an integer-sized pointer).
 *  - every function returns an irs_status: 0 ok, 1 invalid argument (the
 *    reference throws std::invalid_argument -> Python ValueError), 2 runtime
 *    error (std::runtime_error -> RuntimeError).  irs_last_error() returns the
 *    message of the last failing call on the calling thread.
 *  - calls are synchronous: on return device work is complete and outputs are
 *    host-visible, unless the function name ends in `_async`.
 *  - inputs are copied before return; the library never keeps a caller pointer.
 *  - CSR inputs: indptr int64[rows+1], indices int32[nnz] (sorted within a
 *    row), data float32 (iALS) or float64 (kNN / evaluator).
 *  - handles are not thread-safe (neither are the reference's objects).
 */
#ifndef IRSPACK_AMD_H
#define IRSPACK_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef int32_t irs_status;
#define IRS_OK 0
#define IRS_INVALID_ARGUMENT 1
#define IRS_RUNTIME_ERROR 2

/* Layout version of the structs below: 2 since irs_ceilings grew the gather rates, 3 since
 * irs_eval_stats carries the call's times.  A binding compares irs_abi_version() with the header it
 * was written against before passing structs. */
#define IRS_ABI_VERSION 4

const char *irs_last_error(void);
/* Library / device probe.  irs_device_count() returns 0 when no GPU is visible. */
int32_t irs_abi_version(void);
int32_t irs_device_count(void);

/* ------------------------------------------------------------------ iALS
 * enums: cpp_source/als/IALSLearningConfig.hpp:11-12, als/wrapper.cpp:25-40 */
#define IRS_LOSS_ORIGINAL 0
#define IRS_LOSS_IALSPP 1
#define IRS_SOLVER_CHOLESKY 0
#define IRS_SOLVER_CG 1
#define IRS_SOLVER_IALSPP 2

/* IALSModelConfig, IALSLearningConfig.hpp:15-31 (als/wrapper.cpp:42-71) */
typedef struct irs_ials_model_config {
  uint64_t K;
  float alpha0;
  float reg;
  float nu;
  float init_stdev;
  int32_t random_seed;
  int32_t loss_type;
  float lambda_user_feature;  /* feature-aware iALS: used by the host-side ridge update of */
  float lambda_item_feature;  /* the feature weights (see irs_ials_set_prior); carried    */
  uint64_t feature_warmup_epochs; /* here for the pickle state.                            */
} irs_ials_model_config;

/* SolverConfig, IALSLearningConfig.hpp:97-112 (als/wrapper.cpp:92-115) */
typedef struct irs_ials_solver_config {
  uint64_t n_threads;   /* validated (> 0) like the reference; no effect on the GPU */
  int32_t solver_type;
  uint64_t max_cg_steps;
  uint64_t ialspp_subspace_dimension;
  uint64_t ialspp_iteration;
} irs_ials_solver_config;

typedef struct irs_ials_trainer irs_ials_trainer;

/* Row shard of a multi-GPU run: this handle solves users [user_begin,user_end)
 * and items [item_begin,item_end) and keeps full replicas of both factor
 * matrices.  {0,n_users,0,n_items} is the single-GPU case. */
typedef struct irs_ials_shard {
  int64_t user_begin, user_end, item_begin, item_end;
} irs_ials_shard;

/* IALSTrainer(config, X): IALSTrainer.hpp:710-720, als/wrapper.cpp:131-132.
 * Copies X (CSR f32), builds X^T, initialises both factor matrices from the same
 * seed (hpp:64-76, 718-719).  `shard` may be NULL. */
irs_status irs_ials_create(const irs_ials_model_config *config, int64_t n_users,
                           int64_t n_items, const int64_t *indptr,
                           const int32_t *indices, const float *data,
                           int32_t device, const irs_ials_shard *shard,
                           irs_ials_trainer **out);
/* Deserialising constructor IALSTrainer(config, user, item): hpp:746-756,
 * als/wrapper.cpp:167-181.  No interaction matrix is kept. */
irs_status irs_ials_create_from_factors(const irs_ials_model_config *config,
                                        int64_t n_users, int64_t n_items,
                                        const float *user, const float *item,
                                        int32_t device, irs_ials_trainer **out);
irs_status irs_ials_destroy(irs_ials_trainer *t);

/* IALSTrainer::step, hpp:758-789: P_u, user solve, P_i, item solve. */
irs_status irs_ials_step(irs_ials_trainer *t, const irs_ials_solver_config *sc);
/* `.user` / `.item` read-write attributes, als/wrapper.cpp:158-159.
 * which: 0 = user, 1 = item.  out/in are C-contiguous float32 [rows, K]. */
irs_status irs_ials_get_factor(irs_ials_trainer *t, int32_t which, float *out);
irs_status irs_ials_set_factor(irs_ials_trainer *t, int32_t which,
                               const float *in, int64_t rows, int64_t cols);
/* IALSTrainer::user_scores, hpp:942-984.  out is float32 [end-begin, n_items]. */
irs_status irs_ials_user_scores(irs_ials_trainer *t, int64_t begin, int64_t end,
                                const irs_ials_solver_config *sc, float *out);
/* IALSTrainer::transform_user / transform_item, hpp:791-802 (+ X_to_vector
 * hpp:122-141).  side 0: X is [rows, n_items] -> out [rows, K];
 * side 1: X is [n_users, cols] -> out [cols, K]. */
irs_status irs_ials_transform(irs_ials_trainer *t, int32_t side, int64_t rows,
                              int64_t cols, const int64_t *indptr,
                              const int32_t *indices, const float *data,
                              const irs_ials_solver_config *sc, float *out);
/* transform_user_with_feature / transform_item_with_feature (hpp:803-824,
 * X_to_vector_with_prior hpp:143-168): as irs_ials_transform with a feature prior
 * (host float32 [out rows, K]): the rows start from the prior and the solve adds
 * reg_r * prior_r to the right-hand side (step_cholesky_with_prior hpp:333-385, step_cg
 * hpp:212-215).  IALSPP is rejected like in the reference (hpp:659-661). */
irs_status irs_ials_transform_with_prior(irs_ials_trainer *t, int32_t side, int64_t rows,
                                         int64_t cols, const int64_t *indptr,
                                         const int32_t *indices, const float *data,
                                         const float *prior,
                                         const irs_ials_solver_config *sc, float *out);
/* Feature-aware training (IALSTrainer::step, hpp:758-789): the prior of side `which`
 * (host float32 [rows, K] = features @ feature_weight) used by the following
 * irs_ials_half_step_async calls of that side; NULL clears it.  The feature-weight ridge
 * update (hpp:1052-1209) is a small F x F host solve and stays with the caller. */
irs_status irs_ials_set_prior(irs_ials_trainer *t, int32_t which, const float *prior);
/* The two feature products of feature-aware training on the device, so that the factor
 * matrices never leave HBM during an epoch.  irs_ials_set_features stores the feature
 * matrix of side `which` (CSR float32 [rows, n_feat]; a dense matrix is passed as a full
 * CSR; FeatureMatrix, hpp:693-700).  irs_ials_apply_feature_prior computes
 * prior = features @ weight (host float32 [n_feat, K]; feature_times_weight hpp:702-708)
 * and installs it like irs_ials_set_prior.  irs_ials_feature_rhs returns
 * features^T (reg_r * factor_r) as host float32 [n_feat, K], the right-hand side of the
 * feature-weight ridge system (solve_feature_weight hpp:1134-1171). */
irs_status irs_ials_set_features(irs_ials_trainer *t, int32_t which, int64_t rows,
                                 int64_t n_feat, const int64_t *indptr,
                                 const int32_t *indices, const float *data);
irs_status irs_ials_apply_feature_prior(irs_ials_trainer *t, int32_t which,
                                        const float *weight);
irs_status irs_ials_feature_rhs(irs_ials_trainer *t, int32_t which, float *out);
/* IALSTrainer::compute_loss, hpp:836-940. */
irs_status irs_ials_compute_loss(irs_ials_trainer *t,
                                 const irs_ials_solver_config *sc, float *out);

/* -- device-level pieces of IALSTrainer::step, for the multi-GPU host loop and
 *    the benchmark (one process per GPU; collectives run outside this library
 *    on the buffers these calls expose). -- */
irs_status irs_ials_set_stream(irs_ials_trainer *t, void *hip_stream);
/* which: 0 user factors, 1 item factors (float32 [rows, ld]; `rows` is the allocated row
 * count = the matrix's rows rounded up to a multiple of 8, zero padded, so that 1 / 2 / 4 / 8
 * equal row shards tile the buffer and one in-place all-gather can exchange them);
 * 2 / 3: Gramian used by the user / item solve (float32 [ld, ld], unscaled
 * sum over this rank's shard until irs_ials_finish_gramian is called). */
irs_status irs_ials_device_buffer(irs_ials_trainer *t, int32_t which,
                                  void **device_ptr, int64_t *rows, int64_t *ld);
/* Device-to-device copy of rows [row_begin,row_end) of buffer `which` (same
 * numbering as irs_ials_device_buffer) to (to_ext != 0) or from an external
 * device pointer holding (row_end-row_begin) x ld floats, on the trainer's
 * stream.  This is how the host loop stages the Gramian all-reduce and the
 * factor all-gather through torch.distributed tensors. */
irs_status irs_ials_copy_rows_async(irs_ials_trainer *t, int32_t which,
                                    int64_t row_begin, int64_t row_end,
                                    void *ext_device_ptr, int32_t to_ext);
/* Solver::prepare_p (hpp:78-115) split in two so that an all-reduce can sit in
 * between: partial = sum over this shard's rows of the *other* side's factors
 * (side 0: Gramian for the user solve = item rows of this shard). */
irs_status irs_ials_partial_gramian_async(irs_ials_trainer *t, int32_t side);
irs_status irs_ials_finish_gramian_async(irs_ials_trainer *t, int32_t side);
/* The two in one call for a trainer that holds every row (nothing to all-reduce in between): the
 * reduction writes the scaled Gramian in all its layouts itself, one launch less per half-epoch. */
irs_status irs_ials_gramian_async(irs_ials_trainer *t, int32_t side);
/* Solver::step (hpp:664-679) over this shard's rows of `side`. */
irs_status irs_ials_half_step_async(irs_ials_trainer *t, int32_t side,
                                    const irs_ials_solver_config *sc);
/* Waits for the stream and raises what the reference would have thrown from
 * inside the solve (hpp:317-323, 250-254). */
irs_status irs_ials_synchronize(irs_ials_trainer *t);
/* ---- row-sharded epoch (one process per GPU; no reference counterpart: the reference has no
 * distributed layer; the sharded unit is IALSTrainer::step, hpp:758-789).
 * irs_comm is this rank's transport of the two exchanges a sharded epoch needs: the K x K
 * all-reduce of the Gramian and the exchange of the freshly solved rows.
 *   irs_comm_create: two RCCL communicators (solved rows; Gramians).  id256: 256 bytes made by
 *     irs_comm_unique_id on ONE rank and handed to every rank by the caller (torch.distributed /
 *     MPI / a file).
 *   irs_comm_create_local: no RCCL at all - both exchanges are stores into the peers' mapped
 *     memory (below); needs irs_comm_export + irs_comm_attach before the first step.
 * How the solved rows reach the other replicas: */
#define IRS_EXCHANGE_AUTO 0      /* ncclAllGather in place (equal row blocks) or grouped broadcasts */
#define IRS_EXCHANGE_BROADCAST 1 /* one group of in-place ncclBroadcast, one per rank */
#define IRS_EXCHANGE_MESH 2      /* one group of ncclSend / ncclRecv: own block to every peer, every
                                  * peer's block into place - all links of the xGMI mesh at once */
#define IRS_EXCHANGE_PEER 3      /* a copy kernel stores the own block into every peer's mapped factor
                                  * buffer; arrival = sequence numbers in mapped flag words */
typedef struct irs_comm irs_comm;
irs_status irs_comm_unique_id(void *id256);
irs_status irs_comm_create(const void *id256, int32_t rank, int32_t world, int32_t device,
                           irs_comm **out);
irs_status irs_comm_create_local(int32_t rank, int32_t world, int32_t device, irs_comm **out);
irs_status irs_comm_destroy(irs_comm *c);
/* Peer stores: irs_comm_export fills handle256 (256 bytes: hipIpc handles of the trainer's two
 * factor buffers and of the communicator's flag + mailbox block); the caller gathers the blobs of
 * all ranks in rank order (world x 256 bytes) and gives them to irs_comm_attach on every rank,
 * which maps the peers' memory.  The trainer must outlive the communicator's use. */
irs_status irs_comm_export(irs_comm *c, irs_ials_trainer *t, void *handle256);
irs_status irs_comm_attach(irs_comm *c, irs_ials_trainer *t, const void *handles);
/* Selects IRS_EXCHANGE_* for the steps that follow (the same value on every rank). */
irs_status irs_comm_set_exchange(irs_comm *c, int32_t mode);
int32_t irs_comm_get_exchange(irs_comm *c);
/* IALSTrainer::step (hpp:758-789) over the ranks of `c`: the trainer holds the shard
 * [bounds[rank], bounds[rank + 1]) of each side (irs_ials_create with a shard); per half-epoch the
 * partial Gramian of the own rows is all-reduced, the own rows are solved and exchanged into
 * every replica (in place), the next half-epoch's Gramian overlapping the row exchange.  What the
 * reference would throw from inside a solve (hpp:317-323, 250-254) is raised on EVERY rank.
 * user_bounds / item_bounds: world + 1 row offsets, identical on every rank. */
irs_status irs_ials_sharded_step(irs_ials_trainer *t, const irs_ials_solver_config *sc, irs_comm *c,
                                 const int64_t *user_bounds, const int64_t *item_bounds);
/* Which rows of the last half-step were solved in the eigenbasis of the Gramian (measurement /
 * tests only, no reference counterpart): bit 0 = the rows of <= 32 stored entries. */
int32_t irs_ials_last_eigenbasis(irs_ials_trainer *t);
/* Test hook for the eigen-decomposition kernel of the short-row path (csrc/ials_eig_kernels.hpp):
 * P [K, K] row-major symmetric, K <= 128 -> Qrows [K, K] (row k = eigenvector k), lam [K],
 * stats [3] = largest, smallest eigenvalue, sweeps; P_prev (or NULL): a nearby matrix whose
 * eigenvectors warm-start the sweeps, as the previous epoch's Gramian does in a fit. */
irs_status irs_ials_eigen_debug(const float *P, int64_t K, int32_t device, float *Qrows, float *lam,
                                float *stats, const float *P_prev);
/* Per-kernel device times of the launches that follow (HIP events riding on the dispatches).
 * enable: 0 off, 1 every launch, 2 the dominant kernel only (the "ials_solve_*" launch of the side
 * with more rows: event pairs on all ten launches of an epoch cost 0.05 ms of 2.1, which a timed run
 * should not pay for durations it does not need). */
irs_status irs_ials_profile(irs_ials_trainer *t, int32_t enable);
irs_status irs_ials_profile_read(irs_ials_trainer *t, int32_t cap,
                                 char (*names)[48], double *ms,
                                 int64_t *launches, int32_t *count);

/* ------------------------------------------------------------------ kNN
 * cpp_source/knn/wrapper.cpp:11-66; KNNComputer knn.hpp:30-83 */
#define IRS_SIM_COSINE 0
#define IRS_SIM_ASYMMETRIC 1
#define IRS_SIM_JACCARD 2
#define IRS_SIM_TVERSKY 3
#define IRS_SIM_P3ALPHA 4
#define IRS_SIM_RP3BETA 5

typedef struct irs_knn_computer irs_knn_computer;

/* How (indptr, indices, data) store a [rows, cols] sparse matrix.  The reference's binding takes
 * either scipy layout (nanobind's Eigen caster converts: cpp_source/knn/wrapper.cpp:11-19 declares
 * `const CSRMatrix &`, and irspack/recommenders/knn.py:77-79 passes `X_weighted.T` / `X_train_all.T`,
 * which are CSC).  IRS_LAYOUT_CSC: the arrays are the compressed COLUMNS of the matrix, i.e. the CSR
 * arrays of its transpose [cols, rows] - indptr has cols + 1 entries, indices are row numbers.  No
 * host conversion: the library reads them as they are. */
#define IRS_LAYOUT_CSR 0
#define IRS_LAYOUT_CSC 1
/* Feature weighting of the interaction matrix before a similarity computer is built on it
 * (cpp_source/util.hpp:159-188 okapi_BM_25_weight, :190-209 tf_idf_weight; callers
 * irspack/recommenders/knn.py:67-75, user_knn.py:62-72). */
#define IRS_WEIGHT_NONE 0
#define IRS_WEIGHT_TF_IDF 1
#define IRS_WEIGHT_BM25 2
/* Optional description of the arrays handed to irs_knn_create (null = CSR, no weighting).
 * `weighting` is applied ON THE DEVICE, on the way in, to the matrix AS STORED - its compressed rows
 * are the documents, `indices` the terms, exactly what tf_idf_weight / okapi_BM_25_weight do with a
 * CSR matrix: `CosineSimilarityComputer(tf_idf_weight(X).T, ...)` of the reference is
 * irs_knn_create(rows = X.cols, cols = X.rows, X's CSR arrays, {IRS_LAYOUT_CSC, IRS_WEIGHT_TF_IDF}),
 * and the user-kNN `CosineSimilarityComputer(tf_idf_weight(X), ...)` is {IRS_LAYOUT_CSR,
 * IRS_WEIGHT_TF_IDF} - the weighted matrix never exists on the host.  Jaccard / Tversky binarise their
 * input (similarities.hpp:96-107, 143-159), which makes any weighting a no-op; P3alpha / RP3beta
 * take no weighting (invalid argument). */
typedef struct irs_knn_input {
  int32_t layout;    /* IRS_LAYOUT_* */
  int32_t weighting; /* IRS_WEIGHT_* */
  int32_t smooth;    /* tf-idf: idf = log(N / (df + smooth)) (util.hpp:203; the Python default is 1) */
  int32_t reserved;  /* 0 */
  double k1, b;      /* BM25 (util.hpp:159; Python defaults 1.2, 0.75) */
} irs_knn_input;

/* *SimilarityComputer(X, shrinkage, [alpha, beta, normalize], n_threads,
 * max_chunk_size): similarities.hpp:20-28, 61-72, 96-107, 143-159, 198-222,
 * 265-292.  X is float64 [rows = N, cols = n_features], stored as `input` says. */
irs_status irs_knn_create(int32_t sim_type, int64_t rows, int64_t cols,
                          const int64_t *indptr, const int32_t *indices,
                          const double *data, const irs_knn_input *input,
                          double shrinkage, double alpha,
                          double beta, int32_t normalize, int64_t n_threads,
                          int64_t max_chunk_size, int32_t device,
                          irs_knn_computer **out);
/* tf_idf_weight(X, smooth) / okapi_BM_25_weight(X, k1, b) on their own (util.hpp:159-209, bound at
 * util.cpp:29-32): X is CSR float64 [rows, cols]; `out` (caller-allocated, nnz doubles) receives the
 * weighted values in X's entry order (X's pattern is unchanged).  Column counts, row sums and the
 * idf table (libm's log) are made on host threads, the per-entry pass runs on the device
 * (knn_weight_kernel: one IEEE multiply for tf-idf; multiply, multiply, add, divide for BM25, each
 * rounded once - the host loop's values bit for bit). */
irs_status irs_knn_weight(int32_t weighting, int64_t rows, int64_t cols,
                          const int64_t *indptr, const int32_t *indices,
                          const double *data, double k1, double b, int32_t smooth,
                          int32_t device, double *out);
irs_status irs_knn_destroy(irs_knn_computer *c);
/* compute_similarity(X, top_k) (knn.hpp:43-139) / compute_W (similarities.hpp:
 * 224-240, 294-324; as_w != 0, result still row-major [rows, N]).  Two calls:
 * compute runs the device work and reports nnz; fetch copies the CSR out (from a
 * page-locked copy the compute call made while its later row chunks were still
 * running, or from the device).  The caller's arrays are read in place, never
 * written.  `layout` (IRS_LAYOUT_*): how the arrays store the [rows, cols] target - knn.py:79 passes
 * `X_train_all.T`, a CSC matrix; its columns are regrouped into rows on host threads inside the call (indices
 * only when every value is 1).  row_begin/row_end select a shard of target rows (multi-GPU: rows are
 * independent, no collective). */
irs_status irs_knn_compute(irs_knn_computer *c, int64_t rows, int64_t cols,
                           const int64_t *indptr, const int32_t *indices,
                           const double *data, int32_t layout, int64_t top_k, int32_t as_w,
                           int64_t row_begin, int64_t row_end, int64_t *nnz_out);
irs_status irs_knn_fetch(irs_knn_computer *c, int64_t *indptr, int32_t *indices,
                         double *data);
/* The last result once more, as the compressed COLUMNS of the [rows of the call, N] matrix (col_ptr
 * int64[N + 1], row numbers relative to the call's first row, ascending in every column) - what
 * `remove_diagonal(result).tocsc()` of irspack/recommenders/knn.py:78-80 builds on the host (util.hpp:211-226
 * + scipy's conversion, ~50 ms for the ML-20M result), regrouped on the device instead.
 * zero_diagonal_row0 >= 0: the stored entries whose column equals zero_diagonal_row0 + their row are set to
 * 0.0 and KEPT (remove_diagonal; pass the call's row_begin, 0 for a whole matrix); < 0: values unchanged. */
irs_status irs_knn_fetch_csc(irs_knn_computer *c, int64_t zero_diagonal_row0, int64_t *col_ptr,
                             int32_t *row_idx, double *data);
/* Measurement only: multiply-adds of the last call that were added one by one, and the rows of the
 * dense block of the popular items - an opt-in of rounds 4 - 5 that was removed: every multiply-add is
 * added one by one and dense_rows is 0 (kept for ABI 3). */
irs_status irs_knn_last_walked(irs_knn_computer *c, int64_t *walked_macs, int32_t *dense_rows);
/* Milliseconds of device time of the last irs_knn_compute (HIP events: first launch to last row merge of
 * the call's row chunks) and the number of multiply-adds it performed. */
irs_status irs_knn_last_stats(irs_knn_computer *c, double *kernel_ms,
                              int64_t *macs);
/* remove_diagonal, cpp_source/util.hpp:211-226 (in place on `data`). */
irs_status irs_remove_diagonal(int64_t rows, int64_t cols, const int64_t *indptr,
                               const int32_t *indices, double *data);
/* retrieve_recommend_from_score_{f32,f64}, cpp_source/util.hpp:426-504 (bound at
 * util.cpp; caller utils/id_mapping.py:29-44, 312-317): top-`cutoff` candidates of every
 * score row, best first, stopping at -inf.  scores: host row-major [rows, n_items], float32
 * (is_f64 == 0) or float64.  Allowed lists are ragged (list_ptr int64[n_lists + 1],
 * list_items int64; n_lists in {0, 1, rows}); ids outside [0, n_items) are dropped, order
 * and duplicates are kept.  out_idx: int32 [rows, cutoff], -1 padded; the caller reads the
 * scores of the winners from its own array.  Equal scores come out in candidate order (the
 * reference's comparator leaves their order unspecified, util.hpp:483-485). */
irs_status irs_retrieve_recommend(int32_t is_f64, const void *scores, int64_t rows,
                                  int64_t n_items, int64_t n_lists, const int64_t *list_ptr,
                                  const int64_t *list_items, int64_t cutoff,
                                  int64_t n_threads, int32_t device, int32_t *out_idx);

/* ------------------------------------------------------------------ evaluator
 * cpp_source/evaluator.cpp:441-484 */
typedef struct irs_evaluator irs_evaluator;

/* Metrics accumulator state (evaluator.cpp:168-178): the caller owns
 * item_cnt[n_items]. */
typedef struct irs_metrics {
  uint64_t valid_user;
  uint64_t total_user;
  double hit, recall, ndcg, precision, map;
} irs_metrics;

/* EvaluatorCore(ground_truth, recommendable): evaluator.cpp:183-206.
 * Recommendable lists are ragged: rec_ptr int64[n_lists+1], rec_items int64. */
irs_status irs_eval_create(int64_t n_users, int64_t n_items,
                           const int64_t *indptr, const int32_t *indices,
                           int64_t n_lists, const int64_t *rec_ptr,
                           const int64_t *rec_items, int32_t device,
                           irs_evaluator **out);
irs_status irs_eval_destroy(irs_evaluator *e);
/* get_metrics_f32 / get_metrics_f64: evaluator.cpp:256-284 (+ :292-367,
 * Metrics::update :127-166).  scores is a host row-major [rows, n_items] block
 * of float32 (is_f64 == 0) or float64. */
irs_status irs_eval_get_metrics(irs_evaluator *e, int32_t is_f64,
                                const void *scores, int64_t rows, int64_t cutoff,
                                int64_t offset, int64_t n_threads,
                                int32_t recall_with_cutoff, irs_metrics *out,
                                int64_t *item_cnt);
/* The caller loop of evaluation/evaluator.py:371-393 (score chunks) and :417-438 (model blocks)
 * for one block, on the device: the host block is uploaded ONCE (the upload is the copy the
 * reference makes at :387, so the caller's array is never written), the stored entries of the
 * mask rows are set to -inf there (:389 / :432; the caller passes the NONZERO entries only:
 * mask_indptr has rows + 1 entries, mask_indices[mask_indptr[r] - mask_indptr[0] ..) are the
 * columns of row r; NULL = no mask) and the block is ranked once per cutoff (:390-391 /
 * :433-439).  out[c] / item_cnt[c * n_items ..] receive the metrics of cutoffs[c]. */
irs_status irs_eval_get_metrics_masked(irs_evaluator *e, int32_t is_f64, const void *scores,
                                       int64_t rows, const int64_t *mask_indptr,
                                       const int32_t *mask_indices, int32_t n_cutoffs,
                                       const int64_t *cutoffs, int64_t offset,
                                       int64_t n_threads, int32_t recall_with_cutoff,
                                       irs_metrics *out, int64_t *item_cnt);
/* Evaluator.get_score(model) for a SIMILARITY model (evaluation/evaluator.py:400-441 with
 * BaseSimilarityRecommender.get_score_block, recommenders/base.py:406-429: score = X_train[u] @ W), without
 * the host in the loop: users begin .. end of the model are scored on the device - x_* are the profiles
 * X [n_model_users, n_profile_cols] as CSR float64, w_* is W [n_profile_cols, n_items] BY ROWS (CSR): the
 * training matrix and the learnt item-item weights for item-kNN / P3alpha / RP3beta, the learnt user-user
 * weights and the training matrix for user-kNN (base.py:432-453) - with the per-entry order and rounding of
 * scipy's row-by-row sparse product (the block is the host product bit for bit), masked (mask_* as in irs_eval_get_metrics_masked: the rows begin .. end only, NULL = no mask) and
 * ranked once per cutoff; out[c] / item_cnt[c * n_items ..] receive the metrics of cutoffs[c].  `offset`:
 * ground-truth row of user `begin`. */
irs_status irs_eval_get_metrics_similarity(irs_evaluator *e, int64_t begin, int64_t end,
                                           int64_t n_model_users, int64_t n_profile_cols,
                                           const int64_t *x_indptr,
                                           const int32_t *x_indices, const double *x_data,
                                           const int64_t *w_indptr, const int32_t *w_indices,
                                           const double *w_data, const int64_t *mask_indptr,
                                           const int32_t *mask_indices, int32_t n_cutoffs,
                                           const int64_t *cutoffs, int64_t offset,
                                           int32_t recall_with_cutoff, irs_metrics *out,
                                           int64_t *item_cnt);
/* Fused device path used by the Evaluator counterpart when the model is an
 * iALS trainer of this library: scores = user[begin:end] @ item^T (hpp:942-984)
 * are produced, masked (evaluator.py:417-432, mask = CSR rows given here, set
 * to -inf) and ranked on the device without leaving HBM.  mask_indptr may be
 * NULL (no mask, or the mask cached by irs_eval_cache_mask); it has rows+1 entries
 * relative to `begin`. */
irs_status irs_eval_get_metrics_ials(irs_evaluator *e, irs_ials_trainer *t,
                                     int64_t begin, int64_t end,
                                     const int64_t *mask_indptr,
                                     const int32_t *mask_indices, int64_t cutoff,
                                     int64_t offset, int32_t recall_with_cutoff,
                                     irs_metrics *out, int64_t *item_cnt);
/* Keeps the mask of the fused path on the device between calls: the same CSR rows
 * irs_eval_get_metrics_ials takes (rows + 1 entries of indptr relative to its `begin`).
 * While a mask of `rows` rows is cached, a call of irs_eval_get_metrics_ials over that many
 * users with mask_indptr == NULL applies it; rows <= 0 or mask_indptr == NULL drops it. */
irs_status irs_eval_cache_mask(irs_evaluator *e, int64_t rows, const int64_t *mask_indptr,
                               const int32_t *mask_indices);

/* 64-bit content fingerprint of a host buffer, computed on several host threads: the check that the
 * device-resident mask of irs_eval_cache_mask is still the caller's matrix (every byte, every call;
 * no reference counterpart).  Not cryptographic. */
irs_status irs_fingerprint(const void *data, int64_t n_bytes, uint64_t seed, uint64_t *out);

/* What the last irs_eval_get_metrics_ials call did (measurement only, no reference
 * counterpart).  path: 0 = score block + ranking (two passes), 1 = threshold-filtered
 * candidates, 2 = threshold-filtered with norm-bound pruning (3, a single-pass variant, was removed in ABI 3's round). */
typedef struct irs_eval_stats {
  int32_t path;
  int32_t hard_rows;    /* rows ranked from their full score row after the filtered pass */
  int64_t tiles_total;  /* 64 x 64 score tiles of the user block */
  int64_t tiles_scored; /* tiles the scoring kernel computed (= tiles_total unless pruned) */
  int64_t sample_items; /* items of the threshold sample pass (0: none) */
  double call_ms;        /* host clock around the whole C call (mask upload, launches, read-back) */
  double device_span_ms; /* HIP events on the launch stream: first kernel of the call -> last one done
                          * (includes the gaps in which the host reads the hard-row list back) */
} irs_eval_stats;
irs_status irs_eval_last_stats(irs_evaluator *e, irs_eval_stats *out);

/* ------------------------------------------------------------ measurement
 * No reference counterpart: SURVEY.md 8(d) asks for ceilings MEASURED on the box next to the
 * spec peaks.  Runs a 1 GiB device copy and STREAM triad (HBM bytes moved / time), a loop of
 * nothing but v_mfma_f32_16x16x4_f32 on every SIMD, and a loop of random-bank ds_add_u32 on
 * every CU (the kNN count accumulation's instruction), and a random whole-row gather out of a
 * 1 GiB table (the iALS access pattern); all HIP-event timed, best of 3-5. */
typedef struct irs_ceilings {
  double copy_gbs;            /* read + write bytes per second of dst = src, GB/s */
  double triad_gbs;           /* a = b + s c: 3 x bytes, GB/s */
  double mfma_f32_tflops;     /* dense fp32 MFMA rate, TFLOP/s */
  double lds_atomic_u32_gops; /* lane-level LDS atomic adds per second (whole device), G/s */
  double clock_mhz;           /* hipDeviceAttributeClockRate */
  int32_t n_cu;
  double gather256_gbs;       /* uniformly random 256-byte rows of a 1 GiB table (16 lanes per row), GB/s */
  double gather512_gbs;       /* the same with 512-byte rows (K = 128 factor rows) */
} irs_ceilings;
irs_status irs_measure_ceilings(int32_t device, irs_ceilings *out);

#ifdef __cplusplus
}
#endif
#endif /* IRSPACK_AMD_H */
