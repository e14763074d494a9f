// ORACLE — TEST INFRASTRUCTURE ONLY. Not part of the product path.
//
// CPU restatement of the reference's item-kNN path:
//   /root/reference/cpp_source/knn/knn.hpp            (KNNComputer)
//   /root/reference/cpp_source/knn/similarities.hpp   (six similarity types)
//   /root/reference/cpp_source/util.hpp:158-226       (bm25 / tf-idf / remove_diagonal)
// Arithmetic is `double` like the reference (knn/wrapper.cpp:9).
//
// Parity pinning: see oracle/ials_oracle.cpp header; this file is pinned by the
// dense numpy formulas and the tie-break known-answer test the reference holds
// in tests/recommenders/test_knn.py:33-165, restated in tests/test_oracle_knn.py.
//
// Third-party arithmetic: the sparse x sparse product is Eigen 5.0.1's
// conservative_sparse_sparse_product (not vendored).  Its published algorithm
// for a row-major result is: for each stored (u, y) of the target row in
// ascending u, for each stored (j, x) of X_t's row u in ascending j,
// acc[j] += x * y; every touched j is a stored entry of the result (no pruning
// of exact zeros).  That is what `spgemm_row` below does.
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

namespace {

thread_local std::string g_last_error_knn;

enum SimType : int32_t {
  COSINE = 0,
  ASYMMETRIC = 1,
  JACCARD = 2,
  TVERSKY = 3,
  P3ALPHA = 4,
  RP3BETA = 5
};

struct CsrD {
  int64_t rows = 0, cols = 0;
  std::vector<int64_t> indptr;
  std::vector<int32_t> indices;
  std::vector<double> data;
};

CsrD make_csr(int64_t rows, int64_t cols, const int64_t *indptr,
              const int32_t *indices, const double *data) {
  CsrD m;
  m.rows = rows;
  m.cols = cols;
  m.indptr.assign(indptr, indptr + rows + 1);
  m.indices.assign(indices, indices + indptr[rows]);
  m.data.assign(data, data + indptr[rows]);
  return m;
}

CsrD transpose(const CsrD &x) {
  CsrD t;
  t.rows = x.cols;
  t.cols = x.rows;
  t.indptr.assign(t.rows + 1, 0);
  int64_t nnz = x.indptr[x.rows];
  t.indices.resize(nnz);
  t.data.resize(nnz);
  for (int64_t p = 0; p < nnz; p++) t.indptr[x.indices[p] + 1]++;
  for (int64_t c = 0; c < t.rows; c++) t.indptr[c + 1] += t.indptr[c];
  std::vector<int64_t> cur(t.indptr.begin(), t.indptr.end() - 1);
  for (int64_t r = 0; r < x.rows; r++)
    for (int64_t p = x.indptr[r]; p < x.indptr[r + 1]; p++) {
      int64_t d = cur[x.indices[p]]++;
      t.indices[d] = static_cast<int32_t>(r);
      t.data[d] = x.data[p];
    }
  return t;
}

struct Computer {
  // X_arg is [N, n_features]; xt_rows is X_arg^T stored by rows u (i.e. the
  // CSC matrix X_t of knn.hpp:33 walked by its outer index).
  int32_t type;
  int64_t N, n_features;
  double shrinkage, alpha, beta;
  bool normalize;
  size_t n_threads, max_chunk_size;
  CsrD xt_rows;
  std::vector<double> norms;
};

void check_lower(double x, double low, const char *name) {  // argcheck.hpp:13-20
  if (x < low)
    throw std::invalid_argument(std::string(name) +
                                " must be greater than or equal to  " +
                                std::to_string(low));
}

Computer *build(int32_t type, const CsrD &X_arg, double shrinkage, double alpha,
                double beta, bool normalize, int64_t n_threads,
                int64_t max_chunk_size) {
  // KNNComputer ctor, knn.hpp:30-41
  check_lower(shrinkage, 0, "shrinkage");
  if (n_threads < 1)
    throw std::invalid_argument("n_threads must be greater than or equal to  1");
  if (max_chunk_size < 1)
    throw std::invalid_argument(
        "max_chunk_size must be greater than or equal to  1");
  Computer *c = new Computer;
  c->type = type;
  c->N = X_arg.rows;
  c->n_features = X_arg.cols;
  c->shrinkage = shrinkage;
  c->alpha = alpha;
  c->beta = beta;
  c->normalize = normalize;
  c->n_threads = n_threads;
  c->max_chunk_size = max_chunk_size;
  c->norms.assign(c->N, 0.0);
  CsrD X = X_arg;  // per-column statistics of X_t == per-row statistics of X_arg
  try {
    switch (type) {
    case COSINE:  // similarities.hpp:20-28
      for (int64_t i = 0; i < X.rows; i++) {
        double s = 0;
        for (int64_t p = X.indptr[i]; p < X.indptr[i + 1]; p++)
          s += X.data[p] * X.data[p];
        c->norms[i] = std::sqrt(s);
      }
      break;
    case ASYMMETRIC:  // similarities.hpp:61-72
      if (alpha < 0)
        throw std::invalid_argument("alpha must be greater than or equal to  0");
      if (alpha > 1)
        throw std::invalid_argument("alpha must be less than or equal to  1");
      for (int64_t i = 0; i < X.rows; i++) {
        double s = 0;
        for (int64_t p = X.indptr[i]; p < X.indptr[i + 1]; p++)
          s += X.data[p] * X.data[p];
        c->norms[i] = std::pow(s, 1 - alpha);
      }
      break;
    case TVERSKY:  // similarities.hpp:143-159
      check_lower(alpha, 0, "alpha");
      check_lower(beta, 0, "beta");
      [[fallthrough]];  // same binarisation + column sums as Jaccard
    case JACCARD:  // similarities.hpp:96-107
      for (auto &v : X.data) v = 1;
      for (int64_t i = 0; i < X.rows; i++)
        c->norms[i] = static_cast<double>(X.indptr[i + 1] - X.indptr[i]);
      break;
    case RP3BETA:  // similarities.hpp:265-292
      check_lower(alpha, 0, "alpha");
      check_lower(beta, 0, "beta");
      [[fallthrough]];  // same constructor normalisation as P3alpha
    case P3ALPHA: {  // similarities.hpp:198-222
      check_lower(alpha, 0, "alpha");
      // X_t is [n_features, N] col-major; the loops at :214-222 run over
      // X_t's columns (= rows of X_arg) and normalise by norm_temp(iter.col())
      // where iter.col() is the *outer* index, i.e. each row of X_arg is
      // pow-ed and divided by its own sum.
      for (int64_t i = 0; i < X.rows; i++) {
        double s = 0;
        for (int64_t p = X.indptr[i]; p < X.indptr[i + 1]; p++) {
          X.data[p] = std::pow(X.data[p], alpha);
          s += X.data[p];
        }
        for (int64_t p = X.indptr[i]; p < X.indptr[i + 1]; p++) X.data[p] /= s;
      }
      break;
    }
    default:
      throw std::invalid_argument("unknown similarity type");
    }
  } catch (...) {
    delete c;
    throw;
  }
  c->xt_rows = transpose(X);
  return c;
}

// compute_similarity_imple for one target row (similarities.hpp:29-46, 73-87,
// 109-130, 161-184, 242-250, 326-334) followed by the per-row top-k of
// compute_similarity_triple (knn.hpp:111-136).
struct RowScratch {
  std::vector<double> acc;
  std::vector<uint8_t> touched;
  std::vector<int32_t> cols;
  std::vector<double> vals;
  std::vector<int32_t> buffer;
};

void similarity_row(const Computer &c, const CsrD &target, int64_t r,
                    size_t top_k, RowScratch &s, std::vector<int32_t> &out_cols,
                    std::vector<double> &out_vals) {
  const bool binarise = c.type == JACCARD || c.type == TVERSKY;
  s.cols.clear();
  for (int64_t p = target.indptr[r]; p < target.indptr[r + 1]; p++) {
    const int32_t u = target.indices[p];
    const double y = binarise ? 1.0 : target.data[p];
    for (int64_t q = c.xt_rows.indptr[u]; q < c.xt_rows.indptr[u + 1]; q++) {
      const int32_t j = c.xt_rows.indices[q];
      if (!s.touched[j]) {
        s.touched[j] = 1;
        s.cols.push_back(j);
        s.acc[j] = 0.0;
      }
      s.acc[j] += c.xt_rows.data[q] * y;
    }
  }
  std::sort(s.cols.begin(), s.cols.end());
  const size_t nz = s.cols.size();
  s.vals.resize(nz);
  double target_stat = 0;
  switch (c.type) {
  case COSINE: {
    double t = 0;
    for (int64_t p = target.indptr[r]; p < target.indptr[r + 1]; p++)
      t += target.data[p] * target.data[p];
    target_stat = std::sqrt(t);  // similarities.hpp:39
    break;
  }
  case ASYMMETRIC: {
    double t = 0;
    for (int64_t p = target.indptr[r]; p < target.indptr[r + 1]; p++)
      t += target.data[p] * target.data[p];
    target_stat = std::pow(t, c.alpha);  // similarities.hpp:78-79
    break;
  }
  case JACCARD:
  case TVERSKY:
    target_stat =
        static_cast<double>(target.indptr[r + 1] - target.indptr[r]);  // :122,174
    break;
  default:
    break;
  }
  for (size_t i = 0; i < nz; i++) {
    const int32_t j = s.cols[i];
    double v = s.acc[j];
    s.touched[j] = 0;
    switch (c.type) {
    case COSINE:
      if (c.normalize) v /= (c.norms[j] * target_stat + c.shrinkage + 1e-6);
      break;
    case ASYMMETRIC:
      v /= (c.norms[j] * target_stat + c.shrinkage + 1e-6);
      break;
    case JACCARD:
      v /= (c.norms[j] + target_stat - v + c.shrinkage + 1e-6);
      break;
    case TVERSKY:
      v /= (v + c.beta * (c.norms[j] - v) + c.alpha * (target_stat - v) +
            c.shrinkage + 1e-6);
      break;
    default:
      break;
    }
    s.vals[i] = v;
  }
  // knn.hpp:111-136
  const size_t col_size = std::min(nz, top_k);
  s.buffer.resize(nz);
  for (size_t i = 0; i < nz; i++) s.buffer[i] = static_cast<int32_t>(i);
  const auto score_descending = [&s](int32_t a, int32_t b) {
    if (s.vals[a] != s.vals[b]) return s.vals[a] > s.vals[b];
    return s.cols[a] < s.cols[b];
  };
  if (col_size < nz)
    std::nth_element(s.buffer.begin(), s.buffer.begin() + col_size,
                     s.buffer.end(), score_descending);
  std::sort(s.buffer.begin(), s.buffer.begin() + col_size);
  out_cols.clear();
  out_vals.clear();
  for (size_t j = 0; j < col_size; j++) {
    out_cols.push_back(s.cols[s.buffer[j]]);
    out_vals.push_back(s.vals[s.buffer[j]]);
  }
}

struct Result {
  std::vector<int64_t> indptr;
  std::vector<int32_t> indices;
  std::vector<double> data;
};

// KNNComputer::compute_similarity, knn.hpp:43-83: contiguous row ranges per
// thread, concatenated in thread order.
Result compute_similarity(const Computer &c, const CsrD &target, size_t top_k) {
  if (target.cols != c.n_features)
    throw std::invalid_argument("illegal # of feature.");  // knn.hpp:44-45
  const int64_t R = target.rows;
  std::vector<std::vector<int32_t>> rc(R);
  std::vector<std::vector<double>> rv(R);
  const size_t T = std::max<size_t>(1, c.n_threads);
  auto work = [&](int64_t b, int64_t e) {
    RowScratch s;
    s.acc.assign(c.N, 0.0);
    s.touched.assign(c.N, 0);
    for (int64_t r = b; r < e; r++) similarity_row(c, target, r, top_k, s, rc[r], rv[r]);
  };
  std::vector<std::thread> th;
  int64_t start = 0;
  for (size_t t = 0; t < T; t++) {
    int64_t bs = R / T + (static_cast<int64_t>(t) < R % static_cast<int64_t>(T) ? 1 : 0);
    if (t + 1 < T)
      th.emplace_back(work, start, start + bs);
    else
      work(start, start + bs);
    start += bs;
  }
  for (auto &t : th) t.join();
  Result res;
  res.indptr.assign(R + 1, 0);
  for (int64_t r = 0; r < R; r++) res.indptr[r + 1] = res.indptr[r] + rc[r].size();
  res.indices.reserve(res.indptr[R]);
  res.data.reserve(res.indptr[R]);
  for (int64_t r = 0; r < R; r++) {
    res.indices.insert(res.indices.end(), rc[r].begin(), rc[r].end());
    res.data.insert(res.data.end(), rv[r].begin(), rv[r].end());
  }
  return res;
}

// P3alphaComputer::compute_W / RP3betaComputer::compute_W target preparation
// (similarities.hpp:224-240, 294-324).  The transposition of the result to CSC
// is left to the caller.
CsrD prepare_w_target(const Computer &c, const CsrD &arg) {
  CsrD t = arg;
  std::vector<double> norm_temp(arg.cols, 0.0);
  std::vector<double> pop(arg.rows, 0.0);
  if (c.type == RP3BETA) {
    for (int64_t i = 0; i < t.rows; i++)
      for (int64_t p = t.indptr[i]; p < t.indptr[i + 1]; p++) pop[i] += t.data[p];
    for (auto &v : pop) v = std::pow(v, c.beta);
  }
  for (int64_t i = 0; i < t.rows; i++)
    for (int64_t p = t.indptr[i]; p < t.indptr[i + 1]; p++) {
      t.data[p] = std::pow(t.data[p], c.alpha);
      norm_temp[t.indices[p]] += t.data[p];
    }
  for (int64_t i = 0; i < t.rows; i++)
    for (int64_t p = t.indptr[i]; p < t.indptr[i + 1]; p++) {
      if (c.type == RP3BETA)
        t.data[p] /= (norm_temp[t.indices[p]] * pop[i]);
      else
        t.data[p] /= norm_temp[t.indices[p]];
    }
  return t;
}

template <class F> int guard(F &&f) {
  try {
    f();
    return 0;
  } catch (const std::invalid_argument &e) {
    g_last_error_knn = e.what();
    return 1;
  } catch (const std::exception &e) {
    g_last_error_knn = e.what();
    return 2;
  }
}

}  // namespace

extern "C" {

const char *orc_knn_last_error() { return g_last_error_knn.c_str(); }

int orc_knn_create(int32_t type, int64_t rows, int64_t cols,
                   const int64_t *indptr, const int32_t *indices,
                   const double *data, double shrinkage, double alpha,
                   double beta, int32_t normalize, int64_t n_threads,
                   int64_t max_chunk_size, void **out) {
  return guard([&] {
    CsrD X = make_csr(rows, cols, indptr, indices, data);
    *out = build(type, X, shrinkage, alpha, beta, normalize != 0, n_threads,
                 max_chunk_size);
  });
}

void orc_knn_destroy(void *h) { delete static_cast<Computer *>(h); }

// Two-call protocol: compute fills an internal result and returns nnz;
// fetch copies it out.  `as_w` != 0 applies the compute_W target preparation.
struct Pending {
  Result r;
};
static thread_local Pending g_pending;

int orc_knn_compute(void *h, int64_t rows, int64_t cols, const int64_t *indptr,
                    const int32_t *indices, const double *data, int64_t top_k,
                    int32_t as_w, int64_t *nnz_out) {
  Computer *c = static_cast<Computer *>(h);
  return guard([&] {
    CsrD target = make_csr(rows, cols, indptr, indices, data);
    if (as_w) target = prepare_w_target(*c, target);
    g_pending.r = compute_similarity(*c, target, static_cast<size_t>(top_k));
    *nnz_out = g_pending.r.indptr.back();
  });
}

void orc_knn_fetch(int64_t *indptr, int32_t *indices, double *data) {
  std::copy(g_pending.r.indptr.begin(), g_pending.r.indptr.end(), indptr);
  std::copy(g_pending.r.indices.begin(), g_pending.r.indices.end(), indices);
  std::copy(g_pending.r.data.begin(), g_pending.r.data.end(), data);
  g_pending.r = Result();
}

// util.hpp:211-226 remove_diagonal: zero stored diagonal entries in place.
int orc_remove_diagonal(int64_t rows, int64_t cols, const int64_t *indptr,
                        const int32_t *indices, double *data) {
  return guard([&] {
    if (rows != cols) throw std::invalid_argument("X must be square");
    for (int64_t i = 0; i < rows; i++)
      for (int64_t p = indptr[i]; p < indptr[i + 1]; p++)
        if (indices[p] == i) data[p] = 0.0;
  });
}

// util.hpp:190-209 tf_idf_weight (in place on data).
int orc_tf_idf_weight(int64_t rows, int64_t cols, const int64_t *indptr,
                      const int32_t *indices, double *data, int32_t smooth) {
  return guard([&] {
    std::vector<double> idf(cols, 0.0);
    for (int64_t p = 0; p < indptr[rows]; p++) idf[indices[p]] += 1;
    for (auto &v : idf) v = std::log(rows / (v + static_cast<double>(smooth != 0)));
    for (int64_t p = 0; p < indptr[rows]; p++) data[p] *= idf[indices[p]];
  });
}

// util.hpp:158-188 okapi_BM_25_weight (in place on data).
int orc_bm25_weight(int64_t rows, int64_t cols, const int64_t *indptr,
                    const int32_t *indices, double *data, double k1, double b) {
  return guard([&] {
    std::vector<double> idf(cols, 0.0), dl(rows, 0.0);
    for (int64_t i = 0; i < rows; i++)
      for (int64_t p = indptr[i]; p < indptr[i + 1]; p++) {
        idf[indices[p]] += 1;
        dl[i] += data[p];
      }
    double total = 0;
    for (auto v : dl) total += v;
    const double avgdl = total / rows;
    for (auto &v : idf) v = std::log(rows / (v + 1.0) + 1.0);
    for (int64_t i = 0; i < rows; i++) {
      const double regularizer = k1 * (1 - b + b * dl[i] / avgdl);
      for (int64_t p = indptr[i]; p < indptr[i + 1]; p++)
        data[p] = idf[indices[p]] * (data[p] * (k1 + 1)) / (data[p] + regularizer);
    }
  });
}

}  // extern "C"
