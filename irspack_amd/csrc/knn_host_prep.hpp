// Multi-threaded host helpers of the kNN path (no HIP dependency): the validated fp64 CSR copy,
// the counting-sort transpose (X_arg^T, knn.hpp:30-41) and the per-row parallel-for of the
// target pass.  Included by knn.hip; compiled on its own under ThreadSanitizer /
// AddressSanitizer by tests/test_host_sanitizers.py.
#pragma once
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstring>
#include <thread>

#include "host_util.hpp"

namespace irs {
namespace knn {

struct HostCsrD {
  int64_t rows = 0, cols = 0;
  std::vector<int64_t> indptr;
  RawVector<int32_t> indices;  // (sized, then written by several threads: no zero fill)
  RawVector<double> data;
};

// f(i) for every row i of a CSR on several host threads (row blocks of about equal entry
// counts); f must only touch its own row
template <class F>
static void for_rows_parallel(const std::vector<int64_t> &indptr, int64_t rows, F &&f) {
  const int64_t nnz = rows > 0 ? indptr[rows] : 0;
  const int n_thr = static_cast<int>(std::max<int64_t>(
      1, std::min<int64_t>({16, static_cast<int64_t>(std::thread::hardware_concurrency()),
                            nnz / 500000 + 1})));
  auto body = [&](int k) {
    const int64_t lo = std::lower_bound(indptr.begin(), indptr.begin() + rows, nnz * k / n_thr) - indptr.begin();
    const int64_t hi = k + 1 == n_thr ? rows
                                      : std::lower_bound(indptr.begin(), indptr.begin() + rows, nnz * (k + 1) / n_thr) - indptr.begin();
    for (int64_t i = (k == 0 ? 0 : lo); i < hi; i++) f(i);
  };
  std::vector<std::thread> th;
  for (int k = 1; k < n_thr; k++) th.emplace_back(body, k);
  body(0);
  for (auto &w : th) w.join();
}

static HostCsrD host_csr(int64_t rows, int64_t cols, const int64_t *indptr,
                         const int32_t *indices, const double *data) {
  check_arg(rows >= 0 && cols >= 0 && indptr, "bad matrix.");
  HostCsrD m;
  m.rows = rows;
  m.cols = cols;
  m.indptr.assign(indptr, indptr + rows + 1);
  const int64_t nnz = indptr[rows];
  check_arg(indptr[0] == 0 && nnz >= 0, "malformed indptr.");
  // the two copies and the index check, on a few host threads (240 MB for 20 M entries)
  m.indices.resize(nnz);
  m.data.resize(nnz);
  const int n_thr = static_cast<int>(std::max<int64_t>(
      1, std::min<int64_t>({8, static_cast<int64_t>(std::thread::hardware_concurrency()), nnz / 1000000 + 1})));
  std::atomic<int> bad(0);
  auto body = [&](int k) {
    const int64_t b = nnz * k / n_thr, e = nnz * (k + 1) / n_thr;
    if (e > b) {
      std::memcpy(m.indices.data() + b, indices + b, (e - b) * sizeof(int32_t));
      std::memcpy(m.data.data() + b, data + b, (e - b) * sizeof(double));
    }
    int32_t lo = 0, hi = 0;
    for (int64_t q = b; q < e; q++) {
      lo = std::min(lo, indices[q]);
      hi = std::max(hi, indices[q]);
    }
    if (e > b && (lo < 0 || hi >= cols)) bad.store(1);
  };
  {
    std::vector<std::thread> th;
    for (int k = 1; k < n_thr; k++) th.emplace_back(body, k);
    body(0);
    for (auto &w : th) w.join();
  }
  check_arg(bad.load() == 0, "column index out of range.");
  return m;
}

// Transpose on several host threads: every thread owns a contiguous block of rows (about equal
// entry counts), counts its entries per column, and after a prefix over (column, thread) writes
// them to its own slots - the entries of a column stay in row order whatever the thread count.
static HostCsrD transpose(const HostCsrD &x) {
  HostCsrD t;
  t.rows = x.cols;
  t.cols = x.rows;
  t.indptr.assign(t.rows + 1, 0);
  const int64_t nnz = x.indptr[x.rows];
  t.indices.resize(nnz);
  t.data.resize(nnz);
  const int n_thr = static_cast<int>(std::max<int64_t>(
      1, std::min<int64_t>({16, static_cast<int64_t>(std::thread::hardware_concurrency()),
                            nnz / 1000000 + 1, (int64_t(1) << 25) / std::max<int64_t>(x.cols, 1)})));
  std::vector<int64_t> row_lo(n_thr + 1, x.rows);
  row_lo[0] = 0;
  for (int k = 1; k < n_thr; k++)
    row_lo[k] = std::lower_bound(x.indptr.begin(), x.indptr.end() - 1, nnz * k / n_thr) -
                x.indptr.begin();
  // cnt[k][c]: entries of column c in the rows of thread k; turned into write positions below
  std::vector<std::vector<int64_t>> cnt(n_thr, std::vector<int64_t>(t.rows, 0));
  auto run = [&](auto &&body) {
    std::vector<std::thread> th;
    for (int k = 1; k < n_thr; k++) th.emplace_back(body, k);
    body(0);
    for (auto &w : th) w.join();
  };
  run([&](int k) {
    auto &c = cnt[k];
    for (int64_t q = x.indptr[row_lo[k]]; q < x.indptr[row_lo[k + 1]]; q++) c[x.indices[q]]++;
  });
  int64_t pos = 0;
  for (int64_t c = 0; c < t.rows; c++) {
    t.indptr[c] = pos;
    for (int k = 0; k < n_thr; k++) {
      const int64_t here = cnt[k][c];
      cnt[k][c] = pos;
      pos += here;
    }
  }
  t.indptr[t.rows] = pos;
  run([&](int k) {
    auto &cur = cnt[k];
    for (int64_t r = row_lo[k]; r < row_lo[k + 1]; r++)
      for (int64_t q = x.indptr[r]; q < x.indptr[r + 1]; q++) {
        const int64_t d = cur[x.indices[q]]++;
        t.indices[d] = static_cast<int32_t>(r);
        t.data[d] = x.data[q];
      }
  });
  return t;
}


// The pattern of X^T (no values) on several host threads: what a CSC-layout target needs when every
// stored value is 1.  Same scheme as transpose() above; the entries of a column stay in row order.
// `x_indices` must have been range-checked against x_cols.
static HostCsrD transpose_pattern(int64_t x_rows, int64_t x_cols, const int64_t *x_indptr,
                                  const int32_t *x_indices, const double *x_data /* null: pattern only */) {
  HostCsrD t;
  t.rows = x_cols;
  t.cols = x_rows;
  t.indptr.assign(t.rows + 1, 0);
  const int64_t nnz = x_indptr[x_rows];
  t.indices.resize(nnz);
  if (x_data) t.data.resize(nnz);
  // (up to 48 threads: 20 M entries on the pool's 256-thread hosts - 8 threads 25-31 ms, 16: 18-25, 32-48:
  // 13-15, 96+: no better; the pass is bound by the host's memory system)
  const int64_t cap = 48;
  const int n_thr = static_cast<int>(std::max<int64_t>(
      1, std::min<int64_t>({cap, static_cast<int64_t>(std::thread::hardware_concurrency()),
                            nnz / 500000 + 1, (int64_t(1) << 25) / std::max<int64_t>(x_cols, 1)})));
  std::vector<int64_t> row_lo(n_thr + 1, x_rows);
  row_lo[0] = 0;
  for (int k = 1; k < n_thr; k++)
    row_lo[k] = std::lower_bound(x_indptr, x_indptr + x_rows, nnz * k / n_thr) - x_indptr;
  std::vector<std::vector<int32_t>> cnt(n_thr);
  run_on_threads(n_thr, [&](int k) {
    auto &c = cnt[k];
    c.assign(static_cast<size_t>(t.rows), 0);
    for (int64_t q = x_indptr[row_lo[k]]; q < x_indptr[row_lo[k + 1]]; q++) c[x_indices[q]]++;
  });
  // slot of (column c, thread k) = prefix in that order; the threads' counters become write cursors
  // relative to the column's first slot (32 bits: a column holds fewer than 2^31 entries)
  int64_t pos = 0;
  for (int64_t c = 0; c < t.rows; c++) {
    t.indptr[c] = pos;
    int32_t within = 0;
    for (int k = 0; k < n_thr; k++) {
      const int32_t here = cnt[k][c];
      cnt[k][c] = within;
      within += here;
    }
    pos += within;
  }
  t.indptr[t.rows] = pos;
  run_on_threads(n_thr, [&](int k) {
    auto &cur = cnt[k];
    for (int64_t r = row_lo[k]; r < row_lo[k + 1]; r++)
      for (int64_t q = x_indptr[r]; q < x_indptr[r + 1]; q++) {
        const int32_t c = x_indices[q];
        const int64_t d = t.indptr[c] + cur[c]++;
        t.indices[d] = static_cast<int32_t>(r);
        if (x_data) t.data[d] = x_data[q];
      }
  });
  return t;
}

// Stored entries per column of a CSR (range-checked indices) on several host threads.
static std::vector<int64_t> column_counts(int64_t rows, int64_t cols, const int64_t *indptr, const int32_t *indices) {
  const int64_t nnz = indptr[rows];
  const int n_thr = static_cast<int>(std::max<int64_t>(
      1, std::min<int64_t>({16, static_cast<int64_t>(std::thread::hardware_concurrency()),
                            nnz / 1000000 + 1, (int64_t(1) << 25) / std::max<int64_t>(cols, 1)})));
  std::vector<std::vector<int32_t>> cnt(n_thr);
  run_on_threads(n_thr, [&](int k) {
    cnt[k].assign(static_cast<size_t>(cols), 0);
    auto &c = cnt[k];
    for (int64_t q = nnz * k / n_thr; q < nnz * (k + 1) / n_thr; q++) c[indices[q]]++;
  });
  std::vector<int64_t> out(static_cast<size_t>(cols), 0);
  parallel_ranges(cols, [&](int64_t b, int64_t e) {
    for (int k = 0; k < n_thr; k++)
      for (int64_t c = b; c < e; c++) out[c] += cnt[k][c];
  }, 16, 100000);
  return out;
}

// The two tables of the feature weightings (util.hpp:159-209), from the matrix AS STORED (rows = documents):
//   tf-idf : idf[c] = log(N / (df[c] + smooth))                                  (:196-203)
//   BM25   : idf[c] = log(N / (df[c] + 1) + 1), reg[r] = k1 (1 - b + b dl[r] / avgdl),
//            dl[r] = the row's values summed in entry order, avgdl = (dl summed in row order) / N   (:168-181)
// Every operation is a single IEEE double operation in the order the reference's expression has
// (this file is compiled with -ffp-contract=off); log is libm's.  `ones`: every stored value is 1.
struct WeightTables {
  std::vector<double> idf, reg;  // reg: BM25 only
};
static WeightTables weight_tables(bool bm25, int64_t rows, int64_t cols, const int64_t *indptr,
                                  const int32_t *indices, const double *data, bool ones, double k1, double b,
                                  bool smooth) {
  WeightTables w;
  const std::vector<int64_t> df = column_counts(rows, cols, indptr, indices);
  w.idf.resize(static_cast<size_t>(cols));
  const double N = static_cast<double>(rows);
  if (!bm25) {
    const double sm = smooth ? 1.0 : 0.0;
    parallel_ranges(cols, [&](int64_t lo, int64_t hi) {
      for (int64_t c = lo; c < hi; c++) w.idf[c] = std::log(N / (static_cast<double>(df[c]) + sm));
    }, 16, 20000);
    return w;
  }
  parallel_ranges(cols, [&](int64_t lo, int64_t hi) {
    for (int64_t c = lo; c < hi; c++) w.idf[c] = std::log(N / (static_cast<double>(df[c]) + 1.0) + 1.0);
  }, 16, 20000);
  std::vector<double> dl(static_cast<size_t>(rows), 0.0);
  parallel_ranges(rows, [&](int64_t lo, int64_t hi) {
    for (int64_t r = lo; r < hi; r++) {
      double sum = 0.0;
      if (ones) sum = static_cast<double>(indptr[r + 1] - indptr[r]);  // (a sum of 1.0s is exact)
      else
        for (int64_t q = indptr[r]; q < indptr[r + 1]; q++) sum += data[q];
      dl[r] = sum;
    }
  }, 16, 20000);
  double total = 0.0;
  for (double v : dl) total += v;
  const double avgdl = total / N;
  w.reg.resize(static_cast<size_t>(rows));
  parallel_ranges(rows, [&](int64_t lo, int64_t hi) {
    for (int64_t r = lo; r < hi; r++) w.reg[r] = k1 * (1 - b + b * dl[r] / avgdl);
  }, 16, 20000);
  return w;
}

// the per-entry pass of the weightings on the host (paths that never reach the device kernel)
static void weight_values_host(bool bm25, const WeightTables &w, int64_t rows, const int64_t *indptr,
                               const int32_t *indices, const double *data, double k1, double *out) {
  const std::vector<int64_t> ip(indptr, indptr + rows + 1);
  for_rows_parallel(ip, rows, [&](int64_t r) {
    for (int64_t q = indptr[r]; q < indptr[r + 1]; q++) {
      const double v = data[q];
      out[q] = bm25 ? w.idf[indices[q]] * (v * (k1 + 1)) / (v + w.reg[r]) : v * w.idf[indices[q]];
    }
  });
}

}  // namespace knn
}  // namespace irs
