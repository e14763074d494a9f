// Multi-threaded host helpers of the kNN path (no HIP dependency): the validated fp64 CSR copy,
// the counting-sort transpose (X_arg^T, knn.hpp:30-41) and the per-row parallel-for of the
// target pass.  Included by knn.hip; compiled on its own under ThreadSanitizer /
// AddressSanitizer by tests/test_host_sanitizers.py.
#pragma once
#include <algorithm>
#include <atomic>
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

}  // namespace knn
}  // namespace irs
