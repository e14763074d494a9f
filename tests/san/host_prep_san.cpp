// Sanitizer harness of the multi-threaded HOST preparation of libirspack_amd.so (no GPU): the
// validated CSR copy, the counting-sort transposes and the parallel libstdc++ random stream of
// irspack_amd/csrc/host_prep.hpp, and the kNN target pass helpers of knn_host_prep.hpp.  Built
// and run by tests/test_host_sanitizers.py with -fsanitize=thread and -fsanitize=address,undefined.
// Every multi-threaded result is compared with a sequential evaluation.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "../../irspack_amd/csrc/host_prep.hpp"
#include "../../irspack_amd/csrc/knn_host_prep.hpp"

using namespace irs;

#define CHECK(cond)                                                       \
  do {                                                                    \
    if (!(cond)) {                                                        \
      std::fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); \
      std::exit(1);                                                       \
    }                                                                     \
  } while (0)

int main() {
  // ---- a random CSR large enough for every helper to take several threads
  const int64_t rows = 30000, cols = 4000;
  std::mt19937_64 rng(7);
  std::vector<int64_t> indptr(rows + 1, 0);
  std::vector<int32_t> indices;
  std::vector<float> data;
  for (int64_t r = 0; r < rows; r++) {
    const int deg = static_cast<int>(rng() % 200);
    std::vector<char> seen(cols, 0);
    std::vector<int32_t> row;
    for (int d = 0; d < deg; d++) {
      const int32_t c = static_cast<int32_t>(rng() % cols);
      if (!seen[c]) {
        seen[c] = 1;
        row.push_back(c);
      }
    }
    std::sort(row.begin(), row.end());
    for (int32_t c : row) {
      indices.push_back(c);
      data.push_back(static_cast<float>(rng() % 1000) / 100.0f + 0.5f);
    }
    indptr[r + 1] = static_cast<int64_t>(indices.size());
  }
  const int64_t nnz = indptr[rows];
  CHECK(nnz > 2500000);
  // ---- iALS: copy + validation, transpose (counting sort on several threads)
  const ials::HostCsr X = ials::host_csr(rows, cols, indptr.data(), indices.data(), data.data());
  CHECK(X.indptr == indptr);
  CHECK(std::equal(indices.begin(), indices.end(), X.indices.begin()));
  const ials::HostCsr Xt = ials::transpose(X);
  {  // sequential transpose
    std::vector<int64_t> tp(cols + 1, 0);
    for (int64_t p = 0; p < nnz; p++) tp[indices[p] + 1]++;
    for (int64_t c = 0; c < cols; c++) tp[c + 1] += tp[c];
    CHECK(Xt.indptr == tp);
    std::vector<int64_t> cur(tp.begin(), tp.end() - 1);
    std::vector<int32_t> ti(nnz);
    std::vector<float> td(nnz);
    for (int64_t r = 0; r < rows; r++)
      for (int64_t p = indptr[r]; p < indptr[r + 1]; p++) {
        const int64_t d = cur[indices[p]]++;
        ti[d] = static_cast<int32_t>(r);
        td[d] = data[p];
      }
    CHECK(std::equal(ti.begin(), ti.end(), Xt.indices.begin()));
    CHECK(std::equal(td.begin(), td.end(), Xt.data.begin()));
  }
  {  // view mode: the caller's index array is validated in place and not copied; same transpose
    const ials::HostCsr V = ials::host_csr(rows, cols, indptr.data(), indices.data(), data.data(), true);
    CHECK(V.indices.empty() && V.idx() == indices.data() && V.data.size() == static_cast<size_t>(nnz));
    const ials::HostCsr Vt = ials::transpose(V);
    CHECK(Vt.indptr == Xt.indptr && std::equal(Vt.indices.begin(), Vt.indices.end(), Xt.indices.begin()));
    CHECK(std::equal(Vt.data.begin(), Vt.data.end(), Xt.data.begin()));
  }
  {  // binary interactions: the values are classified on the validation pass and never copied
    CHECK(X.flags_known && !X.unit && X.positive && Xt.flags_known && !Xt.unit);
    const std::vector<float> ones(nnz, 1.0f);
    const ials::HostCsr U = ials::host_csr(rows, cols, indptr.data(), indices.data(), ones.data());
    CHECK(U.flags_known && U.unit && U.positive && U.data.empty());
    const ials::HostCsr Ut = ials::transpose(U);
    CHECK(Ut.unit && Ut.data.empty() && Ut.indptr == Xt.indptr);
    CHECK(std::equal(Ut.indices.begin(), Ut.indices.end(), Xt.indices.begin()));
    std::vector<float> neg(ones);
    neg[nnz / 2] = -0.0f;
    const ials::HostCsr G = ials::host_csr(rows, cols, indptr.data(), indices.data(), neg.data());
    CHECK(!G.unit && !G.positive && G.data.size() == static_cast<size_t>(nnz) && !std::signbit(G.data[nnz / 2]));
  }
  {  // a rank's shard: rows [rb, re) of X and the columns [cb, ce) of X as rows of X^T
    const int64_t rb = 1000, re = 17000, cb = 100, ce = 2100;
    const ials::HostCsr S = ials::host_csr_rows(rows, cols, indptr.data(), indices.data(), data.data(), rb, re);
    CHECK(S.indptr[rows] == indptr[re] - indptr[rb] && S.indptr[rb] == 0);
    const ials::HostCsr T = ials::transpose_cols(rows, cols, indptr.data(), indices.data(), data.data(), cb, ce);
    for (int64_t c = cb; c < ce; c++) {
      CHECK(T.indptr[c + 1] - T.indptr[c] == Xt.indptr[c + 1] - Xt.indptr[c]);
      CHECK(std::equal(T.indices.begin() + T.indptr[c], T.indices.begin() + T.indptr[c + 1],
                       Xt.indices.begin() + Xt.indptr[c]));
    }
    CHECK(T.indptr[cb] == 0 && T.indptr[cols] == T.indptr[ce]);
  }
  // ---- the bulk engine is std::mt19937, word for word (odd chunk sizes cross the state blocks)
  for (const uint32_t seed : {42u, 0u, 5489u, 0xffffffffu}) {
    ials::Mt19937Bulk bulk(seed);
    std::mt19937 ref(seed);
    std::vector<uint32_t> got(100003);
    for (const size_t chunk : {size_t(1), size_t(623), size_t(624), size_t(625), size_t(100003), size_t(7)}) {
      bulk.fill(got.data(), chunk);
      for (size_t i = 0; i < chunk; i++) CHECK(got[i] == static_cast<uint32_t>(ref()));
    }
  }
  // ---- the parallel random stream is the sequential libstdc++ one, bit for bit (> 2^18 values)
  CHECK(mtjump::phi_low() == mtjump::phi_low_computed());  // (the tabulated characteristic polynomial)
  for (const int64_t K : {int64_t(64), int64_t(10), int64_t(33), int64_t(70)}) {
    // (K = 70: 4.9 M values - the jump-ahead path by the default threshold and thread / block policy)
    const int64_t n = K == 64 ? 6000 : K == 33 ? 20000 : K == 70 ? 70000 : 40000;
    const float stdev = 0.1f;
    // (K = 10: blocks of 30,000 attempts, i.e. eight blocks with the producer thread one ahead)
    // (K = 33: the jump-ahead path - every thread regenerates its own blocks of the engine's stream from a
    // state computed by polynomial arithmetic over GF(2), mt_jump.hpp - forced by a low threshold)
    const RawVector<float> par = (K == 64 || K == 70) ? ials::draw_factor(stdev, 42, K, n)
                                   : K == 33 ? ials::draw_factor(stdev, 42, K, n, size_t(1) << 24, size_t(1) << 18)
                                             : ials::draw_factor(stdev, 42, K, n, 30000);
    CHECK(par.size() == static_cast<size_t>(n * K) && par.size() >= (size_t(1) << 18));
    std::mt19937 gen(42);
    const float sd = static_cast<float>(static_cast<double>(stdev) / std::sqrt(static_cast<double>(K)));
    std::normal_distribution<float> dist(0.0, sd);
    for (size_t i = 0; i < par.size(); i++) {
      const float want = dist(gen);
      if (par[i] != want) {
        std::fprintf(stderr, "stream differs at %zu: %a vs %a\n", i, par[i], want);
        return 1;
      }
    }
  }
  // ---- kNN: parallel fp64 CSR copy, transpose, per-row pass
  {
    std::vector<double> dd(data.begin(), data.end());
    const knn::HostCsrD D = knn::host_csr(rows, cols, indptr.data(), indices.data(), dd.data());
    const knn::HostCsrD Dt = knn::transpose(D);
    CHECK(Dt.indptr == Xt.indptr);
    CHECK(std::equal(Dt.indices.begin(), Dt.indices.end(), Xt.indices.begin()));
    for (int64_t p = 0; p < nnz; p++) CHECK(Dt.data[p] == static_cast<double>(Xt.data[p]));
    std::vector<double> norms(rows, 0.0);
    knn::for_rows_parallel(D.indptr, rows, [&](int64_t r) {
      double s = 0;
      for (int64_t p = D.indptr[r]; p < D.indptr[r + 1]; p++) s += D.data[p] * D.data[p];
      norms[r] = s;
    });
    for (int64_t r = 0; r < rows; r += 997) {
      double s = 0;
      for (int64_t p = indptr[r]; p < indptr[r + 1]; p++) s += dd[p] * dd[p];
      CHECK(norms[r] == s);
    }
  }
  // ---- kNN, round 6: CSC-layout inputs (pattern-only and valued regrouping), column counts, the feature
  // weightings' tables and their per-entry pass - against sequential restatements of util.hpp:159-209
  {
    std::vector<double> dd(data.begin(), data.end());
    const knn::HostCsrD P = knn::transpose_pattern(rows, cols, indptr.data(), indices.data(), nullptr);
    CHECK(P.indptr == Xt.indptr && P.data.empty());
    CHECK(std::equal(P.indices.begin(), P.indices.end(), Xt.indices.begin()));
    const knn::HostCsrD V = knn::transpose_pattern(rows, cols, indptr.data(), indices.data(), dd.data());
    CHECK(std::equal(V.indices.begin(), V.indices.end(), Xt.indices.begin()));
    for (int64_t p = 0; p < nnz; p++) CHECK(V.data[p] == static_cast<double>(Xt.data[p]));
    const std::vector<int64_t> cnt = knn::column_counts(rows, cols, indptr.data(), indices.data());
    for (int64_t c = 0; c < cols; c++) CHECK(cnt[c] == Xt.indptr[c + 1] - Xt.indptr[c]);
    const double k1 = 1.3, b = 0.6;
    for (const bool bm25 : {false, true}) {
      const knn::WeightTables wt = knn::weight_tables(bm25, rows, cols, indptr.data(), indices.data(), dd.data(),
                                                      false, k1, b, true);
      std::vector<double> got(nnz), idf(cols, 0.0), dl(rows, 0.0);
      knn::weight_values_host(bm25, wt, rows, indptr.data(), indices.data(), dd.data(), k1, got.data());
      for (int64_t r = 0; r < rows; r++)
        for (int64_t p = indptr[r]; p < indptr[r + 1]; p++) {
          idf[indices[p]] += 1;
          dl[r] += dd[p];
        }
      double total = 0;
      for (double v : dl) total += v;
      const double avgdl = total / rows;
      for (auto &v : idf) v = bm25 ? std::log(rows / (v + 1.0) + 1.0) : std::log(rows / (v + 1.0));
      for (int64_t r = 0; r < rows; r += 7) {
        const double reg = k1 * (1 - b + b * dl[r] / avgdl);
        for (int64_t p = indptr[r]; p < indptr[r + 1]; p++) {
          const double want = bm25 ? idf[indices[p]] * (dd[p] * (k1 + 1)) / (dd[p] + reg) : dd[p] * idf[indices[p]];
          CHECK(got[p] == want);
        }
      }
    }
    // all-ones input: the row sums are the entry counts
    const std::vector<double> ones(nnz, 1.0);
    const knn::WeightTables a = knn::weight_tables(true, rows, cols, indptr.data(), indices.data(), ones.data(), true, k1, b, true);
    const knn::WeightTables c = knn::weight_tables(true, rows, cols, indptr.data(), indices.data(), ones.data(), false, k1, b, true);
    CHECK(a.idf == c.idf && a.reg == c.reg);
  }
  std::puts("host_prep_san ok");
  return 0;
}
