// Multi-threaded host preparation of the iALS trainer (no HIP dependency): the CSR copy with
// validation, the counting-sort transpose (X^T, hpp:713) and the parallel but bit-identical
// libstdc++ random stream of Solver::initialize (hpp:64-76).  Included by ials.hip; compiled on
// its own under ThreadSanitizer / AddressSanitizer by tests/test_host_sanitizers.py.
#pragma once
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstring>
#include <random>
#include <thread>
#if defined(__linux__)
#include <sys/mman.h>
#endif

#include "host_util.hpp"
#include "mt_jump.hpp"

namespace irs {
namespace ials {

struct HostCsr {
  int64_t rows = 0, cols = 0;
  std::vector<int64_t> indptr;
  RawVector<int32_t> indices;  // (sized, then written: no zero fill of hundreds of MB)
  RawVector<float> data;       // EMPTY when `unit` (every stored value is exactly 1: nothing to carry)
  // known from the validation pass of host_csr (flags_known): every stored value == 1 / > 0
  bool flags_known = false, unit = false, positive = false;
  // host_csr(..., view = true): the caller's index array itself (validated in place, valid for the duration
  // of the call that made it) instead of a copy in `indices` - nothing to allocate, fault in or unmap
  const int32_t *indices_view = nullptr;
  const int32_t *idx() const { return indices_view ? indices_view : indices.data(); }
};

// Stored values as the device kernels take them: -0.0 becomes +0.0 (x + 0.0f; every other value,
// NaN included, is unchanged).  The general-confidence rank update marks an entry past a row's end
// with c = -0.0 (ials_kernels.hpp: syrk_gather), so the data must not contain that bit pattern.
static inline void canonical_copy(float *dst, const float *src, int64_t n) {
  for (int64_t i = 0; i < n; i++) dst[i] = src[i] + 0.0f;
}

static HostCsr host_csr(int64_t rows, int64_t cols, const int64_t *indptr,
                        const int32_t *indices, const float *data, bool view = false) {
  check_arg(rows >= 0 && cols >= 0, "negative matrix shape.");
  check_arg(indptr != nullptr, "indptr is null.");
  HostCsr m;
  m.rows = rows;
  m.cols = cols;
  m.indptr.assign(indptr, indptr + rows + 1);
  const int64_t nnz = indptr[rows];
  check_arg(indptr[0] == 0 && nnz >= 0, "malformed indptr.");
  check_arg(nnz < (int64_t(1) << 31), "nnz must be below 2^31 (32-bit CSR like Eigen's).");
  for (int64_t r = 0; r < rows; r++) check_arg(indptr[r + 1] >= indptr[r], "malformed indptr.");
  // the two copies and the index check on a few host threads
  if (view) m.indices_view = indices;
  else m.indices.resize(nnz);
  const int n_thr = static_cast<int>(std::max<int64_t>(
      1, std::min<int64_t>({16, static_cast<int64_t>(std::thread::hardware_concurrency()), nnz / 1000000 + 1})));
  std::atomic<int> bad(0), not_unit(0), not_positive(0);
  auto run = [&](auto &&body) {
    std::vector<std::thread> th;
    for (int k = 1; k < n_thr; k++) th.emplace_back(body, k);
    body(0);
    for (auto &w : th) w.join();
  };
  // pass 1: the index copy + range check, and what the values are (all exactly 1: binary interactions -
  // then no copy of them is made at all; all positive)
  run([&](int k) {
    const int64_t b = nnz * k / n_thr, e = nnz * (k + 1) / n_thr;
    if (e <= b) return;
    if (!view) std::memcpy(m.indices.data() + b, indices + b, (e - b) * sizeof(int32_t));
    int32_t lo = 0, hi = 0;
    bool one = true, pos = true;
    for (int64_t q = b; q < e; q++) {
      lo = std::min(lo, indices[q]);
      hi = std::max(hi, indices[q]);
      one &= data[q] == 1.0f;
      pos &= data[q] > 0.0f;
    }
    if (lo < 0 || hi >= cols) bad.store(1);
    if (!one) not_unit.store(1);
    if (!pos) not_positive.store(1);
  });
  check_arg(bad.load() == 0, "column index out of range.");
  m.flags_known = true;
  m.unit = not_unit.load() == 0;
  m.positive = not_positive.load() == 0;
  if (!m.unit) {
    m.data.resize(nnz);
    run([&](int k) {
      const int64_t b = nnz * k / n_thr, e = nnz * (k + 1) / n_thr;
      if (e > b) canonical_copy(m.data.data() + b, data + b, e - b);
    });
  }
  return m;
}

// Rows [rb, re) of a CSR as a matrix of the same shape whose other rows are empty: what a
// rank of a sharded run needs of X (its user rows).  Only the slice is copied and validated.
static HostCsr host_csr_rows(int64_t rows, int64_t cols, const int64_t *indptr,
                             const int32_t *indices, const float *data, int64_t rb, int64_t re) {
  check_arg(rows >= 0 && cols >= 0, "negative matrix shape.");
  check_arg(indptr != nullptr, "indptr is null.");
  const int64_t nnz = indptr[rows];
  check_arg(indptr[0] == 0 && nnz >= 0, "malformed indptr.");
  check_arg(nnz < (int64_t(1) << 31), "nnz must be below 2^31 (32-bit CSR like Eigen's).");
  for (int64_t r = 0; r < rows; r++) check_arg(indptr[r + 1] >= indptr[r], "malformed indptr.");
  HostCsr m;
  m.rows = rows;
  m.cols = cols;
  m.indptr.assign(rows + 1, 0);
  const int64_t b = indptr[rb], e = indptr[re];
  for (int64_t r = rb; r <= rows; r++) m.indptr[r] = std::min(indptr[r], e) - b;
  m.indices.assign(indices + b, indices + e);
  m.data.resize(e - b);
  canonical_copy(m.data.data(), data + b, e - b);
  for (int64_t p = 0; p < e - b; p++)
    check_arg(m.indices[p] >= 0 && m.indices[p] < cols, "column index out of range.");
  return m;
}

// Rows [cb, ce) of X^T (the columns [cb, ce) of X), other rows empty: a rank's item rows.
// One scan of X; every column index is validated on the way.
static HostCsr transpose_cols(int64_t rows, int64_t cols, const int64_t *indptr,
                              const int32_t *indices, const float *data, int64_t cb, int64_t ce) {
  HostCsr t;
  t.rows = cols;
  t.cols = rows;
  t.indptr.assign(cols + 1, 0);
  const int64_t nnz = indptr[rows];
  for (int64_t p = 0; p < nnz; p++) {
    const int32_t c = indices[p];
    check_arg(c >= 0 && c < cols, "column index out of range.");
    if (c >= cb && c < ce) t.indptr[c + 1]++;
  }
  for (int64_t c = 0; c < cols; c++) t.indptr[c + 1] += t.indptr[c];
  t.indices.resize(t.indptr[cols]);
  t.data.resize(t.indptr[cols]);
  std::vector<int64_t> cur(t.indptr.begin() + cb, t.indptr.begin() + ce);
  for (int64_t r = 0; r < rows; r++)
    for (int64_t p = indptr[r]; p < indptr[r + 1]; p++) {
      const int32_t c = indices[p];
      if (c < cb || c >= ce) continue;
      const int64_t d = cur[c - cb]++;
      t.indices[d] = static_cast<int32_t>(r);
      t.data[d] = data[p] + 0.0f;  // (canonical_copy's rule)
    }
  return t;
}

// X.transpose() as compressed row-major (hpp:713)
// A counting sort by column on several host threads: thread k counts the columns of its row
// range, a prefix over (column, thread) gives every thread its first slot in every column, and
// the threads scatter their rows - the entries of a column stay in row order, so the result is
// the sequential one (it was 120 ms of the 270 ms trainer construction on the ML-20M shape).
static HostCsr transpose(const HostCsr &x) {
  HostCsr t;
  t.rows = x.cols;
  t.cols = x.rows;
  t.indptr.assign(t.rows + 1, 0);
  const int64_t nnz = x.indptr[x.rows];
  t.indices.resize(nnz);
  t.flags_known = x.flags_known;
  t.unit = x.unit;
  t.positive = x.positive;
  const bool carry = !(x.flags_known && x.unit);  // (all ones: nothing to move)
  const int32_t *xi = x.idx();
  if (carry) t.data.resize(nnz);
  const int64_t cols = x.cols;
  // threads: bounded by the counter memory (cols x threads x 8 B <= 256 MB) and the work
  int n_thr = static_cast<int>(std::max<int64_t>(
      1, std::min<int64_t>({16, static_cast<int64_t>(std::thread::hardware_concurrency()),
                            nnz / 500000 + 1, (int64_t(1) << 25) / std::max<int64_t>(cols, 1)})));
  std::vector<int64_t> rb(n_thr + 1, x.rows);
  for (int k = 0; k < n_thr; k++)  // row ranges of about equal entry counts
    rb[k] = std::lower_bound(x.indptr.begin(), x.indptr.begin() + x.rows, nnz * k / n_thr) - x.indptr.begin();
  rb[0] = 0;
  std::vector<std::vector<int64_t>> cnt(n_thr);
  auto run = [&](auto &&body) {
    std::vector<std::thread> th;
    for (int k = 1; k < n_thr; k++) th.emplace_back(body, k);
    body(0);
    for (auto &q : th) q.join();
  };
  run([&](int k) {
    cnt[k].assign(cols, 0);
    for (int64_t p = x.indptr[rb[k]]; p < x.indptr[rb[k + 1]]; p++) cnt[k][xi[p]]++;
  });
  int64_t run_sum = 0;
  for (int64_t c = 0; c < cols; c++) {  // slot of (column c, thread k) = prefix in that order
    t.indptr[c] = run_sum;
    for (int k = 0; k < n_thr; k++) {
      const int64_t n = cnt[k][c];
      cnt[k][c] = run_sum;
      run_sum += n;
    }
  }
  t.indptr[cols] = run_sum;
  run([&](int k) {
    std::vector<int64_t> &cur = cnt[k];
    for (int64_t r = rb[k]; r < rb[k + 1]; r++)
      for (int64_t p = x.indptr[r]; p < x.indptr[r + 1]; p++) {
        const int64_t d = cur[xi[p]]++;
        t.indices[d] = static_cast<int32_t>(r);
        if (carry) t.data[d] = x.data[p];
      }
  });
  return t;
}

// std::mt19937's output stream in bulk: the same recurrence and tempering (Matsumoto & Nishimura's
// MT19937, [rand.eng.mers]: w = 32, n = 624, m = 397, r = 31, a = 0x9908b0df, u = 11, s = 7,
// b = 0x9d2c5680, t = 15, c = 0xefc60000, l = 18, seeding multiplier 1812433253), but a whole
// state block is regenerated and tempered by three branch-free loops the compiler vectorises
// (the lag of the recurrence, n - m = 227, is far wider than a vector) instead of one call per
// word with its "block exhausted?" branch: ~0.5 ns a word against 2.5.
// (AVX2 clones of the two loops, chosen at load time: 0.67 ns a word against 1.28 with SSE2 and
// 2.5 for std::mt19937 on this toolchain; the device pass of hipcc parses this header too and
// knows no function multiversioning)
// (... and the ifunc resolver of a clone runs before a sanitizer's runtime is up: plain loops there)
#if defined(__has_feature)
#if __has_feature(thread_sanitizer) || __has_feature(address_sanitizer)
#define IRS_MT_NO_CLONES 1
#endif
#endif
#if defined(__SANITIZE_THREAD__) || defined(__SANITIZE_ADDRESS__)
#define IRS_MT_NO_CLONES 1
#endif
#if defined(__x86_64__) && defined(__linux__) && !defined(__HIP_DEVICE_COMPILE__) && !defined(IRS_MT_NO_CLONES)
#define IRS_MT_CLONES __attribute__((target_clones("avx2", "default")))
#else
#define IRS_MT_CLONES
#endif
class Mt19937Bulk {
 public:
  explicit Mt19937Bulk(uint32_t seed) {
    mt_[0] = seed;
    for (uint32_t i = 1; i < N; i++) mt_[i] = 1812433253u * (mt_[i - 1] ^ (mt_[i - 1] >> 30)) + i;
  }
  // an engine whose next block is regenerated from `window` (the raw state x_k .. x_{k+623}: mt_jump.hpp)
  struct FromWindow {};
  Mt19937Bulk(FromWindow, const uint32_t *window) { std::memcpy(mt_, window, sizeof(mt_)); }
  const uint32_t *window() const { return mt_; }  // (meaningful right after construction)
  // the next `count` outputs of the engine
  IRS_MT_CLONES void fill(uint32_t *out, size_t count) {
    size_t done = 0;
    while (done < count) {
      if (pos_ == N) {
        twist();
        pos_ = 0;
      }
      const size_t take = std::min<size_t>(count - done, N - pos_);
      const uint32_t *src = mt_ + pos_;
      uint32_t *dst = out + done;
      for (size_t i = 0; i < take; i++) {
        uint32_t y = src[i];
        y ^= y >> 11;
        y ^= (y << 7) & 0x9d2c5680u;
        y ^= (y << 15) & 0xefc60000u;
        y ^= y >> 18;
        dst[i] = y;
      }
      pos_ += take;
      done += take;
    }
  }

 private:
  static constexpr uint32_t N = 624, M = 397;
  static uint32_t mix(uint32_t hi, uint32_t lo, uint32_t far) {
    const uint32_t y = (hi & 0x80000000u) | (lo & 0x7fffffffu);
    return far ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
  }
  IRS_MT_CLONES void twist() {
    // x[k + n] = x[k + m] ^ twist(x[k], x[k + 1]); in place, in the order of [rand.eng.mers]
    for (uint32_t k = 0; k < N - M; k++) mt_[k] = mix(mt_[k], mt_[k + 1], mt_[k + M]);
    for (uint32_t k = N - M; k < N - 1; k++) mt_[k] = mix(mt_[k], mt_[k + 1], mt_[k + M - N]);
    mt_[N - 1] = mix(mt_[N - 1], mt_[0], mt_[M - 1]);
  }
  uint32_t mt_[N];
  uint32_t pos_ = N;
};

// Solver::initialize, hpp:64-76: libstdc++ mt19937 + normal_distribution<float>
// on the host, so a libstdc++ build of the reference draws the same stream.
// Both matrices are drawn from generators with the SAME seed (hpp:718-719), so the shorter one
// is a prefix of the longer one's stream: `n` rows are drawn once.
// `block_attempts`: attempts per block of the parallel path (tests shrink it).
// `jump_threshold`: matrices of at least that many values take the jump-ahead path (tests lower it).
// (RawVector: the 5 GB of the 10 M x 128 matrix were zero-filled - and page-faulted - by ONE thread before a
// single variate was written: 1 s of the 1.4 s the draw took; every element is written exactly once below.)
static RawVector<float> draw_factor(float init_stdev, int32_t random_seed, int64_t K, int64_t n,
                                    size_t block_attempts = size_t(1) << 24,
                                    size_t jump_threshold = size_t(1) << 22) {
  RawVector<float> h;
  h.resize(static_cast<size_t>(n) * K);
#if defined(__linux__)
  // gigabytes first touched by dozens of threads at once: 2 MB pages where the kernel grants them (a
  // million 4 KB faults contend for the address-space lock with every other allocating thread of the
  // process - the CSR preparation runs beside this)
  if (h.size() * sizeof(float) >= (size_t(64) << 20)) {
    const uintptr_t b = (reinterpret_cast<uintptr_t>(h.data()) + 4095) & ~uintptr_t(4095);
    const uintptr_t e = reinterpret_cast<uintptr_t>(h.data() + h.size()) & ~uintptr_t(4095);
    if (e > b) (void)madvise(reinterpret_cast<void *>(b), e - b, MADV_HUGEPAGE);
  }
#endif
  if (!(init_stdev > 0)) {  // the reference leaves the matrix uninitialised; we zero it
    std::fill(h.begin(), h.end(), 0.0f);
    return h;
  }
  // std::sqrt(factor.cols()) takes the integral overload (hpp:68-69): the quotient is formed
  // in double and rounded to float once
  const float sd = static_cast<float>(static_cast<double>(init_stdev) / std::sqrt(static_cast<double>(K)));
  const size_t total = h.size();
  if (total < (size_t(1) << 18)) {
    std::mt19937 gen(random_seed);
    std::normal_distribution<float> dist(0.0, sd);
    for (size_t i = 0; i < total; i++) h[i] = dist(gen);
    return h;
  }
  // Large matrices (1.3 G values at the 10 M x 1 M shape: 56 s of sequential
  // std::normal_distribution): the SAME stream, bit for bit, in parallel.  libstdc++'s
  // normal_distribution<float> (bits/random.tcc) is Marsaglia's polar method on
  // generate_canonical<float, 24>: every attempt consumes exactly two 32-bit words of the
  // engine, accepted or not, and an accepted attempt yields two variates (y * m first, the
  // saved x * m next).  So attempt j owns the words 2 j, 2 j + 1 whatever happened before it,
  // and the output position of an accepted attempt is twice the number of accepted attempts
  // before it.  The engine's words are the only serial part: one thread generates block b + 1
  // (Mt19937Bulk) while all the others evaluate the attempts of block b with a prefix count.
  auto canonical = [](uint32_t w) {
    const float r = static_cast<float>(w) / 4294967296.0f;
    return r >= 1.0f ? std::nextafter(1.0f, 0.0f) : r;
  };
  struct Attempt {
    float x, y, r2;
    bool ok;
  };
  bool jump_resume = false;  // the jump path stopped short of the matrix: the engine goes on from here
  uint32_t resume_window[mtjump::N];
  auto attempt = [&](uint32_t w0, uint32_t w1) {
    Attempt a;
    a.x = static_cast<float>(2.0f * canonical(w0) - 1.0);
    a.y = static_cast<float>(2.0f * canonical(w1) - 1.0);
    a.r2 = a.x * a.x + a.y * a.y;
    a.ok = !(a.r2 > 1.0f || a.r2 == 0.0f);
    return a;
  };
  const int n_thr = static_cast<int>(std::max(1u, std::min(64u, std::thread::hardware_concurrency())));
  struct Joiner {  // joins on every way out of a scope: an exception (thread creation can fail) must surface
    std::vector<std::thread> th;  // as an irs_status error, not end in std::terminate on a joinable thread
    ~Joiner() {
      for (auto &t : th)
        if (t.joinable()) t.join();
    }
  };
  // Matrices of 4 M values and more (round 5): even the engine's words are generated in parallel.  MT19937 is linear
  // over GF(2), so the state after J words is g_J(F) state with g_J = t^J mod the characteristic
  // polynomial (mt_jump.hpp).  The stream is cut into blocks of 624 * 2^b words; block j's starting
  // state is a jump of j blocks from the seed state, every thread regenerates ITS blocks twice - once to
  // count the accepted attempts (the output position of an attempt is twice the number of accepted
  // attempts before it), once to emit - and nothing but two counters per block is ever stored.  The
  // number of attempts the matrix needs is estimated from the acceptance rate pi / 4 with a margin of
  // > 100 standard deviations; should it fall short all the same, the block path below continues the
  // stream where the last block ended.
  size_t jumped_values = 0;   // values written by the jump path
  if (total >= jump_threshold && n_thr >= 4) {
    const double need_attempts = static_cast<double>((total + 1) / 2) / 0.78539816339 * 1.002 + 65536.0;
    // threads of this path: a jump costs ~0.5 ms (one or two carry-less products + one Horner pass,
    // mt_jump.hpp) - as much as generating 2^18 attempts' words twice - so no more threads than that pays
    const int jt = static_cast<int>(std::max(4.0, std::min(static_cast<double>(n_thr), need_attempts / 262144.0)));
    int b = 0;  // block = 624 * 2^b words = 312 * 2^b attempts: 2 - 4 blocks per thread (each costs one jump)
    while (312.0 * static_cast<double>(uint64_t(1) << (b + 1)) * jt * 2 <= need_attempts && b < 30) b++;
    const uint64_t block_words = 624ull << b, block_attempts_j = block_words / 2;
    const uint64_t n_blocks = static_cast<uint64_t>(need_attempts / static_cast<double>(block_attempts_j)) + 1;
    for (int i = 0; (uint64_t(1) << i) <= n_blocks; i++) (void)mtjump::pow_block(b + i);  // (the shared powers, once)
    uint32_t seed_window[mtjump::N];
    {
      const Mt19937Bulk seeded(static_cast<uint32_t>(random_seed));
      std::memcpy(seed_window, seeded.window(), sizeof(seed_window));
    }
    std::vector<uint32_t> windows(static_cast<size_t>(n_blocks + 1) * mtjump::N);  // (+ 1: where the stream goes on)
    std::vector<uint64_t> accepted(n_blocks + 1, 0);
    std::atomic<uint64_t> next{0};
    constexpr size_t CH = 624 * 16;  // words per regeneration step
    auto for_blocks = [&](auto &&body) {  // (threads started as a tree: host_util.hpp)
      next.store(0);
      run_on_threads(jt, [&](int) {
        for (;;) {
          const uint64_t j = next.fetch_add(1);
          if (j >= n_blocks) return;
          body(j);
        }
      });
    };
    // pass 1: every block's starting state and its number of accepted attempts
    {
      next.store(0);
      run_on_threads(jt, [&](int) {
        for (;;) {
          const uint64_t j = next.fetch_add(1);
          if (j > n_blocks) return;
          uint32_t *w = windows.data() + static_cast<size_t>(j) * mtjump::N;
          std::memcpy(w, seed_window, sizeof(seed_window));
          if (j > 0) mtjump::apply(mtjump::pow_blocks(b, j), w);
          if (j == n_blocks) continue;
          Mt19937Bulk eng(Mt19937Bulk::FromWindow{}, w);
          uint32_t buf[CH];
          uint64_t c = 0;
          for (uint64_t done = 0; done < block_words; done += CH) {
            const size_t take = static_cast<size_t>(std::min<uint64_t>(CH, block_words - done));
            eng.fill(buf, take);
            for (size_t q = 0; q < take; q += 2) c += attempt(buf[q], buf[q + 1]).ok ? 1 : 0;
          }
          accepted[j + 1] = c;
        }
      });
    }
    for (uint64_t j = 0; j < n_blocks; j++) accepted[j + 1] += accepted[j];
    // pass 2: emit
    for_blocks([&](uint64_t j) {
      size_t pos = 2 * static_cast<size_t>(accepted[j]);
      if (pos >= total) return;
      Mt19937Bulk eng(Mt19937Bulk::FromWindow{}, windows.data() + static_cast<size_t>(j) * mtjump::N);
      uint32_t buf[CH];
      for (uint64_t done = 0; done < block_words && pos < total; done += CH) {
        const size_t take = static_cast<size_t>(std::min<uint64_t>(CH, block_words - done));
        eng.fill(buf, take);
        for (size_t q = 0; q < take && pos < total; q += 2) {
          const Attempt a = attempt(buf[q], buf[q + 1]);
          if (!a.ok) continue;
          const float mult = std::sqrt(-2 * std::log(a.r2) / a.r2);
          h[pos] = a.y * mult * sd + 0.0f;
          if (pos + 1 < total) h[pos + 1] = a.x * mult * sd + 0.0f;
          pos += 2;
        }
      }
    });
    jumped_values = std::min<size_t>(total, 2 * static_cast<size_t>(accepted[n_blocks]));
    if (jumped_values >= total) return h;
    // (the estimate fell short - never seen: the block path goes on from the state behind the last block)
    std::memcpy(seed_window, windows.data() + static_cast<size_t>(n_blocks) * mtjump::N, sizeof(seed_window));
    jump_resume = true;
    std::memcpy(resume_window, seed_window, sizeof(seed_window));
  }
  const size_t BLK = std::max<size_t>(block_attempts, 1024);
  // (sized to what the call needs, no zero fill: a 2^18-value factor used to allocate and clear
  // 128 MB per trainer)
  auto block_size = [&](size_t produced) {  // attempts of the block that starts at `produced`
    const size_t want_pairs = (total - produced + 1) / 2;
    // ~78.5 % of the attempts are accepted; a short block at the end
    return std::min(BLK, static_cast<size_t>(want_pairs * 1.3) + 4096);
  };
  Mt19937Bulk engine = jump_resume ? Mt19937Bulk(Mt19937Bulk::FromWindow{}, resume_window)
                                   : Mt19937Bulk(static_cast<uint32_t>(random_seed));
  RawVector<uint32_t> buf[2];
  const size_t first = block_size(jumped_values);
  buf[0].resize(2 * first);
  engine.fill(buf[0].data(), 2 * first);
  std::vector<size_t> cnt(n_thr + 1);
  size_t produced = jumped_values, avail = first;  // attempts whose words are in buf[cur]
  int cur = 0;
  while (produced < total) {
    // EVERY attempt in the buffer is evaluated (emission stops at `total`): a block that used
    // fewer than it holds would drop the unread words and leave the libstdc++ stream
    const size_t na = avail;
    const uint32_t *words = buf[cur].data();
    // The next block is generated beside this one's evaluation when this one cannot finish the
    // matrix (at most its 2 na values).  Its size is known only afterwards: a full block is
    // generated and the unused tail of the stream is dropped - nothing else draws from it.
    const bool ahead = produced + 2 * na < total;
    size_t ahead_attempts = 0;
    Joiner producer;
    if (ahead) {
      ahead_attempts = block_size(produced + 2 * (na * 7 / 10));  // (no smaller than the next block will ask for)
      buf[1 - cur].resize(2 * ahead_attempts);
      producer.th.emplace_back([&, ahead_attempts] { engine.fill(buf[1 - cur].data(), 2 * ahead_attempts); });
    }
    auto range = [&](int th, size_t &b, size_t &e) {
      b = na * th / n_thr;
      e = na * (th + 1) / n_thr;
    };
    auto count = [&](int th) {
      size_t b, e, c = 0;
      range(th, b, e);
      for (size_t j = b; j < e; j++) c += attempt(words[2 * j], words[2 * j + 1]).ok ? 1 : 0;
      cnt[th + 1] = c;
    };
    auto emit = [&](int th) {
      size_t b, e;
      range(th, b, e);
      size_t pos = produced + 2 * cnt[th];
      for (size_t j = b; j < e && pos < total; j++) {
        const Attempt a = attempt(words[2 * j], words[2 * j + 1]);
        if (!a.ok) continue;
        const float mult = std::sqrt(-2 * std::log(a.r2) / a.r2);
        h[pos] = a.y * mult * sd + 0.0f;
        if (pos + 1 < total) h[pos + 1] = a.x * mult * sd + 0.0f;
        pos += 2;
      }
    };
    auto run = [&](auto fn) {
      Joiner workers;
      for (int k = 1; k < n_thr; k++) workers.th.emplace_back(fn, k);
      fn(0);
    };
    cnt[0] = 0;
    run(count);
    for (int k = 0; k < n_thr; k++) cnt[k + 1] += cnt[k];
    run(emit);
    produced = std::min(total, produced + 2 * cnt[n_thr]);
    if (ahead) {
      producer.th[0].join();
      cur = 1 - cur;
      avail = ahead_attempts;
      if (produced < total && avail < block_size(produced)) {
        // (an acceptance rate below 0.7 over a whole block - never seen: the stream goes on
        // where the producer stopped)
        const size_t need = block_size(produced);
        buf[cur].resize(2 * need);  // (RawVector keeps its contents)
        engine.fill(buf[cur].data() + 2 * avail, 2 * (need - avail));
        avail = need;
      }
    } else if (produced < total) {  // (the estimate fell short: the next block, synchronously; every
      const size_t nb = block_size(produced);  // word of this one was read, so the engine is in step)
      buf[cur].resize(2 * nb);
      engine.fill(buf[cur].data(), 2 * nb);
      avail = nb;
    }
  }
  return h;
}
}  // namespace ials
}  // namespace irs
