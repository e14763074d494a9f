// Item-kNN similarity on gfx950.  Replaces KNN::KNNComputer::compute_similarity /
// compute_similarity_triple (/root/reference/cpp_source/knn/knn.hpp:43-139) and the
// compute_similarity_imple bodies of knn/similarities.hpp (:29-46, 73-87, 109-130,
// 161-184, 242-250, 326-334) behind include/irspack_amd.h.  Arithmetic is fp64
// like the reference (knn/wrapper.cpp:9).
//
// R = target * X_t is a sparse x sparse product whose rows are dense-ish (up to
// N stored entries), so each workgroup owns one (target row, column tile) pair
// and keeps the tile's fp64 accumulators plus a "stored entry" bitmap in LDS:
//   1. for every stored (u, y) of the target row (one wave per u), walk the part
//      of X_arg^T's row u that falls in the tile: acc[j] += x * y  (ds_add_f64),
//      touched[j] = 1  — every touched j is a stored entry of the product, exact
//      zeros included (knn.hpp:111-118 ranks stored entries)
//   2. similarity epilogue per touched j (similarities.hpp, the `+ 1e-6` forms),
//      written with explicit non-contracted fp64 ops
//   3. top-k of the tile by (value desc, column asc) (knn.hpp:119-129): radix
//      select on order-preserving 64-bit keys + index-ordered tie pick
// A second kernel merges the tiles of a row, keeps top_k and sorts them by
// column (knn.hpp:130).  fp64 LDS atomics make the summation order of a sum
// run-dependent; for integer-valued inputs (binary interactions) all sums are
// exact and the result is bit-reproducible.
#include <algorithm>
#include <cmath>
#include <memory>
#include <numeric>

#include "common.hpp"

namespace irs {
namespace knn {

constexpr int TILE = 16384;     // columns per workgroup (128 KB of fp64 accumulators)
constexpr int TOPK_CAP = 1024;  // largest top_k ranked in LDS next to the accumulators
constexpr int MERGE_CAP = 4096; // candidates a row merge can hold
constexpr int THREADS = 1024;

struct Params {
  // X_arg^T by rows u: (column j, value x) sorted by j
  const int64_t *xt_ptr;
  const int32_t *xt_idx;
  const double *xt_val;
  const double *norms;  // per column j
  // target rows
  const int64_t *t_ptr;
  const int32_t *t_idx;
  const double *t_val;
  const double *t_stat;       // per target row (norm / pow / count), see epilogue
  const int32_t *row_order;   // work-sorted list of target rows of this call
  int32_t n_rows;             // rows in this call
  int32_t n_tiles;
  int32_t N;
  int32_t sim_type;
  int32_t normalize;
  double shrinkage, alpha, beta;
  int32_t top_k;
  // per (row slot, tile) winners
  int32_t *cand_idx;   // [n_rows * n_tiles * top_k]
  double *cand_val;
  int32_t *cand_cnt;   // [n_rows * n_tiles]
  // final rows
  int32_t *out_idx;    // [n_rows * top_k]
  double *out_val;
  int32_t *out_cnt;    // [n_rows]
};

__device__ __forceinline__ uint64_t order_key(double s) {
  if (s != s) return 0ull;
  if (s == 0.0) s = 0.0;
  uint64_t u = static_cast<uint64_t>(__double_as_longlong(s));
  return (u & 0x8000000000000000ull) ? ~u : (u | 0x8000000000000000ull);
}

// similarities.hpp epilogues; operations are kept un-fused so that the value is
// the one a default x86-64 build of the reference computes.
__device__ __forceinline__ double epilogue(const Params &p, double v, double norm_j,
                                           double tstat) {
  const double eps = 1e-6;
  switch (p.sim_type) {
    case IRS_SIM_COSINE:
      if (!p.normalize) return v;
      return __ddiv_rn(v, __dadd_rn(__dadd_rn(__dmul_rn(norm_j, tstat), p.shrinkage), eps));
    case IRS_SIM_ASYMMETRIC:
      return __ddiv_rn(v, __dadd_rn(__dadd_rn(__dmul_rn(norm_j, tstat), p.shrinkage), eps));
    case IRS_SIM_JACCARD:
      // norms(j) + target_norm - v + shrinkage + 1e-6   (left to right)
      return __ddiv_rn(
          v, __dadd_rn(__dadd_rn(__dsub_rn(__dadd_rn(norm_j, tstat), v), p.shrinkage), eps));
    case IRS_SIM_TVERSKY: {
      // itv + beta * (norms(j) - itv) + alpha * (target_norm - itv) + shrinkage + 1e-6
      const double a = __dadd_rn(v, __dmul_rn(p.beta, __dsub_rn(norm_j, v)));
      const double b = __dadd_rn(a, __dmul_rn(p.alpha, __dsub_rn(tstat, v)));
      return __ddiv_rn(v, __dadd_rn(__dadd_rn(b, p.shrinkage), eps));
    }
    default:
      return v;
  }
}

__global__ __launch_bounds__(THREADS) void knn_tile_kernel(Params p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  double *acc = reinterpret_cast<double *>(smem);                        // TILE
  uint32_t *bits = reinterpret_cast<uint32_t *>(acc + TILE);             // TILE / 32
  uint64_t *sel_key = reinterpret_cast<uint64_t *>(bits + TILE / 32);    // TOPK_CAP
  int32_t *sel_idx = reinterpret_cast<int32_t *>(sel_key + TOPK_CAP);    // TOPK_CAP
  uint32_t *hist = reinterpret_cast<uint32_t *>(sel_idx + TOPK_CAP);     // 256
  int32_t *wave_cnt = reinterpret_cast<int32_t *>(hist + 256);           // 16
  __shared__ uint64_t sh_prefix;
  __shared__ int32_t sh_need, sh_count, sh_tie_base, sh_total;

  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  constexpr int NW = THREADS / 64;
  const int slot = blockIdx.x / p.n_tiles, tile = blockIdx.x % p.n_tiles;
  const int r = p.row_order[slot];
  const int c0 = tile * TILE, c1 = min(c0 + TILE, p.N);
  const int width = c1 - c0;

  for (int i = tid; i < width; i += THREADS) acc[i] = 0.0;
  for (int i = tid; i < (width + 31) / 32; i += THREADS) bits[i] = 0u;
  __syncthreads();

  // ---- 1. accumulate: one wave per stored (u, y) of the target row
  const int64_t tb = p.t_ptr[r], te = p.t_ptr[r + 1];
  for (int64_t q = tb + wv; q < te; q += NW) {
    const int32_t u = p.t_idx[q];
    const double y = p.t_val[q];
    int64_t lo = p.xt_ptr[u], hi = p.xt_ptr[u + 1];
    const int64_t row_end = hi;
    if (p.n_tiles > 1) {  // lower_bound of c0 in the sorted column list
      while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (p.xt_idx[mid] < c0) lo = mid + 1; else hi = mid;
      }
    }
    for (int64_t e = lo + lane; e < row_end; e += 64) {
      const int32_t j = p.xt_idx[e];
      if (j >= c1) break;
      const int jj = j - c0;
      atomicAdd(&acc[jj], __dmul_rn(p.xt_val[e], y));
      atomicOr(&bits[jj >> 5], 1u << (jj & 31));
    }
  }
  __syncthreads();

  // ---- 2. epilogue on the stored entries; count them
  const double tstat = p.t_stat[r];
  int local = 0;
  for (int i = tid; i < width; i += THREADS) {
    if (bits[i >> 5] & (1u << (i & 31))) {
      acc[i] = epilogue(p, acc[i], p.norms[c0 + i], tstat);
      local++;
    }
  }
  local += __shfl_xor(local, 32, 64);
  local += __shfl_xor(local, 16, 64);
  local += __shfl_xor(local, 8, 64);
  local += __shfl_xor(local, 4, 64);
  local += __shfl_xor(local, 2, 64);
  local += __shfl_xor(local, 1, 64);
  if (lane == 0) wave_cnt[wv] = local;
  __syncthreads();
  if (tid == 0) {
    int s = 0;
    for (int w = 0; w < NW; w++) s += wave_cnt[w];
    sh_total = s;
  }
  __syncthreads();
  const int n_stored = sh_total;
  const int n_sel = min(p.top_k, n_stored);
  int32_t *cidx = p.cand_idx + static_cast<size_t>(blockIdx.x) * p.top_k;
  double *cval = p.cand_val + static_cast<size_t>(blockIdx.x) * p.top_k;
  if (tid == 0) p.cand_cnt[blockIdx.x] = n_sel;
  if (n_sel == 0) return;

  auto stored = [&](int i) { return (bits[i >> 5] >> (i & 31)) & 1u; };

  // ---- 3. radix select of the n_sel-th largest key (value desc, column asc on ties)
  uint64_t prefix = 0;
  int need = n_sel;
  if (n_sel < n_stored) {
    for (int shift = 56; shift >= 0; shift -= 8) {
      if (tid < 256) hist[tid] = 0;
      __syncthreads();
      const uint64_t hi_mask = (shift + 8 >= 64) ? 0ull : (~0ull << (shift + 8));
      for (int i = tid; i < width; i += THREADS) {
        if (!stored(i)) continue;
        const uint64_t k = order_key(acc[i]);
        if ((k & hi_mask) == (prefix & hi_mask)) atomicAdd(&hist[(k >> shift) & 0xff], 1u);
      }
      __syncthreads();
      if (tid == 0) {
        int a = 0, d = 255;
        for (; d >= 0; d--) {
          if (a + static_cast<int>(hist[d]) >= need) break;
          a += hist[d];
        }
        sh_prefix = prefix | (static_cast<uint64_t>(d) << shift);
        sh_need = need - a;
      }
      __syncthreads();
      prefix = sh_prefix;
      need = sh_need;
      __syncthreads();
    }
  } else {
    prefix = 0;  // everything is taken: treat every key as "greater than threshold"
    need = 0;
  }
  const bool take_all = n_sel >= n_stored;
  if (tid == 0) {
    sh_count = 0;
    sh_tie_base = 0;
  }
  __syncthreads();
  for (int base = 0; base < width; base += THREADS) {
    const int i = base + tid;
    bool tie = false;
    if (i < width && stored(i)) {
      const uint64_t k = order_key(acc[i]);
      if (take_all || k > prefix) {
        const int pos = atomicAdd(&sh_count, 1);
        cidx[pos] = c0 + i;
        cval[pos] = acc[i];
      } else if (k == prefix) {
        tie = true;
      }
    }
    const unsigned long long bal = __ballot(tie);
    if (lane == 0) wave_cnt[wv] = __popcll(bal);
    __syncthreads();
    int before_me = sh_tie_base, total = 0;
    for (int w = 0; w < NW; w++) {
      if (w < wv) before_me += wave_cnt[w];
      total += wave_cnt[w];
    }
    const int rank = before_me + __popcll(bal & ((1ull << lane) - 1ull));
    if (tie && rank < need) {
      const int pos = atomicAdd(&sh_count, 1);
      cidx[pos] = c0 + i;
      cval[pos] = acc[i];
    }
    __syncthreads();
    if (tid == 0) sh_tie_base += total;
  }
}

// One 256-thread workgroup per target row: union of the tile winners, keep the
// top_k by (value desc, column asc), emit them sorted by column (knn.hpp:119-136).
__global__ __launch_bounds__(256) void knn_merge_kernel(Params p) {
  __shared__ uint64_t key[MERGE_CAP];
  __shared__ double val[MERGE_CAP];
  __shared__ int32_t idx[MERGE_CAP];
  const int tid = threadIdx.x;
  const int slot = blockIdx.x;
  int n = 0;
  for (int t = 0; t < p.n_tiles; t++) {
    const int b = slot * p.n_tiles + t;
    const int c = p.cand_cnt[b];
    for (int i = tid; i < c; i += 256) {
      const double v = p.cand_val[static_cast<size_t>(b) * p.top_k + i];
      val[n + i] = v;
      key[n + i] = order_key(v);
      idx[n + i] = p.cand_idx[static_cast<size_t>(b) * p.top_k + i];
    }
    n += c;
  }
  int n_pow = 1;
  while (n_pow < n) n_pow <<= 1;
  for (int i = n + tid; i < n_pow; i += 256) {
    key[i] = 0ull;
    val[i] = 0.0;
    idx[i] = 0x7fffffff;
  }
  __syncthreads();
  auto swap_el = [&](int a, int b) {
    const uint64_t tk = key[a]; key[a] = key[b]; key[b] = tk;
    const double tv = val[a]; val[a] = val[b]; val[b] = tv;
    const int32_t ti = idx[a]; idx[a] = idx[b]; idx[b] = ti;
  };
  // pass 1: (value desc, column asc); padding (column = INT_MAX) sorts last among equal keys
  for (int k2 = 2; k2 <= n_pow; k2 <<= 1)
    for (int j2 = k2 >> 1; j2 > 0; j2 >>= 1) {
      for (int i = tid; i < n_pow; i += 256) {
        const int l = i ^ j2;
        if (l > i) {
          const bool up = (i & k2) == 0;
          const bool l_first = (key[l] != key[i]) ? key[l] > key[i] : idx[l] < idx[i];
          const bool i_first = (key[l] != key[i]) ? key[i] > key[l] : idx[i] < idx[l];
          if (up ? l_first : i_first) swap_el(i, l);
        }
      }
      __syncthreads();
    }
  // real entries with key 0 (NaN / padding ties) cannot outrank padding by key; they do by column
  const int keep = min(n, p.top_k);
  int k_pow = 1;
  while (k_pow < keep) k_pow <<= 1;
  for (int i = keep + tid; i < k_pow; i += 256) idx[i] = 0x7fffffff;
  __syncthreads();
  // pass 2: the kept entries by column
  for (int k2 = 2; k2 <= k_pow; k2 <<= 1)
    for (int j2 = k2 >> 1; j2 > 0; j2 >>= 1) {
      for (int i = tid; i < k_pow; i += 256) {
        const int l = i ^ j2;
        if (l > i) {
          const bool up = (i & k2) == 0;
          if (up ? idx[l] < idx[i] : idx[i] < idx[l]) swap_el(i, l);
        }
      }
      __syncthreads();
    }
  for (int i = tid; i < keep; i += 256) {
    p.out_idx[static_cast<size_t>(slot) * p.top_k + i] = idx[i];
    p.out_val[static_cast<size_t>(slot) * p.top_k + i] = val[i];
  }
  if (tid == 0) p.out_cnt[slot] = keep;
}

struct HostCsrD {
  int64_t rows = 0, cols = 0;
  std::vector<int64_t> indptr;
  std::vector<int32_t> indices;
  std::vector<double> data;
};

static HostCsrD host_csr(int64_t rows, int64_t cols, const int64_t *indptr,
                         const int32_t *indices, const double *data) {
  check_arg(rows >= 0 && cols >= 0 && indptr, "bad matrix.");
  HostCsrD m;
  m.rows = rows;
  m.cols = cols;
  m.indptr.assign(indptr, indptr + rows + 1);
  const int64_t nnz = indptr[rows];
  check_arg(indptr[0] == 0 && nnz >= 0, "malformed indptr.");
  m.indices.assign(indices, indices + nnz);
  m.data.assign(data, data + nnz);
  for (int64_t q = 0; q < nnz; q++)
    check_arg(m.indices[q] >= 0 && m.indices[q] < cols, "column index out of range.");
  return m;
}

static HostCsrD transpose(const HostCsrD &x) {
  HostCsrD t;
  t.rows = x.cols;
  t.cols = x.rows;
  t.indptr.assign(t.rows + 1, 0);
  const int64_t nnz = x.indptr[x.rows];
  t.indices.resize(nnz);
  t.data.resize(nnz);
  for (int64_t q = 0; q < nnz; q++) t.indptr[x.indices[q] + 1]++;
  for (int64_t c = 0; c < t.rows; c++) t.indptr[c + 1] += t.indptr[c];
  std::vector<int64_t> cur(t.indptr.begin(), t.indptr.end() - 1);
  for (int64_t r = 0; r < x.rows; r++)
    for (int64_t q = x.indptr[r]; q < x.indptr[r + 1]; q++) {
      const int64_t d = cur[x.indices[q]]++;
      t.indices[d] = static_cast<int32_t>(r);
      t.data[d] = x.data[q];
    }
  return t;
}

static void check_lower(double x, double low, const char *name) {  // argcheck.hpp:13-20
  if (x < low) {
    std::string msg = std::string(name) + " must be greater than or equal to  " + std::to_string(low);
    throw std::invalid_argument(msg);
  }
}

}  // namespace knn
}  // namespace irs

using namespace irs;
using namespace irs::knn;

struct irs_knn_computer {
  int device = 0;
  int32_t sim_type = 0;
  int64_t N = 0, n_features = 0;
  double shrinkage = 0, alpha = 0, beta = 0;
  bool normalize = false;
  std::vector<int64_t> xt_row_len;  // host: stored entries per feature row (work model)
  DeviceBuffer<int64_t> xt_ptr;
  DeviceBuffer<int32_t> xt_idx;
  DeviceBuffer<double> xt_val, norms;
  // last result (host)
  std::vector<int64_t> res_ptr;
  std::vector<int32_t> res_idx;
  std::vector<double> res_val;
  double last_ms = 0;
  int64_t last_macs = 0;
};

extern "C" {

irs_status irs_knn_create(int32_t sim_type, int64_t rows, int64_t cols, const int64_t *indptr,
                          const int32_t *indices, const double *data, double shrinkage,
                          double alpha, double beta, int32_t normalize, int64_t n_threads,
                          int64_t max_chunk_size, int32_t device, irs_knn_computer **out) {
  return guard([&] {
    check_arg(out != nullptr, "null argument.");
    // KNNComputer ctor, knn.hpp:35-38
    check_lower(shrinkage, 0, "shrinkage");
    check_arg(n_threads >= 1, "n_threads must be greater than or equal to  1");
    check_arg(max_chunk_size >= 1, "max_chunk_size must be greater than or equal to  1");
    HostCsrD X = host_csr(rows, cols, indptr, indices, data);
    check_arg(rows < (int64_t(1) << 31), "too many rows.");
    std::vector<double> norms(rows, 0.0);
    switch (sim_type) {
      case IRS_SIM_COSINE:  // similarities.hpp:20-28
        for (int64_t i = 0; i < rows; i++) {
          double s = 0;
          for (int64_t q = X.indptr[i]; q < X.indptr[i + 1]; q++) s += X.data[q] * X.data[q];
          norms[i] = std::sqrt(s);
        }
        break;
      case IRS_SIM_ASYMMETRIC:  // similarities.hpp:61-72
        check_lower(alpha, 0, "alpha");
        if (alpha > 1)
          throw std::invalid_argument("alpha must be less than or equal to  " + std::to_string(1.0));
        for (int64_t i = 0; i < rows; i++) {
          double s = 0;
          for (int64_t q = X.indptr[i]; q < X.indptr[i + 1]; q++) s += X.data[q] * X.data[q];
          norms[i] = std::pow(s, 1 - alpha);
        }
        break;
      case IRS_SIM_TVERSKY:  // similarities.hpp:143-159
        check_lower(alpha, 0, "alpha");
        check_lower(beta, 0, "beta");
        [[fallthrough]];
      case IRS_SIM_JACCARD:  // similarities.hpp:96-107: stored entries become 1
        for (auto &v : X.data) v = 1;
        for (int64_t i = 0; i < rows; i++)
          norms[i] = static_cast<double>(X.indptr[i + 1] - X.indptr[i]);
        break;
      case IRS_SIM_RP3BETA:  // similarities.hpp:265-292
        check_lower(alpha, 0, "alpha");
        check_lower(beta, 0, "beta");
        [[fallthrough]];
      case IRS_SIM_P3ALPHA:  // similarities.hpp:198-222: rows pow-ed and normalised to sum 1
        check_lower(alpha, 0, "alpha");
        for (int64_t i = 0; i < rows; i++) {
          double s = 0;
          for (int64_t q = X.indptr[i]; q < X.indptr[i + 1]; q++) {
            X.data[q] = std::pow(X.data[q], alpha);
            s += X.data[q];
          }
          for (int64_t q = X.indptr[i]; q < X.indptr[i + 1]; q++) X.data[q] /= s;
        }
        break;
      default:
        throw std::invalid_argument("unknown similarity type.");
    }
    require_device(device);
    auto c = std::make_unique<irs_knn_computer>();
    c->device = device;
    c->sim_type = sim_type;
    c->N = rows;
    c->n_features = cols;
    c->shrinkage = shrinkage;
    c->alpha = alpha;
    c->beta = beta;
    c->normalize = normalize != 0;
    HostCsrD Xt = transpose(X);
    c->xt_row_len.resize(Xt.rows);
    for (int64_t u = 0; u < Xt.rows; u++) c->xt_row_len[u] = Xt.indptr[u + 1] - Xt.indptr[u];
    hipStream_t s = nullptr;
    c->xt_ptr.upload(Xt.indptr, s);
    c->xt_idx.upload(Xt.indices, s);
    c->xt_val.upload(Xt.data, s);
    c->norms.upload(norms, s);
    IRS_HIP(hipStreamSynchronize(s));
    *out = c.release();
  });
}

irs_status irs_knn_destroy(irs_knn_computer *c) {
  return guard([&] {
    if (c) {
      (void)hipSetDevice(c->device);
      delete c;
    }
  });
}

irs_status irs_knn_compute(irs_knn_computer *c, int64_t rows, int64_t cols,
                           const int64_t *indptr, const int32_t *indices, const double *data,
                           int64_t top_k, int32_t as_w, int64_t row_begin, int64_t row_end,
                           int64_t *nnz_out) {
  return guard([&] {
    check_arg(c && nnz_out, "null argument.");
    if (cols != c->n_features) throw std::invalid_argument("illegal # of feature.");  // knn.hpp:44-45
    check_arg(top_k >= 0, "top_k must be non-negative.");
    HostCsrD T = host_csr(rows, cols, indptr, indices, data);
    check_arg(0 <= row_begin && row_begin <= row_end && row_end <= rows, "row range out of bounds.");
    const int64_t n = row_end - row_begin;
    // --- target preparation (host; mirrors the prologues of compute_similarity_imple / compute_W)
    if (as_w) {  // similarities.hpp:224-240, 294-324
      std::vector<double> norm_temp(cols, 0.0), pop(rows, 0.0);
      if (c->sim_type == IRS_SIM_RP3BETA) {
        for (int64_t i = 0; i < rows; i++)
          for (int64_t q = T.indptr[i]; q < T.indptr[i + 1]; q++) pop[i] += T.data[q];
        for (auto &v : pop) v = std::pow(v, c->beta);
      }
      for (int64_t q = 0; q < T.indptr[rows]; q++) {
        T.data[q] = std::pow(T.data[q], c->alpha);
        norm_temp[T.indices[q]] += T.data[q];
      }
      for (int64_t i = 0; i < rows; i++)
        for (int64_t q = T.indptr[i]; q < T.indptr[i + 1]; q++) {
          if (c->sim_type == IRS_SIM_RP3BETA)
            T.data[q] /= (norm_temp[T.indices[q]] * pop[i]);
          else
            T.data[q] /= norm_temp[T.indices[q]];
        }
    }
    const bool binarise = c->sim_type == IRS_SIM_JACCARD || c->sim_type == IRS_SIM_TVERSKY;
    std::vector<double> tstat(rows, 0.0);
    for (int64_t i = 0; i < rows; i++) {
      double s = 0;
      for (int64_t q = T.indptr[i]; q < T.indptr[i + 1]; q++) s += T.data[q] * T.data[q];
      switch (c->sim_type) {
        case IRS_SIM_COSINE: tstat[i] = std::sqrt(s); break;                  // :39
        case IRS_SIM_ASYMMETRIC: tstat[i] = std::pow(s, c->alpha); break;      // :78-79
        case IRS_SIM_JACCARD:
        case IRS_SIM_TVERSKY:
          tstat[i] = static_cast<double>(T.indptr[i + 1] - T.indptr[i]);      // :122, :174
          break;
        default: break;
      }
    }
    if (binarise)
      for (auto &v : T.data) v = 1;  // similarities.hpp:113-118, 165-170
    const int64_t out_k = std::min<int64_t>(top_k, c->N);
    c->res_ptr.assign(n + 1, 0);
    c->res_idx.clear();
    c->res_val.clear();
    c->last_ms = 0;
    c->last_macs = 0;
    *nnz_out = 0;
    if (n == 0 || out_k == 0 || c->N == 0) return;
    const int n_tiles = static_cast<int>(ceil_div(c->N, TILE));
    if (out_k > TOPK_CAP)
      throw std::invalid_argument("irspack_amd: top_k above " + std::to_string(TOPK_CAP) +
                                  " is not supported by the device kNN kernel.");
    if (static_cast<int64_t>(n_tiles) * out_k > MERGE_CAP)
      throw std::invalid_argument("irspack_amd: top_k * column tiles exceeds the merge capacity.");
    IRS_HIP(hipSetDevice(c->device));
    hipStream_t s = nullptr;
    // work per target row = multiply-adds of its product row; longest first
    std::vector<int64_t> work(n, 0);
    for (int64_t i = 0; i < n; i++) {
      int64_t w = 0;
      for (int64_t q = T.indptr[row_begin + i]; q < T.indptr[row_begin + i + 1]; q++)
        w += c->xt_row_len[T.indices[q]];
      work[i] = w;
      c->last_macs += w;
    }
    std::vector<int32_t> order(n);
    for (int64_t i = 0; i < n; i++) order[i] = static_cast<int32_t>(row_begin + i);
    std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) {
      return work[a - row_begin] > work[b - row_begin];
    });
    DeviceBuffer<int64_t> t_ptr;
    DeviceBuffer<int32_t> t_idx, d_order, cand_idx, cand_cnt, out_idx, out_cnt;
    DeviceBuffer<double> t_val, t_stat, cand_val, out_val;
    t_ptr.upload(T.indptr, s);
    t_idx.upload(T.indices, s);
    t_val.upload(T.data, s);
    t_stat.upload(tstat, s);
    d_order.upload(order, s);
    const size_t slots = static_cast<size_t>(n) * n_tiles;
    cand_idx.alloc(slots * out_k);
    cand_val.alloc(slots * out_k);
    cand_cnt.alloc(slots);
    out_idx.alloc(static_cast<size_t>(n) * out_k);
    out_val.alloc(static_cast<size_t>(n) * out_k);
    out_cnt.alloc(n);
    Params p;
    p.xt_ptr = c->xt_ptr.ptr;
    p.xt_idx = c->xt_idx.ptr;
    p.xt_val = c->xt_val.ptr;
    p.norms = c->norms.ptr;
    p.t_ptr = t_ptr.ptr;
    p.t_idx = t_idx.ptr;
    p.t_val = t_val.ptr;
    p.t_stat = t_stat.ptr;
    p.row_order = d_order.ptr;
    p.n_rows = static_cast<int32_t>(n);
    p.n_tiles = n_tiles;
    p.N = static_cast<int32_t>(c->N);
    p.sim_type = c->sim_type;
    p.normalize = c->normalize ? 1 : 0;
    p.shrinkage = c->shrinkage;
    p.alpha = c->alpha;
    p.beta = c->beta;
    p.top_k = static_cast<int32_t>(out_k);
    p.cand_idx = cand_idx.ptr;
    p.cand_val = cand_val.ptr;
    p.cand_cnt = cand_cnt.ptr;
    p.out_idx = out_idx.ptr;
    p.out_val = out_val.ptr;
    p.out_cnt = out_cnt.ptr;
    const size_t lds = TILE * sizeof(double) + (TILE / 32) * sizeof(uint32_t) +
                       TOPK_CAP * (sizeof(uint64_t) + sizeof(int32_t)) + 256 * sizeof(uint32_t) +
                       16 * sizeof(int32_t);
    IRS_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(knn_tile_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
    hipEvent_t ev0, ev1;
    IRS_HIP(hipEventCreate(&ev0));
    IRS_HIP(hipEventCreate(&ev1));
    IRS_HIP(hipEventRecord(ev0, s));
    hipLaunchKernelGGL(knn_tile_kernel, dim3(static_cast<unsigned>(slots)), dim3(THREADS), lds, s, p);
    hipLaunchKernelGGL(knn_merge_kernel, dim3(static_cast<unsigned>(n)), dim3(256), 0, s, p);
    IRS_HIP(hipEventRecord(ev1, s));
    IRS_HIP(hipGetLastError());
    std::vector<int32_t> h_cnt(n), h_idx(static_cast<size_t>(n) * out_k);
    std::vector<double> h_val(static_cast<size_t>(n) * out_k);
    IRS_HIP(hipMemcpyAsync(h_cnt.data(), out_cnt.ptr, n * sizeof(int32_t), hipMemcpyDeviceToHost, s));
    IRS_HIP(hipMemcpyAsync(h_idx.data(), out_idx.ptr, h_idx.size() * sizeof(int32_t),
                           hipMemcpyDeviceToHost, s));
    IRS_HIP(hipMemcpyAsync(h_val.data(), out_val.ptr, h_val.size() * sizeof(double),
                           hipMemcpyDeviceToHost, s));
    IRS_HIP(hipStreamSynchronize(s));
    float ms = 0;
    IRS_HIP(hipEventElapsedTime(&ms, ev0, ev1));
    (void)hipEventDestroy(ev0);
    (void)hipEventDestroy(ev1);
    c->last_ms = ms;
    // assemble the CSR in target-row order (slots are work-ordered)
    std::vector<int32_t> slot_of(n);
    for (int64_t sl = 0; sl < n; sl++) slot_of[order[sl] - row_begin] = static_cast<int32_t>(sl);
    for (int64_t i = 0; i < n; i++) c->res_ptr[i + 1] = c->res_ptr[i] + h_cnt[slot_of[i]];
    c->res_idx.resize(c->res_ptr[n]);
    c->res_val.resize(c->res_ptr[n]);
    for (int64_t i = 0; i < n; i++) {
      const size_t src = static_cast<size_t>(slot_of[i]) * out_k;
      std::copy(h_idx.begin() + src, h_idx.begin() + src + h_cnt[slot_of[i]],
                c->res_idx.begin() + c->res_ptr[i]);
      std::copy(h_val.begin() + src, h_val.begin() + src + h_cnt[slot_of[i]],
                c->res_val.begin() + c->res_ptr[i]);
    }
    *nnz_out = c->res_ptr[n];
  });
}

irs_status irs_knn_fetch(irs_knn_computer *c, int64_t *indptr, int32_t *indices, double *data) {
  return guard([&] {
    check_arg(c && indptr, "null argument.");
    std::copy(c->res_ptr.begin(), c->res_ptr.end(), indptr);
    if (!c->res_idx.empty()) {
      check_arg(indices && data, "null argument.");
      std::copy(c->res_idx.begin(), c->res_idx.end(), indices);
      std::copy(c->res_val.begin(), c->res_val.end(), data);
    }
  });
}

irs_status irs_knn_last_stats(irs_knn_computer *c, double *kernel_ms, int64_t *macs) {
  return guard([&] {
    check_arg(c && kernel_ms && macs, "null argument.");
    *kernel_ms = c->last_ms;
    *macs = c->last_macs;
  });
}

// remove_diagonal, util.hpp:211-226: stored diagonal entries become explicit zeros
irs_status irs_remove_diagonal(int64_t rows, int64_t cols, const int64_t *indptr,
                               const int32_t *indices, double *data) {
  return guard([&] {
    check_arg(rows == cols, "X must be square");
    check_arg(indptr && (indptr[rows] == 0 || (indices && data)), "null argument.");
    for (int64_t i = 0; i < rows; i++)
      for (int64_t q = indptr[i]; q < indptr[i + 1]; q++)
        if (indices[q] == i) data[q] = 0.0;
  });
}

}  // extern "C"
