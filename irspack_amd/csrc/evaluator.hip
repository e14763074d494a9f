// Ranking-metric evaluator on gfx950.  Replaces EvaluatorCore::get_metrics /
// get_metrics_local / Metrics::update of /root/reference/cpp_source/evaluator.cpp
// (:256-284, :292-367, :127-166) behind include/irspack_amd.h.
//
// One 256-thread workgroup ranks one user row:
//   1. candidates = all items | global list | per-user list, score != -inf (:324-348)
//   2. top-`cutoff` by the total order (score desc, index asc) — the order
//      std::partial_sort gives on (-score, index) pairs (:329, :353-355):
//      MSB-first 8-bit radix select of the cutoff-th key, then an index-ordered
//      pick of the ties at the threshold, then a bitonic sort of the winners in LDS
//   3. hits by binary search in the (recommendable-filtered) ground-truth row,
//      then the sequential dcg / AP recurrences of Metrics::update
// Per-user terms are summed in user order (double), item counts with int64
// atomics; both are order-independent up to fp64 rounding / exactly.
#include <algorithm>
#include <cmath>
#include <limits>
#include <memory>

#include "common.hpp"

namespace irs {
namespace eval {

constexpr int SEL_CAP = 2048;  // largest cutoff ranked in LDS (16 KB keys + 8 KB indices)

struct RowOut {
  double hit, recall, ndcg, precision, map;
  int32_t valid;   // 1 when the user has ground truth
  int32_t n_rec;
};

// order-preserving integer keys; larger key = better score.  NaN ranks last,
// -0.0 == +0.0 (the reference compares floats, where they tie).
__device__ __forceinline__ uint64_t order_key(float s) {
  if (s != s) return 0ull;
  if (s == 0.f) s = 0.f;
  uint32_t u = __float_as_uint(s);
  u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
  return static_cast<uint64_t>(u);
}
__device__ __forceinline__ uint64_t order_key(double s) {
  if (s != s) return 0ull;
  if (s == 0.0) s = 0.0;
  uint64_t u = static_cast<uint64_t>(__double_as_longlong(s));
  return (u & 0x8000000000000000ull) ? ~u : (u | 0x8000000000000000ull);
}
template <class T> struct KeyBits;
template <> struct KeyBits<float> { static constexpr int value = 32; };
template <> struct KeyBits<double> { static constexpr int value = 64; };
template <class T> __device__ __forceinline__ bool is_neg_inf(T s) {
  return s == -std::numeric_limits<T>::infinity();
}

struct EvalParams {
  const void *scores;        // [rows, n_items] row-major
  int64_t rows, n_items;
  int64_t offset;            // ground-truth row of scores row 0
  const int32_t *gt_ptr;     // filtered ground truth CSR (relevant ∩ recommendable)
  const int32_t *gt_idx;
  int32_t rec_mode;          // 0 all items, 1 global list, 2 per-user lists
  const int64_t *rec_ptr;
  const int32_t *rec_items;
  int32_t cutoff;
  int32_t recall_with_cutoff;
  const double *disc;        // 1 / log2(2 + i)
  const double *idcg_prefix; // sequential prefix sums of disc
  RowOut *out;
  int32_t *rec_out;          // [rows, cutoff] recommended items (-1 padded)
  unsigned long long *item_cnt;
};

template <class T>
__global__ __launch_bounds__(256) void rank_rows_kernel(EvalParams p) {
  __shared__ uint32_t hist[256];
  __shared__ uint64_t sel_key[SEL_CAP];
  __shared__ int32_t sel_idx[SEL_CAP];
  __shared__ uint8_t sel_hit[SEL_CAP];
  __shared__ int32_t scan_buf[256];
  __shared__ uint64_t sh_prefix;
  __shared__ int32_t sh_need, sh_count, sh_tie_base;

  const int tid = threadIdx.x;
  const int64_t row = blockIdx.x;
  const int64_t u = row + p.offset;
  const T *srow = static_cast<const T *>(p.scores) + row * p.n_items;
  RowOut res{0, 0, 0, 0, 0, 0, 0};
  const int gb = p.gt_ptr[u], ge = p.gt_ptr[u + 1];
  const int n_gt = ge - gb;
  int32_t *rec_row = p.rec_out + row * p.cutoff;
  for (int i = tid; i < p.cutoff; i += 256) rec_row[i] = -1;
  if (n_gt == 0) {  // counted in total_user only (:316-321)
    if (tid == 0) p.out[row] = res;
    return;
  }
  // candidate list
  int64_t cb = 0, n_cand = p.n_items;
  const int32_t *list = nullptr;
  if (p.rec_mode == 1) {
    cb = p.rec_ptr[0];
    n_cand = p.rec_ptr[1] - cb;
    list = p.rec_items + cb;
  } else if (p.rec_mode == 2) {
    cb = p.rec_ptr[u];
    n_cand = p.rec_ptr[u + 1] - cb;
    list = p.rec_items + cb;
  }
  auto item_of = [&](int64_t j) -> int32_t { return list ? list[j] : static_cast<int32_t>(j); };

  // --- count rankable candidates
  int local = 0;
  for (int64_t j = tid; j < n_cand; j += 256) local += !is_neg_inf(srow[item_of(j)]);
  scan_buf[tid] = local;
  __syncthreads();
  for (int off = 128; off >= 1; off >>= 1) {
    if (tid < off) scan_buf[tid] += scan_buf[tid + off];
    __syncthreads();
  }
  const int n_rankable = scan_buf[0];
  __syncthreads();
  const int n_rec = min(p.cutoff, n_rankable);
  res.valid = 1;
  res.n_rec = n_rec;
  if (n_rec == 0) {  // :132-135
    if (tid == 0) p.out[row] = res;
    return;
  }

  // --- radix select: key of the n_rec-th best candidate
  uint64_t prefix = 0;
  int need = n_rec;  // how many still to take among keys matching `prefix` so far
  constexpr int BITS = KeyBits<T>::value;
  for (int shift = BITS - 8; shift >= 0; shift -= 8) {
    hist[tid] = 0;
    __syncthreads();
    const uint64_t hi_mask = (shift + 8 >= 64) ? 0ull : (~0ull << (shift + 8));
    for (int64_t j = tid; j < n_cand; j += 256) {
      const T s = srow[item_of(j)];
      if (is_neg_inf(s)) continue;
      const uint64_t k = order_key(s);
      if ((k & hi_mask) == (prefix & hi_mask)) atomicAdd(&hist[(k >> shift) & 0xff], 1u);
    }
    __syncthreads();
    if (tid == 0) {
      int acc = 0, d = 255;
      for (; d >= 0; d--) {
        if (acc + static_cast<int>(hist[d]) >= need) break;
        acc += hist[d];
      }
      sh_prefix = prefix | (static_cast<uint64_t>(d) << shift);
      sh_need = need - acc;
    }
    __syncthreads();
    prefix = sh_prefix;
    need = sh_need;
    __syncthreads();
  }
  const uint64_t thr = prefix;  // threshold key; `need` ties at thr are taken, lowest index first

  // --- gather winners: key > thr anywhere, key == thr in index order
  if (tid == 0) {
    sh_count = 0;
    sh_tie_base = 0;
  }
  __syncthreads();
  for (int64_t base = 0; base < n_cand; base += 256) {
    const int64_t j = base + tid;
    bool tie = false;
    uint64_t k = 0;
    int32_t it = 0;
    if (j < n_cand) {
      it = item_of(j);
      const T s = srow[it];
      if (!is_neg_inf(s)) {
        k = order_key(s);
        if (k > thr) {
          const int pos = atomicAdd(&sh_count, 1);
          sel_key[pos] = k;
          sel_idx[pos] = it;
        } else if (k == thr) {
          tie = true;
        }
      }
    }
    // rank of each tie in index order: ballot inside the wave, wave totals through LDS
    const unsigned long long bal = __ballot(tie);
    const int wv = tid >> 6, ln = tid & 63;
    if (ln == 0) scan_buf[wv] = __popcll(bal);
    __syncthreads();
    int before_me = sh_tie_base;
    for (int ww = 0; ww < wv; ww++) before_me += scan_buf[ww];
    const int total = scan_buf[0] + scan_buf[1] + scan_buf[2] + scan_buf[3];
    const int rank = before_me + __popcll(bal & ((1ull << ln) - 1ull));
    if (tie && rank < need) {
      const int pos = atomicAdd(&sh_count, 1);
      sel_key[pos] = k;
      sel_idx[pos] = it;
    }
    __syncthreads();
    if (tid == 0) sh_tie_base += total;
  }
  __syncthreads();
  // --- bitonic sort of the n_rec winners: key desc, index asc
  int n_pow = 1;
  while (n_pow < n_rec) n_pow <<= 1;
  for (int i = n_rec + tid; i < n_pow; i += 256) {
    sel_key[i] = 0ull;
    sel_idx[i] = 0x7fffffff;
  }
  __syncthreads();
  auto before = [&](int a, int b) {  // element a ranks before element b
    if (sel_key[a] != sel_key[b]) return sel_key[a] > sel_key[b];
    return sel_idx[a] < sel_idx[b];
  };
  for (int k2 = 2; k2 <= n_pow; k2 <<= 1) {
    for (int j2 = k2 >> 1; j2 > 0; j2 >>= 1) {
      for (int i = tid; i < n_pow; i += 256) {
        const int l = i ^ j2;
        if (l > i) {
          const bool up = (i & k2) == 0;
          const bool swap = up ? before(l, i) : before(i, l);
          if (swap) {
            const uint64_t tk = sel_key[i];
            sel_key[i] = sel_key[l];
            sel_key[l] = tk;
            const int32_t ti = sel_idx[i];
            sel_idx[i] = sel_idx[l];
            sel_idx[l] = ti;
          }
        }
      }
      __syncthreads();
    }
  }
  // --- hits, histogram, output list
  for (int i = tid; i < n_rec; i += 256) {
    const int32_t it = sel_idx[i];
    rec_row[i] = it;
    atomicAdd(&p.item_cnt[it], 1ull);  // :146
    int lo = gb, hi = ge;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (p.gt_idx[mid] < it) lo = mid + 1; else hi = mid;
    }
    sel_hit[i] = (lo < ge && p.gt_idx[lo] == it) ? 1 : 0;
  }
  __syncthreads();
  if (tid == 0) {  // Metrics::update :136-165, same sequential order
    double dcg = 0, ap = 0;
    int cum_hit = 0;
    for (int i = 0; i < n_rec; i++) {
      if (sel_hit[i]) {
        dcg += p.disc[i];
        cum_hit++;
        ap += static_cast<double>(cum_hit) / (i + 1);
      }
    }
    const double idcg = p.idcg_prefix[min(n_gt, n_rec)];
    res.hit = cum_hit > 0 ? 1.0 : 0.0;
    res.precision = cum_hit / static_cast<double>(n_rec);
    res.recall = cum_hit / static_cast<double>(
                               p.recall_with_cutoff ? (n_gt > n_rec ? n_rec : n_gt) : n_gt);
    res.ndcg = dcg / idcg;
    res.map = ap / n_gt;
    p.out[row] = res;
  }
}

// sum of the per-user terms in user order (one thread; rows <= a few thousand per call)
__global__ void reduce_rows_kernel(const RowOut *rows, int64_t n, irs_metrics *out) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  irs_metrics m{0, 0, 0, 0, 0, 0, 0};
  for (int64_t i = 0; i < n; i++) {
    m.total_user += 1;
    if (rows[i].valid) {
      m.valid_user += 1;
      m.hit += rows[i].hit;
      m.recall += rows[i].recall;
      m.ndcg += rows[i].ndcg;
      m.precision += rows[i].precision;
      m.map += rows[i].map;
    }
  }
  out->valid_user += m.valid_user;
  out->total_user += m.total_user;
  out->hit += m.hit;
  out->recall += m.recall;
  out->ndcg += m.ndcg;
  out->precision += m.precision;
  out->map += m.map;
}

// scores[row, col] = -inf for the stored entries of the mask rows (evaluator.py:426-432)
__global__ void mask_rows_kernel(float *scores, int64_t rows, int64_t n_items,
                                 const int64_t *mask_ptr, const int32_t *mask_idx) {
  const int64_t row = blockIdx.x;
  if (row >= rows) return;
  for (int64_t q = mask_ptr[row] + threadIdx.x; q < mask_ptr[row + 1]; q += blockDim.x)
    scores[row * n_items + mask_idx[q]] = -std::numeric_limits<float>::infinity();
}

}  // namespace eval
}  // namespace irs

using namespace irs;
using namespace irs::eval;

// exported by ials.hip for the fused path
extern "C" irs_status irs_ials_scores_device_(irs_ials_trainer *t, int64_t begin, int64_t end,
                                              float *device_out, void **stream_out,
                                              int32_t *device_index);

struct irs_evaluator {
  int device = 0;
  int64_t n_users = 0, n_items = 0;
  int rec_mode = 0;
  DeviceBuffer<int32_t> gt_ptr, gt_idx, rec_items;
  DeviceBuffer<int64_t> rec_ptr;
  DeviceBuffer<double> disc, idcg_prefix;
  DeviceBuffer<RowOut> row_out;
  DeviceBuffer<int32_t> rec_out;
  DeviceBuffer<unsigned long long> item_cnt;
  DeviceBuffer<irs_metrics> metrics;
  DeviceBuffer<char> score_buf;
};

namespace {

void validate_call(irs_evaluator *e, int64_t rows, int64_t cutoff, int64_t offset,
                   int64_t n_threads) {
  // evaluator.cpp:209, 263-268
  check_arg(n_threads > 0, "n_threads must be strictly positive.");
  check_arg(offset >= 0 && e->n_users > offset, "got offset >= n_users");
  check_arg(offset + rows <= e->n_users, "offset + scores.shape[0] exceeds n_users");
  check_arg(cutoff > 0, "cutoff must be strictly greather than 0.");
  check_arg(cutoff <= e->n_items, "cutoff must not exeeed the number of items.");
  if (cutoff > SEL_CAP)
    throw std::invalid_argument("irspack_amd: cutoff above " + std::to_string(SEL_CAP) +
                                " is not supported by the device ranking kernel.");
}

// ranks `rows` rows already resident at `d_scores` and accumulates into out / item_cnt
template <class T>
void rank_block(irs_evaluator *e, const void *d_scores, int64_t rows, int64_t cutoff,
                int64_t offset, bool rwc, hipStream_t s) {
  e->row_out.alloc(rows);
  e->rec_out.alloc(rows * cutoff);
  EvalParams p;
  p.scores = d_scores;
  p.rows = rows;
  p.n_items = e->n_items;
  p.offset = offset;
  p.gt_ptr = e->gt_ptr.ptr;
  p.gt_idx = e->gt_idx.ptr;
  p.rec_mode = e->rec_mode;
  p.rec_ptr = e->rec_ptr.ptr;
  p.rec_items = e->rec_items.ptr;
  p.cutoff = static_cast<int32_t>(cutoff);
  p.recall_with_cutoff = rwc ? 1 : 0;
  p.disc = e->disc.ptr;
  p.idcg_prefix = e->idcg_prefix.ptr;
  p.out = e->row_out.ptr;
  p.rec_out = e->rec_out.ptr;
  p.item_cnt = e->item_cnt.ptr;
  hipLaunchKernelGGL((rank_rows_kernel<T>), dim3(rows), dim3(256), 0, s, p);
  hipLaunchKernelGGL(reduce_rows_kernel, dim3(1), dim3(64), 0, s, e->row_out.ptr, rows,
                     e->metrics.ptr);
  IRS_HIP(hipGetLastError());
}

void begin_accumulate(irs_evaluator *e, hipStream_t s) {
  e->metrics.alloc(1);
  e->metrics.zero(s);
  e->item_cnt.zero(s);
}

void finish_accumulate(irs_evaluator *e, irs_metrics *out, int64_t *item_cnt, hipStream_t s) {
  static_assert(sizeof(unsigned long long) == sizeof(int64_t), "");
  IRS_HIP(hipMemcpyAsync(out, e->metrics.ptr, sizeof(irs_metrics), hipMemcpyDeviceToHost, s));
  IRS_HIP(hipMemcpyAsync(item_cnt, e->item_cnt.ptr, e->n_items * sizeof(int64_t),
                         hipMemcpyDeviceToHost, s));
  IRS_HIP(hipStreamSynchronize(s));
}

}  // namespace

extern "C" {

irs_status irs_eval_create(int64_t n_users, int64_t n_items, const int64_t *indptr,
                           const int32_t *indices, int64_t n_lists, const int64_t *rec_ptr,
                           const int64_t *rec_items, int32_t device, irs_evaluator **out) {
  return guard([&] {
    check_arg(out && indptr, "null argument.");
    check_arg(n_users >= 0 && n_items >= 0, "negative shape.");
    // evaluator.cpp:187-190
    check_arg(n_lists == 0 || n_lists == 1 || n_lists == n_users,
              "recommendable.size.() must be in {0, 1, ground_truth.size()}");
    const int64_t nnz = indptr[n_users];
    check_arg(nnz < (int64_t(1) << 31), "nnz must be below 2^31.");
    // sort + validate the recommendable lists (:193-205)
    std::vector<std::vector<int32_t>> lists(n_lists);
    for (int64_t l = 0; l < n_lists; l++) {
      auto &v = lists[l];
      for (int64_t q = rec_ptr[l]; q < rec_ptr[l + 1]; q++) {
        check_arg(rec_items[q] >= 0 && rec_items[q] < n_items,
                  "recommendable items contain a index >= n_items.");
        v.push_back(static_cast<int32_t>(rec_items[q]));
      }
      std::sort(v.begin(), v.end());
      for (size_t i = 1; i < v.size(); i++)
        check_arg(v[i] > v[i - 1], "duplicate recommendable items.");
    }
    // ground truth restricted to the recommendable set (cache_X_map, :208-254)
    std::vector<int32_t> gptr(n_users + 1, 0), gidx;
    gidx.reserve(nnz);
    for (int64_t u = 0; u < n_users; u++) {
      std::vector<int32_t> row(indices + indptr[u], indices + indptr[u + 1]);
      std::sort(row.begin(), row.end());
      row.erase(std::unique(row.begin(), row.end()), row.end());
      for (auto c : row) check_arg(c >= 0 && c < n_items, "column index out of range.");
      if (n_lists == 0) {
        gidx.insert(gidx.end(), row.begin(), row.end());
      } else {
        const auto &rec = n_lists == 1 ? lists[0] : lists[u];
        std::set_intersection(row.begin(), row.end(), rec.begin(), rec.end(),
                              std::back_inserter(gidx));
      }
      gptr[u + 1] = static_cast<int32_t>(gidx.size());
    }
    require_device(device);
    auto e = std::make_unique<irs_evaluator>();
    e->device = device;
    e->n_users = n_users;
    e->n_items = n_items;
    e->rec_mode = n_lists == 0 ? 0 : (n_lists == 1 ? 1 : 2);
    hipStream_t s = nullptr;
    e->gt_ptr.upload(gptr, s);
    e->gt_idx.upload(gidx, s);
    std::vector<int64_t> rp(n_lists + 1, 0);
    std::vector<int32_t> ri;
    for (int64_t l = 0; l < n_lists; l++) {
      ri.insert(ri.end(), lists[l].begin(), lists[l].end());
      rp[l + 1] = static_cast<int64_t>(ri.size());
    }
    e->rec_ptr.upload(rp, s);
    e->rec_items.upload(ri, s);
    // prepare_dcg_discount (:42-48) and its sequential prefix sums (std::accumulate, :137-139)
    const int64_t nd = std::min<int64_t>(n_items, SEL_CAP);
    std::vector<double> disc(nd), pre(nd + 1, 0.0);
    for (int64_t i = 0; i < nd; i++) disc[i] = 1 / std::log2(2 + i);
    for (int64_t i = 0; i < nd; i++) pre[i + 1] = pre[i] + disc[i];
    e->disc.upload(disc, s);
    e->idcg_prefix.upload(pre, s);
    e->item_cnt.alloc(std::max<int64_t>(n_items, 1));
    e->metrics.alloc(1);
    IRS_HIP(hipStreamSynchronize(s));
    *out = e.release();
  });
}

irs_status irs_eval_destroy(irs_evaluator *e) {
  return guard([&] {
    if (e) {
      (void)hipSetDevice(e->device);
      delete e;
    }
  });
}

irs_status irs_eval_get_metrics(irs_evaluator *e, int32_t is_f64, const void *scores,
                                int64_t rows, int64_t cutoff, int64_t offset,
                                int64_t n_threads, int32_t recall_with_cutoff,
                                irs_metrics *out, int64_t *item_cnt) {
  return guard([&] {
    check_arg(e && out && item_cnt, "null argument.");
    check_arg(rows >= 0, "negative row count.");
    validate_call(e, rows, cutoff, offset, n_threads);
    IRS_HIP(hipSetDevice(e->device));
    hipStream_t s = nullptr;
    begin_accumulate(e, s);
    if (rows > 0) {
      const size_t bytes = static_cast<size_t>(rows) * e->n_items * (is_f64 ? 8 : 4);
      e->score_buf.alloc(bytes);
      IRS_HIP(hipMemcpyAsync(e->score_buf.ptr, scores, bytes, hipMemcpyHostToDevice, s));
      if (is_f64)
        rank_block<double>(e, e->score_buf.ptr, rows, cutoff, offset, recall_with_cutoff != 0, s);
      else
        rank_block<float>(e, e->score_buf.ptr, rows, cutoff, offset, recall_with_cutoff != 0, s);
    }
    finish_accumulate(e, out, item_cnt, s);
  });
}

irs_status irs_eval_get_metrics_ials(irs_evaluator *e, irs_ials_trainer *t, int64_t begin,
                                     int64_t end, const int64_t *mask_indptr,
                                     const int32_t *mask_indices, int64_t cutoff,
                                     int64_t offset, int32_t recall_with_cutoff,
                                     irs_metrics *out, int64_t *item_cnt) {
  return guard([&] {
    check_arg(e && t && out && item_cnt, "null argument.");
    check_arg(end >= begin && begin >= 0, "bad user block.");
    const int64_t rows = end - begin;
    validate_call(e, rows, cutoff, offset, 1);
    IRS_HIP(hipSetDevice(e->device));
    const int64_t BLOCK = 1024;  // users scored and ranked per pass
    DeviceBuffer<float> scores;
    scores.alloc(static_cast<size_t>(std::min(BLOCK, std::max<int64_t>(rows, 1))) * e->n_items);
    DeviceBuffer<int64_t> mptr;
    DeviceBuffer<int32_t> midx;
    void *sv = nullptr;
    int32_t dev = 0;
    // a zero-row call only fetches the trainer's stream / device
    if (irs_ials_scores_device_(t, begin, begin, nullptr, &sv, &dev) != IRS_OK)
      throw std::runtime_error(irs_last_error());
    check_arg(dev == e->device, "evaluator and trainer live on different devices.");
    hipStream_t s = static_cast<hipStream_t>(sv);
    begin_accumulate(e, s);
    if (mask_indptr) {
      std::vector<int64_t> mp(mask_indptr, mask_indptr + rows + 1);
      mptr.upload(mp, s);
      midx.upload(mask_indices, static_cast<size_t>(mask_indptr[rows]), s);
      IRS_HIP(hipStreamSynchronize(s));
    }
    for (int64_t b = 0; b < rows; b += BLOCK) {
      const int64_t m = std::min(BLOCK, rows - b);
      if (irs_ials_scores_device_(t, begin + b, begin + b + m, scores.ptr, &sv, &dev) != IRS_OK)
        throw std::runtime_error(irs_last_error());
      if (mask_indptr)
        hipLaunchKernelGGL(mask_rows_kernel, dim3(m), dim3(64), 0, s, scores.ptr, m, e->n_items,
                           mptr.ptr + b, midx.ptr);
      rank_block<float>(e, scores.ptr, m, cutoff, offset + b, recall_with_cutoff != 0, s);
    }
    finish_accumulate(e, out, item_cnt, s);
  });
}

}  // extern "C"
