// ORACLE — TEST INFRASTRUCTURE ONLY. Not part of the product path.
//
// CPU restatement of /root/reference/cpp_source/evaluator.cpp:29-435
// (Metrics, EvaluatorCore, evaluate_list_vs_list).  Metrics are `double`,
// counts `int64`, scores templated float/double like the reference (:31-41).
//
// Parity pinning: see oracle/ials_oracle.cpp header; this file is pinned by
// the sklearn / hand-loop checks the reference holds in
// tests/evaluation/test_evaluator.py:19-152, 358-398 and
// tests/evaluation/test_restricted_evaluator.py:25-108, restated in
// tests/test_oracle_evaluator.py.
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <limits>
#include <numeric>
#include <stdexcept>
#include <string>
#include <thread>
#include <unordered_set>
#include <vector>

namespace {

thread_local std::string g_last_error_ev;

inline void check_arg(bool cond, const char *msg) {  // argcheck.hpp:7-11
  if (!cond) throw std::invalid_argument(msg);
}

struct Metrics {  // evaluator.cpp:49-179
  size_t valid_user = 0, total_user = 0;
  double hit = 0, recall = 0, ndcg = 0, precision = 0, map = 0;
  size_t n_item;
  std::vector<int64_t> item_cnt;
  std::vector<double> dcg_discount;

  explicit Metrics(size_t n) : n_item(n), item_cnt(n, 0), dcg_discount(n) {
    for (size_t i = 0; i < n; i++) dcg_discount[i] = 1 / std::log2(2 + i);  // :42-48
  }
  void merge(const Metrics &o) {  // :76-85
    hit += o.hit;
    recall += o.recall;
    ndcg += o.ndcg;
    total_user += o.total_user;
    valid_user += o.valid_user;
    for (size_t i = 0; i < n_item; i++) item_cnt[i] += o.item_cnt[i];
    precision += o.precision;
    map += o.map;
  }
  // :127-166
  void update(const std::vector<size_t> &rec,
              const std::unordered_set<size_t> &gt, bool recall_with_cutoff) {
    size_t n_gt = gt.size();
    size_t n_rec = rec.size();
    valid_user += 1;
    if (n_rec == 0) return;
    double dcg = 0;
    double idcg = std::accumulate(dcg_discount.begin(),
                                  dcg_discount.begin() + std::min(n_gt, n_rec), 0.);
    double ap = 0;
    size_t cum_hit = 0;
    for (size_t i = 0; i < n_rec; i++) {
      auto r = rec[i];
      item_cnt[r] += 1;
      if (gt.find(r) != gt.cend()) {
        dcg += dcg_discount[i];
        cum_hit++;
        ap += (static_cast<double>(cum_hit) / (i + 1));
      }
    }
    if (cum_hit > 0) hit += 1;
    precision += cum_hit / static_cast<double>(n_rec);
    recall += cum_hit / static_cast<double>(
                            recall_with_cutoff ? (n_gt > n_rec ? n_rec : n_gt) : n_gt);
    ndcg += (dcg / idcg);
    map += ap / n_gt;
  }
  // :87-123; key order: total_user, valid_user, n_items, hit, ndcg, recall,
  // map, precision, appeared_item, entropy, gini_index
  void as_array(double *out) const {
    std::vector<int64_t> cnt(item_cnt);
    double total_item = 0;
    for (auto c : item_cnt) total_item += c;
    std::sort(cnt.begin(), cnt.end());
    double appeared = 0, entropy = 0, gini = 0;
    const int64_t n = static_cast<int64_t>(cnt.size());
    for (int64_t i = 0; i < n; i++) {
      int64_t c = cnt[i];
      if (c == 0) continue;
      double p = c / total_item;
      appeared++;
      entropy += -std::log(p) * p;
      gini += (2 * i - n + 1) * c;
    }
    if (total_item > 0) gini /= (n * total_item);
    size_t denom = valid_user > 0u ? valid_user : 1;
    out[0] = total_user;
    out[1] = valid_user;
    out[2] = n_item;
    out[3] = hit / denom;
    out[4] = ndcg / denom;
    out[5] = recall / denom;
    out[6] = map / denom;
    out[7] = precision / denom;
    out[8] = appeared;
    out[9] = entropy;
    out[10] = gini;
  }
};

struct Core {  // evaluator.cpp:181-374
  int64_t n_users, n_items;
  std::vector<int64_t> indptr;
  std::vector<int32_t> indices;
  std::vector<std::vector<size_t>> recommendable;
  std::vector<std::unordered_set<size_t>> X_as_set;

  void cache_X_map() {  // :208-254
    if (!X_as_set.empty()) return;
    X_as_set.resize(n_users);
    for (int64_t u = 0; u < n_users; u++) {
      auto &target = X_as_set[u];
      std::vector<size_t> idx(indices.begin() + indptr[u], indices.begin() + indptr[u + 1]);
      if (recommendable.empty()) {
        for (auto i : idx) target.insert(i);
      } else {
        const auto &rec = recommendable.size() == 1 ? recommendable[0] : recommendable[u];
        std::vector<size_t> inter;
        std::set_intersection(idx.begin(), idx.end(), rec.begin(), rec.end(),
                              std::back_inserter(inter));
        for (auto i : inter) target.insert(i);
      }
    }
  }

  template <typename T>
  Metrics get_metrics(const T *scores, int64_t rows, size_t cutoff, size_t offset,
                      size_t n_threads, bool recall_with_cutoff) {
    check_arg(n_threads > 0, "n_threads must be strictly positive.");  // :209
    cache_X_map();
    Metrics overall(n_items);
    check_arg(n_threads > 0, "n_threads == 0");
    check_arg(static_cast<size_t>(n_users) > offset, "got offset >= n_users");
    check_arg(static_cast<size_t>(offset + rows) <= static_cast<size_t>(n_users),
              "offset + scores.shape[0] exceeds n_users");
    check_arg(cutoff > 0, "cutoff must be strictly greather than 0.");
    check_arg(cutoff <= static_cast<size_t>(n_items),
              "cutoff must not exeeed the number of items.");
    std::atomic<size_t> cursor(0);
    std::vector<Metrics> locals(n_threads, Metrics(n_items));
    auto work = [&](size_t tid) {  // get_metrics_local, :292-367
      Metrics &m = locals[tid];
      std::vector<std::pair<T, int32_t>> sai;
      std::vector<size_t> rec_index;
      sai.reserve(n_items);
      while (true) {
        size_t u = cursor.fetch_add(1);
        if (u >= static_cast<size_t>(rows)) break;
        const T *buffer = scores + n_items * u;
        const size_t u_orig = u + offset;
        const auto &gt = X_as_set.at(u_orig);
        m.total_user++;
        sai.clear();
        rec_index.clear();
        if (gt.empty()) continue;
        const T ninf = -std::numeric_limits<T>::infinity();
        if (recommendable.empty()) {
          for (int32_t i = 0; i < static_cast<int32_t>(n_items); i++) {
            T s = buffer[i];
            if (s != ninf) sai.emplace_back(-s, i);
          }
        } else {
          const auto &items = recommendable.size() == 1u ? recommendable[0]
                                                         : recommendable[u_orig];
          for (auto i : items) {
            T s = buffer[i];
            if (s != ninf) sai.emplace_back(-s, static_cast<int32_t>(i));
          }
        }
        size_t n_rec = std::min(cutoff, sai.size());
        std::partial_sort(sai.begin(), sai.begin() + n_rec, sai.end());  // :353-355
        for (size_t i = 0; i < n_rec; i++) rec_index.push_back(sai[i].second);
        m.update(rec_index, gt, recall_with_cutoff);
      }
    };
    std::vector<std::thread> th;
    for (size_t t = 1; t < n_threads; t++) th.emplace_back(work, t);
    work(0);
    for (auto &t : th) t.join();
    for (auto &l : locals) overall.merge(l);
    return overall;
  }
};

template <class F> int guard(F &&f) {
  try {
    f();
    return 0;
  } catch (const std::invalid_argument &e) {
    g_last_error_ev = e.what();
    return 1;
  } catch (const std::exception &e) {
    g_last_error_ev = e.what();
    return 2;
  }
}

}  // namespace

extern "C" {

const char *orc_eval_last_error() { return g_last_error_ev.c_str(); }

// EvaluatorCore ctor, evaluator.cpp:183-206.  Recommendable lists arrive as a
// ragged array (rec_ptr has n_lists + 1 entries).
int orc_eval_create(int64_t n_users, int64_t n_items, const int64_t *indptr,
                    const int32_t *indices, int64_t n_lists,
                    const int64_t *rec_ptr, const int64_t *rec_items, void **out) {
  return guard([&] {
    check_arg(n_lists == 0 || n_lists == 1 || n_lists == n_users,
              "recommendable.size.() must be in {0, 1, ground_truth.size()}");
    Core *c = new Core;
    c->n_users = n_users;
    c->n_items = n_items;
    c->indptr.assign(indptr, indptr + n_users + 1);
    c->indices.assign(indices, indices + indptr[n_users]);
    c->recommendable.resize(n_lists);
    try {
      for (int64_t l = 0; l < n_lists; l++) {
        auto &urec = c->recommendable[l];
        for (int64_t p = rec_ptr[l]; p < rec_ptr[l + 1]; p++) {
          check_arg(rec_items[p] >= 0, "recommendable items contain a index >= n_items.");
          urec.push_back(static_cast<size_t>(rec_items[p]));
        }
        std::sort(urec.begin(), urec.end());
        if (!urec.empty()) {
          check_arg(urec.back() < static_cast<size_t>(n_items),
                    "recommendable items contain a index >= n_items.");
          for (size_t i = 1; i < urec.size(); i++)
            check_arg(urec[i] > urec[i - 1], "duplicate recommendable items.");
        }
      }
    } catch (...) {
      delete c;
      throw;
    }
    *out = c;
  });
}

void orc_eval_destroy(void *h) { delete static_cast<Core *>(h); }

void *orc_metrics_create(int64_t n_item) { return new Metrics(n_item); }
void orc_metrics_destroy(void *m) { delete static_cast<Metrics *>(m); }
void orc_metrics_merge(void *a, void *b) {
  static_cast<Metrics *>(a)->merge(*static_cast<Metrics *>(b));
}
void orc_metrics_as_array(void *m, double *out11) {
  static_cast<Metrics *>(m)->as_array(out11);
}
void orc_metrics_item_cnt(void *m, int64_t *out) {
  auto *mm = static_cast<Metrics *>(m);
  std::copy(mm->item_cnt.begin(), mm->item_cnt.end(), out);
}
// raw accumulators: valid_user,total_user,hit,recall,ndcg,precision,map
void orc_metrics_raw(void *m, double *out7) {
  auto *mm = static_cast<Metrics *>(m);
  out7[0] = mm->valid_user;
  out7[1] = mm->total_user;
  out7[2] = mm->hit;
  out7[3] = mm->recall;
  out7[4] = mm->ndcg;
  out7[5] = mm->precision;
  out7[6] = mm->map;
}

// get_metrics_f32 / get_metrics_f64, evaluator.cpp:256-284; result is a new
// Metrics handle.
int orc_eval_get_metrics(void *h, int32_t is_f64, const void *scores,
                         int64_t rows, int64_t cutoff, int64_t offset,
                         int64_t n_threads, int32_t recall_with_cutoff,
                         void **out) {
  Core *c = static_cast<Core *>(h);
  return guard([&] {
    check_arg(cutoff >= 0 && offset >= 0 && n_threads >= 0, "negative argument");
    Metrics m = is_f64 ? c->get_metrics<double>(static_cast<const double *>(scores), rows,
                                                cutoff, offset, n_threads,
                                                recall_with_cutoff != 0)
                       : c->get_metrics<float>(static_cast<const float *>(scores), rows,
                                               cutoff, offset, n_threads,
                                               recall_with_cutoff != 0);
    *out = new Metrics(m);
  });
}

// evaluate_list_vs_list, evaluator.cpp:376-431 (ragged inputs).
int orc_eval_list_vs_list(int64_t n_users, const int64_t *rec_ptr,
                          const int64_t *rec, const int64_t *gt_ptr,
                          const int64_t *gt, int64_t n_items, int64_t n_threads,
                          void **out) {
  return guard([&] {
    (void)n_threads;
    for (int64_t p = 0; p < rec_ptr[n_users]; p++)
      check_arg(rec[p] >= 0 && rec[p] < n_items,
                "found recommendation index larger than n_items.");
    for (int64_t p = 0; p < gt_ptr[n_users]; p++)
      check_arg(gt[p] >= 0 && gt[p] < n_items,
                "found ground truth index larger than n_items.");
    Metrics *overall = new Metrics(n_items);
    for (int64_t u = 0; u < n_users; u++) {
      std::vector<size_t> r(rec + rec_ptr[u], rec + rec_ptr[u + 1]);
      std::unordered_set<size_t> g(gt + gt_ptr[u], gt + gt_ptr[u + 1]);
      overall->total_user++;
      overall->update(r, g, false);
    }
    *out = overall;
  });
}

}  // extern "C"
