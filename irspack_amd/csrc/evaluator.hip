// Ranking-metric evaluator on gfx950.  Replaces EvaluatorCore::get_metrics /
// get_metrics_local / Metrics::update of /root/reference/cpp_source/evaluator.cpp
// (:256-284, :292-367, :127-166) behind include/irspack_amd.h.
//
// One workgroup ranks one user row (1024 threads with the row's keys held in registers or
// LDS after one pass over the scores, else 256 threads re-reading the scores per pass):
//   1. candidates = all items | global list | per-user list, score != -inf (:324-348)
//   2. top-`cutoff` by the total order (score desc, index asc) — the order
//      std::partial_sort gives on (-score, index) pairs (:329, :353-355):
//      MSB-first 8-bit radix select of the cutoff-th key (wave-aggregated histogram, early
//      exit), then an index-ordered pick of the ties at the threshold, then a bitonic sort
//      of the winners (a shuffle network in one wave for cutoff <= 64, else in LDS)
//   3. hits against the (recommendable-filtered) ground-truth row, then the sequential
//      dcg / AP recurrences of Metrics::update
// Per-user terms are folded by a fixed-order wave reduction (double), item counts with
// int64 atomics.  The same kernel serves retrieve_recommend_from_score (no ground truth).
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <limits>
#include <memory>

#include <thread>
#include <cstring>

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
  int32_t retrieve;          // 1: only the ranked lists are wanted (no ground truth, no metrics)
  int32_t recall_with_cutoff;
  const double *disc;        // 1 / log2(2 + i)
  const double *idcg_prefix; // sequential prefix sums of disc
  RowOut *out;
  int32_t *rec_out;          // [rows, cutoff] recommended items (-1 padded)
  unsigned long long *item_cnt;
  int32_t *todo;             // per row: 1 = left to rank_rows_kernel by rank_wave_kernel (or null)
  // score row r stands for row row_map[r] of the call (ground truth, outputs); null: r itself
  const int32_t *row_map = nullptr;
  // cutoff > SEL_CAP (rank_rows_kernel<..., BIG>): the selected (key, index, hit) lists live in
  // global scratch, `big_cap` (a power of two >= cutoff) entries per workgroup of the launch;
  // the launch covers the rows row_base + blockIdx.x
  int64_t row_base = 0;
  uint64_t *big_key = nullptr;
  int32_t *big_idx = nullptr;
  uint8_t *big_hit = nullptr;
  int32_t big_cap = 0;
};

// key value no score maps to (it is the image of a negative NaN pattern, and NaNs are
// canonicalised to 0): marks "not rankable" (-inf) in the cached keys
constexpr uint64_t KEY_SKIP = 1ull;
template <class T> struct KeyStore;
template <> struct KeyStore<float> { using type = uint32_t; };
template <> struct KeyStore<double> { using type = uint64_t; };

// NT threads per user row.  Wave w owns the contiguous candidates [w * span, (w + 1) * span) in
// 64-wide steps: slot q of lane l is candidate w * span + 64 q + l, so (wave, slot, lane)
// order is index order.  Where the keys live between the passes:
//   MODE 2 (NT = 1024, float scores, <= 32 slots per lane): in registers - one pass over the
//          scores in HBM, no LDS traffic for keys (same speed as MODE 1 today: the kernel is
//          bound by its barrier / latency chain, and the 64-VGPR budget that a second
//          workgroup per CU would need spills);
//   MODE 1 (NT = 1024): in dynamic LDS - one pass over the scores, one workgroup per CU;
//   MODE 0 (NT = 256): nowhere, every pass re-reads the scores.
constexpr int RANK_MAXQ = 32;

// BIG (cutoff > SEL_CAP, any cutoff up to n_items like the reference, evaluator.cpp:263-268):
// the selected lists are kept in global scratch instead of LDS (a workgroup's own global stores
// are visible to its other waves after __syncthreads: they share the CU's L1), everything else
// is the same code.
template <class T, int NT, int MODE, int MAXQ = RANK_MAXQ, bool BIG = false>
__global__ __launch_bounds__(NT, MODE == 2 ? 4 : 1) void rank_rows_kernel(EvalParams p) {
  using KeyT = typename KeyStore<T>::type;
  constexpr int NWV = NT / 64;
  constexpr bool CACHED = MODE == 1;
  extern __shared__ __attribute__((aligned(16))) unsigned char rank_dyn[];
  KeyT *keys = reinterpret_cast<KeyT *>(rank_dyn);
  __shared__ uint32_t hist[256];
  __shared__ uint64_t sel_key_lds[BIG ? 1 : SEL_CAP];
  __shared__ int32_t sel_idx_lds[BIG ? 1 : SEL_CAP];
  __shared__ uint8_t sel_hit_lds[BIG ? 1 : SEL_CAP];
  __shared__ int32_t scan_buf[NWV > 4 ? NWV : 4];
  __shared__ uint64_t sh_prefix;
  __shared__ int32_t sh_need, sh_count;
  const int sel_cap = BIG ? p.big_cap : SEL_CAP;
  uint64_t *sel_key = BIG ? p.big_key + static_cast<size_t>(blockIdx.x) * p.big_cap : sel_key_lds;
  int32_t *sel_idx = BIG ? p.big_idx + static_cast<size_t>(blockIdx.x) * p.big_cap : sel_idx_lds;
  uint8_t *sel_hit = BIG ? p.big_hit + static_cast<size_t>(blockIdx.x) * p.big_cap : sel_hit_lds;

  const int tid = threadIdx.x;
  const int wv = tid >> 6, ln = tid & 63;
  const int64_t row = p.row_base + blockIdx.x;
  if (p.todo != nullptr && p.todo[row] == 0) return;  // ranked by rank_wave_kernel
  const int64_t orow = p.row_map ? p.row_map[row] : row;
  const int64_t u = orow + p.offset;
  const T *srow = static_cast<const T *>(p.scores) + row * p.n_items;
  RowOut res{0, 0, 0, 0, 0, 0, 0};
  const int gb = p.retrieve ? 0 : p.gt_ptr[u], ge = p.retrieve ? 0 : p.gt_ptr[u + 1];
  const int n_gt = ge - gb;
  int32_t *rec_row = p.rec_out + orow * p.cutoff;
  for (int i = tid; i < p.cutoff; i += NT) rec_row[i] = -1;
  if (n_gt == 0 && !p.retrieve) {  // counted in total_user only (:316-321)
    if (tid == 0) p.out[orow] = res;
    return;
  }
  // what the last phase needs from HBM is requested now, so that its latency hides behind
  // the ranking: a short ground-truth row, the discounts, the ideal dcg
  int32_t gt_pref = -1;
  double disc_pref = 0.0, idcg_pref = 0.0;
  if (!p.retrieve) {
    if (n_gt <= 64 && ln < n_gt) gt_pref = p.gt_idx[gb + ln];
    if (ln < min(p.cutoff, 64)) disc_pref = p.disc[ln];
    idcg_pref = p.idcg_prefix[min(n_gt, p.cutoff)];
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
  auto key_from_scores = [&](int64_t j) -> uint64_t {
    const T s = srow[item_of(j)];
    return is_neg_inf(s) ? KEY_SKIP : order_key(s);
  };
  auto key_at = [&](int64_t j) -> uint64_t {
    return CACHED ? static_cast<uint64_t>(keys[j]) : key_from_scores(j);
  };

  // --- count rankable candidates (and fill the key cache).  Eight independent loads per
  //     thread are issued before the first use (clamped index, result masked) so the one
  //     pass over the scores in HBM is not a chain of exposed latencies.
  const int64_t span = ((n_cand + NT - 1) / NT) * 64;
  const int64_t wbeg = wv * span, wend = min<int64_t>(wbeg + span, n_cand);
  KeyT kreg[MODE == 2 ? MAXQ : 1];
  int local = 0;
  uint64_t tmax = 0;  // best rankable key of this thread (0: none)
  if constexpr (MODE == 2) {
#pragma unroll
    for (int qb = 0; qb < MAXQ; qb += 8) {  // eight loads in flight, small register peak
      T sv[8];
#pragma unroll
      for (int q = 0; q < 8; q++)
        sv[q] = srow[item_of(max<int64_t>(min<int64_t>(wbeg + 64 * (qb + q) + ln, n_cand - 1), 0))];
#pragma unroll
      for (int q = 0; q < 8; q++) {
        const bool live = wbeg + 64 * (qb + q) + ln < wend;
        const uint64_t k = (!live || is_neg_inf(sv[q])) ? KEY_SKIP : order_key(sv[q]);
        kreg[qb + q] = static_cast<KeyT>(k);
        local += k != KEY_SKIP;
        tmax = (k != KEY_SKIP && k > tmax) ? k : tmax;
      }
    }
  } else {
    for (int64_t base = 0; base < n_cand; base += NT * 8) {
      T sv[8];
#pragma unroll
      for (int q = 0; q < 8; q++)
        sv[q] = srow[item_of(min<int64_t>(base + q * NT + tid, n_cand - 1))];
#pragma unroll
      for (int q = 0; q < 8; q++) {
        const int64_t j = base + q * NT + tid;
        if (j < n_cand) {
          const uint64_t k = is_neg_inf(sv[q]) ? KEY_SKIP : order_key(sv[q]);
          if (CACHED) keys[j] = static_cast<KeyT>(k);
          local += k != KEY_SKIP;
          tmax = (k != KEY_SKIP && k > tmax) ? k : tmax;
        }
      }
    }
  }
  // visit every slot of this wave, all lanes together (f may use wave-wide ballots); f
  // returns false to stop early (a wave-uniform decision)
  auto for_each_slot = [&](auto &&f) {
    if constexpr (MODE == 2) {
      bool go = true;
#pragma unroll
      for (int q = 0; q < MAXQ; q++)
        if (go && wbeg + 64 * q < wend) go = f(wbeg + 64 * q + ln, static_cast<uint64_t>(kreg[q]));
    } else {
      for (int64_t base = wbeg; base < wend; base += 64) {
        const int64_t j = base + ln;
        if (!f(j, j < wend ? key_at(j) : KEY_SKIP)) break;
      }
    }
  };
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) local += __shfl_xor(local, o, 64);
  if (ln == 0) scan_buf[wv] = local;
  if (tid < 256) hist[tid] = 0;
  __syncthreads();
  int n_rankable = 0;
  for (int w = 0; w < NWV; w++) n_rankable += scan_buf[w];
  __syncthreads();
  const int n_rec = min(p.cutoff, n_rankable);
  res.valid = 1;
  res.n_rec = n_rec;
  if (n_rec == 0) {  // :132-135
    if (tid == 0) p.out[orow] = res;
    return;
  }

  // --- short lists (the usual cutoffs): the n_rec-th largest of the per-thread maxima is a
  //     lower bound L of the n_rec-th largest key (those maxima are n_rec different keys
  //     >= L).  L costs a radix select over ONE key per thread, and the keys >= L - the
  //     n_rec winners, all their ties and a few more - are gathered in one sweep and
  //     sorted exactly by (key desc, index asc) below.  Rows where that list does not fit
  //     (e.g. thousands of equal scores) take the general selection.
  constexpr int BITS = KeyBits<T>::value;
  int n_sel = n_rec;  // entries in sel_key / sel_idx; the best n_rec of them are the result
  bool fast = n_rec <= NT;
  if (fast) {
    uint64_t lpre = 0;
    int lneed = n_rec;
    for (int shift = BITS - 8; shift >= 0; shift -= 8) {
      const uint64_t hi_mask = (shift + 8 >= 64) ? 0ull : (~0ull << (shift + 8));
      const bool in = (tmax & hi_mask) == (lpre & hi_mask);
      const uint32_t digit = static_cast<uint32_t>(tmax >> shift) & 0xffu;
      const unsigned long long todo = __ballot(in);
      if (todo) {
        const int leader = __ffsll(static_cast<long long>(todo)) - 1;
        const uint32_t d = __builtin_amdgcn_readlane(digit, leader);
        const unsigned long long same = __ballot(in && digit == d);
        if (same == todo) {
          if (ln == leader) atomicAdd(&hist[d], static_cast<uint32_t>(__popcll(same)));
        } else if (in) {
          atomicAdd(&hist[digit], 1u);
        }
      }
      __syncthreads();
      int bin = 0, incl = 0;
      if (tid < 256) {
        bin = static_cast<int>(hist[255 - tid]);
        incl = bin;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
          const int t = __shfl_up(incl, o, 64);
          if (ln >= o) incl += t;
        }
        if (ln == 63) scan_buf[wv] = incl;
      }
      __syncthreads();
      if (tid < 256) {
        for (int w = 0; w < wv && w < 4; w++) incl += scan_buf[w];
        if (incl - bin < lneed && incl >= lneed) {
          sh_prefix = lpre | (static_cast<uint64_t>(255 - tid) << shift);
          sh_need = (incl == lneed) ? -1 : lneed - (incl - bin);
        }
        hist[tid] = 0;
      }
      __syncthreads();
      lpre = sh_prefix;
      lneed = sh_need;
      if (lneed < 0) break;  // the whole bin is wanted: its lowest possible key bounds L
    }
    const uint64_t L = lpre;
    if (tid == 0) sh_count = 0;
    __syncthreads();
    for_each_slot([&](int64_t j, uint64_t k) {
      if (k != KEY_SKIP && k >= L) {
        const int pos = atomicAdd(&sh_count, 1);
        if (pos < sel_cap) {
          sel_key[pos] = k;
          sel_idx[pos] = item_of(j);
        }
      }
      return true;
    });
    __syncthreads();
    n_sel = sh_count;
    fast = n_sel <= sel_cap;
    __syncthreads();  // sh_count is reset by the general selection
  }
  if (!fast) {
  n_sel = n_rec;
  // --- radix select: key of the n_rec-th best candidate
  uint64_t prefix = 0;
  int need = n_rec;  // how many still to take among keys matching `prefix` so far
  bool inclusive = false;  // true: every key >= prefix is taken and there is no tie pick
  // hist is zero here (cleared before the count barrier) and is cleared again by the bin scan
  // of every pass: three barriers per digit
  for (int shift = BITS - 8; shift >= 0; shift -= 8) {
    const uint64_t hi_mask = (shift + 8 >= 64) ? 0ull : (~0ull << (shift + 8));
    for_each_slot([&](int64_t, uint64_t k) {
      const bool in = k != KEY_SKIP && (k & hi_mask) == (prefix & hi_mask);
      const uint32_t digit = static_cast<uint32_t>(k >> shift) & 0xffu;
      // scores share their leading byte(s): when all candidate lanes of a wave fall into
      // one bin a single lane adds the count (no same-address atomic storm); otherwise
      // the digits are spread and per-lane atomics are cheap
      const unsigned long long todo = __ballot(in);
      if (todo) {
        const int leader = __ffsll(static_cast<long long>(todo)) - 1;
        const uint32_t d = __builtin_amdgcn_readlane(digit, leader);
        const unsigned long long same = __ballot(in && digit == d);
        if (same == todo) {
          if (ln == leader) atomicAdd(&hist[d], static_cast<uint32_t>(__popcll(same)));
        } else if (in) {
          atomicAdd(&hist[digit], 1u);
        }
      }
      return true;
    });
    __syncthreads();
    // the digit d with  #(digits > d) < need <= #(digits >= d): bins scanned in descending
    // order by the first four waves
    int bin = 0, incl = 0;
    if (tid < 256) {
      bin = static_cast<int>(hist[255 - tid]);
      incl = bin;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(incl, o, 64);
        if (ln >= o) incl += t;
      }
      if (ln == 63) scan_buf[wv] = incl;
    }
    __syncthreads();
    if (tid < 256) {
      for (int w = 0; w < wv; w++) incl += scan_buf[w];
      if (incl - bin < need && incl >= need) {
        sh_prefix = prefix | (static_cast<uint64_t>(255 - tid) << shift);
        // the whole bin is wanted: no lower digit has to be resolved (-1 stops the loop)
        sh_need = (incl == need) ? -1 : need - (incl - bin);
      }
      hist[tid] = 0;  // every bin was read before the barrier above
    }
    __syncthreads();
    // (the next write to sh_prefix / sh_need sits two barriers ahead: no barrier after the read)
    prefix = sh_prefix;
    need = sh_need;
    if (need < 0) {
      inclusive = true;
      need = 0;
      break;
    }
  }
  const uint64_t thr = prefix;  // threshold key; `need` ties at thr are taken, lowest index first

  // --- gather winners ((wave, slot, lane) order is index order): sweep 1 appends the keys
  //     above the threshold and counts the wave's ties, sweep 2 gives the `need` lowest-index
  //     ties their slots (evaluator.cpp:329, 353-355)
  if (tid == 0) sh_count = 0;
  __syncthreads();
  int my_ties = 0;
  for_each_slot([&](int64_t j, uint64_t k) {
    bool tie = false;
    if (k != KEY_SKIP) {
      if (inclusive ? k >= thr : k > thr) {
        const int pos = atomicAdd(&sh_count, 1);
        sel_key[pos] = k;
        sel_idx[pos] = item_of(j);
      } else if (k == thr) {
        tie = true;
      }
    }
    my_ties += __popcll(__ballot(tie));
    return true;
  });
  if (ln == 0) scan_buf[wv] = my_ties;
  __syncthreads();
  int seen = 0;
  for (int w = 0; w < wv; w++) seen += scan_buf[w];
  for_each_slot([&](int64_t j, uint64_t k) {
    if (seen >= need) return false;
    const bool tie = k != KEY_SKIP && k == thr;
    const unsigned long long bal = __ballot(tie);
    const int rank = seen + __popcll(bal & ((1ull << ln) - 1ull));
    if (tie && rank < need) {
      const int pos = atomicAdd(&sh_count, 1);
      sel_key[pos] = k;
      sel_idx[pos] = item_of(j);
    }
    seen += __popcll(bal);
    return true;
  });
  __syncthreads();
  }  // general selection
  if (n_sel <= 64) {
    // --- common cutoffs: one wave finishes the row without further barriers.  Lane i holds
    //     candidate i (the best n_rec of the n_sel gathered ones are the list); bitonic network
    //     over the lanes (key desc, index asc), hits by binary search, then the sequential
    //     dcg / AP recurrences of Metrics::update (:136-165).
    if (wv != 0) return;
    uint64_t mk = ln < n_sel ? sel_key[ln] : 0ull;
    int32_t mi = ln < n_sel ? sel_idx[ln] : 0x7fffffff;
#pragma unroll
    for (int k2 = 2; k2 <= 64; k2 <<= 1) {
#pragma unroll
      for (int j2 = k2 >> 1; j2 > 0; j2 >>= 1) {
        const uint64_t ok = __shfl_xor(static_cast<unsigned long long>(mk), j2, 64);
        const int32_t oi = __shfl_xor(mi, j2, 64);
        const bool mine_first = (mk != ok) ? mk > ok : mi < oi;
        const bool lower = (ln & j2) == 0, up = (ln & k2) == 0;
        const bool keep = (lower == up) ? mine_first : !mine_first;
        mk = keep ? mk : ok;
        mi = keep ? mi : oi;
      }
    }
    if (p.retrieve) {
      if (ln < n_rec) rec_row[ln] = mi;
      return;
    }
    bool hit = false;
    if (ln < n_rec) rec_row[ln] = mi;  // (counted by item_hist_kernel, :146)
    if (n_gt <= 64) {
      // short ground-truth row (prefetched at kernel start): compare against every entry
      // instead of a chain of dependent loads
      for (int c = 0; c < n_gt; c++) hit |= __builtin_amdgcn_readlane(gt_pref, c) == mi;
      hit = hit && ln < n_rec;
    } else if (ln < n_rec) {
      int lo = gb, hi = ge;
      while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (p.gt_idx[mid] < mi) lo = mid + 1; else hi = mid;
      }
      hit = lo < ge && p.gt_idx[lo] == mi;
    }
    const double disc_l = ln < n_rec ? disc_pref : 0.0;
    unsigned long long hits = __ballot(hit);
    double dcg = 0, ap = 0;
    int cum_hit = 0;
    while (hits) {  // ascending rank order, like the reference's loop over the list
      const int i = __ffsll(static_cast<long long>(hits)) - 1;
      hits &= hits - 1;
      dcg += __shfl(disc_l, i, 64);
      cum_hit++;
      ap += static_cast<double>(cum_hit) / (i + 1);
    }
    const double idcg = min(n_gt, n_rec) == min(n_gt, p.cutoff)
                            ? idcg_pref
                            : p.idcg_prefix[min(n_gt, n_rec)];
    res.hit = cum_hit > 0 ? 1.0 : 0.0;
    res.precision = cum_hit / static_cast<double>(n_rec);
    res.recall = cum_hit / static_cast<double>(
                               p.recall_with_cutoff ? (n_gt > n_rec ? n_rec : n_gt) : n_gt);
    res.ndcg = dcg / idcg;
    res.map = ap / n_gt;
    if (ln == 0) p.out[orow] = res;
    return;
  }
  // --- bitonic sort of the n_sel gathered candidates: key desc, index asc
  int n_pow = 1;
  while (n_pow < n_sel) n_pow <<= 1;
  for (int i = n_sel + tid; i < n_pow; i += NT) {
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
      for (int i = tid; i < n_pow; i += NT) {
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
  if (p.retrieve) {
    for (int i = tid; i < n_rec; i += NT) rec_row[i] = sel_idx[i];
    return;
  }
  // --- hits, histogram, output list
  for (int i = tid; i < n_rec; i += NT) {
    const int32_t it = sel_idx[i];
    rec_row[i] = it;  // (counted by item_hist_kernel, :146)
    int lo = gb, hi = ge;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (p.gt_idx[mid] < it) lo = mid + 1; else hi = mid;
    }
    sel_hit[i] = (lo < ge && p.gt_idx[lo] == it) ? 1 : 0;
  }
  __syncthreads();
  if (wv == 0) {
    // Metrics::update :136-165 in the same sequential order: the wave reads 64 hit flags at a
    // time and walks the set bits in ascending rank (every lane carries the same sums)
    double dcg = 0, ap = 0;
    int cum_hit = 0;
    for (int base = 0; base < n_rec; base += 64) {
      unsigned long long hits = __ballot(base + ln < n_rec && sel_hit[base + ln] != 0);
      while (hits) {
        const int i = base + __ffsll(static_cast<long long>(hits)) - 1;
        hits &= hits - 1;
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
    if (ln == 0) p.out[orow] = res;
  }
}

}  // namespace eval
}  // namespace irs
#include "eval_fused_kernels.hpp"
namespace irs {
namespace eval {

// ---------------------------------------------------------------------------
// One WAVE per user row, one pass over the scores, no barrier and no LDS: the usual case
// (every item is a candidate, cutoff <= 64).  Lane l streams the scores l, l + 64, ... with
// U loads in flight and keeps its own M best (key, index) pairs sorted in registers.  The
// list is then drawn from the 64 heads: a wave-wide arg-max of (key desc, index asc) per
// rank, the winning lane popping its head; lane i ends up with rank i, exactly the state the
// single-wave finish of rank_rows_kernel starts from.  A lane that runs dry although it saw
// more than M rankable scores may hide a better candidate: the row is then flagged in p.todo
// and ranked by rank_rows_kernel (for cutoff 20 and M = 4 about one row in a thousand).
// 8 waves per SIMD are resident, so the loads of some rows overlap the drawing of others.
// Last phase of a wave-ranked row: lane i holds the item of rank i (mi, i < n_rec).  Hits,
// then the SEQUENTIAL dcg / AP recurrences of Metrics::update (evaluator.cpp:136-165) so that
// the fp64 terms are those of the reference.
__device__ __forceinline__ void wave_metrics_tail(const EvalParams &p, int64_t row, int ln,
                                                  int32_t mi, int n_rec, int gb, int ge, int n_gt,
                                                  int32_t gt_pref, double disc_pref,
                                                  double idcg_pref, RowOut res, int32_t *rec_row) {
  if (p.retrieve) {
    if (ln < n_rec) rec_row[ln] = mi;
    return;
  }
  bool hit = false;
  if (ln < n_rec) rec_row[ln] = mi;  // (counted by item_hist_kernel, :146)
  if (n_gt <= 64) {
    for (int c = 0; c < n_gt; c++) hit |= __builtin_amdgcn_readlane(gt_pref, c) == mi;
    hit = hit && ln < n_rec;
  } else if (ln < n_rec) {
    int lo = gb, hi = ge;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (p.gt_idx[mid] < mi) lo = mid + 1; else hi = mid;
    }
    hit = lo < ge && p.gt_idx[lo] == mi;
  }
  const double disc_l = ln < n_rec ? disc_pref : 0.0;
  unsigned long long hits = __ballot(hit);
  double dcg = 0, ap = 0;
  int cum_hit = 0;
  while (hits) {  // ascending rank order, like the reference's loop over the list (:136-165)
    const int i = __ffsll(static_cast<long long>(hits)) - 1;
    hits &= hits - 1;
    dcg += __shfl(disc_l, i, 64);
    cum_hit++;
    ap += static_cast<double>(cum_hit) / (i + 1);
  }
  const double idcg = min(n_gt, n_rec) == min(n_gt, p.cutoff) ? idcg_pref
                                                              : p.idcg_prefix[min(n_gt, n_rec)];
  res.hit = cum_hit > 0 ? 1.0 : 0.0;
  res.precision = cum_hit / static_cast<double>(n_rec);
  res.recall = cum_hit / static_cast<double>(
                             p.recall_with_cutoff ? (n_gt > n_rec ? n_rec : n_gt) : n_gt);
  res.ndcg = dcg / idcg;
  res.map = ap / n_gt;
  if (ln == 0) p.out[row] = res;
}

// Emit path, last kernel: one wave per user ranks its candidate list (score_emit_kernel:
// every unmasked item with score >= tau_u, in arbitrary order) by (score desc, index asc).
// Lane l keeps the M best of the entries l, l + 64, ... sorted in registers, the list is drawn
// from the 64 heads; a lane that runs dry with entries left over may hide a better one: the row
// is then ranked by rank_cand_slow_kernel.
template <int M>
__global__ __launch_bounds__(256) void rank_cand_kernel(EvalParams p, EmitParams f,
                                                        const int32_t *__restrict__ n_masked) {
  const int ln = threadIdx.x & 63;
  const int64_t row = static_cast<int64_t>(blockIdx.x) * 4 + wave_index_in_block();
  if (row >= p.rows) return;
  const int64_t u = row + p.offset;
  RowOut res{0, 0, 0, 0, 0, 0, 0};
  const int gb = p.gt_ptr[u], ge = p.gt_ptr[u + 1];
  const int n_gt = ge - gb;
  int32_t *rec_row = p.rec_out + row * p.cutoff;
  if (ln < p.cutoff) rec_row[ln] = -1;  // cutoff <= 32
  if (ln == 0) p.todo[row] = 0;
  if (n_gt == 0) {  // counted in total_user only (:316-321)
    if (ln == 0) p.out[row] = res;
    return;
  }
  int32_t gt_pref = -1;
  double disc_pref = 0.0;
  if (n_gt <= 64 && ln < n_gt) gt_pref = p.gt_idx[gb + ln];
  if (ln < p.cutoff) disc_pref = p.disc[ln];
  const double idcg_pref = p.idcg_prefix[min(n_gt, p.cutoff)];
  if (f.hard[row]) return;  // ranked from its full score row afterwards
  const int n = min(f.cand_cnt[row], EM_CAP);
  const int64_t n_rankable = f.n_items - (n_masked ? n_masked[row] : 0);
  const int n_rec = static_cast<int>(min<int64_t>(p.cutoff, n_rankable));
  res.valid = 1;
  res.n_rec = n_rec;
  if (n_rec == 0) {  // :132-135
    if (ln == 0) p.out[row] = res;
    return;
  }
  if (n < n_rec) {  // cannot happen with a valid threshold: have the host repeat the call
    if (ln == 0) atomicOr(f.bad_flag, 2);
    if (ln == 0) p.out[row] = res;
    return;
  }
  const float NEG_INF = -std::numeric_limits<float>::infinity();
  const float *cs = f.cand_score + static_cast<size_t>(row) * EM_CAP;
  const int32_t *ci = f.cand_item + static_cast<size_t>(row) * EM_CAP;
  if (n <= 64) {
    // The usual case after pruning (a few dozen candidates): one candidate per lane, its rank
    // = the number of candidates that go before it (score desc, index asc: a strict order, so
    // the ranks are a permutation), counted against every candidate by v_readlane; the item
    // of rank r is then pushed to lane r.
    const float s = ln < n ? cs[ln] : NEG_INF;
    const int32_t i = ln < n ? ci[ln] : 0x7fffffff;
    int rank = 0;
    for (int k = 0; k < n; k++) {  // n is wave-uniform
      const float sk = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, s), k));
      const int32_t ik = __builtin_amdgcn_readlane(i, k);
      rank += (sk > s || (sk == s && ik < i)) ? 1 : 0;
    }
    if (ln >= n) rank = ln;  // lanes without a candidate keep to themselves
    const int32_t mi = __builtin_amdgcn_ds_permute(rank << 2, i);
    wave_metrics_tail(p, row, ln, mi, n_rec, gb, ge, n_gt, gt_pref, disc_pref, idcg_pref, res, rec_row);
    return;
  }
  float bs[M];
  int32_t bi[M];
#pragma unroll
  for (int t = 0; t < M; t++) {
    bs[t] = NEG_INF;
    bi[t] = 0x7fffffff;
  }
  for (int e0 = 0; e0 < n; e0 += 64) {
    const int e = e0 + ln;
    float s = e < n ? cs[e] : NEG_INF;
    int32_t i = e < n ? ci[e] : 0x7fffffff;
#pragma unroll
    for (int t = 0; t < M; t++) {
      const bool better = s > bs[t] || (s == bs[t] && i < bi[t]);
      const float ts = bs[t];
      const int32_t ti = bi[t];
      bs[t] = better ? s : ts;
      bi[t] = better ? i : ti;
      s = better ? ts : s;
      i = better ? ti : i;
    }
  }
  const int mine_total = (n - ln + 63) / 64;  // entries this lane has seen
  int32_t mi = 0x7fffffff;
  int popped = 0;
  for (int it = 0; it < n_rec; it++) {
    float ws = bs[0];
    int32_t wi = bi[0];
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
      const float os = __shfl_xor(ws, o, 64);
      const int32_t oi = __shfl_xor(wi, o, 64);
      const bool better = os > ws || (os == ws && oi < wi);
      ws = better ? os : ws;
      wi = better ? oi : wi;
    }
    if (ln == it) mi = wi;
    if (bi[0] == wi && bs[0] == ws) {  // item ids are unique: one lane
#pragma unroll
      for (int t = 0; t + 1 < M; t++) {
        bs[t] = bs[t + 1];
        bi[t] = bi[t + 1];
      }
      bs[M - 1] = NEG_INF;
      bi[M - 1] = 0x7fffffff;
      popped++;
    }
    if (it + 1 < n_rec && __any(popped == M && mine_total > M)) {
      if (ln == 0) p.todo[row] = 1;
      return;
    }
  }
  wave_metrics_tail(p, row, ln, mi, n_rec, gb, ge, n_gt, gt_pref, disc_pref, idcg_pref, res, rec_row);
}

// The rows rank_cand_kernel flagged: exact rank of every candidate by counting (LDS copy).
__global__ __launch_bounds__(256) void rank_cand_slow_kernel(EvalParams p, EmitParams f,
                                                             const int32_t *__restrict__ n_masked) {
  __shared__ float cs[4][EM_CAP];
  __shared__ int32_t ci[4][EM_CAP];
  __shared__ int32_t sel[4][64];
  const int ln = threadIdx.x & 63, wv = wave_index_in_block();
  const int64_t row = static_cast<int64_t>(blockIdx.x) * 4 + wv;
  if (row >= p.rows || p.todo[row] == 0) return;
  const int64_t u = row + p.offset;
  RowOut res{0, 0, 0, 0, 0, 0, 0};
  const int gb = p.gt_ptr[u], ge = p.gt_ptr[u + 1];
  const int n_gt = ge - gb;
  int32_t *rec_row = p.rec_out + row * p.cutoff;
  int32_t gt_pref = -1;
  double disc_pref = 0.0;
  if (n_gt <= 64 && ln < n_gt) gt_pref = p.gt_idx[gb + ln];
  if (ln < p.cutoff) disc_pref = p.disc[ln];
  const double idcg_pref = p.idcg_prefix[min(n_gt, p.cutoff)];
  const int n = min(f.cand_cnt[row], EM_CAP);
  const int64_t n_rankable = f.n_items - (n_masked ? n_masked[row] : 0);
  const int n_rec = static_cast<int>(min<int64_t>(p.cutoff, n_rankable));
  res.valid = 1;
  res.n_rec = n_rec;
  for (int e = ln; e < n; e += 64) {
    cs[wv][e] = f.cand_score[static_cast<size_t>(row) * EM_CAP + e];
    ci[wv][e] = f.cand_item[static_cast<size_t>(row) * EM_CAP + e];
  }
  __threadfence_block();
  for (int e = ln; e < n; e += 64) {
    const float s = cs[wv][e];
    const int32_t i = ci[wv][e];
    int rank = 0;
    for (int k = 0; k < n; k++) {
      const float sk = cs[wv][k];
      const int32_t ik = ci[wv][k];
      rank += (sk > s || (sk == s && ik < i)) ? 1 : 0;
    }
    if (rank < n_rec) sel[wv][rank] = i;
  }
  __threadfence_block();
  const int32_t mi = ln < n_rec ? sel[wv][ln] : 0x7fffffff;
  wave_metrics_tail(p, row, ln, mi, n_rec, gb, ge, n_gt, gt_pref, disc_pref, idcg_pref, res, rec_row);
}

template <class T, int M>
__global__ __launch_bounds__(256) void rank_wave_kernel(EvalParams p) {
  constexpr int U = 16;
  const int ln = threadIdx.x & 63;
  const int64_t row = static_cast<int64_t>(blockIdx.x) * 4 + wave_index_in_block();
  if (row >= p.rows) return;
  const int64_t orow = p.row_map ? p.row_map[row] : row;
  const int64_t u = orow + p.offset;
  const T *srow = static_cast<const T *>(p.scores) + row * p.n_items;
  RowOut res{0, 0, 0, 0, 0, 0, 0};
  const int gb = p.retrieve ? 0 : p.gt_ptr[u], ge = p.retrieve ? 0 : p.gt_ptr[u + 1];
  const int n_gt = ge - gb;
  int32_t *rec_row = p.rec_out + orow * p.cutoff;
  if (ln < p.cutoff) rec_row[ln] = -1;  // cutoff <= 64
  if (ln == 0) p.todo[row] = 0;
  if (n_gt == 0 && !p.retrieve) {  // counted in total_user only (:316-321)
    if (ln == 0) p.out[orow] = res;
    return;
  }
  int32_t gt_pref = -1;
  double disc_pref = 0.0, idcg_pref = 0.0;
  if (!p.retrieve) {
    if (n_gt <= 64 && ln < n_gt) gt_pref = p.gt_idx[gb + ln];
    if (ln < p.cutoff) disc_pref = p.disc[ln];
    idcg_pref = p.idcg_prefix[min(n_gt, p.cutoff)];
  }
  // The lists hold the raw scores: float comparison is the reference's order (-0.0 == +0.0
  // ties; equal scores keep index order because insertion is strict), -inf marks an empty
  // slot and is never inserted - exactly the scores that are not rankable.  A NaN (ranks last
  // among the rankable, evaluator.hip header) would never be inserted either, so a row that
  // contains one goes to the general kernel.
  const T NEG_INF = -std::numeric_limits<T>::infinity();
  T bs[M];
  int32_t bi[M];
#pragma unroll
  for (int t = 0; t < M; t++) {
    bs[t] = NEG_INF;
    bi[t] = 0x7fffffff;
  }
  int n_rankable = 0;             // wave-uniform
  unsigned long long nan_any = 0;  // lanes that met a NaN
  const int32_t n = static_cast<int32_t>(p.n_items);
  auto insert = [&](T cs, int32_t ci) {
    if (__any(cs > bs[M - 1])) {
      // from the first entry the new score beats, the tail SHIFTS down one place (sticky
      // `gt`): an entry pushed down by one place still goes before its equals, whose
      // indices are higher
      bool gt = false;
#pragma unroll
      for (int t = 0; t < M; t++) {
        gt = gt || cs > bs[t];
        const T ts = bs[t];
        const int32_t ti = bi[t];
        bs[t] = gt ? cs : ts;
        bi[t] = gt ? ci : ti;
        cs = gt ? ts : cs;
        ci = gt ? ti : ci;
      }
    }
  };
  int32_t base = 0;
  for (; base + 64 * U <= n; base += 64 * U) {  // full groups: no bounds to check
    const T *sp = srow + base + ln;
    T sv[U];
#pragma unroll
    for (int q = 0; q < U; q++) sv[q] = sp[64 * q];
#pragma unroll
    for (int q = 0; q < U; q++) {
      nan_any |= __ballot(sv[q] != sv[q]);
      n_rankable += __popcll(__ballot(sv[q] != NEG_INF));
      insert(sv[q], base + 64 * q + ln);
    }
  }
  if (base < n) {  // the last, partial group
    T sv[U];
#pragma unroll
    for (int q = 0; q < U; q++) sv[q] = srow[min(base + 64 * q + ln, n - 1)];
#pragma unroll
    for (int q = 0; q < U; q++) {
      const int32_t j = base + 64 * q + ln;
      const T s1 = j < n ? sv[q] : NEG_INF;
      nan_any |= __ballot(s1 != s1);
      n_rankable += __popcll(__ballot(s1 != NEG_INF));
      insert(s1, j);
    }
  }
  if (nan_any) {
    if (ln == 0) p.todo[row] = 1;
    return;
  }
  const int n_rec = min(p.cutoff, n_rankable);
  res.valid = 1;
  res.n_rec = n_rec;
  if (n_rec == 0) {  // :132-135
    if (ln == 0) p.out[orow] = res;
    return;
  }
  // --- draw the list: rank `it` is the best head
  int32_t mi = 0x7fffffff;
  int popped = 0;
  for (int it = 0; it < n_rec; it++) {
    T ws = bs[0];
    int32_t wi = bi[0];
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
      const T os = __shfl_xor(ws, o, 64);
      const int32_t oi = __shfl_xor(wi, o, 64);
      const bool better = os > ws || (os == ws && oi < wi);
      ws = better ? os : ws;
      wi = better ? oi : wi;
    }
    if (ln == it) mi = wi;
    const bool mine = bi[0] == wi && bs[0] == ws;  // indices are unique: one lane
    if (mine) {
#pragma unroll
      for (int t = 0; t + 1 < M; t++) {
        bs[t] = bs[t + 1];
        bi[t] = bi[t + 1];
      }
      bs[M - 1] = NEG_INF;
      bi[M - 1] = 0x7fffffff;
      popped++;
    }
    // a lane that ran dry may hide the next best candidate (it kept only its M best)
    if (it + 1 < n_rec && __any(popped == M)) {
      if (ln == 0) p.todo[row] = 1;
      return;
    }
  }
  wave_metrics_tail(p, orow, ln, mi, n_rec, gb, ge, n_gt, gt_pref, disc_pref, idcg_pref, res, rec_row);
}

// First level of the sum for large calls: workgroup b folds the rows [b * chunk, (b + 1) * chunk)
// into one RowOut-like partial with the SAME code as reduce_rows_kernel, so that the second
// level (reduce over the partials) keeps a fixed order.  See reduce_rows_kernel.
struct RowPartial {
  long long valid;
  double hit, recall, ndcg, precision, map;
};

// Sum of the per-user terms of one call (one 1024-thread workgroup) in a fixed order, so the
// fp64 result is reproducible run to run; it is not the strict user order of a
// single-threaded reference run (the reference's own order depends on n_threads,
// evaluator.cpp:280-312).
__global__ __launch_bounds__(1024) void reduce_rows_kernel(const RowOut *rows, int64_t n,
                                                           irs_metrics *out,
                                                           RowPartial *partials = nullptr,
                                                           int64_t chunk = 0) {
  // thread t adds rows t, t + 1024, ... in that order; the 64 lanes of a wave fold by a fixed
  // butterfly and wave 0 adds the 16 wave sums in wave order.  With `partials` workgroup b does
  // that for its chunk of rows only and leaves the sums there (reduce_partials_kernel adds the
  // chunks in order): one workgroup walking 138 k rows alone took 0.11 ms.
  __shared__ double part[16][5];
  __shared__ long long part_valid[16];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  long long valid = 0;
  double hit = 0, recall = 0, ndcg = 0, precision = 0, map = 0;
  if (partials) {
    rows += static_cast<int64_t>(blockIdx.x) * chunk;
    n = min(chunk, n - static_cast<int64_t>(blockIdx.x) * chunk);
  }
  for (int64_t i = tid; i < n; i += 1024) {
    const RowOut r = rows[i];
    if (r.valid) {
      valid += 1;
      hit += r.hit;
      recall += r.recall;
      ndcg += r.ndcg;
      precision += r.precision;
      map += r.map;
    }
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) {
    valid += __shfl_xor(valid, o, 64);
    hit += __shfl_xor(hit, o, 64);
    recall += __shfl_xor(recall, o, 64);
    ndcg += __shfl_xor(ndcg, o, 64);
    precision += __shfl_xor(precision, o, 64);
    map += __shfl_xor(map, o, 64);
  }
  if (lane == 0) {
    part_valid[wv] = valid;
    part[wv][0] = hit;
    part[wv][1] = recall;
    part[wv][2] = ndcg;
    part[wv][3] = precision;
    part[wv][4] = map;
  }
  __syncthreads();
  if (tid == 0) {
    long long v = 0;
    double acc[5] = {0, 0, 0, 0, 0};
    for (int w = 0; w < 16; w++) {
      v += part_valid[w];
      for (int c = 0; c < 5; c++) acc[c] += part[w][c];
    }
    if (partials) {
      partials[blockIdx.x] = RowPartial{v, acc[0], acc[1], acc[2], acc[3], acc[4]};
      return;
    }
    out->valid_user += v;
    out->total_user += n;
    out->hit += acc[0];
    out->recall += acc[1];
    out->ndcg += acc[2];
    out->precision += acc[3];
    out->map += acc[4];
  }
}

__global__ void reduce_partials_kernel(const RowPartial *partials, int nb, int64_t n, irs_metrics *out) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  long long v = 0;
  double acc[5] = {0, 0, 0, 0, 0};
  for (int b = 0; b < nb; b++) {  // chunk order
    v += partials[b].valid;
    acc[0] += partials[b].hit;
    acc[1] += partials[b].recall;
    acc[2] += partials[b].ndcg;
    acc[3] += partials[b].precision;
    acc[4] += partials[b].map;
  }
  out->valid_user += v;
  out->total_user += n;
  out->hit += acc[0];
  out->recall += acc[1];
  out->ndcg += acc[2];
  out->precision += acc[3];
  out->map += acc[4];
}

// item_cnt[i] += how often item i stands in the recommended lists of a call (Metrics::update,
// evaluator.cpp:146).  `rec` = the lists the ranking kernels wrote ([rows, cutoff], -1 where
// there is no entry).  Everybody's list is full of the same popular items, so one global atomic
// per entry queues up on a few addresses (1 ms for 2.7 M entries); each workgroup counts its
// slice in an LDS table first (direct-mapped, a colliding item goes to memory at once) and adds
// one total per item it met.
constexpr int HIST_SLOTS = 8192;
__global__ __launch_bounds__(1024) void item_hist_kernel(const int32_t *__restrict__ rec, int64_t n,
                                                         unsigned long long *__restrict__ item_cnt) {
  __shared__ int32_t ids[HIST_SLOTS];
  __shared__ uint32_t cnt[HIST_SLOTS];
  for (int i = threadIdx.x; i < HIST_SLOTS; i += 1024) {
    ids[i] = -1;
    cnt[i] = 0;
  }
  __syncthreads();
  const int64_t per = (n + gridDim.x - 1) / gridDim.x;
  const int64_t b = per * blockIdx.x, e = min(b + per, n);
  for (int64_t i = b + threadIdx.x; i < e; i += 1024) {
    const int32_t it = rec[i];
    if (it < 0) continue;
    const uint32_t slot = (static_cast<uint32_t>(it) * 2654435761u) >> 19;  // 13 bits
    const int32_t prev = atomicCAS(&ids[slot], -1, it);
    if (prev == -1 || prev == it) atomicAdd(&cnt[slot], 1u);
    else atomicAdd(&item_cnt[it], 1ull);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < HIST_SLOTS; i += 1024)
    if (ids[i] >= 0 && cnt[i] > 0) atomicAdd(&item_cnt[ids[i]], static_cast<unsigned long long>(cnt[i]));
}

// Catalogues whose counters fit one CU's LDS (n_items <= 36,864; ML-20M: 26,744): a direct table per
// workgroup - one LDS atomic per entry, no tags, no collisions (the hashed table above sends every
// collision to a hot global address: 109 us for the 2.8 M entries of an ML-20M call, this form 12).
constexpr int64_t HIST_DIRECT_MAX = 36864;
__global__ __launch_bounds__(1024) void item_hist_direct_kernel(const int32_t *__restrict__ rec, int64_t n,
                                                                int32_t n_items,
                                                                unsigned long long *__restrict__ item_cnt) {
  extern __shared__ uint32_t hist_cnt[];
  for (int i = threadIdx.x; i < n_items; i += 1024) hist_cnt[i] = 0u;
  __syncthreads();
  const int64_t per = (n + gridDim.x - 1) / gridDim.x;
  const int64_t b = per * blockIdx.x, e = min(b + per, n);
  for (int64_t i = b + threadIdx.x; i < e; i += 1024) {
    const int32_t it = rec[i];
    if (it >= 0 && it < n_items) atomicAdd(&hist_cnt[it], 1u);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < n_items; i += 1024) {
    const uint32_t c = hist_cnt[i];
    if (c) atomicAdd(&item_cnt[i], static_cast<unsigned long long>(c));
  }
}

inline void launch_item_hist(const int32_t *rec, int64_t n, unsigned long long *item_cnt, hipStream_t s,
                             int64_t n_items = 0) {
  if (n <= 0) return;
  if (n_items > 0 && n_items <= HIST_DIRECT_MAX && n >= 65536) {
    const size_t lds = static_cast<size_t>(n_items) * sizeof(uint32_t);
    IRS_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(item_hist_direct_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize,
                                static_cast<int>(HIST_DIRECT_MAX * sizeof(uint32_t))));
    const int64_t grid = std::min<int64_t>(128, (n + 16383) / 16384);
    hipLaunchKernelGGL(item_hist_direct_kernel, dim3(static_cast<unsigned>(grid)), dim3(1024), lds, s, rec, n,
                       static_cast<int32_t>(n_items), item_cnt);
    return;
  }
  const int64_t grid = std::min<int64_t>(1024, (n + 16383) / 16384);
  hipLaunchKernelGGL(item_hist_kernel, dim3(static_cast<unsigned>(grid)), dim3(1024), 0, s, rec, n, item_cnt);
}

// scores[row, col] = -inf for the stored entries of the mask rows (evaluator.py:426-432)
// (`n_items` = row stride = number of leading items the block holds: entries beyond are skipped)
// (`row_list`, when given: score row r is masked with the mask row row_list[r])
__global__ void mask_rows_kernel(float *scores, int64_t rows, int64_t n_items,
                                 const int64_t *mask_ptr, const int32_t *mask_idx,
                                 const int32_t *row_list = nullptr) {
  const int64_t row = blockIdx.x;
  if (row >= rows) return;
  const int64_t mrow = row_list ? row_list[row] : row;
  for (int64_t q = mask_ptr[mrow] + threadIdx.x; q < mask_ptr[mrow + 1]; q += blockDim.x) {
    const int32_t j = mask_idx[q];
    if (j < n_items) scores[row * n_items + j] = -std::numeric_limits<float>::infinity();
  }
}

// the same for a block of either score type whose mask rows arrive with the call
// (irs_eval_get_metrics_masked): mask_ptr is relative to the first row of the block
template <class T>
__global__ void mask_block_kernel(T *scores, int64_t rows, int64_t n_items,
                                  const int64_t *mask_ptr, const int32_t *mask_idx) {
  const int64_t row = blockIdx.x;
  if (row >= rows) return;
  for (int64_t q = mask_ptr[row] + threadIdx.x; q < mask_ptr[row + 1]; q += blockDim.x) {
    const int32_t j = mask_idx[q];
    if (j >= 0 && j < n_items) scores[row * n_items + j] = -std::numeric_limits<T>::infinity();
  }
}


// ---------------------------------------------------------------------------------------------------------
// Scores of a SIMILARITY model on the device (round 6): score[u][:] = X[u][:] @ W
// (BaseSimilarityRecommender.get_score_block, base.py:406-429: `X_train_all[begin:end].dot(W)` through
// scipy's row-by-row sparse product).  One wave per (user, tile of SIM_TILE columns): the tile's float64
// sums live in the wave's own LDS slab; the wave walks the user's stored (i, x) in order and, for each,
// the stored entries (j, w) of row i of W, adding x * w to column j when it lies in the tile - product and
// sum rounded separately (__dmul_rn / __dadd_rn), entries of the profile in storage order.  That IS the
// order in which scipy's csr_matmat accumulates `sums[j] += x * w` for a result row, so the block is the
// host product bit for bit (a column gets at most one update per profile entry - W's rows hold distinct
// columns - and LDS operations of one wave execute in order: no atomics, no barriers, no dependence on
// arrival order).  When W's rows hold strictly increasing columns (every recommender of this package) a
// table made per call (sim_tile_ptr_kernel) gives each (row of W, tile) its range of entries, so a tile
// reads a row's entries INSIDE it - one strip, sixteen rows in flight; otherwise every tile re-scans the
// whole rows (two strips, eight rows in flight).
constexpr int SIM_TILE = 2048;  // columns per wave: 16 KB of LDS, ten waves per CU (4096: 65 ms for the ML-20M model, 2048: 59, 1024: 66)

__device__ __forceinline__ int64_t readlane_i64(int64_t v, int src) {
  const uint32_t lo = __builtin_amdgcn_readlane(static_cast<uint32_t>(v), src);
  const uint32_t hi = __builtin_amdgcn_readlane(static_cast<uint32_t>(static_cast<uint64_t>(v) >> 32), src);
  return static_cast<int64_t>((static_cast<uint64_t>(hi) << 32) | lo);
}
__device__ __forceinline__ double readlane_f64(double v, int src) {
  return __longlong_as_double(readlane_i64(__double_as_longlong(v), src));
}

// tptr[i * (n_tiles + 1) + t] = first entry of row i of W (columns increasing) whose column is >= t * SIM_TILE
__global__ __launch_bounds__(256) void sim_tile_ptr_kernel(const int64_t *__restrict__ w_ptr,
                                                           const int32_t *__restrict__ w_idx, int64_t n_rows,
                                                           int32_t n_tiles, int32_t *__restrict__ tptr) {
  const int64_t id = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (id >= n_rows * (n_tiles + 1)) return;
  const int64_t i = id / (n_tiles + 1);
  const int32_t t = static_cast<int32_t>(id % (n_tiles + 1));
  int64_t lo = w_ptr[i], hi = w_ptr[i + 1];
  const int64_t bound = static_cast<int64_t>(t) * SIM_TILE;
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if (w_idx[mid] < bound) lo = mid + 1;
    else hi = mid;
  }
  tptr[id] = static_cast<int32_t>(lo);
}

template <bool TILED>  // TILED: w_tptr given - a row's entries inside the tile are one strip (a second one is not fetched)
__global__ __launch_bounds__(64) void sim_score_kernel(const int64_t *__restrict__ x_ptr, const int32_t *__restrict__ x_idx,
                                                       const double *__restrict__ x_val,  // null: all ones
                                                       const int64_t *__restrict__ w_ptr, const int32_t *__restrict__ w_idx,
                                                       const double *__restrict__ w_val, int64_t w_last, int64_t row0,
                                                       int64_t n_items, int32_t n_tiles, double *__restrict__ out,
                                                       const int32_t *__restrict__ w_tptr,
                                                       const int32_t *__restrict__ order) {
  __shared__ double acc[SIM_TILE];
  const int lane = threadIdx.x;
  const int64_t unit = blockIdx.x;
  // row of the block: the rows are LAUNCHED longest profile first (`order`; a wave lasts as long as its
  // user's profile, and a 9,000-item profile at the end of a block was a tail of its own)
  const int64_t r = order[unit / n_tiles];
  const int32_t tile = static_cast<int32_t>(unit % n_tiles);
  const int32_t c0 = tile * SIM_TILE, width = static_cast<int32_t>(min<int64_t>(SIM_TILE, n_items - c0));
  for (int k = lane; k < width; k += 64) acc[k] = 0.0;
  const int64_t qb = x_ptr[row0 + r], qe = x_ptr[row0 + r + 1];
  auto add = [&](int32_t jg, double x, double w) {  // (in program order per wave: LDS operations do not overtake)
    const int32_t j = jg - c0;
    if (static_cast<uint32_t>(j) < static_cast<uint32_t>(width)) acc[j] = __dadd_rn(acc[j], __dmul_rn(x, w));
  };
  // The walk is a chain of dependent loads (profile entry -> row bounds of W -> the row's columns and values):
  // 64 profile entries are fetched at once (one per lane, with their row bounds), and the first 128 entries of
  // the rows of D consecutive profile entries are in flight while an earlier row is added - one exposed round
  // trip per 64 profile entries instead of three per entry (195 -> ~40 ms for the ML-20M model).
  constexpr int D = TILED ? 16 : 8;  // rows in flight (TILED: one strip per row, twice the rows)
  for (int64_t q0 = qb; q0 < qe; q0 += 64) {
    const int64_t q = min(q0 + lane, qe - 1);
    const int32_t i_l = x_idx[q];
    const double x_l = x_val ? x_val[q] : 1.0;
    // (rows of W with increasing columns: only the row's entries INSIDE this tile, from the table
    // sim_tile_ptr_kernel made - a seventh of a row on the ML-20M shape, one strip instead of two)
    int64_t eb_l, ee_l;
    if constexpr (TILED) {
      const int32_t *tp = w_tptr + static_cast<int64_t>(i_l) * (n_tiles + 1) + tile;
      eb_l = tp[0];
      ee_l = tp[1];
    } else {
      eb_l = w_ptr[i_l];
      ee_l = w_ptr[i_l + 1];
    }
    const int n = static_cast<int>(min<int64_t>(64, qe - q0));
    int32_t ja[D], jb[D];
    double wa[D], wb[D];
    auto fetch = [&](int slot, int k) {  // the first two strips of row k (clamped loads, masked when used)
      const int64_t eb = readlane_i64(eb_l, k), ee = readlane_i64(ee_l, k);
      // (unconditional: a load under a branch would drain the queue; w_last = the last valid entry, >= 0)
      const int64_t e0 = min(eb + lane, w_last), e1 = min(eb + 64 + lane, w_last);
      ja[slot] = w_idx[e0];
      wa[slot] = w_val[e0];
      if constexpr (!TILED) {
        jb[slot] = w_idx[e1];
        wb[slot] = w_val[e1];
      }
      (void)ee;
      (void)e1;
    };
#pragma unroll
    for (int d = 0; d < D; d++) fetch(d, min(d, n - 1));
    for (int k0 = 0; k0 < n; k0 += D) {
#pragma unroll
      for (int d = 0; d < D; d++) {
        const int k = k0 + d;
        if (k < n) {  // (wave-uniform)
          const int64_t eb = readlane_i64(eb_l, k), ee = readlane_i64(ee_l, k);
          const double x = readlane_f64(x_l, k);
          const int32_t j0 = ja[d], j1 = TILED ? 0 : jb[d];
          const double w0 = wa[d], w1 = TILED ? 0.0 : wb[d];
          if (k + D < n) fetch(d, k + D);  // (the slot's registers were copied: its next row starts now)
          if (eb + lane < ee) add(j0, x, w0);
          if constexpr (!TILED) {
            if (eb + 64 + lane < ee) add(j1, x, w1);
          }
          // (rows above 128 entries; TILED: more than 64 of a row's entries inside one tile)
          for (int64_t e = eb + (TILED ? 64 : 128) + lane; e < ee; e += 64) add(w_idx[e], x, w_val[e]);
        }
      }
    }
  }
  double *dst = out + r * n_items + c0;
  for (int k = lane; k < width; k += 64) dst[k] = acc[k];
}

}  // namespace eval
}  // namespace irs

using namespace irs;
using namespace irs::eval;

// exported by ials.hip for the fused path
extern "C" irs_status irs_ials_scores_device_(irs_ials_trainer *t, int64_t begin, int64_t end,
                                              float *device_out, void **stream_out,
                                              int32_t *device_index);
extern "C" irs_status irs_ials_scores_prefix_device_(irs_ials_trainer *t, int64_t begin, int64_t end,
                                                     int64_t n_prefix, const float *item_rows,
                                                     const float *user_rows, float *device_out);
extern "C" irs_status irs_ials_factors_device_(irs_ials_trainer *t, const float **user,
                                               const float **item, int32_t *KP, int64_t *n_users,
                                               int64_t *n_items, void **stream_out,
                                               int32_t *device_index);

struct irs_evaluator {
  int device = 0;
  int64_t n_users = 0, n_items = 0;
  int rec_mode = 0;
  DeviceBuffer<int32_t> gt_ptr, gt_idx, rec_items;
  DeviceBuffer<int64_t> rec_ptr;
  DeviceBuffer<double> disc, idcg_prefix;
  DeviceBuffer<RowOut> row_out;
  DeviceBuffer<int32_t> rec_out, todo;
  DeviceBuffer<unsigned long long> item_cnt;
  DeviceBuffer<irs_metrics> metrics;
  DeviceBuffer<char> score_buf;
  // mask of the fused path kept on the device between calls (irs_eval_cache_mask)
  DeviceBuffer<int64_t> mask_ptr;
  DeviceBuffer<int32_t> mask_idx;
  int64_t mask_rows = -1;
  DeviceBuffer<float> fused_scores;  // score block of the two-pass path, kept between calls
  // fused call (eval_fused_kernels.hpp): the cached mask as a bitmap + scratch of the emit path
  DeviceBuffer<uint64_t> mask_bits;
  DeviceBuffer<int32_t> mask_count;
  int64_t mask_bits_rows = -1;  // rows the bitmap was built for (-1: none)
  DeviceBuffer<float> cand_score, tau;
  DeviceBuffer<int32_t> cand_item, cand_cnt, bad_flag;
  // bounded variant of the emit path: norms, the two sort permutations, per-tile limits
  DeviceBuffer<float> inorm, inorm_sorted, unorm, radius, radius_sorted, sample_item, hard_user;
  DeviceBuffer<int32_t> iota, iperm, iinv, uperm, limit_tiles, hard, hard_list, wg_prefix;
  DeviceBuffer<int4> wg_desc;
  DeviceBuffer<char> sort_tmp;
  int32_t *flags_host = nullptr;  // page-locked: the emit path's flags / hard-row count / scored tiles, one copy per pass
  DeviceBuffer<RowPartial> row_partials;  // chunk sums of the two-level reduction
  DeviceBuffer<int32_t> mask_row;         // row of every mask entry (built with the bitmap)
  std::vector<int64_t> mask_ptr_host;     // host copy of the mask's row pointers
  irs_eval_stats stats{};  // of the last irs_eval_get_metrics_ials call
  hipEvent_t ev_first = nullptr, ev_last = nullptr;  // span of that call's device work
  bool span_open = false;
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
}

// cutoff > SEL_CAP: the lists of a launch's workgroups in global scratch (13 bytes per entry,
// big_cap entries per row), at most ~1 GiB of it: the rows go through in launches of
// `per` rows.  Rare (the reference's callers use cutoffs of 5..100), so the goal is only to be
// correct for every cutoff the reference accepts and not slower than its partial_sort.
template <class T> void launch_rank_big(EvalParams p, int64_t max_cand, hipStream_t s) {
  int64_t cap = 1;
  while (cap < p.cutoff) cap <<= 1;
  const int64_t per = std::max<int64_t>(1, std::min<int64_t>(p.rows, (int64_t(1) << 30) / (cap * 13)));
  DeviceBuffer<uint64_t> bk;
  DeviceBuffer<int32_t> bi;
  DeviceBuffer<uint8_t> bh;
  bk.alloc(static_cast<size_t>(per) * cap);
  bi.alloc(static_cast<size_t>(per) * cap);
  bh.alloc(static_cast<size_t>(per) * cap);
  p.big_key = bk.ptr;
  p.big_idx = bi.ptr;
  p.big_hit = bh.ptr;
  p.big_cap = static_cast<int32_t>(cap);
  p.todo = nullptr;
  const size_t key_bytes = static_cast<size_t>(std::max<int64_t>(max_cand, 1)) * sizeof(typename KeyStore<T>::type);
  for (int64_t b = 0; b < p.rows; b += per) {
    const unsigned m = static_cast<unsigned>(std::min<int64_t>(per, p.rows - b));
    p.row_base = b;
    if (key_bytes <= 128 * 1024) {
      auto kernel = rank_rows_kernel<T, 1024, 1, RANK_MAXQ, true>;
      IRS_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel),
                                  hipFuncAttributeMaxDynamicSharedMemorySize,
                                  static_cast<int>(key_bytes)));
      hipLaunchKernelGGL(kernel, dim3(m), dim3(1024), key_bytes, s, p);
    } else {
      hipLaunchKernelGGL((rank_rows_kernel<T, 256, 0, RANK_MAXQ, true>), dim3(m), dim3(256), 0, s, p);
    }
  }
  IRS_HIP(hipGetLastError());
  IRS_HIP(hipStreamSynchronize(s));  // the scratch is released on return
}

template <class T> void launch_rank(EvalParams p, int64_t max_cand, hipStream_t s, int32_t *todo) {
  if (p.cutoff > SEL_CAP) {
    launch_rank_big<T>(p, max_cand, s);
    return;
  }
  // the usual case first (all items are candidates, cutoff <= 64): one wave per row; the
  // rows it flags (and every row otherwise) go through the general kernel
  p.todo = nullptr;
  const bool wave_ok = [] {  // IRSPACK_AMD_EVAL_WAVE=0: general kernel only (read per call: tests toggle it)
    const char *e = std::getenv("IRSPACK_AMD_EVAL_WAVE");
    return !(e && e[0] == '0');
  }();
  if (wave_ok && p.rec_mode == 0 && p.cutoff <= 64 && p.n_items < (int64_t(1) << 31) - 64 * 16 &&
      todo != nullptr) {
    p.todo = todo;
    const dim3 grid(static_cast<unsigned>((p.rows + 3) / 4));
    if (p.cutoff <= 24)
      hipLaunchKernelGGL((rank_wave_kernel<T, 4>), grid, dim3(256), 0, s, p);
    else
      hipLaunchKernelGGL((rank_wave_kernel<T, 8>), grid, dim3(256), 0, s, p);
  }
  // key cache in LDS when a row's candidates fit next to the 28 KB of static LDS
  const size_t key_bytes = static_cast<size_t>(std::max<int64_t>(max_cand, 1)) * sizeof(typename KeyStore<T>::type);
  if (std::is_same<T, float>::value && max_cand <= 1024 * RANK_MAXQ) {
    hipLaunchKernelGGL((rank_rows_kernel<T, 1024, 2>), dim3(p.rows), dim3(1024), 0, s, p);
  } else if (key_bytes <= 128 * 1024) {
    auto kernel = rank_rows_kernel<T, 1024, 1>;
    IRS_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize,
                                static_cast<int>(key_bytes)));
    hipLaunchKernelGGL(kernel, dim3(p.rows), dim3(1024), key_bytes, s, p);
  } else {
    hipLaunchKernelGGL((rank_rows_kernel<T, 256, 0>), dim3(p.rows), dim3(256), 0, s, p);
  }
}

// ranks `rows` rows already resident at `d_scores` and accumulates into out / item_cnt
template <class T>
void rank_block(irs_evaluator *e, const void *d_scores, int64_t rows, int64_t cutoff,
                int64_t offset, bool rwc, hipStream_t s) {
  e->row_out.alloc(rows);
  e->rec_out.alloc(rows * cutoff);
  e->todo.alloc(rows);
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
  p.retrieve = 0;
  p.recall_with_cutoff = rwc ? 1 : 0;
  p.disc = e->disc.ptr;
  p.idcg_prefix = e->idcg_prefix.ptr;
  p.out = e->row_out.ptr;
  p.rec_out = e->rec_out.ptr;
  p.item_cnt = e->item_cnt.ptr;
  launch_rank<T>(p, e->n_items, s, e->todo.ptr);
  launch_item_hist(e->rec_out.ptr, rows * cutoff, e->item_cnt.ptr, s, e->n_items);
  hipLaunchKernelGGL(reduce_rows_kernel, dim3(1), dim3(1024), 0, s, e->row_out.ptr, rows,
                     e->metrics.ptr);
  IRS_HIP(hipGetLastError());
}

// Metrics::merge, evaluator.cpp:76-85: plain sums (the item histogram is merged by the caller)
static void merge_metrics(irs_metrics &into, const irs_metrics &part) {
  into.valid_user += part.valid_user;
  into.total_user += part.total_user;
  into.hit += part.hit;
  into.recall += part.recall;
  into.ndcg += part.ndcg;
  into.precision += part.precision;
  into.map += part.map;
}

// merge_metrics on the device (one thread: the same sums in the same order) and the item histogram beside it:
// a call that walks its users in blocks folds each block's result here instead of reading it back
__global__ void metrics_fold_kernel(const irs_metrics *__restrict__ part, irs_metrics *__restrict__ into) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  into->valid_user += part->valid_user;
  into->total_user += part->total_user;
  into->hit += part->hit;
  into->recall += part->recall;
  into->ndcg += part->ndcg;
  into->precision += part->precision;
  into->map += part->map;
}
__global__ void counts_fold_kernel(const unsigned long long *__restrict__ part, int64_t n,
                                   unsigned long long *__restrict__ into) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i < n) into[i] += part[i];
}

void begin_accumulate(irs_evaluator *e, hipStream_t s) {
  e->metrics.alloc(1);
  e->metrics.zero(s);
  e->item_cnt.zero(s);
}

void finish_accumulate(irs_evaluator *e, irs_metrics *out, int64_t *item_cnt, hipStream_t s) {
  static_assert(sizeof(unsigned long long) == sizeof(int64_t), "");
  if (e->span_open) IRS_HIP(hipEventRecord(e->ev_last, s));
  IRS_HIP(hipMemcpyAsync(out, e->metrics.ptr, sizeof(irs_metrics), hipMemcpyDeviceToHost, s));
  IRS_HIP(hipMemcpyAsync(item_cnt, e->item_cnt.ptr, e->n_items * sizeof(int64_t),
                         hipMemcpyDeviceToHost, s));
  IRS_HIP(hipStreamSynchronize(s));
}


bool ensure_mask_bitmap(irs_evaluator *e, int64_t rows, int64_t words, const int64_t *d_mptr,
                        const int32_t *d_midx, hipStream_t s, bool cacheable = true);

// IRSPACK_AMD_EVAL_EMIT=0 switches the threshold-filtered path off (A/B against the two-pass one).
bool emit_enabled() {
  const char *e = std::getenv("IRSPACK_AMD_EVAL_EMIT");
  return e ? std::atoi(e) != 0 : true;
}

int device_cu_count(int device) {
  static std::atomic<int> cached[64] = {};
  const int slot = device >= 0 && device < 64 ? device : 0;
  int n = cached[slot].load(std::memory_order_relaxed);
  if (n <= 0) {
    IRS_HIP(hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, device));
    n = std::max(n, 1);
    cached[slot].store(n, std::memory_order_relaxed);
  }
  return n;
}

// IRSPACK_AMD_EVAL_SAMPLE_FUSED=0: the sample pass as three launches per 32,768 users (round 5).
bool sample_fused_enabled() {
  const char *e = std::getenv("IRSPACK_AMD_EVAL_SAMPLE_FUSED");
  return e ? std::atoi(e) != 0 : true;
}

// IRSPACK_AMD_EVAL_BOUND=0: the emit path scores every tile (no norm-bound pruning).
bool bound_enabled() {
  const char *e = std::getenv("IRSPACK_AMD_EVAL_BOUND");
  return e ? std::atoi(e) != 0 : true;
}

// Builds (or reuses) the bitmap + per-row count of the mask CSR; false when it would not fit.
bool ensure_mask_bitmap(irs_evaluator *e, int64_t rows, int64_t words, const int64_t *d_mptr,
                        const int32_t *d_midx, hipStream_t s, bool cacheable) {
  if (static_cast<double>(rows) * words * 8.0 > 2147483648.0) return false;
  const bool cached = cacheable && d_mptr == e->mask_ptr.ptr && e->mask_bits_rows == rows;
  if (!cached) {
    e->mask_bits.alloc(static_cast<size_t>(rows) * words);
    e->mask_count.alloc(rows);
    IRS_HIP(hipMemsetAsync(e->mask_bits.ptr, 0, static_cast<size_t>(rows) * words * 8, s));
    hipLaunchKernelGGL(mask_bitmap_kernel, dim3(ceil_div(rows, 4)), dim3(256), 0, s, d_mptr, d_midx,
                       rows, words, e->mask_bits.ptr);
    hipLaunchKernelGGL(mask_count_kernel, dim3(ceil_div(rows, 4)), dim3(256), 0, s, e->mask_bits.ptr,
                       rows, words, e->mask_count.ptr);
    // row ids per entry + the row pointers on the host (entry ranges of the sample blocks)
    e->mask_ptr_host.resize(rows + 1);
    IRS_HIP(hipMemcpyAsync(e->mask_ptr_host.data(), d_mptr, (rows + 1) * sizeof(int64_t),
                           hipMemcpyDeviceToHost, s));
    IRS_HIP(hipStreamSynchronize(s));
    e->mask_row.alloc(static_cast<size_t>(std::max<int64_t>(e->mask_ptr_host[rows], 1)));
    hipLaunchKernelGGL(mask_row_ids_kernel, dim3(ceil_div(rows, 4)), dim3(256), 0, s, d_mptr, rows,
                       e->mask_row.ptr);
    // only the resident mask, covered by ONE pass, is cached
    e->mask_bits_rows = (cacheable && d_mptr == e->mask_ptr.ptr) ? rows : -1;
  }
  return true;
}

// The threshold-filtered path of irs_eval_get_metrics_ials (eval_fused_kernels.hpp, second
// half).  Returns false when the call is outside its domain or had to be abandoned; the caller
// then runs the two-pass path.
// One pass of the path over the users [begin, begin + rows): `first` = the first pass of a call
// (it prepares the item side: norms, order, sample rows), `single` = the only one (the mask
// bitmap of a resident mask may then be kept for the next call).
bool emit_block(irs_evaluator *e, irs_ials_trainer *t, int64_t begin, int64_t rows,
                const int64_t *d_mptr, const int32_t *d_midx, int64_t cutoff, int64_t offset,
                bool rwc, hipStream_t s, bool first, bool single) {
  if (!emit_enabled() || e->rec_mode != 0 || cutoff > FZ_MAX_CUTOFF || rows <= 0) return false;
  const float *user = nullptr, *item = nullptr;
  int32_t KP = 0, dev = 0;
  int64_t nu = 0, ni = 0;
  void *sv = nullptr;
  if (irs_ials_factors_device_(t, &user, &item, &KP, &nu, &ni, &sv, &dev) != IRS_OK)
    throw std::runtime_error(irs_last_error());
  if (KP > 256) return false;  // K > 256: the two-pass path (run-time-sized scoring kernel)
  // worth it only when the sample is a small part of the catalogue
  if (ni != e->n_items || ni < 4 * EM_SAMPLE || rows > (int64_t(1) << 31) / EM_CAP) return false;
  const bool bounded = bound_enabled() && ni < (int64_t(1) << 31) && rows < (int64_t(1) << 31);
  // items of the sample pass.  Sorted by norm (bounded variant) a few hundred items already
  // hold nearly every user's threshold; IRSPACK_AMD_EVAL_SAMPLE overrides (multiples of 64).
  // The users it leaves without one (they have seen almost all of the sample) get a second
  // chance on EM_SAMPLE2 items.
  const int64_t env_sample = [] {  // (read per call: tests toggle it)
    const char *v = std::getenv("IRSPACK_AMD_EVAL_SAMPLE");
    return v ? std::max<int64_t>(64, std::atoll(v) / 64 * 64) : int64_t(0);
  }();
  const int64_t n_sample = std::min<int64_t>(env_sample ? env_sample : (bounded ? EM_SAMPLE_SORTED : EM_SAMPLE),
                                             EM_SAMPLE2);
  const int64_t words = ceil_div(ni, 64);
  const uint64_t *bits = nullptr;
  const int32_t *n_masked = nullptr;
  if (d_mptr) {
    if (!ensure_mask_bitmap(e, rows, words, d_mptr, d_midx, s, single)) return false;
    bits = e->mask_bits.ptr;
    n_masked = e->mask_count.ptr;
  }
  // [0] flags, [1] hard rows at the end, [2] hard rows after the sample pass, [3] workgroups
  // ([4..5]: the scored-tile count, 64 bits - one block, so that the call's single read-back is ONE copy, into
  // page-locked memory: two pageable copies and their staging cost ~50 us of a 1.2 ms call)
  e->bad_flag.alloc(8);
  IRS_HIP(hipMemsetAsync(e->bad_flag.ptr, 0, 8 * sizeof(int32_t), s));
  unsigned long long *const tiles_scored_dev = reinterpret_cast<unsigned long long *>(e->bad_flag.ptr + 4);
  if (!e->flags_host) IRS_HIP(hipHostMalloc(reinterpret_cast<void **>(&e->flags_host), 8 * sizeof(int32_t), hipHostMallocDefault));
  e->hard.alloc(rows);
  e->hard_list.alloc(rows);
  IRS_HIP(hipMemsetAsync(e->hard.ptr, 0, rows * sizeof(int32_t), s));
  // ---- 0. bounded variant: items in order of decreasing norm (the sample is then the items
  //         of largest norm, which hold most of every user's final list)
  const float norm_c = 1.0f + (KP + 16) * 2.5e-7f;
  if (bounded && !first) {  // (the item side is ready; iota must still cover this pass's rows)
    e->iota.alloc(std::max(ni, rows));
    hipLaunchKernelGGL(iota_kernel, dim3(ceil_div(std::max(ni, rows), 256)), dim3(256), 0, s,
                       e->iota.ptr, std::max(ni, rows));
  }
  if (bounded && first) {
    e->iota.alloc(std::max(ni, rows));
    hipLaunchKernelGGL(iota_kernel, dim3(ceil_div(std::max(ni, rows), 256)), dim3(256), 0, s,
                       e->iota.ptr, std::max(ni, rows));
    e->inorm.alloc(ni);
    e->inorm_sorted.alloc(ni);
    e->iperm.alloc(ni);
    e->iinv.alloc(ni);
    hipLaunchKernelGGL(row_norm_up_kernel, dim3(ceil_div(ni, 16)), dim3(256), 0, s, item, int64_t(0),
                       ni, KP, norm_c, e->inorm.ptr, e->bad_flag.ptr);
    sort_pairs_f32(true, e->inorm.ptr, e->inorm_sorted.ptr, e->iota.ptr, e->iperm.ptr, ni,
                   e->sort_tmp, s);
    hipLaunchKernelGGL(inverse_perm_kernel, dim3(ceil_div(ni, 256)), dim3(256), 0, s, e->iperm.ptr, ni,
                       e->iinv.ptr);
    e->sample_item.alloc(static_cast<size_t>(EM_SAMPLE2) * KP);
    hipLaunchKernelGGL(gather_rows_kernel, dim3(ceil_div(int64_t(EM_SAMPLE2) * (KP / 4), 256)), dim3(256),
                       0, s, item, e->iperm.ptr, static_cast<int64_t>(EM_SAMPLE2), KP,
                       e->sample_item.ptr, static_cast<const int32_t *>(nullptr));
  }
  // ---- 1. sample pass: thresholds from the first n_sample items, in blocks of users
  e->tau.alloc(rows);
  {
    const int64_t SB = 32768;  // users per sample block (<= 256 MB of scores)
    const int64_t HCAP = std::min<int64_t>(EM_HARD_CAP, rows);
    e->fused_scores.alloc(std::max(static_cast<size_t>(std::min(SB, rows)) * n_sample,
                                   static_cast<size_t>(HCAP) * EM_SAMPLE2));
    const int32_t *no_list = nullptr;
    // the whole pass in one launch (sample_tau_fused_kernel); IRSPACK_AMD_EVAL_SAMPLE_FUSED=0: the three
    // launches per block of users below (A/B)
    const bool fused_sample = bounded && n_sample == SF_ITEMS && (KP == 16 || KP == 32 || KP == 64 || KP == 128) &&
                              sample_fused_enabled();
    if (fused_sample) {
      SampleParams sp{};
      sp.user = user;
      sp.sample = e->sample_item.ptr;
      sp.begin = begin;
      sp.rows = rows;
      sp.mask_ptr = d_mptr;
      sp.mask_idx = d_midx;
      sp.iinv = e->iinv.ptr;
      sp.cutoff = static_cast<int32_t>(cutoff);
      sp.tau = e->tau.ptr;
      sp.hard = e->hard.ptr;
      sp.bad_flag = e->bad_flag.ptr;
      auto launch = [&](auto kernel, int per_cu) {
        IRS_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(SF_LDS_BYTES)));
        const int64_t grid = std::min<int64_t>(ceil_div(rows, SF_USERS), int64_t(per_cu) * device_cu_count(dev));
        hipLaunchKernelGGL(kernel, dim3(static_cast<unsigned>(grid)), dim3(512), SF_LDS_BYTES, s, sp);
      };
      switch (KP) {
        case 16: launch(sample_tau_fused_kernel<16>, 2); break;
        case 32: launch(sample_tau_fused_kernel<32>, 2); break;
        case 64: launch(sample_tau_fused_kernel<64>, 2); break;
        default: launch(sample_tau_fused_kernel<128>, 1); break;
      }
    }
    for (int64_t b = fused_sample ? rows : 0; b < rows; b += SB) {
      const int64_t m = std::min(SB, rows - b);
      if (irs_ials_scores_prefix_device_(t, begin + b, begin + b + m, n_sample,
                                         bounded ? e->sample_item.ptr : nullptr, nullptr,
                                         e->fused_scores.ptr) != IRS_OK)
        throw std::runtime_error(irs_last_error());
      if (d_mptr && bounded) {
        const int64_t q0 = e->mask_ptr_host[b], q1 = e->mask_ptr_host[b + m];
        if (q1 > q0)
          hipLaunchKernelGGL(mask_entries_perm_kernel, dim3(ceil_div(q1 - q0, 256)), dim3(256), 0, s,
                             e->fused_scores.ptr, n_sample, b, q0, q1, e->mask_row.ptr, d_midx,
                             e->iinv.ptr);
      } else if (d_mptr)
        hipLaunchKernelGGL(mask_rows_kernel, dim3(m), dim3(64), 0, s, e->fused_scores.ptr, m,
                           n_sample, d_mptr + b, d_midx);
      hipLaunchKernelGGL((sample_tau_kernel<8>), dim3(static_cast<unsigned>(ceil_div(m, 4))), dim3(256),
                         0, s, e->fused_scores.ptr, m, n_sample,
                         static_cast<int32_t>(cutoff), e->tau.ptr + b, e->bad_flag.ptr,
                         e->hard.ptr + b, no_list, no_list);
    }
    if (bounded && n_sample < EM_SAMPLE2) {
      // second chance: up to HCAP hard rows against the EM_SAMPLE2 items of largest norm (no
      // host round trip: the kernels run on HCAP rows and stop at the device-side count)
      int32_t *n_hard1 = e->bad_flag.ptr + 2;
      hipLaunchKernelGGL(collect_hard_kernel, dim3(ceil_div(rows, 256)), dim3(256), 0, s, e->hard.ptr,
                         e->gt_ptr.ptr, offset, rows, e->hard_list.ptr, n_hard1,
                         static_cast<int32_t>(HCAP));
      e->hard_user.alloc(static_cast<size_t>(HCAP) * KP);
      hipLaunchKernelGGL(gather_rows_kernel, dim3(ceil_div(HCAP * (KP / 4), 256)), dim3(256), 0, s,
                         user + begin * KP, e->hard_list.ptr, HCAP, KP, e->hard_user.ptr,
                         static_cast<const int32_t *>(n_hard1));
      if (irs_ials_scores_prefix_device_(t, 0, HCAP, EM_SAMPLE2, e->sample_item.ptr, e->hard_user.ptr,
                                         e->fused_scores.ptr) != IRS_OK)
        throw std::runtime_error(irs_last_error());
      if (d_mptr)
        hipLaunchKernelGGL(mask_rows_perm_kernel, dim3(HCAP), dim3(64), 0, s, e->fused_scores.ptr, HCAP,
                           static_cast<int64_t>(EM_SAMPLE2), d_mptr, d_midx, e->iinv.ptr,
                           static_cast<const int32_t *>(e->hard_list.ptr), static_cast<const int32_t *>(n_hard1));
      hipLaunchKernelGGL((sample_tau_kernel<8>), dim3(static_cast<unsigned>(ceil_div(HCAP, 4))), dim3(256),
                         0, s, e->fused_scores.ptr, HCAP, static_cast<int64_t>(EM_SAMPLE2),
                         static_cast<int32_t>(cutoff), e->tau.ptr, e->bad_flag.ptr, e->hard.ptr,
                         static_cast<const int32_t *>(e->hard_list.ptr), static_cast<const int32_t *>(n_hard1));
    }
  }
  // ---- 2. the whole score matrix, candidates only
  e->cand_score.alloc(static_cast<size_t>(rows) * EM_CAP);
  e->cand_item.alloc(static_cast<size_t>(rows) * EM_CAP);
  e->cand_cnt.alloc(rows);
  IRS_HIP(hipMemsetAsync(e->cand_cnt.ptr, 0, rows * sizeof(int32_t), s));
  EmitParams f;
  f.user = user;
  f.item = item;
  f.begin = begin;
  f.rows = rows;
  f.n_items = ni;
  f.mask_bits = bits;
  f.words = words;
  f.tau = e->tau.ptr;
  f.cand_score = e->cand_score.ptr;
  f.cand_item = e->cand_item.ptr;
  f.cand_cnt = e->cand_cnt.ptr;
  f.bad_flag = e->bad_flag.ptr;
  f.iperm = f.uperm = f.limit_tiles = nullptr;
  f.wg_desc = nullptr;
  f.n_wg = nullptr;
  f.hard = e->hard.ptr;
  int32_t n_wg = 0;
  if (bounded) {
    // users in order of increasing pruning radius, and what each 64-user tile still needs
    e->unorm.alloc(rows);
    e->radius.alloc(rows);
    e->radius_sorted.alloc(rows);
    e->uperm.alloc(rows);
    e->limit_tiles.alloc(ceil_div(rows, 64));
    hipLaunchKernelGGL(row_norm_up_kernel, dim3(ceil_div(rows, 16)), dim3(256), 0, s, user, begin, rows,
                       KP, norm_c, e->unorm.ptr, e->bad_flag.ptr);
    hipLaunchKernelGGL(prune_radius_kernel, dim3(ceil_div(rows, 256)), dim3(256), 0, s, e->tau.ptr,
                       e->unorm.ptr, e->gt_ptr.ptr, offset, rows, e->hard.ptr, e->radius.ptr);
    sort_pairs_f32(false, e->radius.ptr, e->radius_sorted.ptr, e->iota.ptr, e->uperm.ptr, rows,
                   e->sort_tmp, s);
    hipLaunchKernelGGL(tile_limit_kernel, dim3(ceil_div(ceil_div(rows, 64), 256)), dim3(256), 0, s,
                       e->radius_sorted.ptr, rows, e->inorm_sorted.ptr, ni, e->limit_tiles.ptr,
                       tiles_scored_dev);
    f.iperm = e->iperm.ptr;
    f.uperm = e->uperm.ptr;
    f.limit_tiles = e->limit_tiles.ptr;
    // the work list: only workgroups with a live tile are launched (the others would still
    // queue for a workgroup slot and its LDS just to leave); its length comes back to the host
    const int64_t n_ut = ceil_div(rows, 64);
    e->wg_prefix.alloc(n_ut + 1);
    hipLaunchKernelGGL(wg_scan_kernel, dim3(1), dim3(1024), 0, s, e->limit_tiles.ptr, n_ut,
                       e->wg_prefix.ptr, e->bad_flag.ptr + 3);
    // at most ceil(item tiles / 4) entries per user tile.  Up to 2^24 entries (64 MB) the list is sized for
    // that bound and its length stays on the device (score_emit_kernel walks it with a grid stride);
    // beyond, the length comes back first
    const int64_t wg_bound = n_ut * ceil_div(ceil_div(ni, 64), 4);
    if (wg_bound <= (int64_t(1) << 24)) {
      n_wg = static_cast<int32_t>(std::min<int64_t>(wg_bound, 32768));  // the grid
      e->wg_desc.alloc(std::max<int64_t>(wg_bound, 1));
      f.n_wg = e->bad_flag.ptr + 3;
    } else {
      IRS_HIP(hipMemcpyAsync(&n_wg, e->bad_flag.ptr + 3, sizeof(int32_t), hipMemcpyDeviceToHost, s));
      IRS_HIP(hipStreamSynchronize(s));
      e->wg_desc.alloc(std::max<int64_t>(n_wg, 1));
    }
    hipLaunchKernelGGL(wg_fill_kernel, dim3(ceil_div(n_ut, 4)), dim3(256), 0, s, e->wg_prefix.ptr,
                       e->limit_tiles.ptr, n_ut, e->wg_desc.ptr);
    f.wg_desc = e->wg_desc.ptr;
  }
  if (!bounded || n_wg > 0) {
    const int64_t tiles = bounded ? int64_t(n_wg) * 4 : ceil_div(rows, 64) * ceil_div(ni, 64);
    const size_t lds = 4 * 64 * FZ_SROW * sizeof(float) + 4 * 64 * sizeof(int32_t);
    auto launch = [&](auto kernel) {
      IRS_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
      hipLaunchKernelGGL(kernel, dim3(static_cast<unsigned>(ceil_div(tiles, 4))), dim3(256), lds, s, f);
    };
#define IRS_EMIT_CASE(KK)                                        \
  case KK:                                                       \
    if (bounded) launch(score_emit_kernel<KK, true>);            \
    else launch(score_emit_kernel<KK, false>);                   \
    break;
    switch (KP) {
      IRS_EMIT_CASE(16)
      IRS_EMIT_CASE(32)
      IRS_EMIT_CASE(64)
      IRS_EMIT_CASE(128)
      IRS_EMIT_CASE(192)
      IRS_EMIT_CASE(256)
      default: return false;
    }
#undef IRS_EMIT_CASE
  }
  // ---- 3. rank the candidates, metrics
  e->row_out.alloc(rows);
  e->rec_out.alloc(rows * cutoff);
  e->todo.alloc(rows);
  EvalParams p;
  p.scores = nullptr;
  p.rows = rows;
  p.n_items = e->n_items;
  p.offset = offset;
  p.gt_ptr = e->gt_ptr.ptr;
  p.gt_idx = e->gt_idx.ptr;
  p.rec_mode = 0;
  p.rec_ptr = e->rec_ptr.ptr;
  p.rec_items = e->rec_items.ptr;
  p.cutoff = static_cast<int32_t>(cutoff);
  p.retrieve = 0;
  p.recall_with_cutoff = rwc ? 1 : 0;
  p.disc = e->disc.ptr;
  p.idcg_prefix = e->idcg_prefix.ptr;
  p.out = e->row_out.ptr;
  p.rec_out = e->rec_out.ptr;
  p.item_cnt = e->item_cnt.ptr;
  p.todo = e->todo.ptr;
  const dim3 grid(static_cast<unsigned>(ceil_div(rows, 4)));
  hipLaunchKernelGGL((rank_cand_kernel<8>), grid, dim3(256), 0, s, p, f, n_masked);
  hipLaunchKernelGGL(rank_cand_slow_kernel, grid, dim3(256), 0, s, p, f, n_masked);
  hipLaunchKernelGGL(collect_hard_kernel, dim3(ceil_div(rows, 256)), dim3(256), 0, s, e->hard.ptr,
                     e->gt_ptr.ptr, offset, rows, e->hard_list.ptr, e->bad_flag.ptr + 1,
                     static_cast<int32_t>(rows));
  IRS_HIP(hipGetLastError());
  IRS_HIP(hipMemcpyAsync(e->flags_host, e->bad_flag.ptr, 8 * sizeof(int32_t), hipMemcpyDeviceToHost, s));
  IRS_HIP(hipStreamSynchronize(s));
  const int32_t bad[3] = {e->flags_host[0], e->flags_host[1], e->flags_host[2]};
  unsigned long long tiles_scored = 0;
  std::memcpy(&tiles_scored, e->flags_host + 4, sizeof(tiles_scored));
  {
    const int64_t total = ceil_div(rows, 64) * ceil_div(ni, 64);
    const int64_t scored = bounded ? static_cast<int64_t>(tiles_scored) : total;
    if (first) e->stats = irs_eval_stats{bounded ? 2 : 1, 0, 0, 0, n_sample};
    e->stats.hard_rows += bad[1];
    e->stats.tiles_total += total;
    e->stats.tiles_scored += scored;
  }
  // non-finite scores (the two-pass path defines the order of NaN), or so many hard rows that
  // one by one is the slower way: the caller runs the two-pass path
  if (bad[0] || bad[1] > 1024) return false;
  if (bad[1] > 0) {
    // the hard rows from their full score rows, in chunks of <= 1 GB of scores: gather the
    // user factors -> scores -> mask -> the general ranking, all through the row list
    const int64_t n_hard = bad[1];
    const int64_t chunk = std::max<int64_t>(1, std::min<int64_t>(n_hard, (int64_t(1) << 28) / ni));
    e->hard_user.alloc(static_cast<size_t>(chunk) * KP);
    e->fused_scores.alloc(static_cast<size_t>(chunk) * ni);
    for (int64_t b = 0; b < n_hard; b += chunk) {
      const int64_t m = std::min(chunk, n_hard - b);
      const int32_t *list = e->hard_list.ptr + b;
      hipLaunchKernelGGL(gather_rows_kernel, dim3(ceil_div(m * (KP / 4), 256)), dim3(256), 0, s,
                         user + begin * KP, list, m, KP, e->hard_user.ptr,
                         static_cast<const int32_t *>(nullptr));
      if (irs_ials_scores_prefix_device_(t, 0, m, ni, nullptr, e->hard_user.ptr, e->fused_scores.ptr) != IRS_OK)
        throw std::runtime_error(irs_last_error());
      if (d_mptr)
        hipLaunchKernelGGL(mask_rows_kernel, dim3(m), dim3(256), 0, s, e->fused_scores.ptr, m, ni, d_mptr,
                           d_midx, list);
      EvalParams q = p;
      q.scores = e->fused_scores.ptr;
      q.rows = m;
      q.row_map = list;
      // (no `todo` scratch = the general kernel for every row: a few dozen rows leave the one-wave-per-row
      // kernel latency bound - 51 us for 76 rows of an ML-20M call, the 1024-thread form takes 12)
      launch_rank<float>(q, ni, s, n_hard >= 2048 ? e->todo.ptr : nullptr);
    }
    IRS_HIP(hipGetLastError());
  }
  launch_item_hist(e->rec_out.ptr, rows * cutoff, e->item_cnt.ptr, s, e->n_items);
  if (rows >= 32768) {  // two levels: 4096-row chunks, then the chunks in order
    const int64_t chunk = 4096;
    const int nb = static_cast<int>(ceil_div(rows, chunk));
    e->row_partials.alloc(nb);
    hipLaunchKernelGGL(reduce_rows_kernel, dim3(nb), dim3(1024), 0, s, e->row_out.ptr, rows,
                       e->metrics.ptr, e->row_partials.ptr, chunk);
    hipLaunchKernelGGL(reduce_partials_kernel, dim3(1), dim3(64), 0, s, e->row_partials.ptr, nb, rows,
                       e->metrics.ptr);
  } else {
    hipLaunchKernelGGL(reduce_rows_kernel, dim3(1), dim3(1024), 0, s, e->row_out.ptr, rows,
                       e->metrics.ptr, static_cast<RowPartial *>(nullptr), int64_t(0));
  }
  IRS_HIP(hipGetLastError());
  return true;
}


// The threshold-filtered path over all users of a call, in passes that keep the scratch bounded:
// the candidate lists take 8 KB per user and the mask bitmap n_items / 8 bytes per user (at most
// 2 GiB per pass), so a call over millions of users or a catalogue of a million items runs as
// several passes (IRSPACK_AMD_EVAL_PASS_ROWS overrides the pass size).  False: outside the
// path's domain or abandoned - the caller resets the sums and runs the two-pass path.
bool emit_path(irs_evaluator *e, irs_ials_trainer *t, int64_t begin, int64_t rows,
               const int64_t *d_mptr, const int32_t *d_midx, int64_t cutoff, int64_t offset,
               bool rwc, hipStream_t s) {
  if (rows <= 0) return false;
  const char *env_v = std::getenv("IRSPACK_AMD_EVAL_PASS_ROWS");
  const int64_t env_rows = env_v ? std::max<int64_t>(64, std::atoll(env_v) / 64 * 64) : int64_t(0);
  const int64_t words = ceil_div(std::max<int64_t>(e->n_items, 1), 64);
  int64_t pass = std::min<int64_t>(524288, (int64_t(1) << 31) / (words * 8) / 64 * 64);
  if (env_rows) pass = std::min(pass, env_rows);
  if (pass < 64) return false;
  if (rows <= pass) return emit_block(e, t, begin, rows, d_mptr, d_midx, cutoff, offset, rwc, s, true, true);
  for (int64_t b = 0; b < rows; b += pass) {
    const int64_t m = std::min(pass, rows - b);
    if (!emit_block(e, t, begin + b, m, d_mptr ? d_mptr + b : nullptr, d_midx, cutoff, offset + b, rwc,
                    s, b == 0, false))
      return false;
  }
  return true;
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
    const int64_t nd = std::max<int64_t>(n_items, 1);  // any cutoff up to n_items (:263-268)
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

irs_status irs_eval_last_stats(irs_evaluator *e, irs_eval_stats *out) {
  return guard([&] {
    check_arg(e && out, "null argument.");
    *out = e->stats;
  });
}

irs_status irs_eval_destroy(irs_evaluator *e) {
  return guard([&] {
    if (e) {
      (void)hipSetDevice(e->device);
      if (e->ev_first) (void)hipEventDestroy(e->ev_first);
      if (e->ev_last) (void)hipEventDestroy(e->ev_last);
      if (e->flags_host) (void)hipHostFree(e->flags_host);
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

irs_status irs_eval_get_metrics_masked(irs_evaluator *e, int32_t is_f64, const void *scores,
                                       int64_t rows, const int64_t *mask_indptr,
                                       const int32_t *mask_indices, int32_t n_cutoffs,
                                       const int64_t *cutoffs, int64_t offset,
                                       int64_t n_threads, int32_t recall_with_cutoff,
                                       irs_metrics *out, int64_t *item_cnt) {
  return guard([&] {
    check_arg(e && out && item_cnt && cutoffs, "null argument.");
    check_arg(rows >= 0 && n_cutoffs >= 0, "negative count.");
    check_arg(rows == 0 || scores != nullptr, "null score block.");
    for (int32_t c = 0; c < n_cutoffs; c++) validate_call(e, rows, cutoffs[c], offset, n_threads);
    IRS_HIP(hipSetDevice(e->device));
    hipStream_t s = nullptr;
    DeviceBuffer<int64_t> mptr;
    DeviceBuffer<int32_t> midx;
    if (rows > 0) {
      const size_t bytes = static_cast<size_t>(rows) * e->n_items * (is_f64 ? 8 : 4);
      e->score_buf.alloc(bytes);
      IRS_HIP(hipMemcpyAsync(e->score_buf.ptr, scores, bytes, hipMemcpyHostToDevice, s));
      const int64_t mnnz = mask_indptr ? mask_indptr[rows] - mask_indptr[0] : 0;
      if (mnnz > 0) {
        check_arg(mask_indices != nullptr, "mask_indices is null.");
        std::vector<int64_t> mp(rows + 1);
        for (int64_t r = 0; r <= rows; r++) {
          mp[r] = mask_indptr[r] - mask_indptr[0];
          check_arg(mp[r] >= (r ? mp[r - 1] : 0), "mask_indptr must not decrease.");
        }
        mptr.upload(mp, s);
        midx.upload(mask_indices, static_cast<size_t>(mnnz), s);
        if (is_f64)
          hipLaunchKernelGGL(mask_block_kernel<double>, dim3(static_cast<unsigned>(rows)), dim3(64), 0, s,
                             reinterpret_cast<double *>(e->score_buf.ptr), rows, e->n_items, mptr.ptr,
                             midx.ptr);
        else
          hipLaunchKernelGGL(mask_block_kernel<float>, dim3(static_cast<unsigned>(rows)), dim3(64), 0, s,
                             reinterpret_cast<float *>(e->score_buf.ptr), rows, e->n_items, mptr.ptr,
                             midx.ptr);
        IRS_HIP(hipGetLastError());
      }
    }
    for (int32_t c = 0; c < n_cutoffs; c++) {
      begin_accumulate(e, s);
      if (rows > 0) {
        if (is_f64)
          rank_block<double>(e, e->score_buf.ptr, rows, cutoffs[c], offset, recall_with_cutoff != 0, s);
        else
          rank_block<float>(e, e->score_buf.ptr, rows, cutoffs[c], offset, recall_with_cutoff != 0, s);
      }
      finish_accumulate(e, out + c, item_cnt + static_cast<int64_t>(c) * e->n_items, s);
    }
  });
}

irs_status irs_eval_get_metrics_similarity(irs_evaluator *e, int64_t begin, int64_t end, int64_t n_model_users,
                                           int64_t n_profile_cols,
                                           const int64_t *x_indptr, const int32_t *x_indices, const double *x_data,
                                           const int64_t *w_indptr, const int32_t *w_indices, const double *w_data,
                                           const int64_t *mask_indptr, const int32_t *mask_indices,
                                           int32_t n_cutoffs, const int64_t *cutoffs, int64_t offset,
                                           int32_t recall_with_cutoff, irs_metrics *out, int64_t *item_cnt) {
  return guard([&] {
    check_arg(e && out && item_cnt && cutoffs && x_indptr && w_indptr, "null argument.");
    check_arg(0 <= begin && begin <= end && end <= n_model_users, "user range out of bounds.");
    check_arg(n_cutoffs >= 0, "negative count.");
    const int64_t rows = end - begin, ni = e->n_items;
    for (int32_t c = 0; c < n_cutoffs; c++) validate_call(e, rows, cutoffs[c], offset, 1);
    // the two CSR matrices: the profiles X [n_model_users, n_profile_cols] (rows begin .. end are read) and
    // W [n_profile_cols, n_items]: item-kNN / P3alpha / RP3beta: X = the training matrix, W item x item;
    // user-kNN (base.py:432-453: U[u] @ X): X = the user-user weights, W = the training matrix
    const int64_t np_ = n_profile_cols;
    check_arg(np_ >= 0 && np_ < (int64_t(1) << 31), "bad profile width.");
    check_arg(x_indptr[0] == 0 && w_indptr[0] == 0, "malformed indptr.");
    for (int64_t r = begin; r < end; r++) check_arg(x_indptr[r + 1] >= x_indptr[r], "malformed indptr.");
    for (int64_t i = 0; i < np_; i++) check_arg(w_indptr[i + 1] >= w_indptr[i], "malformed indptr.");
    const int64_t xq0 = x_indptr[begin], xq1 = x_indptr[end], x_nnz = xq1 - xq0, w_nnz = w_indptr[np_];
    check_arg((x_nnz == 0 || (x_indices && x_data)) && (w_nnz == 0 || (w_indices && w_data)), "null argument.");
    std::atomic<int> bad(0), x_not_ones(0);
    parallel_ranges(x_nnz, [&](int64_t lo, int64_t hi) {
      int32_t mn = 0, mx = 0;
      uint64_t diff = 0;
      const uint64_t *vb = reinterpret_cast<const uint64_t *>(x_data + xq0);
      for (int64_t q = lo; q < hi; q++) {
        mn = std::min(mn, x_indices[xq0 + q]);
        mx = std::max(mx, x_indices[xq0 + q]);
        diff |= vb[q] ^ 0x3ff0000000000000ull;
      }
      if (hi > lo && (mn < 0 || mx >= np_)) bad.store(1);
      if (diff) x_not_ones.store(1);
    });
    parallel_ranges(w_nnz, [&](int64_t lo, int64_t hi) {
      int32_t mn = 0, mx = 0;
      for (int64_t q = lo; q < hi; q++) {
        mn = std::min(mn, w_indices[q]);
        mx = std::max(mx, w_indices[q]);
      }
      if (hi > lo && (mn < 0 || mx >= ni)) bad.store(1);
    });
    check_arg(bad.load() == 0, "column index out of range.");
    // rows of W with strictly increasing columns (what every recommender of this package stores) are cut
    // into per-tile ranges on the device; any other W takes the whole-row walk
    std::atomic<int> w_unsorted{0};
    parallel_ranges(np_, [&](int64_t lo, int64_t hi) {
      int any = 0;
      for (int64_t r = lo; r < hi; r++)
        for (int64_t q = w_indptr[r] + 1; q < w_indptr[r + 1]; q++) any |= w_indices[q] <= w_indices[q - 1];
      if (any) w_unsorted.store(1);
    });
    const bool w_tiled = w_unsorted.load() == 0 && w_nnz < (int64_t(1) << 31);
    IRS_HIP(hipSetDevice(e->device));
    hipStream_t s = nullptr;
    // uploads: the profile rows (values only when they are not all ones), W by rows, the mask rows
    DeviceBuffer<int64_t> d_xp, d_wp, d_mp;
    DeviceBuffer<int32_t> d_xi, d_wi, d_mi;
    DeviceBuffer<double> d_xv, d_wv;
    std::vector<int64_t> xp(static_cast<size_t>(rows) + 1);
    for (int64_t r = 0; r <= rows; r++) xp[r] = x_indptr[begin + r] - xq0;
    d_xp.upload(xp, s);
    d_xi.upload(x_indices + xq0, static_cast<size_t>(x_nnz), s);
    if (x_not_ones.load()) d_xv.upload(x_data + xq0, static_cast<size_t>(x_nnz), s);
    d_wp.upload(w_indptr, static_cast<size_t>(np_) + 1, s);
    d_wi.upload(w_indices, static_cast<size_t>(w_nnz), s);
    d_wv.upload(w_data, static_cast<size_t>(w_nnz), s);
    const int64_t m_nnz = mask_indptr ? mask_indptr[rows] - mask_indptr[0] : 0;
    std::vector<int64_t> mp;
    // (the usual caller masks with the rows it scores from - `X_train[u] @ W`, seen items removed: the same
    // arrays, handed over twice; they are uploaded once - 80 MB less over PCIe on the ML-20M shape)
    bool mask_is_x = false;
    const int64_t *mask_ptr_dev = nullptr;
    const int32_t *mask_idx_dev = nullptr;
    if (m_nnz > 0) {
      check_arg(mask_indices != nullptr, "mask_indices is null.");
      mp.resize(static_cast<size_t>(rows) + 1);
      for (int64_t r = 0; r <= rows; r++) {
        mp[r] = mask_indptr[r] - mask_indptr[0];
        check_arg(mp[r] >= (r ? mp[r - 1] : 0), "mask_indptr must not decrease.");
      }
      mask_is_x = mask_indices == x_indices + xq0 && m_nnz == x_nnz && mp == xp;
      if (mask_is_x) {
        mask_ptr_dev = d_xp.ptr;
        mask_idx_dev = d_xi.ptr;
      } else {
        d_mp.upload(mp, s);
        d_mi.upload(mask_indices, static_cast<size_t>(m_nnz), s);
        mask_ptr_dev = d_mp.ptr;
        mask_idx_dev = d_mi.ptr;
      }
    }
    // blocks of users whose dense float64 scores fit 4 GB; every cutoff ranks the same block.  (A block ends
    // with a read-back of its metrics and item counts: ~0.7 ms of launches, synchronisation and host merge.
    // All 138,493 users of the ML-20M shape, 29.6 GB of scores: 1 GB blocks 59.8 ms, 2 GB 50.1, 4 GB 44.5,
    // 8 GB 42.6, one block 40.5 - against 90 / 190 / 310 ms for the first call's allocation at 1 / 4 / 8 GB.)
    int64_t per = std::max<int64_t>(1, std::min<int64_t>(rows, (int64_t(1) << 32) / std::max<int64_t>(8 * ni, 1)));
    if (const char *v = std::getenv("IRSPACK_AMD_EVAL_SIM_BLOCK_ROWS"))  // (tests: several blocks on a small call)
      per = std::max<int64_t>(1, std::min<int64_t>(per, std::atoll(v)));
    const int32_t n_tiles = static_cast<int32_t>(ceil_div(std::max<int64_t>(ni, 1), SIM_TILE));
    // per cutoff: the running totals of the blocks, on the device (read back once, after the last block)
    const size_t nc_ = static_cast<size_t>(std::max(n_cutoffs, 1));
    DeviceBuffer<irs_metrics> tot_m;
    DeviceBuffer<unsigned long long> tot_cnt;
    tot_m.alloc(nc_);
    tot_cnt.alloc(nc_ * static_cast<size_t>(std::max<int64_t>(ni, 1)));
    tot_m.zero(s);
    tot_cnt.zero(s);
    std::fill(item_cnt, item_cnt + static_cast<int64_t>(std::max(n_cutoffs, 0)) * ni, int64_t(0));
    if (rows > 0) e->score_buf.alloc(static_cast<size_t>(per) * ni * 8);
    DeviceBuffer<int32_t> d_wt;
    if (w_tiled && rows > 0 && np_ > 0) {
      const int64_t n_tp = np_ * (n_tiles + 1);
      d_wt.alloc(static_cast<size_t>(n_tp));
      hipLaunchKernelGGL(sim_tile_ptr_kernel, dim3(static_cast<unsigned>(ceil_div(n_tp, 256))), dim3(256), 0, s,
                         static_cast<const int64_t *>(d_wp.ptr), static_cast<const int32_t *>(d_wi.ptr), np_, n_tiles,
                         d_wt.ptr);
    }
    // launch order inside every block: rows by stored profile length, longest first (counting sort, stable)
    DeviceBuffer<int32_t> d_order;
    if (rows > 0) {
      std::vector<int32_t> order(static_cast<size_t>(rows));
      constexpr int64_t CAP = 1 << 16;
      std::vector<int32_t> start(CAP + 2);
      for (int64_t b = 0; b < rows; b += per) {
        const int64_t m = std::min(per, rows - b);
        std::fill(start.begin(), start.end(), 0);
        auto len = [&](int64_t r) { return std::min<int64_t>(CAP, xp[b + r + 1] - xp[b + r]); };
        for (int64_t r = 0; r < m; r++) start[CAP - len(r) + 1]++;
        for (size_t i = 1; i < start.size(); i++) start[i] += start[i - 1];
        for (int64_t r = 0; r < m; r++) order[b + start[CAP - len(r)]++] = static_cast<int32_t>(r);
      }
      d_order.upload(order, s);
    }
    for (int64_t b = 0; b < rows; b += per) {
      const int64_t m = std::min(per, rows - b);
      double *scores = reinterpret_cast<double *>(e->score_buf.ptr);
      hipLaunchKernelGGL(d_wt.ptr ? sim_score_kernel<true> : sim_score_kernel<false>, dim3(static_cast<unsigned>(m * n_tiles)), dim3(64), 0, s,
                         static_cast<const int64_t *>(d_xp.ptr), static_cast<const int32_t *>(d_xi.ptr),
                         x_not_ones.load() ? static_cast<const double *>(d_xv.ptr) : static_cast<const double *>(nullptr),
                         static_cast<const int64_t *>(d_wp.ptr), static_cast<const int32_t *>(d_wi.ptr),
                         static_cast<const double *>(d_wv.ptr), std::max<int64_t>(w_nnz - 1, 0), b, ni, n_tiles, scores,
                         static_cast<const int32_t *>(d_wt.ptr), static_cast<const int32_t *>(d_order.ptr) + b);
      if (m_nnz > 0)
        hipLaunchKernelGGL(mask_block_kernel<double>, dim3(static_cast<unsigned>(m)), dim3(64), 0, s, scores, m, ni,
                           mask_ptr_dev + b, mask_idx_dev);
      IRS_HIP(hipGetLastError());
      for (int32_t c = 0; c < n_cutoffs; c++) {
        begin_accumulate(e, s);
        rank_block<double>(e, scores, m, cutoffs[c], offset + b, recall_with_cutoff != 0, s);
        // Metrics::merge of the block into the cutoff's totals, block after block (evaluator.cpp:76-85)
        hipLaunchKernelGGL(metrics_fold_kernel, dim3(1), dim3(64), 0, s, static_cast<const irs_metrics *>(e->metrics.ptr),
                           tot_m.ptr + c);
        if (ni > 0)
          hipLaunchKernelGGL(counts_fold_kernel, dim3(static_cast<unsigned>(ceil_div(ni, 256))), dim3(256), 0, s,
                             static_cast<const unsigned long long *>(e->item_cnt.ptr), ni, tot_cnt.ptr + static_cast<size_t>(c) * ni);
      }
    }
    if (e->span_open) IRS_HIP(hipEventRecord(e->ev_last, s));
    if (n_cutoffs > 0) {
      IRS_HIP(hipMemcpyAsync(out, tot_m.ptr, sizeof(irs_metrics) * static_cast<size_t>(n_cutoffs), hipMemcpyDeviceToHost, s));
      if (ni > 0)
        IRS_HIP(hipMemcpyAsync(item_cnt, tot_cnt.ptr, sizeof(int64_t) * static_cast<size_t>(n_cutoffs) * ni,
                               hipMemcpyDeviceToHost, s));
    }
    IRS_HIP(hipStreamSynchronize(s));
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
    // users scored and ranked per pass: enough rows to fill the device with one wave per row
    // (32 waves x 256 CUs), within a 2 GiB score block
    const int64_t fit = (int64_t(1) << 31) / (std::max<int64_t>(e->n_items, 1) * 4);
    int64_t block_cap = 16384;
    if (const char *eb = std::getenv("IRSPACK_AMD_EVAL_BLOCK")) block_cap = std::max<int64_t>(256, std::atoll(eb));
    const int64_t BLOCK = std::min<int64_t>(block_cap, std::max<int64_t>(1024, fit / 1024 * 1024));
    DeviceBuffer<int64_t> mptr;
    DeviceBuffer<int32_t> midx;
    void *sv = nullptr;
    int32_t dev = 0;
    // a zero-row call only fetches the trainer's stream / device
    if (irs_ials_scores_device_(t, begin, begin, nullptr, &sv, &dev) != IRS_OK)
      throw std::runtime_error(irs_last_error());
    check_arg(dev == e->device, "evaluator and trainer live on different devices.");
    hipStream_t s = static_cast<hipStream_t>(sv);
    const auto host_t0 = std::chrono::steady_clock::now();
    if (!e->ev_first) {
      IRS_HIP(hipEventCreate(&e->ev_first));
      IRS_HIP(hipEventCreate(&e->ev_last));
    }
    IRS_HIP(hipEventRecord(e->ev_first, s));
    e->span_open = true;
    struct CloseSpan {  // the call's times, whichever path returns (irs_eval_stats: call_ms, device_span_ms)
      irs_evaluator *e;
      std::chrono::steady_clock::time_point t0;
      ~CloseSpan() {
        float ms = 0.0f;
        if (e->span_open && hipEventElapsedTime(&ms, e->ev_first, e->ev_last) != hipSuccess) {
          (void)hipGetLastError();
          ms = 0.0f;
        }
        e->span_open = false;
        e->stats.device_span_ms = ms;
        e->stats.call_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
      }
    } close_span{e, host_t0};
    begin_accumulate(e, s);
    const int64_t *d_mptr = nullptr;
    const int32_t *d_midx = nullptr;
    if (mask_indptr) {
      std::vector<int64_t> mp(mask_indptr, mask_indptr + rows + 1);
      mptr.upload(mp, s);
      midx.upload(mask_indices, static_cast<size_t>(mask_indptr[rows]), s);
      IRS_HIP(hipStreamSynchronize(s));
      d_mptr = mptr.ptr;
      d_midx = midx.ptr;
    } else if (e->mask_rows == rows) {  // the cached mask (irs_eval_cache_mask)
      d_mptr = e->mask_ptr.ptr;
      d_midx = e->mask_idx.ptr;
    }
    if (emit_path(e, t, begin, rows, d_mptr, d_midx, cutoff, offset, recall_with_cutoff != 0, s)) {
      finish_accumulate(e, out, item_cnt, s);
      return;
    }
    begin_accumulate(e, s);  // (an abandoned attempt may have touched the sums)
    e->stats = irs_eval_stats{0, 0, ceil_div(rows, 64) * ceil_div(e->n_items, 64),
                              ceil_div(rows, 64) * ceil_div(e->n_items, 64), 0};
    DeviceBuffer<float> &scores = e->fused_scores;
    scores.alloc(static_cast<size_t>(std::min(BLOCK, std::max<int64_t>(rows, 1))) * e->n_items);
    for (int64_t b = 0; b < rows; b += BLOCK) {
      const int64_t m = std::min(BLOCK, rows - b);
      if (irs_ials_scores_device_(t, begin + b, begin + b + m, scores.ptr, &sv, &dev) != IRS_OK)
        throw std::runtime_error(irs_last_error());
      if (d_mptr)
        hipLaunchKernelGGL(mask_rows_kernel, dim3(m), dim3(64), 0, s, scores.ptr, m, e->n_items,
                           d_mptr + b, d_midx);
      rank_block<float>(e, scores.ptr, m, cutoff, offset + b, recall_with_cutoff != 0, s);
    }
    finish_accumulate(e, out, item_cnt, s);
  });
}

irs_status irs_eval_cache_mask(irs_evaluator *e, int64_t rows, const int64_t *mask_indptr,
                               const int32_t *mask_indices) {
  return guard([&] {
    check_arg(e != nullptr, "null argument.");
    e->mask_bits_rows = -1;  // the bitmap follows the cached mask
    if (mask_indptr == nullptr || rows <= 0) {  // drop the cached mask
      e->mask_rows = -1;
      return;
    }
    check_arg(mask_indices != nullptr && mask_indptr[0] == 0, "malformed mask.");
    IRS_HIP(hipSetDevice(e->device));
    hipStream_t s = nullptr;
    std::vector<int64_t> mp(mask_indptr, mask_indptr + rows + 1);
    e->mask_ptr.upload(mp, s);
    e->mask_idx.upload(mask_indices, static_cast<size_t>(std::max<int64_t>(mask_indptr[rows], 1)), s);
    IRS_HIP(hipStreamSynchronize(s));
    e->mask_rows = rows;
  });
}

// 64-bit content fingerprint of a host buffer on several threads (the strict check of the
// evaluator's device-resident mask: every byte of ~140 MB per call, 4.5 ms with one thread of
// xxh3).  Not cryptographic: per 1 MiB piece a multiply-rotate mix of its 8-byte words (any changed
// word changes the piece's value), the pieces combined with their position.
irs_status irs_fingerprint(const void *data, int64_t n_bytes, uint64_t seed, uint64_t *out) {
  return guard([&] {
    check_arg(out != nullptr && n_bytes >= 0 && (data != nullptr || n_bytes == 0), "bad argument.");
    const unsigned char *p = static_cast<const unsigned char *>(data);
    constexpr int64_t PIECE = int64_t(1) << 20;
    const int64_t n_pieces = ceil_div(n_bytes, PIECE);
    int n_thr = static_cast<int>(std::max<int64_t>(
        1, std::min<int64_t>({16, static_cast<int64_t>(std::thread::hardware_concurrency()), n_pieces / 4 + 1})));
    // (IRSPACK_AMD_FINGERPRINT_THREADS: tests force the thread count - the value must not depend on it)
    if (const char *te = std::getenv("IRSPACK_AMD_FINGERPRINT_THREADS"))
      n_thr = static_cast<int>(std::max<int64_t>(1, std::min<int64_t>(64, std::atoll(te))));
    auto mix = [](uint64_t h, uint64_t w) {
      h ^= w * 0x9E3779B97F4A7C15ull;
      h = (h << 29) | (h >> 35);
      return h * 0xD6E8FEB86659FD93ull;
    };
    auto piece_hash = [&](int64_t k) {
      const int64_t b = k * PIECE, e = std::min(n_bytes, b + PIECE);
      uint64_t h[4] = {seed ^ 0x243F6A8885A308D3ull, seed ^ 0x13198A2E03707344ull, seed ^ 0xA4093822299F31D0ull,
                       seed ^ 0x082EFA98EC4E6C89ull};  // four independent lanes: the loop pipelines
      int64_t i = b;
      for (; i + 32 <= e; i += 32) {
        uint64_t w[4];
        std::memcpy(w, p + i, 32);
        h[0] = mix(h[0], w[0]);
        h[1] = mix(h[1], w[1]);
        h[2] = mix(h[2], w[2]);
        h[3] = mix(h[3], w[3]);
      }
      uint64_t tail[4] = {0, 0, 0, 0};
      std::memcpy(tail, p + i, static_cast<size_t>(e - i));
      for (int q = 0; q < 4; q++) h[q] = mix(h[q], tail[q] + static_cast<uint64_t>(e - i));
      return mix(mix(h[0], h[1]), mix(h[2], h[3]) + static_cast<uint64_t>(k));
    };
    std::vector<uint64_t> part(static_cast<size_t>(n_thr), 0);
    auto work = [&](int t) {
      uint64_t acc = 0;
      for (int64_t k = t; k < n_pieces; k += n_thr) acc += mix(piece_hash(k), static_cast<uint64_t>(k) + 1);
      part[t] = acc;
    };
    {
      std::vector<std::thread> th;
      struct Join {
        std::vector<std::thread> &v;
        ~Join() { for (auto &t : v) if (t.joinable()) t.join(); }
      } join{th};
      for (int t = 1; t < n_thr; t++) th.emplace_back(work, t);
      work(0);
    }
    uint64_t h = mix(seed, static_cast<uint64_t>(n_bytes));
    for (int t = 0; t < n_thr; t++) h += part[t];  // (a sum: independent of the thread count's dealing)
    *out = h;
  });
}

irs_status irs_retrieve_recommend(int32_t is_f64, const void *scores, int64_t rows,
                                  int64_t n_items, int64_t n_lists, const int64_t *list_ptr,
                                  const int64_t *list_items, int64_t cutoff, int64_t n_threads,
                                  int32_t device, int32_t *out_idx) {
  return guard([&] {
    // util.hpp:426-439
    check_arg(n_threads > 0, "n_threads must not be 0.");
    check_arg(rows >= 0 && n_items >= 0, "negative shape.");
    check_arg(n_lists == 0 || n_lists == 1 || n_lists == rows,
              "allowed_indices, if not empty, must have a size equal to X.rows()");
    check_arg(out_idx != nullptr && (rows == 0 || scores != nullptr), "null argument.");
    if (rows == 0 || cutoff <= 0) return;
    require_device(device);
    hipStream_t s = nullptr;
    // candidate lists keep their order and duplicates; out-of-range ids are dropped (:468-472)
    std::vector<int64_t> lptr(n_lists + 1, 0);
    std::vector<int32_t> litems;
    int64_t max_cand = n_lists == 0 ? n_items : 0;
    for (int64_t l = 0; l < n_lists; l++) {
      for (int64_t q = list_ptr[l]; q < list_ptr[l + 1]; q++)
        if (list_items[q] >= 0 && list_items[q] < n_items)
          litems.push_back(static_cast<int32_t>(list_items[q]));
      lptr[l + 1] = static_cast<int64_t>(litems.size());
      max_cand = std::max(max_cand, lptr[l + 1] - lptr[l]);
    }
    DeviceBuffer<int64_t> d_lptr;
    DeviceBuffer<int32_t> d_litems, d_rec;
    DeviceBuffer<RowOut> d_rows;
    DeviceBuffer<char> d_scores;
    if (n_lists > 0) {
      d_lptr.upload(lptr, s);
      if (litems.empty()) litems.push_back(0);
      d_litems.upload(litems, s);
    }
    const size_t bytes = static_cast<size_t>(rows) * n_items * (is_f64 ? 8 : 4);
    d_scores.alloc(std::max<size_t>(bytes, 8));
    IRS_HIP(hipMemcpyAsync(d_scores.ptr, scores, bytes, hipMemcpyHostToDevice, s));
    d_rec.alloc(static_cast<size_t>(rows) * cutoff);
    d_rows.alloc(rows);
    DeviceBuffer<int32_t> d_todo;
    d_todo.alloc(rows);
    EvalParams p{};
    p.scores = d_scores.ptr;
    p.rows = rows;
    p.n_items = n_items;
    p.offset = 0;
    p.rec_mode = n_lists == 0 ? 0 : (n_lists == 1 ? 1 : 2);
    p.rec_ptr = d_lptr.ptr;
    p.rec_items = d_litems.ptr;
    p.cutoff = static_cast<int32_t>(cutoff);
    p.retrieve = 1;
    p.out = d_rows.ptr;
    p.rec_out = d_rec.ptr;
    if (is_f64)
      launch_rank<double>(p, max_cand, s, d_todo.ptr);
    else
      launch_rank<float>(p, max_cand, s, d_todo.ptr);
    IRS_HIP(hipGetLastError());
    IRS_HIP(hipMemcpyAsync(out_idx, d_rec.ptr, static_cast<size_t>(rows) * cutoff * sizeof(int32_t),
                           hipMemcpyDeviceToHost, s));
    IRS_HIP(hipStreamSynchronize(s));
  });
}

}  // extern "C"
