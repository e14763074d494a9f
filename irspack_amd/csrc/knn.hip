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
#include <atomic>
#include <thread>
#include <chrono>
#include <cmath>
#include <cstring>
#include <memory>
#include <mutex>
#include <numeric>

#include "common.hpp"
#include "knn_host_prep.hpp"

namespace irs {
namespace knn {

constexpr int TILE = 16384;     // columns per workgroup (128 KB of fp64 accumulators)
constexpr int TOPK_CAP = 2048;  // largest top_k of the LDS row merge (two candidate sets must fit one
                                // round); above it the rows are merged by knn_merge_big_kernel
constexpr int MERGE_CAP = 4096; // candidates a row merge can hold
constexpr int THREADS = 1024;

struct Params {
  // X_arg^T by rows u: (column j, value x) sorted by j; xt_tptr[u * (n_tiles + 1) + t]
  // is the position of the first entry of row u with column >= t * TILE (the last one is
  // the row end), so a (row, tile) workgroup walks exactly its own slice of every row
  const uint32_t *xt_tptr;
  const uint32_t *xt_idx16;  // tile-relative columns, 16 bits each, two per dword
  const double *xt_val;
  // Not null: every feature row u of X_arg^T holds ONE value xt_rowval[u] in all its stored entries
  // (TF-IDF of binary interactions: the weight is the row's idf).  The kernels then run their all-ones
  // form on y' = x_u y - the same products x y, bit for bit - and never read the 8-byte value stream
  // (four fifths of the bytes of a weighted walk).
  const double *xt_rowval;
  const double *norms;  // per column j
  // Not null: the sums of column j are multiplied by col_scale[j] (one rounding) before the similarity's
  // epilogue - the column factor of a SEPARABLE weighting w[u][j] = rowval[u] * col_scale[j] (tf-idf / BM25 of
  // binary interactions, fused at construction: DESIGN 3.4).  Exact kernels only (never FAST).
  const double *col_scale;
  // target rows
  const int64_t *t_ptr;
  const int32_t *t_idx;
  const double *t_val;
  const double *t_stat;       // per target row (norm / pow / count), see epilogue
  const double *t_scale;      // per target row: power of two that maps its products to the
                              // 64-bit fixed-point accumulators (weighted paths); NEGATIVE: the
                              // row's products span a wide range - two limbs, see `hi_stash`
  uint64_t *hi_stash;         // [workgroups x TILE]: the first limbs of a wide-range pair while the
                              // second ones are accumulated
  const int32_t *row_order;   // work-sorted list of target rows of this call
  int32_t n_rows;             // rows in this call
  int32_t n_tiles;
  int32_t N;
  int32_t sim_type;
  int32_t normalize;
  double shrinkage, alpha, beta;
  float shrink_f, alpha_f, beta_f;  // the same rounded to float32 (FAST path: scalar operands, no conversions)
  int32_t top_k;
  int32_t tile_k;      // winners one (row, tile) pair can hold: min(top_k, TILE)
  // per (row slot, tile) winners, in ascending column order
  int32_t *cand_idx;   // [n_rows * n_tiles * tile_k]
  double *cand_val;
  int32_t *cand_cnt;   // [n_rows * n_tiles]
  // final rows
  int32_t *out_idx;    // [n_rows * top_k]
  double *out_val;
  int32_t *out_cnt;    // [n_rows]
  // persistent launch: workgroups draw (row slot, tile) pairs from this counter
  int32_t *cursor;
  int32_t n_slots;     // n_rows * n_tiles
  // FAST: pairs whose candidates did not fit are listed here and taken by a second launch of the
  // exact kernel (redo != 0: the pairs are redo_list[0 .. *redo_count), drawn through cursor[1])
  int32_t *redo_list;
  int32_t *redo_count;
  int32_t redo;
  int32_t merge_cap;   // entries of the merge buffer: power of two, >= 2 * top_k
};

__device__ __forceinline__ int64_t readlane_i64(int64_t v, int src) {
  const uint32_t lo = __builtin_amdgcn_readlane(static_cast<uint32_t>(v), src);
  const uint32_t hi = __builtin_amdgcn_readlane(static_cast<uint32_t>(static_cast<uint64_t>(v) >> 32), src);
  return static_cast<int64_t>((static_cast<uint64_t>(hi) << 32) | lo);
}
__device__ __forceinline__ double readlane_f64(double v, int src) {
  return __longlong_as_double(readlane_i64(__double_as_longlong(v), src));
}

__device__ __forceinline__ uint64_t order_key(double s) {
  if (s != s) return 0ull;
  if (s == 0.0) s = 0.0;
  uint64_t u = static_cast<uint64_t>(__double_as_longlong(s));
  return (u & 0x8000000000000000ull) ? ~u : (u | 0x8000000000000000ull);
}

__device__ __forceinline__ double key_to_double(uint64_t k) {  // inverse of order_key (non-NaN)
  const uint64_t u = (k & 0x8000000000000000ull) ? (k & 0x7fffffffffffffffull) : ~k;
  return __longlong_as_double(static_cast<long long>(u));
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

// float32 approximation of `epilogue` for a COUNT v >= 1 (FAST path).  Every term of the
// denominator is non-negative (the host checks shrinkage, alpha, beta >= 0) and the only
// subtractions are of integers below 2^24 (exact in float32) or leave a result no smaller than
// half of their operands' sum, so the relative error stays below ~12 roundings of 2^-24 < 2^-20.
__device__ __forceinline__ float approx_epilogue(const Params &p, float v, float norm_j, float tstat,
                                                 float shrink, float alpha, float beta) {
  if (p.sim_type == IRS_SIM_COSINE && !p.normalize) return v;  // the count itself (exact below 2^24)
  float d;
  switch (p.sim_type) {
    case IRS_SIM_JACCARD: d = (norm_j + tstat) - v; break;
    case IRS_SIM_TVERSKY: d = v + beta * (norm_j - v) + alpha * (tstat - v); break;
    default: d = norm_j * tstat; break;  // cosine (normalised), asymmetric cosine
  }
  d = d + shrink + 1e-6f;
  return v * __builtin_amdgcn_rcpf(d);
}
constexpr int FAST_CAP = 2048;   // candidates the exact ranking of the FAST path holds (32 KB of LDS)
#ifndef IRS_KNN_FAST_SLACK
#define IRS_KNN_FAST_SLACK 96
#endif
constexpr int FAST_SLACK = IRS_KNN_FAST_SLACK;  // the approximate select stops once at most top_k + this many keys remain

// touched-ness of a column without a bitmap atomic: accumulators start at -0.0, and
// -0.0 + x is x for every x that is not itself -0.0, so a column stays at the bit
// pattern of -0.0 exactly when nothing (or only -0.0 products) was added.  The host
// selects this only when no product can be a zero (no stored zeros, no underflow);
// otherwise the bitmap is maintained with atomics next to the sums.
constexpr uint64_t NEG_ZERO_BITS = 0x8000000000000000ull;

#ifdef IRS_KNN_PHASES
__device__ unsigned long long knn_fast_stat[8];  // pairs, exact-path pairs, sum of candidates, pairs with <= 128 / 256 / 512 / 1024 candidates
__device__ unsigned long long knn_phase_clk[8];
#define PHASE_MARK(i) do { if (threadIdx.x == 0) { const unsigned long long t_ = wall_clock64(); atomicAdd(&knn_phase_clk[i], t_ - ph_t); ph_t = t_; } } while (0)
#else
#define PHASE_MARK(i)
#endif

// ACC32 (every stored value of both operands is exactly 1, so every product is 1): the sums
// are counts, accumulated with 32-bit LDS atomics (one bank per lane instead of two: 1.5x the
// fp64 atomic rate) in the upper half of the accumulator block and widened to fp64 in place
// afterwards - the same values as the fp64 sums, which are exact for integers.
// (A COMPACT form of the count path - 512 threads and 66 KB of LDS per workgroup, two workgroups per CU so
// that one pair's epilogue / selection overlaps the other's accumulation - was built in round 3, measured
// at 10.69 against 10.54 ms and kept as an opt-in; removed in round 5.  DESIGN.md 3.4 has the numbers.)
// FAST (round 3; counts only, the four similarities that end in a division): the selection runs
// on float32 APPROXIMATIONS of the similarities (8 vector instructions per column instead of the
// ~45 of the un-fused fp64 epilogue, 32-bit keys), and only the columns that can still be among
// the top_k - approximation within 2^-17 of the top_k-th best approximation - get the exact
// fp64 value and are ranked exactly (value desc, column asc).  With |approx - exact| <=
// eps * exact for every stored column (eps < 2^-20, see approx_epilogue and
// tests/test_knn_approx_bound.py) a column of the exact
// top_k has approx >= t (1 - eps) / (1 + eps), t = the top_k-th largest approximation, so the
// candidate set is a superset of the exact winners whatever the ties; when it does not fit
// FAST_CAP entries (thousands of near-ties) the pair takes the exact path below.
template <bool ONES, bool SENTINEL, bool ACC32 = false, bool FAST = false>
__global__ __launch_bounds__(THREADS, 1) void knn_tile_kernel(Params p) {
  static_assert(!ACC32 || (ONES && SENTINEL), "counts need all-ones operands");
  static_assert(!FAST || ACC32, "the approximate selection is built for the count path");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  double *acc = reinterpret_cast<double *>(smem);                        // TILE
  uint32_t *bits = reinterpret_cast<uint32_t *>(acc + TILE);             // TILE / 32
  uint32_t *hist = bits + TILE / 32;                                     // 256
  int32_t *wave_cnt = reinterpret_cast<int32_t *>(hist + 256);           // 16
  // one sink per lane for the lanes of a strip that lie outside their slice: the atomic is
  // issued without a branch (ACC32 uses the idle upper half of the accumulator block)
  constexpr uint32_t SINK_BYTES = TILE * 8 + (TILE / 32) * 4 + 256 * 4 + 16 * 4;
  __shared__ uint64_t sh_prefix;
  __shared__ int32_t sh_need, sh_count, sh_total;

  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  constexpr int NW = THREADS / 64;
  __shared__ int32_t sh_slot;
  // One workgroup per CU stays resident and draws (row slot, tile) pairs, heaviest rows
  // first, from a global counter: a 1024-thread / 130 KB workgroup costs ~20 us to launch
  // and retire, as much as the whole accumulation of a light row.  The draw for the next
  // pair is issued before the current one is processed, so its latency is hidden.
  int32_t next_slot = 0;
  int32_t *const cursor = p.redo ? p.cursor + 1 : p.cursor;
  const int n_draw = p.redo ? *p.redo_count : p.n_slots;
  if (tid == 0) next_slot = atomicAdd(cursor, 1);
  for (;;) {
  __syncthreads();  // the previous pair is finished by every wave (LDS and sh_* reuse)
  if (tid == 0) {
    sh_slot = next_slot;
    next_slot = atomicAdd(cursor, 1);
  }
  __syncthreads();
  if (sh_slot >= n_draw) break;
  const int bid = p.redo ? p.redo_list[sh_slot] : sh_slot;
  const int slot = bid / p.n_tiles, tile = bid % p.n_tiles;
  const int r = p.row_order[slot];
  const int c0 = tile * TILE, c1 = min(c0 + TILE, p.N);
  const int width = c1 - c0;
#ifdef IRS_KNN_PHASES
  unsigned long long ph_t = wall_clock64();
#endif

  uint32_t *cnt = reinterpret_cast<uint32_t *>(acc);  // ACC32: TILE counters
  const uint32_t *const xt_tptr_l = p.xt_tptr;
  const uint32_t *const xt_idx16_l = p.xt_idx16;
  if (ACC32) {
    for (int i = tid; i < width; i += THREADS) cnt[i] = 0u;
  } else {
    for (int i = tid; i < width; i += THREADS) acc[i] = SENTINEL ? -0.0 : 0.0;
  }
  for (int i = tid; i < TILE / 32; i += THREADS) bits[i] = 0u;
  __syncthreads();
  PHASE_MARK(0);

  // ---- 1. accumulate.  A wave takes 16 stored (u, y) of the target row at a time: lanes
  //      0..15 own one each.  The walk is a chain of dependent loads (t_idx -> slice bounds
  //      -> columns -> LDS atomic), software pipelined three blocks deep; per slice the wave
  //      spends ~16 instructions: positions are 32-bit (nnz < 2^31), bounds travel by
  //      v_readlane, the column fields are stored as LDS byte offsets (column * 4) and the
  //      atomic is issued by every lane (lanes outside the slice add to their sink).
  // Wide-range rows (weights spanning many decades: tf-idf / BM25 / RP3beta powers on large
  // matrices): one scale per row leaves a column whose sum is tiny next to the row's bound with
  // few significant bits.  Their sums are kept in TWO 64-bit limbs - q_hi = round(v 2^s) as always,
  // q_lo = round((v 2^s - q_hi) 2^40) - accumulated in two passes over the same slices into the
  // same LDS block (the first limbs wait in global scratch meanwhile): integer adds, so still
  // independent of the order of arrival, and 2^-40 of the one-limb rounding step.
  const double ts_ = ACC32 ? 1.0 : p.t_scale[r];
  const bool wide = !ACC32 && ts_ < 0.0;
  const double fx_scale = ACC32 ? 1.0 : fabs(ts_);
  int pass = 0;
  auto add = [&](uint32_t off4, bool ok, double v) {
    if (ACC32) {
      // (lanes outside their slice are masked off - v_cmp + exec save / restore on the scalar unit,
      // no branch: one vector instruction fewer per atomic than selecting a per-lane sink
      // address, 8.77 -> 8.35 ms per ML-20M call)
      if (ok) atomicAdd(reinterpret_cast<uint32_t *>(smem + off4), 1u);
    } else {
      // Weighted products are summed in 64-bit FIXED POINT (v * 2^s rounded to an integer, s
      // per target row such that the row's largest possible sum stays below 2^61): integer
      // adds commute, so the sums - and with them the top-k - do not depend on the order in
      // which the LDS atomics land, run to run.  |error| of a column <= (its product count)
      // * 2^-(s+1).  With the sentinel (positive data) a product never rounds to 0: the
      // column must leave the "untouched" pattern.
      const uint32_t a = ok ? 2 * off4 : SINK_BYTES + 8 * lane;
      const double scaled = v * fx_scale;  // (a power of two: exact)
      long long q = __double2ll_rn(scaled);
      if (SENTINEL) q = q == 0 ? 1 : q;
      // second limb: what the first one left out (|scaled - q| <= 1; above 2^53 both are the same integer)
      if (pass == 1) q = __double2ll_rn((scaled - static_cast<double>(q)) * 0x1p40);
      atomicAdd(reinterpret_cast<unsigned long long *>(smem + a), static_cast<unsigned long long>(q));
      if (!SENTINEL && ok && pass == 0) {
        const uint32_t j = off4 >> 2, bit = 1u << (j & 31);
        if (!(bits[j >> 5] & bit)) atomicOr(&bits[j >> 5], bit);
      }
    }
  };
  const int64_t tb = p.t_ptr[r], te = p.t_ptr[r + 1];
  const int tp_stride = p.n_tiles + 1;
  const int64_t stride = 16 * NW;
  // every load below is unconditional (clamped address, result masked afterwards): a load
  // under a branch makes hipcc drain the whole vector-memory queue (s_waitcnt vmcnt(0))
  // right behind it, which would serialise the pipeline
  auto load_uy = [&](int64_t q0, int64_t &u, double &y, bool &v) {
    const int64_t q = q0 + lane;
    v = lane < 16 && q < te;
    const int64_t qc = min(q, te - 1);  // te > tb here
    u = p.t_idx[qc];
    y = ACC32 ? 1.0 : p.t_val[qc];  // ACC32: the value stream is not uploaded
    if (!ACC32 && ONES && p.xt_rowval != nullptr) y = __dmul_rn(p.xt_rowval[u], y);
  };
  auto load_bounds = [&](int64_t u, bool v, uint32_t &lo, int &len) {
    const uint32_t *tp = xt_tptr_l + u * tp_stride + tile;
    lo = tp[0];
    uint32_t hi = tp[1];
    asm volatile("" : "+v"(hi));  // keep the load out of the select below (see above)
    len = v ? static_cast<int>(hi - lo) : 0;
  };
  // Columns are stored tile-relative in 16 bits, two per dword: lane l of a strip takes the
  // entries a + 2l and a + 2l + 1 (a = slice start rounded down to even), so one 256-byte
  // wave load covers 128 entries of a slice.  The 16 slices of a block are cut into such
  // strips and the strips are processed in batches of 16 (16 loads in flight, then 32
  // atomics), whatever slice they belong to: lanes 0..15 each describe one strip of the
  // batch (first dword, offset of its first entry in the slice, slice length, y) by a
  // binary search in the running strip count of the block.
  struct Block {   // per-lane (0..15) state of one block of 16 stored (u, y)
    uint32_t lo;   // slice start in xt_idx16 / xt_val (entries)
    int len;       // slice length, 0 past the end of the row
    double y;
    int pin;       // inclusive prefix of the strip counts over lanes 0..15
    int total, nb; // strips, batches (wave-uniform)
  };
  auto count_strips = [&](Block &bk) {
    const int ns = bk.len > 0 ? (bk.len + static_cast<int>(bk.lo & 1u) + 127) >> 7 : 0;
    int pin = lane < 16 ? ns : 0;
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) {
      const int t = __shfl_up(pin, o, 64);
      if (lane >= o) pin += t;
    }
    bk.pin = pin;
    bk.total = __builtin_amdgcn_readlane(pin, 15);
    bk.nb = max(1, (bk.total + 15) >> 4);
  };
  struct Batch {  // per-lane (0..15): one strip
    uint32_t dw;  // first dword of the strip in xt_idx16
    uint32_t ol;  // (slice length << 16) | (slice offset of the strip's first entry + 1): both are at
                  // most TILE, the offset is -1 when the slice starts odd, the length 0 without a strip
                  // - one word, one v_readlane per strip instead of two
    double y;
  };
  auto make_batch = [&](uint32_t lo, int len, double y, int pin, int total, int j) {
    const int m = 16 * j + (lane & 15);
    int u = 0;  // smallest slice with pin[u] > m
#pragma unroll
    for (int step = 8; step >= 1; step >>= 1) {
      const int t = __shfl(pin, u + step - 1, 64);
      if (t <= m) u += step;
    }
    u = min(u, 15);
    const int before = __shfl(pin, max(u - 1, 0), 64);
    const int sidx = m - (u > 0 ? before : 0);
    const uint32_t lo_u = __shfl(lo, u, 64);
    Batch bt;
    // (strip loads that start on 128-byte lines were timed - wrong sums, same 8.45 ms: the line
    // traffic of the unaligned strips is not what bounds the accumulation)
    bt.dw = (lo_u >> 1) + 64u * static_cast<uint32_t>(sidx);
    const uint32_t o0p1 = static_cast<uint32_t>(128 * sidx + 1) - (lo_u & 1u);
    // (shuffles stay outside conditionals: ds_bpermute returns 0 from an inactive source lane)
    const int len_u = __shfl(len, u, 64);
    bt.ol = m < total ? ((static_cast<uint32_t>(len_u) << 16) | o0p1) : 0u;
    bt.y = ACC32 ? 1.0 : __shfl(y, u, 64);
    if (!(m < total)) bt.dw = lo_u >> 1;  // a valid address for the (ignored) load
    return bt;
  };
  auto issue_idx = [&](const Batch &bt, uint32_t (&w)[16]) {
#pragma unroll
    for (int t = 0; t < 16; t++) {
      // (uniform strip pointer + lane: one 64-bit shift-add on a precomputed per-lane base; the
      // 32-bit index `dw + lane` cost an add and a 64-bit scale per strip)
      const uint32_t dw = __builtin_amdgcn_readlane(bt.dw, t);
      const uint32_t *strip = xt_idx16_l + dw;
      w[t] = strip[lane];  // padded arrays
    }
  };
  auto accumulate = [&](const Batch &bt, const uint32_t (&w)[16]) {
    constexpr int SUB = ONES ? 16 : 8;  // the value stream costs 4 registers per strip
#pragma unroll
    for (int h = 0; h < 16; h += SUB) {
      double xa[ONES ? 1 : SUB], xb[ONES ? 1 : SUB];
      if (!ONES) {
#pragma unroll
        for (int t = 0; t < SUB; t++) {
          const uint32_t dw = __builtin_amdgcn_readlane(bt.dw, h + t);
          const double2 xv = *reinterpret_cast<const double2 *>(
              p.xt_val + 2u * (dw + static_cast<uint32_t>(lane)));
          xa[t] = xv.x;
          xb[t] = xv.y;
        }
      }
#pragma unroll
      for (int t = 0; t < SUB; t++) {
        // (a scalar fast path for strips that lie wholly inside their slice - no lane masks - was
        // tried in round 3: 10.58 -> 11.48 ms; the branch per strip costs the counted vmcnt waits)
        const uint32_t ol = __builtin_amdgcn_readlane(bt.ol, h + t);
        const int o = (2 * lane - 1) + static_cast<int>(ol & 0xffffu);
        const int ln = static_cast<int>(ol >> 16);
        const uint32_t wd = w[h + t];
        const bool oka = static_cast<uint32_t>(o) < static_cast<uint32_t>(ln);
        const bool okb = o < ln - 1;
        // ONES: x == 1 exactly, x * y == y and the value stream is never read
        const double y_t = ACC32 ? 1.0 : readlane_f64(bt.y, h + t);
        add(wd & 0xffffu, oka, ONES ? y_t : __dmul_rn(xa[ONES ? 0 : t], y_t));
        add(wd >> 16, okb, ONES ? y_t : __dmul_rn(xb[ONES ? 0 : t], y_t));
      }
    }
  };
  // Pipeline: while batch b is added, the column words of batch b + 1 are in flight, and so
  // are the slice bounds of block k + 2 and the (u, y) of block k + 3.  The block-level
  // loads are issued with every batch (their results are used when the block advances): a
  // load under a branch would make hipcc drain the queue.
  const int64_t q00 = tb + 16 * wv;
  uint64_t *const stash = ACC32 ? nullptr : p.hi_stash + static_cast<size_t>(blockIdx.x) * TILE;
  for (pass = 0; pass < (wide ? 2 : 1); pass++) {
  if (!ACC32 && pass == 1) {  // first limbs out, accumulators back to zero
    __syncthreads();
    for (int i = tid; i < width; i += THREADS) {
      stash[i] = static_cast<uint64_t>(__double_as_longlong(acc[i]));
      acc[i] = 0.0;
    }
    __syncthreads();
  }
  if (q00 < te) {
    Block A, B;
    int64_t uA, uB, uC;
    double yC;
    bool vA, vB, vC;
    load_uy(q00, uA, A.y, vA);
    load_uy(q00 + stride, uB, B.y, vB);
    load_uy(q00 + 2 * stride, uC, yC, vC);
    load_bounds(uA, vA, A.lo, A.len);
    load_bounds(uB, vB, B.lo, B.len);
    count_strips(A);
    count_strips(B);
    Batch cur = make_batch(A.lo, A.len, A.y, A.pin, A.total, 0);
    uint32_t w0[16], w1[16];
    issue_idx(cur, w0);
    int j = 0;
    int64_t q0 = q00;
    // one batch: issue the next batch's words into `wn`, add the current one from `wc`; the two
    // buffers swap roles by calling it twice per trip (no 16-register copy per batch).  Returns
    // true when the row is finished.
    auto one_batch = [&](uint32_t (&wc)[16], uint32_t (&wn)[16]) -> bool {
      const bool adv = j + 1 >= A.nb;  // wave-uniform
      // The block-level loads (slice bounds of block k + 2, (u, y) of block k + 3) are issued HERE,
      // every iteration, and consumed in `if (adv)` after the batch has been added.  (Earlier
      // versions pinned them with an `asm("" : "+v")` use, which made the wave wait for them
      // on the spot - s_waitcnt vmcnt(0) at the top of the loop.  Without it the compiler keeps
      // them here and waits at the use; the measured time is the same, 8.45 ms: the loop is not
      // bound by that round trip either.)
      const uint32_t *tpN = xt_tptr_l + uC * tp_stride + tile;
      const uint32_t loN = tpN[0], hiN = tpN[1];
      const int64_t qD = q0 + 3 * stride + lane;
      const bool vD = lane < 16 && qD < te;
      const int64_t qDc = min(qD, te - 1);
      const int64_t uD = p.t_idx[qDc];
      double yD = 1.0;
      if (!ACC32) {
        yD = p.t_val[qDc];
        if (ONES && p.xt_rowval != nullptr) yD = __dmul_rn(p.xt_rowval[uD], yD);  // (as in load_uy)
      }
      const int jn = adv ? 0 : j + 1;
      const Batch nxt = make_batch(adv ? B.lo : A.lo, adv ? B.len : A.len, adv ? B.y : A.y,
                                   adv ? B.pin : A.pin, adv ? B.total : A.total, jn);
      issue_idx(nxt, wn);
      accumulate(cur, wc);
      if (adv) {
        A = B;
        B.lo = loN;
        B.len = vC ? static_cast<int>(hiN - loN) : 0;
        B.y = yC;
        count_strips(B);
        uC = uD;
        yC = yD;
        vC = vD;
        q0 += stride;
        if (q0 >= te) return true;
      }
      j = jn;
      cur = nxt;
      return false;
    };
    for (;;) {
      if (one_batch(w0, w1)) break;
      if (one_batch(w1, w0)) break;
    }
  }
  }  // pass
  // (the column norms of this thread's first eight columns are fetched while the slower
  // waves finish accumulating)
  constexpr int PER = TILE / THREADS;  // 16
  // (opaque to the optimiser: everything derived from it - 16 column offsets, their clamped
  // forms, LDS addresses - would otherwise be hoisted out of the persistent loop and held in
  // registers across the accumulation, which has none to spare)
  int cbase_ = wv * (PER * 64) + lane;
  asm volatile("" : "+v"(cbase_));
  const int cbase = cbase_;
  // (FAST: all sixteen - the accumulation's registers are dead by now - and the exact path of a
  // FAST kernel reloads its own)
  double nrm0[FAST ? PER : 8];
#pragma unroll
  for (int k = 0; k < (FAST ? PER : 8); k++) nrm0[k] = p.norms[c0 + min(cbase + 64 * k, width - 1)];
  __syncthreads();
  PHASE_MARK(1);

  // ---- 2. epilogue.  Thread (wave w, lane l) owns the columns w * 1024 + 64 k + l,
  //      k = 0..15, so (wave, k, lane) order is column order.  The similarity of every stored
  //      entry is written back (the winners' values are read from there) and its
  //      order-preserving key stays in registers: the selection below never reads LDS keys.
  static_assert(PER * THREADS == TILE && PER <= 32 && PER % 8 == 0, "column ownership (the `have` mask is 32 bits)");
  const double tstat = p.t_stat[r];
  const double fx_inv = 1.0 / fx_scale;  // a power of two: exact
  uint32_t have = 0;  // bit k: column k of this thread is a stored entry of the product row
  int32_t *cidx = p.cand_idx + static_cast<size_t>(bid) * p.tile_k;
  double *cval = p.cand_val + static_cast<size_t>(bid) * p.tile_k;
  bool fast_done = false;
  if constexpr (FAST) {
    // ---- 2F. approximate similarities (float32) of the stored columns: positive, so their bit
    //      patterns order like the values
    //      (the counters stay where they are, in the lower half of the accumulator block: the
    //      candidate list below lives in the upper half, and the exact path re-reads them)
    uint32_t k32[PER];
    int local_bad = 0;
    {
      const float tstat_f = static_cast<float>(tstat), shrink_f = p.shrink_f;
      const float alpha_f = p.alpha_f, beta_f = p.beta_f;
#pragma unroll
      for (int k = 0; k < PER; k++) {
        const int i = cbase + 64 * k;
        const uint32_t c32 = cnt[min(i, TILE - 1)];
        const bool st = i < width && c32 != 0u;
        if (st) have |= 1u << k;
        const float a = approx_epilogue(p, static_cast<float>(c32), static_cast<float>(nrm0[FAST ? k : 0]), tstat_f,
                                        shrink_f, alpha_f, beta_f);
        k32[k] = st ? __float_as_uint(a) : 0u;
        // (safety net behind the host's checks: a stored column whose approximation is not a
        // positive normal number - the error bound does not cover it - sends the pair to the
        // exact kernel)
        if (st && !(a >= 1.17549435e-38f && a <= 3.0e38f)) local_bad = 1;
      }
    }
    // (barriers are most of what is left of a pair's fixed cost: the statistics have their own
    // LDS words, the histogram is returned to zero by the wave that scans it, and the candidates
    // are counted by the compaction itself - two barriers per digit pass, five around them)
    __shared__ uint32_t f_and[16], f_or[16];
    __shared__ int32_t f_cnt[16];
    uint32_t a_and = ~0u, a_or = 0u;
#pragma unroll
    for (int k = 0; k < PER; k++) {
      if ((have >> k) & 1u) {
        a_and &= k32[k];
        a_or |= k32[k];
      }
    }
    int local = __popc(have);
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
      local += __shfl_xor(local, o, 64);
      a_and &= __shfl_xor(a_and, o, 64);
      a_or |= __shfl_xor(a_or, o, 64);
    }
    local_bad = __any(local_bad) ? 1 : 0;
    if (lane == 0) {
      f_and[wv] = a_and;
      f_or[wv] = a_or;
      f_cnt[wv] = local | (local_bad << 30);
    }
    if (tid < 256) hist[tid] = 0;
    if (tid == 0) sh_count = 0;
    __syncthreads();
    int n_stored = 0;
    a_and = ~0u;
    a_or = 0u;
    int any_bad = 0;
#pragma unroll
    for (int w = 0; w < NW; w++) {
      n_stored += f_cnt[w] & 0x3fffffff;
      any_bad |= f_cnt[w] >> 30;
      a_and &= f_and[w];
      a_or |= f_or[w];
    }
    if (any_bad) {  // (block-uniform)
      if (tid == 0) p.redo_list[atomicAdd(p.redo_count, 1)] = bid;
      continue;
    }
    const int n_sel = min(p.top_k, n_stored);
    if (n_sel == 0) {  // (block-uniform; the next pair starts with a barrier)
      if (tid == 0) p.cand_cnt[bid] = 0;
      continue;
    }
    PHASE_MARK(3);
    // ---- 3F. a lower bound `prefix` of the n_sel-th largest approximation: the radix select of
    //      the exact path on 32-bit keys (it stops as soon as the bin of that key is known)
    uint32_t prefix = 0;
    const bool take_all = n_sel >= n_stored;
    if (!take_all) {
      int need = n_sel;
      const uint32_t diff = a_and ^ a_or;
      if (diff == 0) {
        prefix = a_and;
      } else {
        // digits of up to 8 bits from the highest differing bit downwards (NOT byte aligned: the
        // top byte of a positive float is 7 exponent bits - a handful of values, i.e. a
        // same-address atomic storm in the histogram; starting at the first bit that differs
        // spreads the first digit over exponent AND mantissa bits)
        int top = 32 - __clz(static_cast<int>(diff));           // bits [top - 1 .. 0] may differ
        const int low_bit = __ffs(static_cast<int>(a_or)) - 1;  // below it every key is zero
        prefix = top >= 32 ? 0u : a_and & (~0u << top);
        while (top > 0) {
          const int lo = max(top - 8, 0);
          const uint32_t dmask = (1u << (top - lo)) - 1u;
          const uint32_t hi_mask = top >= 32 ? 0u : (~0u << top);
#pragma unroll
          for (int k = 0; k < PER; k++) {
            const bool in = ((have >> k) & 1u) && (k32[k] & hi_mask) == prefix;
            const uint32_t digit = (k32[k] >> lo) & dmask;
            const unsigned long long todo = __ballot(in);
            if (todo) {
              const int leader = __ffsll(static_cast<long long>(todo)) - 1;
              const uint32_t d = __builtin_amdgcn_readlane(digit, leader);
              const unsigned long long same = __ballot(in && digit == d);
              if (same == todo) {
                if (lane == leader) atomicAdd(&hist[d], static_cast<uint32_t>(__popcll(same)));
              } else if (in) {
                atomicAdd(&hist[digit], 1u);
              }
            }
          }
          __syncthreads();
          if (wv == 0) {
            // lane l scans the bins 255 - 4 l .. 252 - 4 l (descending digits) and clears them
            int bn[4], sum = 0;
#pragma unroll
            for (int j = 0; j < 4; j++) {
              bn[j] = static_cast<int>(hist[255 - 4 * lane - j]);
              hist[255 - 4 * lane - j] = 0u;
              sum += bn[j];
            }
            int incl = sum;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
              const int t = __shfl_up(incl, o, 64);
              if (lane >= o) incl += t;
            }
            int run = incl - sum;  // keys in higher bins
#pragma unroll
            for (int j = 0; j < 4; j++) {
              if (run < need && run + bn[j] >= need) {
                sh_prefix = prefix | (static_cast<uint32_t>(255 - 4 * lane - j) << lo);
                sh_need = (run + bn[j] == need) ? -1 : need - run;
                sh_total = n_sel - need + run + bn[j];  // keys >= the lower bound of this bin
              }
              run += bn[j];
            }
          }
          __syncthreads();
          prefix = static_cast<uint32_t>(sh_prefix);
          need = sh_need;
          const int above = sh_total;
          top = lo;
          // every key of the chosen bin is >= prefix, which is all a lower bound needs: stop when
          // the bin is wanted as a whole, no lower digit differs, or few enough keys lie at or
          // above it (they all become candidates)
          if (need < 0 || top <= low_bit || above <= n_sel + FAST_SLACK) break;
        }
      }
    }
    // candidates: approximation >= prefix (1 - 2^-17), one more unit in the last place for the
    // rounding of this very product
    uint32_t thr = 0u;
    if (!take_all) {
      const float lf = __uint_as_float(prefix) * (1.0f - 0x1p-17f);
      thr = __float_as_uint(lf);
      thr = thr > 0u ? thr - 1u : 0u;
    }
    uint32_t candmask = 0;
#pragma unroll
    for (int k = 0; k < PER; k++)
      if (((have >> k) & 1u) && k32[k] >= thr) candmask |= 1u << k;
    PHASE_MARK(4);
    {
      // (upper half of the accumulator block, behind the 256 bytes of sinks: idle on the count path)
      uint64_t *candK = reinterpret_cast<uint64_t *>(smem + TILE * 4 + 1024);
      uint32_t *candC = reinterpret_cast<uint32_t *>(candK + FAST_CAP);
      uint32_t *candV = candC + FAST_CAP;  // the count
      // owners hand (column, count) to the list; thread t then takes candidate t: ONE gathered
      // norm load and one fp64 epilogue per candidate, all of them side by side (under the
      // owners' branches every load would expose its own latency)
      {
        // one returning atomic per wave: the wave's candidates take consecutive slots
        int mine = __popc(candmask), total = mine;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
          const int t = __shfl_up(total, o, 64);
          if (lane >= o) total += t;
        }
        int base = 0;
        if (lane == 63) base = total > 0 ? atomicAdd(&sh_count, total) : 0;
        base = __shfl(base, 63, 64) + total - mine;
#pragma unroll
        for (int k = 0; k < PER; k++) {
          if ((candmask >> k) & 1u) {
            if (base < FAST_CAP) {
              candC[base] = static_cast<uint32_t>(cbase + 64 * k);
              candV[base] = cnt[cbase + 64 * k];
            }
            base++;
          }
        }
      }
      __syncthreads();
      const int n_cand = sh_count;  // (the compaction counted them)
#ifdef IRS_KNN_PHASES
      if (tid == 0) {
        atomicAdd(&knn_fast_stat[0], 1ull);
        if (n_cand > FAST_CAP) atomicAdd(&knn_fast_stat[1], 1ull);
        atomicAdd(&knn_fast_stat[2], static_cast<unsigned long long>(n_cand));
        if (n_cand <= 128) atomicAdd(&knn_fast_stat[3], 1ull);
        if (n_cand <= 256) atomicAdd(&knn_fast_stat[4], 1ull);
        if (n_cand <= 512) atomicAdd(&knn_fast_stat[5], 1ull);
        if (n_cand <= 1024) atomicAdd(&knn_fast_stat[6], 1ull);
      }
#endif
      if (n_cand <= FAST_CAP) {  // (block-uniform)
      for (int t = tid; t < n_cand; t += THREADS) {
        const double sv = epilogue(p, static_cast<double>(candV[t]), p.norms[c0 + candC[t]], tstat);
        candK[t] = order_key(sv);
      }
      __syncthreads();
      // rank = candidates that go before (value desc, column asc): eight threads per candidate,
      // each against every eighth entry of the list.  The winners leave in RANK order (the row
      // merge sorts the tiles' lists anyway; calls whose merge relies on column-sorted lists,
      // top_k > TOPK_CAP, do not take this path).
      constexpr int TPC = 8, CPR = THREADS / TPC;  // candidates per round
      int tid_ = tid;
      asm volatile("" : "+v"(tid_));  // (nothing derived from it is worth a register across the accumulation)
      const int sub = tid_ & (TPC - 1);
      for (int t0 = 0; t0 < n_cand; t0 += CPR) {
        const int t = t0 + tid_ / TPC;
        const bool live = t < n_cand;
        const uint64_t mk = candK[live ? t : 0];
        const uint32_t mc = candC[live ? t : 0];
        int before = 0;
        for (int q = sub; q < n_cand; q += TPC) {
          const uint64_t ok = candK[q];
          const uint32_t oc = candC[q];
          before += (ok > mk || (ok == mk && oc < mc)) ? 1 : 0;
        }
        before += __shfl_xor(before, 1, 64);
        before += __shfl_xor(before, 2, 64);
        before += __shfl_xor(before, 4, 64);
        if (live && sub == 0 && before < n_sel) {
          cidx[before] = c0 + static_cast<int>(mc);
          cval[before] = key_to_double(mk);  // (order_key is invertible on the finite values)
        }
      }
      if (tid == 0) p.cand_cnt[bid] = n_sel;
      fast_done = true;
      }
      PHASE_MARK(5);
    }
  }
  if constexpr (FAST) {
    // (rare: thousands of near-ties at the threshold) the pair goes to the exact kernel's list
    if (!fast_done && tid == 0) p.redo_list[atomicAdd(p.redo_count, 1)] = bid;
  } else {
  uint64_t key[PER];
  have = 0;
  uint32_t cv[ACC32 ? PER : 1];
  if (ACC32) {
    // count i sits in the bytes of acc[i / 2]: every count is read before any sum is written
#pragma unroll
    for (int k = 0; k < PER; k++) {
      const int i = cbase + 64 * k;
      cv[k] = cnt[min(i, TILE - 1)];
      if (i < width && cv[k] != 0u) have |= 1u << k;
    }
    __syncthreads();
  }
#pragma unroll
  for (int h = 0; h < PER; h += 8) {  // eight columns at a time (register budget)
    double nrm[8], raw[8];
#pragma unroll
    for (int k = 0; k < 8; k++)
      nrm[k] = h == 0 ? nrm0[k] : p.norms[c0 + min(cbase + 64 * (h + k), width - 1)];
#pragma unroll
    for (int k = 0; k < 8; k++) {
      const int i = cbase + 64 * (h + k);
      if (ACC32) {
        raw[k] = static_cast<double>(cv[ACC32 ? h + k : 0]);
      } else {
        // fixed-point sum -> double (one rounding).  Sentinel: the accumulator started at the
        // bit pattern of -0.0 = INT64_MIN, the sum is the difference.
        uint64_t fx = static_cast<uint64_t>(__double_as_longlong(acc[min(i, TILE - 1)]));
        double low = 0.0;
        if (wide) {  // (block-uniform) the accumulator holds the second limb, the stash the first
          low = static_cast<double>(static_cast<long long>(fx)) * 0x1p-40;
          fx = stash[min(i, TILE - 1)];
        }
        raw[k] = (static_cast<double>(static_cast<long long>(fx ^ (SENTINEL ? NEG_ZERO_BITS : 0ull))) + low) *
                 fx_inv;
        const bool st = SENTINEL ? fx != NEG_ZERO_BITS
                                 : ((bits[min(i, TILE - 1) >> 5] >> (i & 31)) & 1u) != 0u;
        if (i < width && st) have |= 1u << (h + k);
      }
    }
    if (!FAST && p.col_scale != nullptr) {  // (block-uniform)
#pragma unroll
      for (int k = 0; k < 8; k++)
        raw[k] = __dmul_rn(raw[k], p.col_scale[c0 + min(cbase + 64 * (h + k), width - 1)]);
    }
#pragma unroll
    for (int k = 0; k < 8; k++) {
      const double sv = epilogue(p, raw[k], nrm[k], tstat);
      key[h + k] = order_key(sv);
      if ((have >> (h + k)) & 1u) acc[cbase + 64 * (h + k)] = sv;
    }
  }
  // stored entries of the tile, and the bits in which their keys differ (the radix select
  // starts at the first digit that is not common to all of them)
  uint64_t k_and = ~0ull, k_or = 0ull;
#pragma unroll
  for (int k = 0; k < PER; k++) {
    if ((have >> k) & 1u) {
      k_and &= key[k];
      k_or |= key[k];
    }
  }
  int local = __popc(have);
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) {
    local += __shfl_xor(local, o, 64);
    k_and &= __shfl_xor(k_and, o, 64);
    k_or |= __shfl_xor(k_or, o, 64);
  }
  uint64_t *red = reinterpret_cast<uint64_t *>(hist);  // 16 x (and, or) + counts behind
  if (lane == 0) {
    red[2 * wv] = k_and;
    red[2 * wv + 1] = k_or;
    wave_cnt[wv] = local;
  }
  __syncthreads();
  int n_stored = 0;
  k_and = ~0ull;
  k_or = 0ull;
#pragma unroll
  for (int w = 0; w < NW; w++) {
    n_stored += wave_cnt[w];
    k_and &= red[2 * w];
    k_or |= red[2 * w + 1];
  }
  __syncthreads();  // red / wave_cnt are reused below
  const int n_sel = min(p.top_k, n_stored);
  if (tid == 0) p.cand_cnt[bid] = n_sel;
  if (n_sel == 0) continue;
  PHASE_MARK(3);

  // ---- 3. radix select of the n_sel-th largest key (value desc, column asc on ties):
  //      `prefix` becomes that key and `need` the number of its ties that are taken.
  uint64_t prefix = 0;
  int need = n_sel;
  const bool take_all = n_sel >= n_stored;
  if (!take_all) {
    const uint64_t diff = k_and ^ k_or;
    if (diff == 0) {
      prefix = k_and;  // all stored keys are equal: everything ties
    } else {
      int shift = ((63 - __clzll(static_cast<long long>(diff))) >> 3) << 3;
      const int low_shift = ((__ffsll(static_cast<long long>(k_or)) - 1) >> 3) << 3;
      prefix = shift + 8 >= 64 ? 0ull : k_and & (~0ull << (shift + 8));
      for (; shift >= 0; shift -= 8) {
        if (tid < 256) hist[tid] = 0;
        __syncthreads();
        const uint64_t hi_mask = (shift + 8 >= 64) ? 0ull : (~0ull << (shift + 8));
#pragma unroll
        for (int k = 0; k < PER; k++) {
          const bool in = ((have >> k) & 1u) && (key[k] & hi_mask) == prefix;
          const uint32_t digit = static_cast<uint32_t>(key[k] >> shift) & 0xffu;
          // similarities share their leading bytes: when every candidate lane of the wave
          // has the same digit, one lane adds the count (no same-address atomic storm);
          // otherwise the digits are spread and per-lane atomics are cheap
          const unsigned long long todo = __ballot(in);
          if (todo) {
            const int leader = __ffsll(static_cast<long long>(todo)) - 1;
            const uint32_t d = __builtin_amdgcn_readlane(digit, leader);
            const unsigned long long same = __ballot(in && digit == d);
            if (same == todo) {
              if (lane == leader) atomicAdd(&hist[d], static_cast<uint32_t>(__popcll(same)));
            } else if (in) {
              atomicAdd(&hist[digit], 1u);
            }
          }
        }
        __syncthreads();
        // the digit d with  #(digits > d) < need <= #(digits >= d): scan the bins in
        // descending digit order with the first four waves
        int bin = 0, incl = 0;
        if (tid < 256) {
          bin = static_cast<int>(hist[255 - tid]);
          incl = bin;
#pragma unroll
          for (int o = 1; o < 64; o <<= 1) {
            const int t = __shfl_up(incl, o, 64);
            if (lane >= o) incl += t;
          }
          if (lane == 63) wave_cnt[wv] = incl;
        }
        __syncthreads();
        if (tid < 256) {
          for (int w = 0; w < wv; w++) incl += wave_cnt[w];
          const int excl = incl - bin;
          if (excl < need && incl >= need) {
            sh_prefix = prefix | (static_cast<uint64_t>(255 - tid) << shift);
            // the whole bin is wanted: every key >= the bin's lowest possible key is taken
            // and no further digit has to be resolved (-1 tells the loop to stop)
            sh_need = (incl == need) ? -1 : need - excl;
            sh_total = bin;
          }
        }
        __syncthreads();
        prefix = sh_prefix;
        need = sh_need;
        const int in_bin = sh_total;
        __syncthreads();
        // below the lowest set bit of any key every digit is zero: the prefix is the key
        if (need < 0 || shift == 0 || shift <= low_shift) break;
        if (in_bin <= 64) {
          // few candidates left: one wave ranks them instead of more digit passes
          const uint64_t lo_mask = ~0ull << shift;
          if (tid == 0) sh_count = 0;
          __syncthreads();
#pragma unroll
          for (int k = 0; k < PER; k++)
            if (((have >> k) & 1u) && (key[k] & lo_mask) == prefix)
              red[atomicAdd(&sh_count, 1)] = key[k];
          __syncthreads();
          if (wv == 0) {
            const uint64_t mine = lane < in_bin ? red[lane] : 0ull;
            int gt = 0, ge = 0;
            for (int q = 0; q < in_bin; q++) {
              const uint64_t other = red[q];
              gt += other > mine;
              ge += other >= mine;
            }
            if (lane < in_bin && gt < need && need <= ge) {  // equal keys write equal values
              sh_prefix = mine;
              sh_need = need - gt;
            }
          }
          __syncthreads();
          prefix = sh_prefix;
          need = sh_need;
          __syncthreads();
          break;
        }
      }
      if (need < 0) {
        if (prefix == 0) {  // lowest possible key: nothing is excluded, everything else ties
          need = n_sel;
        } else {
          prefix -= 1;  // k > prefix - 1  <=>  k >= prefix
          need = 0;
        }
      }
    }
  } else {
    need = 0;  // everything is taken
  }
  PHASE_MARK(4);
  // Ordered compaction: (wave, slot k, lane) order is column order, so the winners are
  // written in ascending column order without atomics and the tile's list is column-sorted
  // (the row merge for top_k > TOPK_CAP relies on it).  Pass A counts this wave's ties with
  // the threshold key; pass B ranks the ties in column order (the `need` lowest columns win,
  // knn.hpp:119-136) and counts this wave's winners; pass C writes them.
  const unsigned long long lt_mask = (1ull << lane) - 1ull;
  int my_ties = 0;
  if (!take_all && need > 0) {
#pragma unroll
    for (int k = 0; k < PER; k++)
      my_ties += __popcll(__ballot(((have >> k) & 1u) && key[k] == prefix));
  }
  __syncthreads();  // wave_cnt (stored entries per wave) has been read by everyone
  if (lane == 0) wave_cnt[wv] = my_ties;
  __syncthreads();
  int ties_before = 0;
  for (int w = 0; w < wv; w++) ties_before += wave_cnt[w];
  uint32_t winmask = 0;
  int my_wins = 0;
#pragma unroll
  for (int k = 0; k < PER; k++) {
    const bool st = (have >> k) & 1u;
    const bool tie = st && !take_all && key[k] == prefix;
    const unsigned long long bal = __ballot(tie);
    const int rank = ties_before + __popcll(bal & lt_mask);
    const bool win = st && (take_all || key[k] > prefix || (tie && rank < need));
    ties_before += __popcll(bal);
    if (win) winmask |= 1u << k;
    my_wins += __popcll(__ballot(win));
  }
  __syncthreads();  // the tie counts have been read
  if (lane == 0) wave_cnt[wv] = my_wins;
  __syncthreads();
  int pos0 = 0;
  for (int w = 0; w < wv; w++) pos0 += wave_cnt[w];
#pragma unroll
  for (int k = 0; k < PER; k++) {
    const bool win = (winmask >> k) & 1u;
    const unsigned long long bal = __ballot(win);
    if (win) {
      const int pos = pos0 + __popcll(bal & lt_mask);
      cidx[pos] = c0 + cbase + 64 * k;
      cval[pos] = acc[cbase + 64 * k];
    }
    pos0 += __popcll(bal);
  }
  PHASE_MARK(5);
  }  // exact path
  }  // persistent loop
}

// One 256-thread workgroup per target row: union of the tile winners, keep the
// top_k by (value desc, column asc), emit them sorted by column (knn.hpp:119-136).
// The tiles are merged in rounds: as many tiles as fit next to the best-so-far are appended,
// the buffer is sorted and cut back to top_k (any number of tiles, top_k <= MERGE_CAP / 2).
__global__ __launch_bounds__(256) void knn_merge_kernel(Params p) {
  // merge_cap entries (a power of two sized to the request, at most MERGE_CAP): 20 B each
  extern __shared__ __attribute__((aligned(16))) unsigned char merge_smem[];
  uint64_t *key = reinterpret_cast<uint64_t *>(merge_smem);
  double *val = reinterpret_cast<double *>(key + p.merge_cap);
  int32_t *idx = reinterpret_cast<int32_t *>(val + p.merge_cap);
  const int tid = threadIdx.x;
  const int slot = blockIdx.x;
  auto swap_el = [&](int a, int b) {
    const uint64_t tk = key[a]; key[a] = key[b]; key[b] = tk;
    const double tv = val[a]; val[a] = val[b]; val[b] = tv;
    const int32_t ti = idx[a]; idx[a] = idx[b]; idx[b] = ti;
  };
  int n = 0;  // entries currently in the buffer (the best so far after a round)
  int t = 0;
  do {
    // append tiles while they fit (cand_cnt <= top_k <= MERGE_CAP / 2: at least one fits)
    while (t < p.n_tiles) {
      const int b = slot * p.n_tiles + t;
      const int c = p.cand_cnt[b];
      if (n + c > p.merge_cap) break;
      for (int i = tid; i < c; i += 256) {
        const double v = p.cand_val[static_cast<size_t>(b) * p.tile_k + i];
        val[n + i] = v;
        key[n + i] = order_key(v);
        idx[n + i] = p.cand_idx[static_cast<size_t>(b) * p.tile_k + i];
      }
      n += c;
      t++;
    }
    int n_pow = 1;
    while (n_pow < n) n_pow <<= 1;
    for (int i = n + tid; i < n_pow; i += 256) {
      key[i] = 0ull;
      val[i] = 0.0;
      idx[i] = 0x7fffffff;
    }
    __syncthreads();
    // (value desc, column asc); padding (column = INT_MAX) sorts last among equal keys
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
    n = min(n, p.top_k);  // real entries with key 0 (NaN) outrank padding by column
  } while (t < p.n_tiles);
  const int keep = n;
  int k_pow = 1;
  while (k_pow < keep) k_pow <<= 1;
  for (int i = keep + tid; i < k_pow; i += 256) idx[i] = 0x7fffffff;
  __syncthreads();
  // the kept entries by column
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

// Row merge for top_k > TOPK_CAP (e.g. P3alpha / RP3beta with their default top_k = None, or a
// user-kNN over every user): the tiles' winner lists are column-sorted and the tiles are in
// column order, so the row only needs the top_k-th best key as a threshold - an MSB-first
// 8-bit radix select over the keys in global memory - and one ordered compaction: everything
// above the threshold plus the `need` lowest-column ties (knn.hpp:119-136).  The output comes
// out sorted by column.  One 256-thread workgroup per target row.
__global__ __launch_bounds__(256) void knn_merge_big_kernel(Params p) {
  __shared__ int32_t hist[256];
  __shared__ int32_t wsum[4];
  __shared__ uint64_t sh_prefix;
  __shared__ int32_t sh_need;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int slot = blockIdx.x;
  const int32_t *cnt = p.cand_cnt + static_cast<size_t>(slot) * p.n_tiles;
  int n_total = 0;
  for (int t = 0; t < p.n_tiles; t++) n_total += cnt[t];
  int32_t *oidx = p.out_idx + static_cast<size_t>(slot) * p.top_k;
  double *oval = p.out_val + static_cast<size_t>(slot) * p.top_k;
  const size_t cbase = static_cast<size_t>(slot) * p.n_tiles * p.tile_k;
  uint64_t thr = 0;
  int need = 0;
  const bool all = n_total <= p.top_k;
  if (!all) {
    uint64_t prefix = 0;
    int remaining = p.top_k;  // rank of the wanted key among those matching the prefix so far
    for (int shift = 56; shift >= 0; shift -= 8) {
      hist[tid] = 0;
      __syncthreads();
      const uint64_t hi_mask = shift == 56 ? 0ull : (~0ull << (shift + 8));
      for (int t = 0; t < p.n_tiles; t++) {
        const double *v = p.cand_val + cbase + static_cast<size_t>(t) * p.tile_k;
        for (int i = tid; i < cnt[t]; i += 256) {
          const uint64_t k = order_key(v[i]);
          if ((k & hi_mask) == prefix) atomicAdd(&hist[(k >> shift) & 255u], 1);
        }
      }
      __syncthreads();
      if (tid == 0) {
        int d = 255, above = 0;
        while (above + hist[d] < remaining) { above += hist[d]; d--; }
        sh_prefix = prefix | (static_cast<uint64_t>(d) << shift);
        sh_need = remaining - above;
      }
      __syncthreads();
      prefix = sh_prefix;
      remaining = sh_need;
      __syncthreads();
    }
    thr = prefix;
    need = remaining;  // ties with the threshold key that are taken (lowest columns)
  }
  // ordered compaction, 256 entries per round in list order
  int out_n = 0, ties_seen = 0;
  for (int t = 0; t < p.n_tiles; t++) {
    const double *v = p.cand_val + cbase + static_cast<size_t>(t) * p.tile_k;
    const int32_t *ix = p.cand_idx + cbase + static_cast<size_t>(t) * p.tile_k;
    for (int i0 = 0; i0 < cnt[t]; i0 += 256) {
      const int i = i0 + tid;
      const bool in = i < cnt[t];
      const double val = in ? v[i] : 0.0;
      const int32_t col = in ? ix[i] : 0;
      const uint64_t k = order_key(val);
      const bool tie = in && !all && k == thr;
      // rank of this tie among the row's ties, in column order
      const unsigned long long tb = __ballot(tie);
      if (lane == 0) wsum[wv] = __popcll(tb);
      __syncthreads();
      int tbefore = ties_seen;
      for (int w = 0; w < wv; w++) tbefore += wsum[w];
      const int tie_total = wsum[0] + wsum[1] + wsum[2] + wsum[3];
      __syncthreads();
      const int trank = tbefore + __popcll(tb & ((1ull << lane) - 1ull));
      const bool win = in && (all || k > thr || (tie && trank < need));
      const unsigned long long wb = __ballot(win);
      if (lane == 0) wsum[wv] = __popcll(wb);
      __syncthreads();
      int wbefore = out_n;
      for (int w = 0; w < wv; w++) wbefore += wsum[w];
      const int win_total = wsum[0] + wsum[1] + wsum[2] + wsum[3];
      __syncthreads();
      if (win) {
        const int pos = wbefore + __popcll(wb & ((1ull << lane) - 1ull));
        oidx[pos] = col;
        oval[pos] = val;
      }
      out_n += win_total;
      ties_seen += tie_total;
    }
  }
  if (tid == 0) p.out_cnt[slot] = out_n;
}

// The merged rows (work-ordered slots, top_k entries apart) -> one contiguous CSR in target-row
// order on the device, so that the result crosses PCIe once, straight into the caller's arrays.
__global__ __launch_bounds__(256) void knn_compact_kernel(const int32_t *__restrict__ out_idx,
                                                          const double *__restrict__ out_val,
                                                          const int32_t *__restrict__ slot_of,
                                                          const int64_t *__restrict__ res_ptr,
                                                          int64_t n_rows, int32_t top_k,
                                                          int32_t *__restrict__ dst_idx,
                                                          double *__restrict__ dst_val) {
  const int64_t row = static_cast<int64_t>(blockIdx.x) * 4 + wave_index_in_block();
  if (row >= n_rows) return;
  const int lane = threadIdx.x & 63;
  const int64_t b = res_ptr[row], e = res_ptr[row + 1];
  const size_t src = static_cast<size_t>(slot_of[row]) * top_k;
  for (int64_t i = lane; i < e - b; i += 64) {
    dst_idx[b + i] = out_idx[src + i];
    dst_val[b + i] = out_val[src + i];
  }
}

// irs_knn_create on the device (all-ones matrices): the slice pointers and the packed 16-bit offsets
// of X_arg^T from its device-resident transpose (transpose_csr_device, device_sort.hip).
// xt_ptr[u] = first entry of feature row u; tptr[u][t] = first entry of the row with column >= t TILE.
__global__ __launch_bounds__(256) void xt_slices_kernel(const uint32_t *__restrict__ xt_ptr,
                                                        const int32_t *__restrict__ t_idx, int64_t n_rows,
                                                        int32_t n_tiles, uint32_t *__restrict__ tptr) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  if (i >= n_rows * (n_tiles + 1)) return;
  const int64_t u = i / (n_tiles + 1);
  const int32_t t = static_cast<int32_t>(i % (n_tiles + 1));
  const uint32_t b = xt_ptr[u], e = xt_ptr[u + 1];
  if (t == n_tiles) {
    tptr[i] = e;
    return;
  }
  const int32_t key = t * TILE;
  uint32_t lo = b, hi = e;  // the first position whose column is >= key
  while (lo < hi) {
    const uint32_t mid = lo + ((hi - lo) >> 1);
    if (t_idx[mid] < key) lo = mid + 1;
    else hi = mid;
  }
  tptr[i] = lo;
}

// per feature row of X_arg^T: largest and smallest non-zero |x| (0 when the row has none): the bounds
// the target pass builds the fixed-point scales from
// ... and the row's first value + whether any row holds two different ones (bit patterns compared)
__global__ __launch_bounds__(256) void xt_row_range_kernel(const uint32_t *__restrict__ xt_ptr,
                                                           const double *__restrict__ xt_val, int64_t n_rows,
                                                           double *__restrict__ row_max, double *__restrict__ row_min,
                                                           double *__restrict__ row_val, int32_t *__restrict__ not_const) {
  const int64_t u = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  if (u >= n_rows) return;
  double mx = 0.0, mn = __longlong_as_double(0x7ff0000000000000ll);
  const uint32_t b = xt_ptr[u], e = xt_ptr[u + 1];
  const double first = b < e ? xt_val[b] : 0.0;
  bool same = true;
  for (uint32_t q = b; q < e; q++) {
    const double v = xt_val[q], a = fabs(v);
    mx = fmax(mx, a);
    if (a > 0.0) mn = fmin(mn, a);
    same &= __double_as_longlong(v) == __double_as_longlong(first);
  }
  row_max[u] = mx;
  row_min[u] = isfinite(mn) ? mn : 0.0;
  row_val[u] = first;
  if (!same) atomicOr(not_const, 1);
}

// two tile-relative LDS byte offsets (column x 4, 16 bits each) per dword; zeros behind the last entry
__global__ __launch_bounds__(256) void xt_pack16_kernel(const int32_t *__restrict__ t_idx, int64_t nnz,
                                                        int64_t n_words, uint32_t *__restrict__ idx_p) {
  const int64_t w = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  if (w >= n_words) return;
  const int64_t q = 2 * w;
  const uint32_t lo = q < nnz ? static_cast<uint32_t>((t_idx[q] % TILE) * 4) : 0u;
  const uint32_t hi = q + 1 < nnz ? static_cast<uint32_t>((t_idx[q + 1] % TILE) * 4) : 0u;
  idx_p[w] = lo | (hi << 16);
}


// ---------------------------------------------------------------------------------------------------------
// Feature weighting on the device (util.hpp:159-209; host tables: knn_host_prep.hpp weight_tables).
// One wave per stored row of the matrix (its documents): w[q] = v * idf[col] (tf-idf, :206) or
// idf[col] * (v * (k1 + 1)) / (v + reg[row]) (BM25, :183-184) - each operation rounded once, in the
// reference's order (__dmul_rn / __dadd_rn / __ddiv_rn: no contraction), so the values are the host
// loop's bit for bit.  `vals` null: every stored value is 1.  flags (or-ed): 1 some weight != 1,
// 2 some |weight| outside (1e-150, 1e150) (or zero), 4 some weight <= 0 - what the host pass of
// irs_knn_create derives from the caller's values when it does not weight.
template <bool BM25>
__global__ __launch_bounds__(256) void knn_weight_kernel(const int32_t *__restrict__ indptr,
                                                         const int32_t *__restrict__ indices,
                                                         const double *__restrict__ vals,
                                                         const double *__restrict__ idf,
                                                         const double *__restrict__ reg, double k1p1,
                                                         int64_t n_rows, double *__restrict__ out,
                                                         int32_t *__restrict__ flags) {
  const int64_t r = static_cast<int64_t>(blockIdx.x) * 4 + wave_index_in_block();
  if (r >= n_rows) return;
  const int lane = threadIdx.x & 63;
  const int32_t b = indptr[r], e = indptr[r + 1];
  const double rg = BM25 ? reg[r] : 0.0;
  int32_t f = 0;
  for (int32_t q = b + lane; q < e; q += 64) {
    const double v = vals ? vals[q] : 1.0;
    const double w = idf[indices[q]];
    double o;
    if (BM25) o = __ddiv_rn(__dmul_rn(w, __dmul_rn(v, k1p1)), __dadd_rn(v, rg));
    else o = __dmul_rn(v, w);
    out[q] = o;
    const double a = fabs(o);
    f |= (o != 1.0 ? 1 : 0) | (!(a > 1e-150 && a < 1e150) ? 2 : 0) | (!(o > 0.0) ? 4 : 0);
  }
  if (flags) {
    f = (__ballot(f & 1) ? 1 : 0) | (__ballot(f & 2) ? 2 : 0) | (__ballot(f & 4) ? 4 : 0);  // wave-wide or
    if (lane == 0 && f && (__hip_atomic_load(flags, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & f) != f)
      atomicOr(flags, f);
  }
}

// ss[r] = the squares of row r's values added one by one in storage order, product and sum rounded
// separately (the `ss += x * x` loop of the norms, similarities.hpp:20-28 / :61-72, compiled without
// contraction).  One wave per row: the lanes load and square a strip of 64 values, then every lane runs
// the same 64-step chain over the strip's squares (lane broadcasts) - the order of the sum is the
// sequential one, only the loads are parallel.  Entries past the row's end add +0.0, which changes no
// partial sum (they are >= +0.0).
__global__ __launch_bounds__(256) void knn_row_sumsq_kernel(const uint32_t *__restrict__ ptr,
                                                            const double *__restrict__ vals, int64_t n_rows,
                                                            double *__restrict__ ss) {
  const int64_t r = static_cast<int64_t>(blockIdx.x) * 4 + wave_index_in_block();
  if (r >= n_rows) return;
  const int lane = threadIdx.x & 63;
  const uint32_t b = ptr[r], e = ptr[r + 1];
  double sum = 0.0;
  for (uint32_t q0 = b; q0 < e; q0 += 64) {
    const uint32_t q = q0 + lane;
    const double v = q < e ? vals[q] : 0.0;
    const double sq = __dmul_rn(v, v);
    const uint32_t n = min(64u, e - q0);
    if (n == 64) {
#pragma unroll
      for (int j = 0; j < 64; j++) sum = __dadd_rn(sum, __shfl(sq, j));
    } else {
      for (uint32_t j = 0; j < n; j++) sum = __dadd_rn(sum, __shfl(sq, static_cast<int>(j)));
    }
  }
  if (lane == 0) ss[r] = sum;
}

// the padded value stream of X_arg^T from values that are already on the device (zeros behind the last entry)
__global__ __launch_bounds__(256) void knn_pad_copy_kernel(const double *__restrict__ src, int64_t nnz,
                                                           int64_t padded, double *__restrict__ dst) {
  const int64_t q = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  if (q < padded) dst[q] = q < nnz ? src[q] : 0.0;
}


// remove_diagonal (util.hpp:211-226) on the TRANSPOSED result: column c's entries hold ascending row numbers;
// the one whose row is c - row0 (if stored) is set to 0.0 and kept.  One thread per column.
__global__ __launch_bounds__(256) void knn_zero_diagonal_csc_kernel(const int32_t *__restrict__ col_ptr,
                                                                    const int32_t *__restrict__ row_idx,
                                                                    int64_t n_cols, int64_t row0,
                                                                    double *__restrict__ val) {
  const int64_t c = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  if (c >= n_cols) return;
  const int64_t want = c - row0;
  int32_t lo = col_ptr[c], hi = col_ptr[c + 1];
  while (lo < hi) {
    const int32_t mid = lo + ((hi - lo) >> 1);
    if (row_idx[mid] < want) lo = mid + 1;
    else hi = mid;
  }
  if (lo < col_ptr[c + 1] && row_idx[lo] == want) val[lo] = 0.0;
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
  DeviceBuffer<uint32_t> xt_tptr;  // [n_features, n_tiles + 1], see Params
  DeviceBuffer<uint32_t> xt_idx16;
  DeviceBuffer<double> xt_val, norms;
  DeviceBuffer<double> xt_rowval;  // see Params (allocated only when every feature row is constant)
  bool xt_row_const = false;
  bool xt_all_ones = false;
  DeviceBuffer<double> col_scale;  // see Params (allocated only for a separable fused weighting)
  bool col_scaled = false;
  double norm_max = 0.0;  // largest column norm (the approximate selection needs counts below 2^24 for Tversky)
  bool xt_nonzero = false;  // |x| in (1e-150, 1e150) for every stored x
  bool xt_positive = false; // every stored x > 0
  std::vector<double> xt_rowmax;  // host: max |x| per feature row (bound of a target row's sums)
  std::vector<double> xt_rowmin;  // host: min |x| per feature row (smallest product of a target row)
  // last result (host)
  std::vector<int64_t> res_ptr;
  DeviceBuffer<int32_t> res_idx;  // the last result stays on the device until irs_knn_fetch
  DeviceBuffer<double> res_val;
  int64_t res_nnz = 0;
  double last_ms = 0;
  int64_t last_macs = 0, last_walked = 0;
  // scratch of a compute call, kept between calls (hipMalloc / hipFree of ~200 MB per call
  // cost more than a millisecond, and hipFree synchronises the device)
  struct Scratch {
    DeviceBuffer<int64_t> t_ptr, res_ptr;
    DeviceBuffer<int32_t> t_idx, order, cand_idx, cand_cnt, out_idx, out_cnt, cursor, slot_of, redo_list;
    DeviceBuffer<double> t_val, t_stat, t_scale, cand_val, out_val;
    DeviceBuffer<uint64_t> hi_stash;
    // ONE allocation behind the buffers every call needs (carved per call; t_val / hi_stash, which
    // only some calls need, are allocations of their own): seventeen hipMalloc calls were 11 ms of the
    // first call of a computer - and `learn()` of a kNN recommender makes exactly one call
    DeviceBuffer<char> arena;
  } scratch;
  // a compute call: set-up, the counts of the finished chunks and the end of the call; the kernels of its
  // row chunks, one chunk after the other; the chunks' small inputs; the column indices (uploader
  // thread); compaction + device-to-host copy of the chunks that are done.
  // (Measured and dropped, round 5: the chunks' tile kernels on two or three streams in turn with the
  // merges on a stream of their own, so that a chunk's last heavy pairs overlap the next chunk.  The
  // ML-20M call went from 9.5 ms (four chunks, one stream) to 11.8 ms of device time: the 130 KB
  // workgroups of two persistent kernels and the small merge blocks take each other's compute units.)
  hipStream_t stream = nullptr, stream_k = nullptr, stream_in = nullptr, stream_up = nullptr, stream_out = nullptr;
  hipEvent_t ev_in = nullptr, ev_setup = nullptr;
  // the last result once more in page-locked host memory (kept between calls): the device-to-host copy
  // runs at the link's rate at the end of the compute call, irs_knn_fetch is then a multi-threaded
  // host copy into the caller's (pageable, usually untouched) arrays
  char *stage = nullptr;
  size_t stage_bytes = 0, stage_idx_offset = 0;  // values at 0, column indices at stage_idx_offset
  bool staged = false;
  int64_t n_calls = 0;  // compute calls so far
  ~irs_knn_computer() {
    if (stage) (void)hipHostFree(stage);
    if (ev_in) (void)hipEventDestroy(ev_in);
    if (ev_setup) (void)hipEventDestroy(ev_setup);
    for (hipStream_t st : {stream, stream_k, stream_in, stream_up, stream_out})
      if (st) (void)hipStreamDestroy(st);
  }
};

// The arena and the page-locked staging buffer of the last computer that was destroyed, per device: a
// tuning loop constructs, uses once and drops one computer after the other, and mapping ~250 MB of
// device memory (14 ms) + page-locking 32 MB (6 ms) cost more than the ML-20M call itself (11 ms).
namespace {
struct KnnBufferCache {
  std::mutex mu;
  struct Slot {
    int device = -1;
    char *arena = nullptr;
    size_t arena_bytes = 0;
    char *stage = nullptr;
    size_t stage_bytes = 0;
    // (five stream creations are another 5 - 10 ms of a computer's first call)
    hipStream_t streams[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    hipEvent_t ev_in = nullptr, ev_setup = nullptr;
  } slot;
  ~KnnBufferCache() {  // (process exit: the runtime may be gone already - nothing to free safely)
  }
} knn_buffer_cache;
}  // namespace

extern "C" {

// host threads of the target pass (IRSPACK_AMD_KNN_THREADS overrides)
static int64_t host_thread_cap() {
  const int64_t cap = [] {  // (read per call: tests toggle it)
    const char *e = std::getenv("IRSPACK_AMD_KNN_THREADS");
    return e ? std::max<int64_t>(1, std::atoll(e)) : int64_t(32);
  }();
  return cap;
}

// IRSPACK_AMD_KNN_TIMING=1 prints the host phases of a compute call to stderr
struct PhaseTimer {
  bool on = std::getenv("IRSPACK_AMD_KNN_TIMING") != nullptr;
  std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
  void mark(const char *what) {
    if (!on) return;
    const auto t1 = std::chrono::steady_clock::now();
    fprintf(stderr, "knn host phase %-18s %8.2f ms\n", what,
            std::chrono::duration<double, std::milli>(t1 - t0).count());
    t0 = t1;
  }
};

irs_status irs_knn_create(int32_t sim_type, int64_t rows, int64_t cols, const int64_t *indptr,
                          const int32_t *indices, const double *data, const irs_knn_input *input,
                          double shrinkage, double alpha, double beta, int32_t normalize, int64_t n_threads,
                          int64_t max_chunk_size, int32_t device, irs_knn_computer **out) {
  return guard([&] {
    check_arg(out != nullptr, "null argument.");
    // KNNComputer ctor, knn.hpp:35-38
    check_lower(shrinkage, 0, "shrinkage");
    check_arg(n_threads >= 1, "n_threads must be greater than or equal to  1");
    check_arg(max_chunk_size >= 1, "max_chunk_size must be greater than or equal to  1");
    PhaseTimer pt;
    // (the parameter checks of the similarities' constructors first: similarities.hpp:61-72, 143-159, 198-222, 265-292)
    switch (sim_type) {
      case IRS_SIM_COSINE:
      case IRS_SIM_JACCARD:
        break;
      case IRS_SIM_ASYMMETRIC:
        check_lower(alpha, 0, "alpha");
        if (alpha > 1)
          throw std::invalid_argument("alpha must be less than or equal to  " + std::to_string(1.0));
        break;
      case IRS_SIM_TVERSKY:
        check_lower(alpha, 0, "alpha");
        check_lower(beta, 0, "beta");
        break;
      case IRS_SIM_RP3BETA:
        check_lower(alpha, 0, "alpha");
        check_lower(beta, 0, "beta");
        break;
      case IRS_SIM_P3ALPHA:
        check_lower(alpha, 0, "alpha");
        break;
      default:
        throw std::invalid_argument("unknown similarity type.");
    }
    int32_t layout = input ? input->layout : IRS_LAYOUT_CSR;
    int32_t weighting = input ? input->weighting : IRS_WEIGHT_NONE;
    check_arg(layout == IRS_LAYOUT_CSR || layout == IRS_LAYOUT_CSC, "unknown matrix layout.");
    check_arg(weighting == IRS_WEIGHT_NONE || weighting == IRS_WEIGHT_TF_IDF || weighting == IRS_WEIGHT_BM25,
              "unknown feature weighting.");
    check_arg(rows >= 0 && cols >= 0 && indptr, "bad matrix.");
    check_arg(rows < (int64_t(1) << 31) && cols < (int64_t(1) << 31), "too many rows.");
    // M: the matrix as the arrays store it - X_arg [rows, cols] itself, or X_arg^T for the CSC layout
    int64_t m_rows = layout == IRS_LAYOUT_CSC ? cols : rows, m_cols = layout == IRS_LAYOUT_CSC ? rows : cols;
    check_arg(indptr[0] == 0 && indptr[m_rows] >= 0, "malformed indptr.");
    for (int64_t i = 0; i < m_rows; i++) check_arg(indptr[i + 1] >= indptr[i], "malformed indptr.");
    const int64_t nnz_in = indptr[m_rows];
    check_arg(nnz_in == 0 || (indices && data), "bad matrix.");
    // ---- Cosine / asymmetric cosine / Jaccard / Tversky, round 5: the caller's arrays are validated and
    // classified in place (no private copy), the column indices - and the values, if they are not all
    // ones - travel to the device beside that pass, and X_arg^T, its slice pointers, its packed offsets
    // and its per-row value ranges are built THERE (stable radix sort of the entry numbers by column +
    // one gather, three small kernels): 68 -> 5 ms for binary interactions on the ML-20M shape, where the
    // host spent 17 ms on the copy, 29 ms on the transpose and 15 ms on slices + packing.  Same
    // similarities bit for bit (test_device_create_is_the_host_create).  P3alpha / RP3beta: the rows are
    // pow-ed and normalised on the host first (libm's pow), into the one private array of this path.
    // IRSPACK_AMD_KNN_DEVICE_CREATE=0: host path for everything (A/B).
    const bool binarise_create = sim_type == IRS_SIM_JACCARD || sim_type == IRS_SIM_TVERSKY;
    const bool transforms = sim_type == IRS_SIM_P3ALPHA || sim_type == IRS_SIM_RP3BETA;
    const bool device_candidate = nnz_in > 0 && cols > 0 && nnz_in < (int64_t(1) << 31) - 1024 &&
                                  env_flag("IRSPACK_AMD_KNN_DEVICE_CREATE", true);
    check_arg(!(transforms && weighting != IRS_WEIGHT_NONE), "P3alpha / RP3beta take no feature weighting.");
    if (binarise_create) weighting = IRS_WEIGHT_NONE;  // (every stored value becomes 1: similarities.hpp:96-107, 143-159)
    // The paths that read X_arg's rows on the host (P3alpha / RP3beta's row normalisation; the host
    // construction; empty matrices) get them: weighting and regrouping on host threads, into private arrays.
    HostCsrD conv;
    RawVector<double> conv_w;
    if ((!device_candidate || transforms) && (layout == IRS_LAYOUT_CSC || weighting != IRS_WEIGHT_NONE)) {
      std::atomic<int> bad_idx(0);
      parallel_ranges(nnz_in, [&](int64_t b, int64_t e) {
        int32_t lo = 0, hi = 0;
        for (int64_t q = b; q < e; q++) {
          lo = std::min(lo, indices[q]);
          hi = std::max(hi, indices[q]);
        }
        if (e > b && (lo < 0 || hi >= m_cols)) bad_idx.store(1);
      });
      check_arg(bad_idx.load() == 0, "column index out of range.");
      if (weighting != IRS_WEIGHT_NONE) {
        const bool bm = weighting == IRS_WEIGHT_BM25;
        const WeightTables wt = weight_tables(bm, m_rows, m_cols, indptr, indices, data, false, input->k1, input->b,
                                              input->smooth != 0);
        conv_w.resize(static_cast<size_t>(nnz_in));
        weight_values_host(bm, wt, m_rows, indptr, indices, data, input->k1, conv_w.data());
        data = conv_w.data();
        weighting = IRS_WEIGHT_NONE;
      }
      if (layout == IRS_LAYOUT_CSC) {
        conv = transpose_pattern(m_rows, m_cols, indptr, indices, data);
        indptr = conv.indptr.data();
        indices = conv.indices.data();
        data = conv.data.data();
        layout = IRS_LAYOUT_CSR;
        m_rows = rows;
        m_cols = cols;
      }
      pt.mark("create: host regroup");
    }
    if (device_candidate) {
      require_device(device);
      IRS_HIP(hipSetDevice(device));
      // M = the matrix as the arrays store it: X_arg itself (CSR layout) or X_arg^T (CSC layout)
      const bool csc = layout == IRS_LAYOUT_CSC;
      DeviceBuffer<int32_t> d_indptr, d_indices, d_tidx;
      DeviceBuffer<double> d_values;
      // P3alpha / RP3beta (similarities.hpp:198-222, 265-292): the rows pow-ed and normalised to sum 1, on
      // the host (libm's pow: the values the oracle has) into the one private array of this path
      const double *vals = data;
      RawVector<double> tvals;
      const std::vector<int64_t> ipv(indptr, indptr + m_rows + 1);
      std::string upload_error;
      std::atomic<int> kind(0);  // 0: not classified yet, 1: all ones, 2: the values travel too, 3: give up
      std::thread uploader([&] {  // (pageable memory: the copies occupy a host thread)
        try {
          IRS_HIP(hipSetDevice(device));
          std::vector<int32_t> ip32(m_rows + 1);
          for (int64_t i = 0; i <= m_rows; i++) ip32[i] = static_cast<int32_t>(indptr[i]);
          d_indptr.alloc(static_cast<size_t>(m_rows) + 1);
          d_indices.alloc(static_cast<size_t>(nnz_in));
          IRS_HIP(hipMemcpy(d_indptr.ptr, ip32.data(), ip32.size() * sizeof(int32_t), hipMemcpyHostToDevice));
          IRS_HIP(hipMemcpy(d_indices.ptr, indices, static_cast<size_t>(nnz_in) * sizeof(int32_t), hipMemcpyHostToDevice));
          int k = 0;
          while ((k = kind.load(std::memory_order_acquire)) == 0) std::this_thread::yield();
          if (k == 2) {
            d_values.alloc(static_cast<size_t>(nnz_in));
            IRS_HIP(hipMemcpy(d_values.ptr, vals, static_cast<size_t>(nnz_in) * sizeof(double), hipMemcpyHostToDevice));
          }
        } catch (const std::exception &e) {
          upload_error = e.what();
        }
      });
      struct Joiner {
        std::thread &t;
        std::atomic<int> &kind;
        ~Joiner() {
          int zero = 0;
          kind.compare_exchange_strong(zero, 3);  // (an exception before the classification: the uploader must not wait)
          if (t.joinable()) t.join();
        }
      } upload_join{uploader, kind};
      std::atomic<int> bad(0), not_ones(0), not_safe(0), not_pos(0);
      if (transforms) {
        tvals.resize(static_cast<size_t>(nnz_in));
        for_rows_parallel(ipv, m_rows, [&](int64_t i) {
          double sum = 0;
          for (int64_t q = indptr[i]; q < indptr[i + 1]; q++) {
            tvals[q] = std::pow(data[q], alpha);
            sum += tvals[q];
          }
          for (int64_t q = indptr[i]; q < indptr[i + 1]; q++) tvals[q] /= sum;
        });
        vals = tvals.data();
        pt.mark("create: transform");
      }
      {
        const int n_thr = static_cast<int>(std::max<int64_t>(
            1, std::min<int64_t>({16, static_cast<int64_t>(std::thread::hardware_concurrency()), nnz_in / 500000 + 1})));
        run_on_threads(n_thr, [&](int k) {
          const int64_t b = nnz_in * k / n_thr, e = nnz_in * (k + 1) / n_thr;
          int32_t lo = 0, hi = 0;
          for (int64_t q = b; q < e; q++) {
            lo = std::min(lo, indices[q]);
            hi = std::max(hi, indices[q]);
          }
          if (e > b && (lo < 0 || hi >= m_cols)) bad.store(1);
          if (!binarise_create) {  // (a branch-free pass the compiler vectorises)
            uint64_t diff = 0;
            const uint64_t one_bits = 0x3ff0000000000000ull;
            const uint64_t *vb = reinterpret_cast<const uint64_t *>(vals);
            for (int64_t q = b; q < e; q++) diff |= vb[q] ^ one_bits;
            if (diff) {
              not_ones.store(1);
              bool safe = true, pos = true;
              for (int64_t q = b; q < e; q++) {
                const double v = vals[q], a = std::fabs(v);
                safe &= a > 1e-150 && a < 1e150;
                pos &= v > 0.0;
              }
              if (!safe) not_safe.store(1);
              if (!pos) not_pos.store(1);
            }
          }
        });
      }
      check_arg(bad.load() == 0, "column index out of range.");
      const bool input_ones = not_ones.load() == 0;  // (binarising similarities: always)
      kind.store(input_ones ? 1 : 2, std::memory_order_release);
      pt.mark("create: validate");
      // The feature weighting (util.hpp:159-209), fused: the two small tables on host threads while the
      // uploads run, the per-entry pass on the device - the weighted matrix is never materialised on the host.
      WeightTables wt;
      const bool bm25 = weighting == IRS_WEIGHT_BM25;
      if (weighting != IRS_WEIGHT_NONE) {
        wt = weight_tables(bm25, m_rows, m_cols, indptr, indices, data, input_ones, input->k1, input->b,
                           input->smooth != 0);
        pt.mark("create: weight tables");
      }
      {
        auto c = std::make_unique<irs_knn_computer>();
        c->device = device;
        c->sim_type = sim_type;
        c->N = rows;
        c->n_features = cols;
        c->shrinkage = shrinkage;
        c->alpha = alpha;
        c->beta = beta;
        c->normalize = normalize != 0;
        // the norms (similarities.hpp:20-28, 61-72, 96-107, 143-159); rows of ones: sqrt / pow of the entry
        // count, which is what their sums of 1.0 * 1.0 are.  From the caller's arrays on host threads when
        // they hold X_arg's rows and nothing weights them; else from the device (below).
        std::vector<double> norms(rows, 0.0);
        auto finish_norm = [&](double ss) {
          switch (sim_type) {
            case IRS_SIM_COSINE: return std::sqrt(ss);
            case IRS_SIM_ASYMMETRIC: return std::pow(ss, 1 - alpha);
            default: return ss;  // (Jaccard / Tversky: the entry count; never weighted)
          }
        };
        const bool norms_on_device = !transforms && (weighting != IRS_WEIGHT_NONE || (csc && !input_ones));
        if (!transforms && !norms_on_device) {  // (P3alpha / RP3beta have none)
          if (csc) {  // all ones: the entry count of every column of M
            const std::vector<int64_t> cnt = column_counts(m_rows, m_cols, indptr, indices);
            parallel_ranges(rows, [&](int64_t lo, int64_t hi) {
              for (int64_t i = lo; i < hi; i++) norms[i] = finish_norm(static_cast<double>(cnt[i]));
            }, 16, 20000);
          } else {
            for_rows_parallel(ipv, rows, [&](int64_t i) {
              double ss = static_cast<double>(indptr[i + 1] - indptr[i]);
              if (!input_ones) {
                ss = 0;
                for (int64_t q = indptr[i]; q < indptr[i + 1]; q++) ss += data[q] * data[q];
              }
              norms[i] = finish_norm(ss);
            });
          }
        }
        uploader.join();
        if (!upload_error.empty()) throw std::runtime_error(upload_error);
        pt.mark("create: norms + upload");
        hipStream_t s = nullptr;
        bool weighted = !input_ones;  // does the value stream of the kernels carry anything but ones?
        int32_t w_flags = 0;
        if (weighting != IRS_WEIGHT_NONE) {
          DeviceBuffer<double> d_idf, d_reg, d_w;
          DeviceBuffer<int32_t> d_flags;
          d_idf.upload(wt.idf, s);
          if (bm25) d_reg.upload(wt.reg, s);
          d_w.alloc(static_cast<size_t>(nnz_in));
          d_flags.alloc(1);
          IRS_HIP(hipMemsetAsync(d_flags.ptr, 0, sizeof(int32_t), s));
          const double *in_vals = input_ones ? nullptr : d_values.ptr;
          const dim3 grid(static_cast<unsigned>(ceil_div(m_rows, 4)));
          if (bm25)
            hipLaunchKernelGGL(knn_weight_kernel<true>, grid, dim3(256), 0, s, static_cast<const int32_t *>(d_indptr.ptr),
                               static_cast<const int32_t *>(d_indices.ptr), in_vals, static_cast<const double *>(d_idf.ptr),
                               static_cast<const double *>(d_reg.ptr), input->k1 + 1, m_rows, d_w.ptr, d_flags.ptr);
          else
            hipLaunchKernelGGL(knn_weight_kernel<false>, grid, dim3(256), 0, s, static_cast<const int32_t *>(d_indptr.ptr),
                               static_cast<const int32_t *>(d_indices.ptr), in_vals, static_cast<const double *>(d_idf.ptr),
                               static_cast<const double *>(nullptr), 0.0, m_rows, d_w.ptr, d_flags.ptr);
          IRS_HIP(hipGetLastError());
          IRS_HIP(hipMemcpyAsync(&w_flags, d_flags.ptr, sizeof(int32_t), hipMemcpyDeviceToHost, s));
          IRS_HIP(hipStreamSynchronize(s));
          d_values.release();
          d_values.ptr = d_w.ptr;  // (the weighted values take the place of the caller's)
          d_values.count = d_w.count;
          d_values.owned = true;
          d_w.ptr = nullptr;
          d_w.count = 0;
          weighted = (w_flags & 1) != 0;
          not_safe.store((w_flags & 2) ? 1 : 0);
          not_pos.store((w_flags & 4) ? 1 : 0);
          pt.mark("create: weight");
        }
        // SEPARABLE weightings.  On binary interactions both weightings factor over the stored matrix M
        // (rows r = documents, columns t = terms): tf-idf w[r][t] = idf[t] (util.hpp:206 with x = 1), BM25
        // w[r][t] = (idf[t] (k1 + 1)) / (1 + reg[r]) (:183-184).  The kernels then never read a value stream:
        // the factor that belongs to X_arg^T's ROWS (the features) becomes `xt_rowval` - the all-ones kernels
        // on y' = x_u y, round 5 - and the factor that belongs to its COLUMNS (the computer's items / users)
        // becomes `col_scale`, applied to the finished sum with one rounding.  Pure tf-idf of an item-kNN
        // (CSC layout: idf belongs to the columns) therefore runs the exact COUNT kernel.  The sums differ
        // from the sums of the individually rounded weights by rounding only (<= 2 ulp per term, far inside
        // the 1e-12 of the parity bar; columns whose weights and counts are equal tie EXACTLY and are ordered
        // by column - DESIGN 3.4, `test_separable_weighting`).  The norms still come from the individually
        // rounded weights (d_values), like the reference's.  IRSPACK_AMD_KNN_SEPARABLE=0: the general form (A/B).
        const bool separable = weighting != IRS_WEIGHT_NONE && input_ones && env_flag("IRSPACK_AMD_KNN_SEPARABLE", true);
        std::vector<double> sep_rowval, sep_colscale;  // (empty: that factor is 1)
        if (separable) {
          const double k1p1 = bm25 ? input->k1 + 1 : 1.0;
          std::vector<double> &term_f = csc ? sep_colscale : sep_rowval;  // per column of M
          std::vector<double> &doc_f = csc ? sep_rowval : sep_colscale;   // per row of M
          term_f.resize(static_cast<size_t>(m_cols));
          for (int64_t t = 0; t < m_cols; t++) term_f[t] = bm25 ? wt.idf[t] * k1p1 : wt.idf[t];
          if (bm25) {
            doc_f.resize(static_cast<size_t>(m_rows));
            for (int64_t r = 0; r < m_rows; r++) doc_f[r] = 1.0 / (1.0 + wt.reg[r]);
          }
        }
        const bool stream_vals = weighted && !separable;  // do the kernels read a float64 value per entry?
        const size_t padded = (static_cast<size_t>(nnz_in) + 256 + 1) & ~size_t(1);
        std::vector<int32_t> t_count;
        DeviceBuffer<char> tmp;
        DeviceBuffer<double> d_ss;
        if (norms_on_device && !csc) {  // rows of X_arg as stored: before the transpose reorders the values
          d_ss.alloc(static_cast<size_t>(std::max<int64_t>(rows, 1)));
          hipLaunchKernelGGL(knn_row_sumsq_kernel, dim3(static_cast<unsigned>(ceil_div(rows, 4))), dim3(256), 0, s,
                             reinterpret_cast<const uint32_t *>(d_indptr.ptr), static_cast<const double *>(d_values.ptr),
                             rows, d_ss.ptr);
        }
        const int32_t *xt_idx_src = nullptr;  // column indices of X_arg^T's entries (device)
        if (!csc) {
          d_tidx.alloc(static_cast<size_t>(nnz_in));
          if (stream_vals) {  // the transposed values straight into the padded value stream of the kernels
            c->xt_val.alloc(padded);
            IRS_HIP(hipMemsetAsync(c->xt_val.ptr + nnz_in, 0, (padded - static_cast<size_t>(nnz_in)) * sizeof(double), s));
            transpose_csr_device(d_indptr.ptr, d_indices.ptr, static_cast<const double *>(d_values.ptr), rows, cols,
                                 nnz_in, d_tidx.ptr, c->xt_val.ptr, t_count, tmp, s);
          } else {
            c->xt_val.alloc(2);  // (the ONES kernels never read the value stream)
            transpose_csr_device(d_indptr.ptr, d_indices.ptr, static_cast<const double *>(nullptr), rows, cols, nnz_in,
                                 d_tidx.ptr, static_cast<double *>(nullptr), t_count, tmp, s);
          }
          xt_idx_src = d_tidx.ptr;
          pt.mark("create: transpose");
        } else {
          // CSC layout: the arrays ARE X_arg^T - nothing to transpose for the kernels' streams; only the
          // norms of a weighted matrix need X_arg's rows together (their squares are added in row order)
          t_count.resize(static_cast<size_t>(cols));
          for (int64_t u = 0; u < cols; u++) t_count[u] = static_cast<int32_t>(indptr[u + 1] - indptr[u]);
          if (stream_vals) {
            c->xt_val.alloc(padded);
            hipLaunchKernelGGL(knn_pad_copy_kernel, dim3(static_cast<unsigned>(ceil_div(static_cast<int64_t>(padded), 256))),
                               dim3(256), 0, s, static_cast<const double *>(d_values.ptr), nnz_in,
                               static_cast<int64_t>(padded), c->xt_val.ptr);
          } else {
            c->xt_val.alloc(2);
          }
          if (norms_on_device) {
            DeviceBuffer<double> d_tval;
            DeviceBuffer<uint32_t> d_rptr;
            std::vector<int32_t> r_count;
            d_tval.alloc(static_cast<size_t>(nnz_in));
            d_tidx.alloc(static_cast<size_t>(nnz_in));
            transpose_csr_device(d_indptr.ptr, d_indices.ptr, static_cast<const double *>(d_values.ptr), m_rows, m_cols,
                                 nnz_in, d_tidx.ptr, d_tval.ptr, r_count, tmp, s);
            std::vector<uint32_t> rptr(static_cast<size_t>(rows) + 1, 0u);
            for (int64_t i = 0; i < rows; i++) rptr[i + 1] = rptr[i] + static_cast<uint32_t>(r_count[i]);
            d_rptr.upload(rptr, s);
            d_ss.alloc(static_cast<size_t>(std::max<int64_t>(rows, 1)));
            hipLaunchKernelGGL(knn_row_sumsq_kernel, dim3(static_cast<unsigned>(ceil_div(rows, 4))), dim3(256), 0, s,
                               static_cast<const uint32_t *>(d_rptr.ptr), static_cast<const double *>(d_tval.ptr), rows,
                               d_ss.ptr);
            IRS_HIP(hipGetLastError());
            IRS_HIP(hipStreamSynchronize(s));  // (rptr and the scratch go out of scope)
          }
          xt_idx_src = d_indices.ptr;
          pt.mark("create: csc streams");
        }
        if (norms_on_device) {
          std::vector<double> ss(static_cast<size_t>(rows), 0.0);
          IRS_HIP(hipGetLastError());
          if (rows > 0) IRS_HIP(hipMemcpyAsync(ss.data(), d_ss.ptr, static_cast<size_t>(rows) * sizeof(double),
                                              hipMemcpyDeviceToHost, s));
          IRS_HIP(hipStreamSynchronize(s));
          parallel_ranges(rows, [&](int64_t lo, int64_t hi) {
            for (int64_t i = lo; i < hi; i++) norms[i] = finish_norm(ss[i]);
          }, 16, 20000);
          pt.mark("create: device norms");
        }
        const int64_t n_tiles = std::max<int64_t>(1, ceil_div(rows, TILE));
        std::vector<uint32_t> xt_ptr(static_cast<size_t>(cols) + 1, 0u);
        c->xt_row_len.resize(cols);
        c->xt_rowmax.assign(cols, 0.0);
        c->xt_rowmin.assign(cols, 0.0);
        for (int64_t u = 0; u < cols; u++) {
          xt_ptr[u + 1] = xt_ptr[u] + static_cast<uint32_t>(t_count[u]);
          c->xt_row_len[u] = t_count[u];
          if (!stream_vals && t_count[u] > 0)
            c->xt_rowmax[u] = c->xt_rowmin[u] = sep_rowval.empty() ? 1.0 : std::fabs(sep_rowval[u]);
        }
        DeviceBuffer<uint32_t> d_xt_ptr;
        DeviceBuffer<double> d_range;
        d_xt_ptr.upload(xt_ptr, s);
        c->xt_tptr.alloc(static_cast<size_t>(cols) * (n_tiles + 1));
        const int64_t n_pairs = cols * (n_tiles + 1);
        hipLaunchKernelGGL(xt_slices_kernel, dim3(static_cast<unsigned>(ceil_div(n_pairs, 256))), dim3(256), 0, s,
                           static_cast<const uint32_t *>(d_xt_ptr.ptr), xt_idx_src, cols,
                           static_cast<int32_t>(n_tiles), c->xt_tptr.ptr);
        c->xt_idx16.alloc(padded / 2);
        hipLaunchKernelGGL(xt_pack16_kernel, dim3(static_cast<unsigned>(ceil_div(static_cast<int64_t>(padded / 2), 256))),
                           dim3(256), 0, s, xt_idx_src, nnz_in,
                           static_cast<int64_t>(padded / 2), c->xt_idx16.ptr);
        DeviceBuffer<int32_t> d_not_const;
        int32_t not_const = 1;
        if (stream_vals) {
          d_range.alloc(2 * static_cast<size_t>(cols));
          c->xt_rowval.alloc(static_cast<size_t>(cols));
          d_not_const.alloc(1);
          IRS_HIP(hipMemsetAsync(d_not_const.ptr, 0, sizeof(int32_t), s));
          hipLaunchKernelGGL(xt_row_range_kernel, dim3(static_cast<unsigned>(ceil_div(cols, 256))), dim3(256), 0, s,
                             static_cast<const uint32_t *>(d_xt_ptr.ptr), static_cast<const double *>(c->xt_val.ptr), cols,
                             d_range.ptr, d_range.ptr + cols, c->xt_rowval.ptr, d_not_const.ptr);
          IRS_HIP(hipMemcpyAsync(&not_const, d_not_const.ptr, sizeof(int32_t), hipMemcpyDeviceToHost, s));
          IRS_HIP(hipMemcpyAsync(c->xt_rowmax.data(), d_range.ptr, static_cast<size_t>(cols) * sizeof(double),
                                 hipMemcpyDeviceToHost, s));
          IRS_HIP(hipMemcpyAsync(c->xt_rowmin.data(), d_range.ptr + cols, static_cast<size_t>(cols) * sizeof(double),
                                 hipMemcpyDeviceToHost, s));
        }
        IRS_HIP(hipGetLastError());
        c->xt_all_ones = !stream_vals && sep_rowval.empty();
        c->xt_nonzero = not_safe.load() == 0;
        c->xt_positive = not_pos.load() == 0;
        if (separable) {  // (the flags describe what the kernels multiply: the row factors)
          bool safe = true, pos = true;
          for (const double v : sep_rowval) {
            const double a = std::fabs(v);
            safe &= a > 1e-150 && a < 1e150;
            pos &= v > 0.0;
          }
          c->xt_nonzero = safe;
          c->xt_positive = pos;
          if (!sep_rowval.empty()) {
            c->xt_row_const = true;
            c->xt_rowval.upload(sep_rowval, s);
          }
          if (!sep_colscale.empty()) {
            c->col_scaled = true;
            c->col_scale.upload(sep_colscale, s);
          }
        }
        c->norm_max = norms.empty() ? 0.0 : *std::max_element(norms.begin(), norms.end());
        c->norms.upload(norms, s);
        IRS_HIP(hipStreamSynchronize(s));  // the host vectors and the device scratch go out of scope
        if (stream_vals && not_const == 0) {  // one value per feature row: the value stream is not needed
          c->xt_row_const = true;
          c->xt_val.release();
          c->xt_val.alloc(2);
        } else if (!c->xt_row_const) {
          c->xt_rowval.release();
        }
        pt.mark("create: slices + pack");
        if (!tvals.empty()) {  // (returning 160 MB to the system takes 10+ ms: not on the caller's clock)
          auto *junk = new RawVector<double>();
          junk->swap(tvals);
          std::thread([junk] { delete junk; }).detach();
        }
        *out = c.release();
        return;
      }
    }
    HostCsrD X = host_csr(rows, cols, indptr, indices, data);
    pt.mark("create: copy");
    std::vector<double> norms(rows, 0.0);
    switch (sim_type) {
      case IRS_SIM_COSINE:  // similarities.hpp:20-28
        for_rows_parallel(X.indptr, rows, [&](int64_t i) {
          double s = 0;
          for (int64_t q = X.indptr[i]; q < X.indptr[i + 1]; q++) s += X.data[q] * X.data[q];
          norms[i] = std::sqrt(s);
        });
        break;
      case IRS_SIM_ASYMMETRIC:  // similarities.hpp:61-72
        for_rows_parallel(X.indptr, rows, [&](int64_t i) {
          double s = 0;
          for (int64_t q = X.indptr[i]; q < X.indptr[i + 1]; q++) s += X.data[q] * X.data[q];
          norms[i] = std::pow(s, 1 - alpha);
        });
        break;
      case IRS_SIM_TVERSKY:  // similarities.hpp:143-159
      case IRS_SIM_JACCARD:  // similarities.hpp:96-107: stored entries become 1
        for_rows_parallel(X.indptr, rows, [&](int64_t i) {
          for (int64_t q = X.indptr[i]; q < X.indptr[i + 1]; q++) X.data[q] = 1;
          norms[i] = static_cast<double>(X.indptr[i + 1] - X.indptr[i]);
        });
        break;
      case IRS_SIM_RP3BETA:  // similarities.hpp:265-292
      case IRS_SIM_P3ALPHA:  // similarities.hpp:198-222: rows pow-ed and normalised to sum 1
        for_rows_parallel(X.indptr, rows, [&](int64_t i) {
          double s = 0;
          for (int64_t q = X.indptr[i]; q < X.indptr[i + 1]; q++) {
            X.data[q] = std::pow(X.data[q], alpha);
            s += X.data[q];
          }
          for (int64_t q = X.indptr[i]; q < X.indptr[i + 1]; q++) X.data[q] /= s;
        });
        break;
      default:
        throw std::invalid_argument("unknown similarity type.");
    }
    pt.mark("create: norms");
    require_device(device);
    pt.mark("create: device");
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
    pt.mark("create: transpose");
    // the per-row and per-entry passes below run on several host threads over row blocks of
    // about equal entry counts
    const int64_t xt_nnz = Xt.indptr[Xt.rows];
    const int n_thr = static_cast<int>(std::max<int64_t>(
        1, std::min<int64_t>({16, static_cast<int64_t>(std::thread::hardware_concurrency()),
                              xt_nnz / 500000 + 1})));
    std::vector<int64_t> blk(n_thr + 1, Xt.rows);
    for (int k = 0; k < n_thr; k++)
      blk[k] = std::lower_bound(Xt.indptr.begin(), Xt.indptr.begin() + Xt.rows, xt_nnz * k / n_thr) -
               Xt.indptr.begin();
    blk[0] = 0;
    auto on_threads = [&](auto &&body) {
      std::vector<std::thread> th;
      for (int k = 1; k < n_thr; k++) th.emplace_back(body, k);
      body(0);
      for (auto &w : th) w.join();
    };
    c->xt_row_len.resize(Xt.rows);
    c->xt_rowmax.assign(Xt.rows, 0.0);
    c->xt_rowmin.assign(Xt.rows, 0.0);
    hipStream_t s = nullptr;
    const int64_t n_tiles = std::max<int64_t>(1, ceil_div(rows, TILE));
    check_arg(xt_nnz < (int64_t(1) << 31) - 1024, "nnz must be below 2^31.");
    std::vector<uint32_t> tptr(static_cast<size_t>(Xt.rows) * (n_tiles + 1));
    // columns relative to their tile as 16-bit LDS byte offsets (column * 4), two per dword;
    // 256 padding entries: the accumulate loop reads whole 128-entry strips from the (even)
    // start of a slice
    static_assert(TILE * 4 <= 65536, "tile-relative column offsets are stored in 16 bits");
    const size_t nnz = Xt.indices.size(), padded = (nnz + 256 + 1) & ~size_t(1);
    std::vector<uint32_t> idx_p(padded / 2, 0u);
    uint16_t *idx16 = reinterpret_cast<uint16_t *>(idx_p.data());  // little endian: entry e = half e
    std::atomic<int> not_ones(0), not_safe(0), not_pos(0);
    on_threads([&](int k) {
      bool ones = true, safe = true, pos = true;
      for (int64_t u = blk[k]; u < blk[k + 1]; u++) {
        c->xt_row_len[u] = Xt.indptr[u + 1] - Xt.indptr[u];
        double mx = 0.0, mn = std::numeric_limits<double>::infinity();
        for (int64_t q = Xt.indptr[u]; q < Xt.indptr[u + 1]; q++) {
          const double v = Xt.data[q], a = std::fabs(v);
          mx = std::max(mx, a);
          if (a > 0.0) mn = std::min(mn, a);
          ones &= v == 1.0;
          safe &= a > 1e-150 && a < 1e150;
          pos &= v > 0.0;
          idx16[q] = static_cast<uint16_t>((Xt.indices[q] % TILE) * 4);
        }
        c->xt_rowmax[u] = mx;
        c->xt_rowmin[u] = std::isfinite(mn) ? mn : 0.0;
        const int32_t *b = Xt.indices.data() + Xt.indptr[u], *e = Xt.indices.data() + Xt.indptr[u + 1];
        uint32_t *dst = tptr.data() + u * (n_tiles + 1);
        for (int64_t t = 0; t < n_tiles; t++)
          dst[t] = static_cast<uint32_t>(Xt.indptr[u] +
                                         (std::lower_bound(b, e, static_cast<int32_t>(t * TILE)) - b));
        dst[n_tiles] = static_cast<uint32_t>(Xt.indptr[u + 1]);
      }
      if (!ones) not_ones.store(1);
      if (!safe) not_safe.store(1);
      if (!pos) not_pos.store(1);
    });
    c->xt_tptr.upload(tptr, s);
    pt.mark("create: slices");
    {
      c->xt_idx16.upload(idx_p, s);
      c->xt_all_ones = not_ones.load() == 0;
      c->xt_nonzero = not_safe.load() == 0;
      c->xt_positive = not_pos.load() == 0;
      if (c->xt_all_ones) {  // the ONES kernels never read the value stream
        c->xt_val.alloc(2);
      } else {
        std::vector<double> val_p(Xt.data.begin(), Xt.data.end());
        val_p.resize(padded, 0.0);
        c->xt_val.upload(val_p, s);
      }
      IRS_HIP(hipStreamSynchronize(s));  // the host vectors go out of scope
    }
    c->norm_max = norms.empty() ? 0.0 : *std::max_element(norms.begin(), norms.end());
    c->norms.upload(norms, s);
    IRS_HIP(hipStreamSynchronize(s));
    pt.mark("create: pack+upload");
    // (the host staging - half a gigabyte for 20 M entries - is released here rather than at
    // scope exit so that the phase shows up in the timing)
    // and on a thread of its own: returning 480 MB to the kernel (munmap) takes 56 ms, nothing
    // waits for it
    {
      auto *junk = new std::pair<HostCsrD, HostCsrD>();
      junk->first.indices.swap(X.indices);
      junk->first.data.swap(X.data);
      junk->second.indices.swap(Xt.indices);
      junk->second.data.swap(Xt.data);
      std::thread([junk] { delete junk; }).detach();
    }
    pt.mark("create: release host");
    *out = c.release();
  });
}

irs_status irs_knn_weight(int32_t weighting, int64_t rows, int64_t cols, const int64_t *indptr,
                          const int32_t *indices, const double *data, double k1, double b, int32_t smooth,
                          int32_t device, double *out) {
  return guard([&] {
    check_arg(weighting == IRS_WEIGHT_TF_IDF || weighting == IRS_WEIGHT_BM25, "unknown feature weighting.");
    check_arg(rows >= 0 && cols >= 0 && indptr && indptr[0] == 0, "bad matrix.");
    for (int64_t i = 0; i < rows; i++) check_arg(indptr[i + 1] >= indptr[i], "malformed indptr.");
    const int64_t nnz = indptr[rows];
    if (nnz == 0) return;
    check_arg(indices && data && out, "bad matrix.");
    check_arg(nnz < (int64_t(1) << 31), "nnz must be below 2^31.");
    PhaseTimer pt;
    require_device(device);
    IRS_HIP(hipSetDevice(device));
    const bool bm25 = weighting == IRS_WEIGHT_BM25;
    // the uploads (pageable memory: they occupy a host thread) beside the validation and the tables
    DeviceBuffer<int32_t> d_indptr, d_indices;
    DeviceBuffer<double> d_values, d_out;
    std::string upload_error;
    std::atomic<int> kind(0);  // 0: values not classified yet, 1: all ones, 2: they travel, 3: give up
    std::thread uploader([&] {
      try {
        IRS_HIP(hipSetDevice(device));
        std::vector<int32_t> ip32(rows + 1);
        for (int64_t i = 0; i <= rows; i++) ip32[i] = static_cast<int32_t>(indptr[i]);
        d_indptr.alloc(static_cast<size_t>(rows) + 1);
        d_indices.alloc(static_cast<size_t>(nnz));
        d_out.alloc(static_cast<size_t>(nnz));
        IRS_HIP(hipMemcpy(d_indptr.ptr, ip32.data(), ip32.size() * sizeof(int32_t), hipMemcpyHostToDevice));
        IRS_HIP(hipMemcpy(d_indices.ptr, indices, static_cast<size_t>(nnz) * sizeof(int32_t), hipMemcpyHostToDevice));
        int k = 0;
        while ((k = kind.load(std::memory_order_acquire)) == 0) std::this_thread::yield();
        if (k == 2) {
          d_values.alloc(static_cast<size_t>(nnz));
          IRS_HIP(hipMemcpy(d_values.ptr, data, static_cast<size_t>(nnz) * sizeof(double), hipMemcpyHostToDevice));
        }
      } catch (const std::exception &e) {
        upload_error = e.what();
      }
    });
    struct Joiner {
      std::thread &t;
      std::atomic<int> &kind;
      ~Joiner() {
        int zero = 0;
        kind.compare_exchange_strong(zero, 3);
        if (t.joinable()) t.join();
      }
    } upload_join{uploader, kind};
    std::atomic<int> bad(0), not_ones(0);
    parallel_ranges(nnz, [&](int64_t lo_q, int64_t hi_q) {
      int32_t lo = 0, hi = 0;
      uint64_t diff = 0;
      const uint64_t *vb = reinterpret_cast<const uint64_t *>(data);
      for (int64_t q = lo_q; q < hi_q; q++) {
        lo = std::min(lo, indices[q]);
        hi = std::max(hi, indices[q]);
        diff |= vb[q] ^ 0x3ff0000000000000ull;
      }
      if (lo < 0 || hi >= cols) bad.store(1);
      if (diff) not_ones.store(1);
    });
    check_arg(bad.load() == 0, "column index out of range.");
    const bool ones = not_ones.load() == 0;
    kind.store(ones ? 1 : 2, std::memory_order_release);
    pt.mark("weight: validate");
    const WeightTables wt = weight_tables(bm25, rows, cols, indptr, indices, data, ones, k1, b, smooth != 0);
    pt.mark("weight: tables");
    uploader.join();
    if (!upload_error.empty()) throw std::runtime_error(upload_error);
    hipStream_t s = nullptr;
    DeviceBuffer<double> d_idf, d_reg;
    d_idf.upload(wt.idf, s);
    if (bm25) d_reg.upload(wt.reg, s);
    const dim3 grid(static_cast<unsigned>(ceil_div(rows, 4)));
    const double *in_vals = ones ? nullptr : d_values.ptr;
    if (bm25)
      hipLaunchKernelGGL(knn_weight_kernel<true>, grid, dim3(256), 0, s, static_cast<const int32_t *>(d_indptr.ptr),
                         static_cast<const int32_t *>(d_indices.ptr), in_vals, static_cast<const double *>(d_idf.ptr),
                         static_cast<const double *>(d_reg.ptr), k1 + 1, rows, d_out.ptr, static_cast<int32_t *>(nullptr));
    else
      hipLaunchKernelGGL(knn_weight_kernel<false>, grid, dim3(256), 0, s, static_cast<const int32_t *>(d_indptr.ptr),
                         static_cast<const int32_t *>(d_indices.ptr), in_vals, static_cast<const double *>(d_idf.ptr),
                         static_cast<const double *>(nullptr), 0.0, rows, d_out.ptr, static_cast<int32_t *>(nullptr));
    IRS_HIP(hipGetLastError());
    pt.mark("weight: upload + launch");
    // the result into the caller's (pageable, usually untouched) array: a few device-to-host copies
    // side by side - one pageable copy is staged through a single pinned bounce buffer by the runtime
    const int n_copy = static_cast<int>(std::max<int64_t>(1, std::min<int64_t>(4, nnz / (int64_t(1) << 21))));
    IRS_HIP(hipStreamSynchronize(s));
    std::atomic<int> copy_failed(0);
    run_on_threads(n_copy, [&](int k) {
      const int64_t q0 = nnz * k / n_copy, q1 = nnz * (k + 1) / n_copy;
      if (hipSetDevice(device) != hipSuccess ||
          (q1 > q0 && hipMemcpy(out + q0, d_out.ptr + q0, static_cast<size_t>(q1 - q0) * sizeof(double),
                                hipMemcpyDeviceToHost) != hipSuccess))
        copy_failed.store(1);
    });
    if (copy_failed.load()) throw std::runtime_error("copying the weighted values to the host failed.");
    pt.mark("weight: result copy");
  });
}

irs_status irs_knn_destroy(irs_knn_computer *c) {
  return guard([&] {
    if (c) {
      (void)hipSetDevice(c->device);
      {  // leave the larger buffers to the next computer on this device
        for (hipStream_t st : {c->stream, c->stream_k, c->stream_in, c->stream_up, c->stream_out})
          if (st) (void)hipStreamSynchronize(st);
        std::lock_guard<std::mutex> lock(knn_buffer_cache.mu);
        auto &slot = knn_buffer_cache.slot;
        irs_knn_computer::Scratch &sc = c->scratch;
        if (sc.arena.ptr && sc.arena.owned && (slot.device != c->device || sc.arena.count > slot.arena_bytes)) {
          if (slot.arena) (void)hipFree(slot.arena);
          slot.arena = sc.arena.ptr;
          slot.arena_bytes = sc.arena.count;
          sc.arena.ptr = nullptr;  // (the views into it die with the computer)
          sc.arena.count = 0;
          if (slot.device != c->device) {  // (what the slot holds belongs to another device)
            if (slot.stage) (void)hipHostFree(slot.stage);
            slot.stage = nullptr;
            slot.stage_bytes = 0;
            for (auto &st : slot.streams) {
              if (st) (void)hipStreamDestroy(st);
              st = nullptr;
            }
            if (slot.ev_in) (void)hipEventDestroy(slot.ev_in);
            if (slot.ev_setup) (void)hipEventDestroy(slot.ev_setup);
            slot.ev_in = slot.ev_setup = nullptr;
          }
          slot.device = c->device;
        }
        if (c->stream && slot.device == c->device && !slot.streams[0]) {
          hipStream_t *mine[5] = {&c->stream, &c->stream_k, &c->stream_in, &c->stream_up, &c->stream_out};
          for (int k = 0; k < 5; k++) {
            slot.streams[k] = *mine[k];
            *mine[k] = nullptr;
          }
          slot.ev_in = c->ev_in;
          slot.ev_setup = c->ev_setup;
          c->ev_in = c->ev_setup = nullptr;
        }
        if (c->stage && slot.device == c->device && c->stage_bytes > slot.stage_bytes) {
          if (slot.stage) (void)hipHostFree(slot.stage);
          slot.stage = c->stage;
          slot.stage_bytes = c->stage_bytes;
          c->stage = nullptr;
          c->stage_bytes = 0;
        }
      }
      delete c;
    }
  });
}

irs_status irs_knn_compute(irs_knn_computer *c, int64_t rows, int64_t cols,
                           const int64_t *indptr, const int32_t *indices, const double *data,
                           int32_t layout, int64_t top_k, int32_t as_w, int64_t row_begin, int64_t row_end,
                           int64_t *nnz_out) {
  return guard([&] {
    check_arg(c && nnz_out, "null argument.");
    if (cols != c->n_features) throw std::invalid_argument("illegal # of feature.");  // knn.hpp:44-45
    check_arg(top_k >= 0, "top_k must be non-negative.");
    PhaseTimer pt;
    check_arg(rows >= 0 && cols >= 0 && indptr && indptr[0] == 0, "bad matrix.");
    check_arg(0 <= row_begin && row_begin <= row_end && row_end <= rows, "row range out of bounds.");
    const int64_t n = row_end - row_begin;
    check_arg(layout == IRS_LAYOUT_CSR || layout == IRS_LAYOUT_CSC, "unknown matrix layout.");
    // A CSC-layout target (knn.py:79 passes `X_train_all.T`): the arrays are the rows of target^T
    // [cols, rows]; the call walks the target's ROWS, so its columns are regrouped here on host threads
    // (a counting sort that keeps a row's entries in ascending column order), into private arrays - the
    // indices only when every stored value is 1 or the similarity binarises them (the host pass and the
    // kernels then never read a value).
    HostCsrD regrouped;
    bool target_unit = false;
    if (layout == IRS_LAYOUT_CSC) {
      check_arg(rows < (int64_t(1) << 31) && cols < (int64_t(1) << 31), "too many rows.");
      for (int64_t u = 0; u < cols; u++) check_arg(indptr[u + 1] >= indptr[u], "malformed indptr.");
      const int64_t t_nnz = indptr[cols];
      check_arg(t_nnz < (int64_t(1) << 31), "nnz must be below 2^31.");
      check_arg(t_nnz == 0 || (indices && data), "bad matrix.");
      std::atomic<int> bad_idx(0), not_unit(0);
      parallel_ranges(t_nnz, [&](int64_t b, int64_t e) {
        int32_t lo = 0, hi = 0;
        uint64_t diff = 0;
        const uint64_t *vb = reinterpret_cast<const uint64_t *>(data);
        for (int64_t q = b; q < e; q++) {
          lo = std::min(lo, indices[q]);
          hi = std::max(hi, indices[q]);
          diff |= vb[q] ^ 0x3ff0000000000000ull;
        }
        if (e > b && (lo < 0 || hi >= rows)) bad_idx.store(1);
        if (diff) not_unit.store(1);
      });
      check_arg(bad_idx.load() == 0, "malformed matrix: column index out of range or indptr not monotone.");
      const bool sim_binarises = c->sim_type == IRS_SIM_JACCARD || c->sim_type == IRS_SIM_TVERSKY;
      target_unit = !as_w && (sim_binarises || not_unit.load() == 0);
      regrouped = transpose_pattern(cols, rows, indptr, indices, target_unit ? nullptr : data);
      indptr = regrouped.indptr.data();
      indices = regrouped.indices.data();
      data = target_unit ? nullptr : regrouped.data.data();
      pt.mark("regroup columns");
    }
    // --- target preparation (host; mirrors the prologues of compute_similarity_imple / compute_W).
    // Only compute_W transforms the values (a private copy); otherwise the caller's arrays are
    // read in place, and only the rows of this call are looked at.
    HostCsrD T;  // as_w only
    const int64_t *ip = indptr;
    const int32_t *ix = indices;
    const double *dv = data;
    if (as_w) {  // similarities.hpp:224-240, 294-324
      T = host_csr(rows, cols, indptr, indices, data);
      std::vector<double> norm_temp(cols, 0.0), pop(rows, 0.0);
      // (the pow calls and the divisions run on several host threads; the column sums are
      // added in entry order on one, like the reference's loop, so they are the same numbers)
      if (c->sim_type == IRS_SIM_RP3BETA) {
        for_rows_parallel(T.indptr, rows, [&](int64_t i) {
          double sum = 0;
          for (int64_t q = T.indptr[i]; q < T.indptr[i + 1]; q++) sum += T.data[q];
          pop[i] = std::pow(sum, c->beta);
        });
      }
      for_rows_parallel(T.indptr, rows, [&](int64_t i) {
        for (int64_t q = T.indptr[i]; q < T.indptr[i + 1]; q++) T.data[q] = std::pow(T.data[q], c->alpha);
      });
      for (int64_t q = 0; q < T.indptr[rows]; q++) norm_temp[T.indices[q]] += T.data[q];
      for_rows_parallel(T.indptr, rows, [&](int64_t i) {
        for (int64_t q = T.indptr[i]; q < T.indptr[i + 1]; q++) {
          if (c->sim_type == IRS_SIM_RP3BETA)
            T.data[q] /= (norm_temp[T.indices[q]] * pop[i]);
          else
            T.data[q] /= norm_temp[T.indices[q]];
        }
      });
      ip = T.indptr.data();
      ix = T.indices.data();
      dv = T.data.data();
    }
    pt.mark("compute_W prologue");
    // (binarise: the target's values count as 1 - the similarity says so, or a CSC-layout target was all ones)
    const bool binarise = c->sim_type == IRS_SIM_JACCARD || c->sim_type == IRS_SIM_TVERSKY || target_unit;
    // The rows of the call are taken in CHUNKS (contiguous row ranges of about equal entry counts): per
    // chunk one host pass on several threads - index check, the per-row statistic of the epilogue, the
    // multiply-add count that orders the launch, whether the values are all ones / free of zeros (which
    // accumulator the kernel may use) - then the chunk's launches.  The pass over chunk k + 1 and the
    // upload of its column indices (a second host thread) run while the device works on chunk k: of
    // the 2.3 ms the pass takes on the ML-20M shape only the first chunk's share comes before the first
    // kernel, and a finished chunk's rows are compacted and copied to the host beside the later chunks'
    // kernels.  Every chunk chooses its kernel variant from its own rows (each variant gives the
    // reference's values; section 3.4 of DESIGN.md).  Three chunks on large calls: a chunk lasts at least
    // as long as its heaviest (row, tile) pair - 2.1 of the ML-20M call's 8.4 ms - so more, shorter
    // chunks lengthen the device time (4: +0.5 ms, 8: +6 ms) by more than they take off the start.
    // IRSPACK_AMD_KNN_CHUNKS overrides the count (1: the whole call at once; tests force small ones).
    std::vector<double> tstat(std::max<int64_t>(n, 1), 0.0);
    std::vector<double> tscale(std::max<int64_t>(n, 1), 1.0);
    std::vector<int64_t> work(std::max<int64_t>(n, 1), 0);
    const int64_t e_begin = ip[row_begin], e_end = ip[row_end];
    check_arg(e_begin >= 0 && e_end >= e_begin, "malformed indptr.");
    check_arg(e_end < (int64_t(1) << 31), "nnz must be below 2^31.");
    // the row pointers of the call must be monotone BEFORE anything walks or uploads them: the
    // threads below cut [e_begin, e_end) by lower_bound, which on a non-monotone array yields
    // rows whose entries lie outside that range (an out-of-bounds host read instead of this error)
    for (int64_t i = row_begin; i < row_end; i++) check_arg(ip[i + 1] >= ip[i], "malformed indptr.");
    // IRSPACK_AMD_KNN_WIDE: 0 = never two limbs (A/B), 1 = every weighted row (tests); else by range
    const char *wide_env = std::getenv("IRSPACK_AMD_KNN_WIDE");
    const bool wide_ok = !(wide_env && wide_env[0] == '0');
    const double wide_ratio = (wide_env && wide_env[0] == '1') ? 0.0 : 2.3e5;
    const int64_t out_k = std::min<int64_t>(top_k, c->N);
    const bool have_work = n > 0 && out_k > 0 && c->N > 0;
    const size_t ne = static_cast<size_t>(e_end - e_begin);
    int n_chunks = 1;
    {
      const char *e = std::getenv("IRSPACK_AMD_KNN_CHUNKS");  // (read per call: tests toggle it)
      if (e) n_chunks = std::min(64, std::max(1, std::atoi(e)));
      else if (have_work && e_end - e_begin >= (int64_t(1) << 22) && n >= 4096) n_chunks = 3;
      n_chunks = static_cast<int>(std::min<int64_t>(n_chunks, std::max<int64_t>(n, 1)));
    }
    std::vector<int64_t> cb(n_chunks + 1);  // chunk k = rows [cb[k], cb[k + 1])
    cb[0] = row_begin;
    cb[n_chunks] = row_end;
    for (int k = 1; k < n_chunks; k++) {
      const int64_t e_cut = e_begin + static_cast<int64_t>(static_cast<double>(e_end - e_begin) * k / n_chunks);
      cb[k] = std::max<int64_t>(cb[k - 1], std::lower_bound(ip + row_begin, ip + row_end, e_cut) - ip);
    }
    // The row pointers and column indices of the call's rows travel to the device on a second
    // host thread, chunk by chunk, while this one walks them (the values follow later, and only if
    // the kernel reads them).  An index out of range is found by the walk before its chunk's kernel runs.
    irs_knn_computer::Scratch &sc = c->scratch;
    std::string upload_error;
    std::thread uploader;
    std::atomic<int> uploaded(0);  // chunks whose indices are on the device
    if (have_work) {
      IRS_HIP(hipSetDevice(c->device));
      if (!c->stream) {  // (a dropped computer's streams first)
        std::lock_guard<std::mutex> lock(knn_buffer_cache.mu);
        auto &slot = knn_buffer_cache.slot;
        if (slot.device == c->device && slot.streams[0]) {
          hipStream_t *mine[5] = {&c->stream, &c->stream_k, &c->stream_in, &c->stream_up, &c->stream_out};
          for (int k = 0; k < 5; k++) {
            *mine[k] = slot.streams[k];
            slot.streams[k] = nullptr;
          }
          c->ev_in = slot.ev_in;
          c->ev_setup = slot.ev_setup;
          slot.ev_in = slot.ev_setup = nullptr;
        }
      }
      // (each on its own: a creation that failed half way must not leave a later call with null streams)
      for (hipStream_t *st : {&c->stream, &c->stream_k, &c->stream_in, &c->stream_up, &c->stream_out})
        if (!*st) IRS_HIP(hipStreamCreateWithFlags(st, hipStreamNonBlocking));
      for (hipEvent_t *ev : {&c->ev_in, &c->ev_setup})
        if (!*ev) IRS_HIP(hipEventCreateWithFlags(ev, hipEventDisableTiming));
      {  // carve the call's buffers out of the arena (the last result, which lives there too, is gone now)
        const size_t nn = static_cast<size_t>(n);
        const size_t n_tiles_a = static_cast<size_t>(ceil_div(std::max<int64_t>(c->N, 1), TILE));
        const size_t tile_k_a = static_cast<size_t>(std::min<int64_t>(out_k, TILE));
        int64_t widest = 0;  // rows of the largest chunk: the candidate lists are one chunk's at a time
        for (int k = 0; k < n_chunks; k++) widest = std::max(widest, cb[k + 1] - cb[k]);
        const size_t slots_a = static_cast<size_t>(widest) * n_tiles_a, cap_a = nn * static_cast<size_t>(out_k);
        size_t off = 0;
        auto take = [&](size_t bytes) {
          const size_t at = off;
          off += (std::max<size_t>(bytes, 1) + 255) & ~size_t(255);
          return at;
        };
        const size_t o_t_ptr = take((nn + 1) * 8), o_t_idx = take(std::max<size_t>(ne, 1) * 4), o_stat = take(nn * 8),
                     o_scale = take(nn * 8), o_order = take(nn * 4), o_slot = take(nn * 4),
                     o_res_ptr = take((nn + 1) * 8), o_cidx = take(slots_a * tile_k_a * 4),
                     o_cval = take(slots_a * tile_k_a * 8), o_ccnt = take(slots_a * 4), o_oidx = take(cap_a * 4),
                     o_oval = take(cap_a * 8), o_ocnt = take(nn * 4), o_ridx = take(cap_a * 4),
                     o_rval = take(cap_a * 8), o_cursor = take(3 * static_cast<size_t>(n_chunks) * 4),
                     o_redo = take(slots_a * 4);
        if (sc.arena.count < off) {  // (a dropped computer's buffers first)
          std::lock_guard<std::mutex> lock(knn_buffer_cache.mu);
          auto &slot = knn_buffer_cache.slot;
          if (slot.device == c->device && slot.arena && slot.arena_bytes >= off) {
            sc.arena.release();
            sc.arena.ptr = slot.arena;
            sc.arena.count = slot.arena_bytes;
            sc.arena.owned = true;
            slot.arena = nullptr;
            slot.arena_bytes = 0;
          }
          if (slot.device == c->device && slot.stage && slot.stage_bytes > c->stage_bytes) {
            if (c->stage) (void)hipHostFree(c->stage);
            c->stage = slot.stage;
            c->stage_bytes = slot.stage_bytes;
            slot.stage = nullptr;
            slot.stage_bytes = 0;
          }
        }
        sc.arena.alloc(off);
        char *base = sc.arena.ptr;
        sc.t_ptr.view(base + o_t_ptr, nn + 1);
        sc.t_idx.view(base + o_t_idx, std::max<size_t>(ne, 1));
        sc.t_stat.view(base + o_stat, nn);
        sc.t_scale.view(base + o_scale, nn);
        sc.order.view(base + o_order, nn);
        sc.slot_of.view(base + o_slot, nn);
        sc.res_ptr.view(base + o_res_ptr, nn + 1);
        sc.cand_idx.view(base + o_cidx, slots_a * tile_k_a);
        sc.cand_val.view(base + o_cval, slots_a * tile_k_a);
        sc.cand_cnt.view(base + o_ccnt, slots_a);
        sc.out_idx.view(base + o_oidx, cap_a);
        sc.out_val.view(base + o_oval, cap_a);
        sc.out_cnt.view(base + o_ocnt, nn);
        c->res_idx.view(base + o_ridx, cap_a);
        c->res_val.view(base + o_rval, cap_a);
        sc.cursor.view(base + o_cursor, 3 * static_cast<size_t>(n_chunks));
        sc.redo_list.view(base + o_redo, slots_a);
      }
      uploader = std::thread([&] {
        try {
          IRS_HIP(hipSetDevice(c->device));
          std::vector<int64_t> rel(n + 1);
          for (int64_t i = 0; i <= n; i++) rel[i] = ip[row_begin + i] - e_begin;
          IRS_HIP(hipMemcpyAsync(sc.t_ptr.ptr, rel.data(), rel.size() * sizeof(int64_t), hipMemcpyHostToDevice,
                                 c->stream_up));
          for (int k = 0; k < n_chunks; k++) {
            const int64_t q0 = ip[cb[k]], q1 = ip[cb[k + 1]];
            if (q1 > q0)
              IRS_HIP(hipMemcpyAsync(sc.t_idx.ptr + (q0 - e_begin), ix + q0,
                                     static_cast<size_t>(q1 - q0) * sizeof(int32_t), hipMemcpyHostToDevice,
                                     c->stream_up));
            IRS_HIP(hipStreamSynchronize(c->stream_up));
            uploaded.store(k + 1, std::memory_order_release);
          }
        } catch (const std::exception &e) {
          upload_error = e.what();
          (void)hipStreamSynchronize(c->stream_up);  // `rel` goes out of scope
          uploaded.store(n_chunks, std::memory_order_release);
        }
      });
    }
    // (declared after every host array the device work reads: on an exception the uploader is joined
    // and the streams are drained before those arrays go)
    struct Drain {
      std::thread &t;
      irs_knn_computer *c;
      bool device;
      ~Drain() {
        if (t.joinable()) t.join();
        if (device)
          for (hipStream_t st : {c->stream_in, c->stream_k, c->stream_out, c->stream})
            (void)hipStreamSynchronize(st);
      }
    };
    c->res_ptr.assign(n + 1, 0);
    c->res_nnz = 0;
    c->staged = false;
    c->last_ms = 0;
    c->last_macs = 0;
    c->last_walked = 0;
    *nnz_out = 0;
    const int n_tiles = static_cast<int>(ceil_div(std::max<int64_t>(c->N, 1), TILE));
    const bool big = out_k > TOPK_CAP;  // row merge by threshold select (knn_merge_big_kernel)
    const int64_t tile_k = std::min<int64_t>(out_k, TILE);
    std::vector<int32_t> order(std::max<int64_t>(n, 1));  // per chunk: its rows (chunk-relative ids), heaviest first
    std::vector<double> ones_host;  // (binarise on a path that reads values: sized once for the largest chunk
    size_t ones_entries = 0;        //  - copies of it may be in flight, it must never move)
    // per chunk: "its merged rows are in out_idx / out_val"; the call's device span lies between ev_first
    // (before the first launch) and ev_last (after every chunk)
    std::vector<hipEvent_t> ev_done(n_chunks, nullptr);
    hipEvent_t ev_first = nullptr, ev_last = nullptr;
    struct Events {
      std::vector<hipEvent_t> &v;
      hipEvent_t &a, &b;
      ~Events() {
        for (auto e : v) if (e) (void)hipEventDestroy(e);
        if (a) (void)hipEventDestroy(a);
        if (b) (void)hipEventDestroy(b);
      }
    } events_guard{ev_done, ev_first, ev_last};
    std::vector<int32_t> slot_of(std::max<int64_t>(n, 1)), h_cnt(std::max<int64_t>(n, 1));
    Drain drain{uploader, c, have_work};
    hipStream_t s = c->stream, s_in = c->stream_in, s_out = c->stream_out;
    DeviceBuffer<int64_t> &t_ptr = sc.t_ptr;
    DeviceBuffer<int32_t> &t_idx = sc.t_idx, &d_order = sc.order, &out_idx = sc.out_idx, &out_cnt = sc.out_cnt;
    DeviceBuffer<double> &t_val = sc.t_val, &t_stat = sc.t_stat, &t_scale = sc.t_scale, &out_val = sc.out_val;
    int n_cu = 0;
    size_t max_slots = 0;
    bool use_stage = false;
    if (have_work) {
      int64_t max_rows = 0, max_entries = 0;
      for (int k = 0; k < n_chunks; k++) {
        max_rows = std::max(max_rows, cb[k + 1] - cb[k]);
        max_entries = std::max(max_entries, ip[cb[k + 1]] - ip[cb[k]]);
      }
      ones_entries = static_cast<size_t>(max_entries);
      max_slots = static_cast<size_t>(max_rows) * n_tiles;
      const size_t cap = static_cast<size_t>(n) * out_k;  // entries the result can have
      // (c->res_idx / res_val hold `cap` entries: the chunks are compacted as they finish, before the total is known)
      // results of up to 1 GB also travel to a page-locked staging buffer, chunk by chunk as the chunks finish
      // (IRSPACK_AMD_KNN_STAGE=0: irs_knn_fetch copies from the device into the caller's arrays, as before round 5)
      const size_t bytes = cap * (sizeof(double) + sizeof(int32_t));
      // (not on a computer's FIRST call: page-locking 32 MB costs several milliseconds, more than the
      // staged copy saves once - `learn()` of a kNN recommender makes exactly one call)
      if (bytes <= (size_t(1) << 30) && (c->n_calls > 0 || c->stage_bytes >= bytes) &&
          env_flag("IRSPACK_AMD_KNN_STAGE", true)) {
        if (c->stage_bytes < bytes) {
          if (c->stage) (void)hipHostFree(c->stage);
          c->stage = nullptr;
          c->stage_bytes = 0;
          if (hipHostMalloc(reinterpret_cast<void **>(&c->stage), bytes, hipHostMallocDefault) == hipSuccess)
            c->stage_bytes = bytes;
          else
            (void)hipGetLastError();  // (no page-locked memory to be had: fetch from the device)
        }
        use_stage = c->stage_bytes >= bytes;
        c->stage_idx_offset = cap * sizeof(double);
      }
      IRS_HIP(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, c->device));
      // [3 k + 0] pairs, [3 k + 1] pairs of the redo list, [3 k + 2] length of the redo list of chunk k
      IRS_HIP(hipMemsetAsync(sc.cursor.ptr, 0, 3 * static_cast<size_t>(n_chunks) * sizeof(int32_t), s));
      IRS_HIP(hipEventRecord(c->ev_setup, s));
      IRS_HIP(hipStreamWaitEvent(c->stream_k, c->ev_setup, 0));
      IRS_HIP(hipEventCreate(&ev_first));
      IRS_HIP(hipEventCreate(&ev_last));
    }
    // A chunk whose kernels are done: its counts come back, its rows of the CSR get their places (the
    // chunks are contiguous row ranges: the offsets continue from the chunk before), its winners are
    // compacted on the device and copied to the staging buffer - all on a stream of its own beside the
    // kernels of the later chunks.
    int retired = 0;
    bool ev_first_recorded = false;
    auto retire = [&](int j) {
      const int64_t rel0 = cb[j] - row_begin, nc = cb[j + 1] - cb[j];
      if (nc <= 0 || !ev_done[j]) return;
      // (the counts on the call's own stream, idle until the end: behind the earlier chunks' result copies
      // on s_out - copy kernels that crawl while the tile kernels hold the compute units - this thread
      // would wait milliseconds and launch the next chunk late)
      IRS_HIP(hipStreamWaitEvent(s, ev_done[j], 0));
      IRS_HIP(hipMemcpyAsync(h_cnt.data() + rel0, out_cnt.ptr + rel0, nc * sizeof(int32_t), hipMemcpyDeviceToHost, s));
      IRS_HIP(hipStreamSynchronize(s));
      IRS_HIP(hipStreamWaitEvent(s_out, ev_done[j], 0));
      for (int64_t i = rel0; i < rel0 + nc; i++) c->res_ptr[i + 1] = c->res_ptr[i] + h_cnt[slot_of[i]];
      const int64_t q0 = c->res_ptr[rel0], q1 = c->res_ptr[rel0 + nc];
      if (q1 == q0) return;
      IRS_HIP(hipMemcpyAsync(sc.res_ptr.ptr + rel0, c->res_ptr.data() + rel0, (nc + 1) * sizeof(int64_t),
                             hipMemcpyHostToDevice, s_out));
      hipLaunchKernelGGL(knn_compact_kernel, dim3(static_cast<unsigned>(ceil_div(nc, 4))), dim3(256), 0, s_out,
                         out_idx.ptr, out_val.ptr, sc.slot_of.ptr + rel0, sc.res_ptr.ptr + rel0, nc,
                         static_cast<int32_t>(out_k), c->res_idx.ptr, c->res_val.ptr);
      IRS_HIP(hipGetLastError());
      if (use_stage) {
        IRS_HIP(hipMemcpyAsync(c->stage + static_cast<size_t>(q0) * sizeof(double), c->res_val.ptr + q0,
                               static_cast<size_t>(q1 - q0) * sizeof(double), hipMemcpyDeviceToHost, s_out));
        IRS_HIP(hipMemcpyAsync(c->stage + c->stage_idx_offset + static_cast<size_t>(q0) * sizeof(int32_t),
                               c->res_idx.ptr + q0, static_cast<size_t>(q1 - q0) * sizeof(int32_t),
                               hipMemcpyDeviceToHost, s_out));
      }
    };
    pt.mark("set-up");
    for (int ck = 0; ck < n_chunks; ck++) {
      const int64_t r_lo = cb[ck], r_hi = cb[ck + 1], nc = r_hi - r_lo, rel0 = r_lo - row_begin;
      if (nc <= 0 && n > 0) continue;
      const int64_t ce_begin = ip[r_lo], ce_end = ip[r_hi];
      std::atomic<int> bad_index(0), not_ones(0), unsafe(0), not_positive(0), any_wide(0);
      {
        const int n_thr = static_cast<int>(std::max<int64_t>(
            1, std::min<int64_t>({host_thread_cap(), static_cast<int64_t>(std::thread::hardware_concurrency()),
                                  (ce_end - ce_begin) / 200000 + 1})));
        auto body = [&](int th) {
          // contiguous row chunks of about equal entry counts
          const int64_t lo_e = ce_begin + (ce_end - ce_begin) * th / n_thr;
          const int64_t hi_e = ce_begin + (ce_end - ce_begin) * (th + 1) / n_thr;
          int64_t r0 = std::lower_bound(ip + r_lo, ip + r_hi, lo_e) - ip;
          int64_t r1 = th + 1 == n_thr ? r_hi : std::lower_bound(ip + r_lo, ip + r_hi, hi_e) - ip;
          if (th == 0) r0 = r_lo;
          bool bad = false, ones = true, safe = true, positive = true;
          // The usual case - every stored value of the chunk is exactly 1 (binary interactions) -
          // is recognised by one tight pass over the values (a branch-free AND the compiler
          // vectorises); the row loop then only walks the indices.
          bool chunk_ones = true;
          if (!binarise && r1 > r0) {
            uint64_t diff = 0;
            const uint64_t one_bits = 0x3ff0000000000000ull;
            const uint64_t *vb = reinterpret_cast<const uint64_t *>(dv);
            for (int64_t q = ip[r0]; q < ip[r1]; q++) diff |= vb[q] ^ one_bits;
            chunk_ones = diff == 0;
          }
          const bool unit = binarise || chunk_ones;
          for (int64_t i = r0; i < r1; i++) {
            if (ip[i + 1] < ip[i]) { bad = true; break; }
            double ss = 0, bound = 0, minprod = std::numeric_limits<double>::infinity();
            int64_t w = 0;
            if (unit) {
              if (cols <= 0 && ip[i + 1] > ip[i]) { bad = true; break; }
              int32_t lo_j = 0, hi_j = 0;  // running min / max: one range test per row
              for (int64_t q = ip[i]; q < ip[i + 1]; q++) {
                const int32_t j = ix[q];
                lo_j = std::min(lo_j, j);
                hi_j = std::max(hi_j, j);
                const int32_t jc = std::min(std::max(j, 0), static_cast<int32_t>(cols - 1));
                w += c->xt_row_len[jc];
                bound += c->xt_rowmax[jc];
                if (c->xt_rowmin[jc] > 0.0) minprod = std::min(minprod, c->xt_rowmin[jc]);
              }
              if (lo_j < 0 || hi_j >= cols) { bad = true; break; }
              ss = static_cast<double>(ip[i + 1] - ip[i]);
            } else {
              for (int64_t q = ip[i]; q < ip[i + 1]; q++) {
                const int32_t j = ix[q];
                if (j < 0 || j >= cols) { bad = true; break; }
                const double x = dv[q];  // similarities.hpp:113-118, 165-170
                ss += x * x;
                w += c->xt_row_len[j];
                ones &= x == 1.0;
                positive &= x > 0.0;
                const double ax = std::fabs(x);
                safe &= ax > 1e-150 && ax < 1e150;  // no stored zero, no underflow of x * y
                bound += ax * c->xt_rowmax[j];      // >= |any sum of this product row|
                if (ax > 0.0 && c->xt_rowmin[j] > 0.0) minprod = std::min(minprod, ax * c->xt_rowmin[j]);
              }
            }
            if (bad) break;
            switch (c->sim_type) {
              case IRS_SIM_COSINE: tstat[i - row_begin] = std::sqrt(ss); break;             // :39
              case IRS_SIM_ASYMMETRIC: tstat[i - row_begin] = std::pow(ss, c->alpha); break; // :78-79
              case IRS_SIM_JACCARD:
              case IRS_SIM_TVERSKY:
                tstat[i - row_begin] = static_cast<double>(ip[i + 1] - ip[i]);            // :122, :174
                break;
              default: break;
            }
            work[i - row_begin] = w;
            // fixed-point scale of the row: 2^s with bound * 2^s < 2^61
            if (bound > 0 && std::isfinite(bound)) {
              int ex = 0;
              (void)std::frexp(bound, &ex);  // bound < 2^ex
              tscale[i - row_begin] = std::ldexp(1.0, 61 - ex);
              // One limb rounds every product to a multiple of 2^-s ~ bound 2^-61: 1e-13 of the row's
              // smallest possible product as long as bound / minprod < 2.3e5.  Beyond that (weights
              // over many decades, very long rows) the row is summed in two limbs (negative scale):
              // twice the accumulation time, 2^-40 of the rounding step.
              if (wide_ok && std::isfinite(minprod) && bound > wide_ratio * minprod) {
                tscale[i - row_begin] = -tscale[i - row_begin];
                any_wide.store(1, std::memory_order_relaxed);
              }
            }
          }
          if (!positive) not_positive.store(1);
          if (bad) bad_index.store(1);
          if (!ones) not_ones.store(1);
          if (!safe) unsafe.store(1);
        };
        run_on_threads(n_thr, body);
      }
      check_arg(bad_index.load() == 0, "malformed matrix: column index out of range or indptr not monotone.");
      const bool t_all_ones = not_ones.load() == 0, t_safe = unsafe.load() == 0;
      pt.mark("target pass");
      if (!have_work) continue;
      // rows of the chunk, heaviest product row first (ids relative to the chunk's first row)
      int64_t chunk_macs = 0;
      for (int64_t i = 0; i < nc; i++) chunk_macs += work[rel0 + i];
      c->last_macs += chunk_macs;
      {  // Heaviest rows first, for the load balance of the persistent launch only (results do
         // not depend on it): a counting sort by 1/64-octave of the work, rows of a bucket in
         // row order - O(n) instead of a 1.2 ms comparison sort.
        constexpr int NB = 64 * 64;
        auto bucket = [&](int64_t w) {
          const int b = static_cast<int>(std::log2(static_cast<double>(w) + 1.0) * 64.0);
          return NB - 1 - std::min(std::max(b, 0), NB - 1);
        };
        std::vector<int32_t> start(NB + 1, 0);
        std::vector<int32_t> bk(nc);
        for (int64_t i = 0; i < nc; i++) {
          bk[i] = bucket(work[rel0 + i]);
          start[bk[i] + 1]++;
        }
        for (int b = 0; b < NB; b++) start[b + 1] += start[b];
        for (int64_t i = 0; i < nc; i++) order[rel0 + start[bk[i]]++] = static_cast<int32_t>(i);
      }
      pt.mark("work + order");
      // which accumulator: 32-bit counts when every product is 1, else fp64 sums with the -0.0
      // sentinel unless some product could be a zero
      // (the sentinel of the fixed-point sums needs positive data: a sum must not return to the
      // "untouched" pattern by cancellation)
      const bool sentinel = c->xt_nonzero && t_safe && c->xt_positive && not_positive.load() == 0;
      const bool acc32 = c->xt_all_ones && sentinel && t_all_ones;
      // the chunk's inputs, on their own stream (the kernels of the chunk before are still running on `s`)
      const size_t cne = static_cast<size_t>(ce_end - ce_begin);
      if (acc32) {
        t_val.alloc(1);
      } else {
        t_val.alloc(std::max<size_t>(ne, 1));  // (whole call: the row pointers are relative to its first entry)
        if (binarise) {
          if (ones_host.empty()) ones_host.assign(std::max<size_t>(ones_entries, 1), 1.0);
          if (cne) IRS_HIP(hipMemcpyAsync(t_val.ptr + (ce_begin - e_begin), ones_host.data(), cne * sizeof(double),
                                          hipMemcpyHostToDevice, s_in));
        } else if (cne) {
          IRS_HIP(hipMemcpyAsync(t_val.ptr + (ce_begin - e_begin), dv + ce_begin, cne * sizeof(double),
                                 hipMemcpyHostToDevice, s_in));
        }
      }
      IRS_HIP(hipMemcpyAsync(t_stat.ptr + rel0, tstat.data() + rel0, nc * sizeof(double), hipMemcpyHostToDevice, s_in));
      IRS_HIP(hipMemcpyAsync(t_scale.ptr + rel0, tscale.data() + rel0, nc * sizeof(double), hipMemcpyHostToDevice, s_in));
      IRS_HIP(hipMemcpyAsync(d_order.ptr + rel0, order.data() + rel0, nc * sizeof(int32_t), hipMemcpyHostToDevice, s_in));
      // (slots are work-ordered inside their chunk; the compaction reads the inverse map)
      for (int64_t sl = 0; sl < nc; sl++) slot_of[rel0 + order[rel0 + sl]] = static_cast<int32_t>(rel0 + sl);
      IRS_HIP(hipMemcpyAsync(sc.slot_of.ptr + rel0, slot_of.data() + rel0, nc * sizeof(int32_t), hipMemcpyHostToDevice, s_in));
      const size_t slots = static_cast<size_t>(nc) * n_tiles;
      Params p;
      p.xt_tptr = c->xt_tptr.ptr;
      p.xt_idx16 = c->xt_idx16.ptr;
      p.xt_val = c->xt_val.ptr;
      p.xt_rowval = c->xt_row_const ? c->xt_rowval.ptr : nullptr;
      p.norms = c->norms.ptr;
      p.col_scale = c->col_scaled ? c->col_scale.ptr : nullptr;
      p.t_ptr = t_ptr.ptr + rel0;  // (values relative to the CALL's first entry, like t_idx / t_val)
      p.t_idx = t_idx.ptr;
      p.t_val = t_val.ptr;
      p.t_stat = t_stat.ptr + rel0;
      p.t_scale = t_scale.ptr + rel0;
      p.hi_stash = nullptr;
      hipStream_t ks = c->stream_k;
      IRS_HIP(hipEventRecord(c->ev_in, s_in));
      IRS_HIP(hipStreamWaitEvent(ks, c->ev_in, 0));
      c->last_walked += chunk_macs;  // (every multiply-add is added one by one)
      p.row_order = d_order.ptr + rel0;
      p.n_rows = static_cast<int32_t>(nc);
      p.n_tiles = n_tiles;
      p.N = static_cast<int32_t>(c->N);
      p.sim_type = c->sim_type;
      p.normalize = c->normalize ? 1 : 0;
      p.shrinkage = c->shrinkage;
      p.alpha = c->alpha;
      p.beta = c->beta;
      p.shrink_f = static_cast<float>(c->shrinkage);
      p.alpha_f = static_cast<float>(c->alpha);
      p.beta_f = static_cast<float>(c->beta);
      p.top_k = static_cast<int32_t>(out_k);
      p.tile_k = static_cast<int32_t>(tile_k);
      p.cand_idx = sc.cand_idx.ptr;  // (scratch of one chunk at a time: the chunks follow one another on the stream)
      p.cand_val = sc.cand_val.ptr;
      p.cand_cnt = sc.cand_cnt.ptr;
      p.out_idx = out_idx.ptr + static_cast<size_t>(rel0) * out_k;
      p.out_val = out_val.ptr + static_cast<size_t>(rel0) * out_k;
      p.out_cnt = out_cnt.ptr + rel0;
      const size_t lds = TILE * sizeof(double) + (TILE / 32) * sizeof(uint32_t) +
                         256 * sizeof(uint32_t) + 16 * sizeof(int32_t) + 64 * sizeof(double);
      p.cursor = sc.cursor.ptr + 3 * ck;
      p.redo_count = p.cursor + 2;
      p.redo_list = nullptr;
      p.redo = 0;
      {  // merge buffer: all tiles' candidates at once when they fit, else rounds of MERGE_CAP
        int64_t want = std::max<int64_t>(2 * out_k, std::min<int64_t>(int64_t(n_tiles) * out_k, MERGE_CAP));
        int cap = 64;
        while (cap < want) cap <<= 1;
        p.merge_cap = std::min(cap, MERGE_CAP);
      }
      p.n_slots = static_cast<int32_t>(slots);
      // persistent launch: one resident workgroup per CU (its LDS footprint allows no second)
      const unsigned grid = static_cast<unsigned>(std::min<size_t>(slots, static_cast<size_t>(std::max(n_cu, 1))));
      if (!acc32 && any_wide.load()) {  // first limbs of the wide-range pairs, one tile per resident workgroup
        sc.hi_stash.alloc(static_cast<size_t>(std::max(n_cu, 1)) * TILE);
        p.hi_stash = sc.hi_stash.ptr;
      }
      auto launch = [&](auto kernel) {
        IRS_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel),
                                    hipFuncAttributeMaxDynamicSharedMemorySize,
                                    static_cast<int>(lds)));
        hipLaunchKernelGGL(kernel, dim3(grid), dim3(THREADS), lds, ks, p);
      };
      // the chunk's column indices must have arrived (the uploader works ahead of the walk)
      while (uploaded.load(std::memory_order_acquire) <= ck) std::this_thread::yield();
      if (!upload_error.empty()) throw std::runtime_error(upload_error);
      pt.mark("uploads + flags");
      if (!ev_first_recorded) {
        IRS_HIP(hipEventRecord(ev_first, ks));
        ev_first_recorded = true;
      }
      if (c->xt_all_ones) {
        // counts + a similarity that ends in a division: selection on float32 approximations, exact
        // fp64 values for the candidates only (knn_tile_kernel, FAST).  The error bound of the
        // approximation needs non-negative terms in the denominator.  IRSPACK_AMD_KNN_FAST=0: A/B.
        const char *fast_env = std::getenv("IRSPACK_AMD_KNN_FAST");  // (read per call: tests toggle it)
        // (un-normalised cosine: the similarity IS the count - no division to save, but the 32-bit
        // select and the candidate ranking replace the 64-bit select over every column)
        const bool divides = c->sim_type == IRS_SIM_COSINE || c->sim_type == IRS_SIM_ASYMMETRIC ||
                             c->sim_type == IRS_SIM_JACCARD || c->sim_type == IRS_SIM_TVERSKY;
        // (Tversky: norms(j) - v and target_norm - v are exact in float32 only for counts below
        // 2^24; with larger ones and weights of 16 the approximation was measured at 2.1e-6 from the
        // float64 value - tests/test_knn_approx_bound.py - too close to the candidate margin)
        const bool tv_ok = c->sim_type != IRS_SIM_TVERSKY ||
                           (p.alpha >= 0.0 && p.alpha <= 16.0 && p.beta >= 0.0 && p.beta <= 16.0 &&
                            c->norm_max < 16777216.0 &&
                            *std::max_element(tstat.begin() + rel0, tstat.begin() + rel0 + nc) < 16777216.0);
        // (asymmetric cosine: norms are s^(1 - alpha) and s^alpha - inside [1, s] only for alpha in
        // [0, 1]; outside, a float32 norm may underflow where the float64 one does not)
        const bool as_ok = c->sim_type != IRS_SIM_ASYMMETRIC || (p.alpha >= 0.0 && p.alpha <= 1.0);
        // (the candidate list holds FAST_CAP columns: a request for more than half of that would send
        // most pairs to the redo list, i.e. accumulate them twice)
        // (a column scale - a separable weighting - may be zero or negative: exact values for every column)
        const bool fast = acc32 && !c->col_scaled && !big && p.top_k <= FAST_CAP / 2 && divides && tv_ok && as_ok && p.shrinkage >= 0.0 &&
                          p.shrinkage < 1e30 &&
                          !(fast_env && fast_env[0] == '0');
        if (fast) {
          p.redo_list = sc.redo_list.ptr;
          launch(knn_tile_kernel<true, true, true, true>);
          p.redo = 1;  // the pairs the approximate selection handed back (usually none), exactly
          launch(knn_tile_kernel<true, true, true>);
          p.redo = 0;
        } else if (acc32) launch(knn_tile_kernel<true, true, true>);
        else if (sentinel) launch(knn_tile_kernel<true, true>);
        else launch(knn_tile_kernel<true, false>);
      } else if (c->xt_row_const) {  // (the all-ones kernels on y' = x_u y: Params::xt_rowval)
        if (sentinel) launch(knn_tile_kernel<true, true>); else launch(knn_tile_kernel<true, false>);
      } else {
        if (sentinel) launch(knn_tile_kernel<false, true>); else launch(knn_tile_kernel<false, false>);
      }
      if (big) {
        hipLaunchKernelGGL(knn_merge_big_kernel, dim3(static_cast<unsigned>(nc)), dim3(256), 0, ks, p);
      } else {
        IRS_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(knn_merge_kernel),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, MERGE_CAP * 20));
        hipLaunchKernelGGL(knn_merge_kernel, dim3(static_cast<unsigned>(nc)), dim3(256),
                           static_cast<size_t>(p.merge_cap) * 20, ks, p);
      }
      IRS_HIP(hipEventCreateWithFlags(&ev_done[ck], hipEventDisableTiming));
      IRS_HIP(hipEventRecord(ev_done[ck], ks));
      IRS_HIP(hipGetLastError());
      pt.mark("launches");
      // chunks that have finished meanwhile
      while (retired < ck && (!ev_done[retired] || hipEventQuery(ev_done[retired]) == hipSuccess)) retire(retired++);
      (void)hipGetLastError();  // (hipErrorNotReady of the query)
    }
    if (!have_work) return;
    for (; retired < n_chunks; retired++) retire(retired);  // (waits for each in turn)
    for (int ck = 0; ck < n_chunks; ck++)
      if (ev_done[ck]) IRS_HIP(hipStreamWaitEvent(s, ev_done[ck], 0));
    IRS_HIP(hipEventRecord(ev_last, s));
    IRS_HIP(hipStreamSynchronize(s_out));
    IRS_HIP(hipStreamSynchronize(s));
    if (ev_first_recorded) {  // first launch .. last merge, the waits for the host in between included
      float ms = 0;
      IRS_HIP(hipEventElapsedTime(&ms, ev_first, ev_last));
      c->last_ms = ms;
    }
    c->res_nnz = c->res_ptr[n];
    c->staged = use_stage && c->res_nnz > 0;
    c->n_calls++;
    pt.mark("kernels + retire");
#ifdef IRS_KNN_PHASES
    {
      unsigned long long h[8] = {0}, z[8] = {0};
      IRS_HIP(hipMemcpyFromSymbol(h, HIP_SYMBOL(knn_phase_clk), sizeof(h)));
      IRS_HIP(hipMemcpyToSymbol(HIP_SYMBOL(knn_phase_clk), z, sizeof(z)));
      fprintf(stderr, "knn phases (10 ns ticks, sum over the WGs): init %llu acc %llu bitmap %llu epi %llu select %llu write %llu\n",
              h[0], h[1], h[2], h[3], h[4], h[5]);
      IRS_HIP(hipMemcpyFromSymbol(h, HIP_SYMBOL(knn_fast_stat), sizeof(h)));
      IRS_HIP(hipMemcpyToSymbol(HIP_SYMBOL(knn_fast_stat), z, sizeof(z)));
      fprintf(stderr, "knn fast path: pairs %llu exact-path %llu candidates %llu; pairs with <= 128 / 256 / 512 / 1024 candidates: %llu %llu %llu %llu\n",
              h[0], h[1], h[2], h[3], h[4], h[5], h[6]);
    }
#endif
    *nnz_out = c->res_ptr[n];
    pt.mark("assemble");
  });
}

irs_status irs_knn_fetch(irs_knn_computer *c, int64_t *indptr, int32_t *indices, double *data) {
  return guard([&] {
    check_arg(c && indptr, "null argument.");
    std::copy(c->res_ptr.begin(), c->res_ptr.end(), indptr);
    if (c->res_nnz > 0 && c->staged) {  // page-locked copy of the result -> the caller's arrays
      check_arg(indices && data, "null argument.");
      const double *sv = reinterpret_cast<const double *>(c->stage);
      const int32_t *si = reinterpret_cast<const int32_t *>(c->stage + c->stage_idx_offset);
      parallel_ranges(c->res_nnz, [&](int64_t lo, int64_t hi) {
        std::memcpy(data + lo, sv + lo, static_cast<size_t>(hi - lo) * sizeof(double));
        std::memcpy(indices + lo, si + lo, static_cast<size_t>(hi - lo) * sizeof(int32_t));
      }, 8, 200000);
    } else if (c->res_nnz > 0) {  // device -> the caller's arrays, one copy each
      check_arg(indices && data, "null argument.");
      IRS_HIP(hipSetDevice(c->device));
      IRS_HIP(hipMemcpy(indices, c->res_idx.ptr, static_cast<size_t>(c->res_nnz) * sizeof(int32_t),
                        hipMemcpyDeviceToHost));
      IRS_HIP(hipMemcpy(data, c->res_val.ptr, static_cast<size_t>(c->res_nnz) * sizeof(double),
                        hipMemcpyDeviceToHost));
    }
  });
}

irs_status irs_knn_fetch_csc(irs_knn_computer *c, int64_t zero_diagonal_row0, int64_t *col_ptr, int32_t *row_idx,
                             double *data) {
  return guard([&] {
    check_arg(c && col_ptr, "null argument.");
    const int64_t n = static_cast<int64_t>(c->res_ptr.size()) - 1, N = c->N, nnz = c->res_nnz;
    std::fill(col_ptr, col_ptr + N + 1, int64_t(0));
    if (nnz <= 0) return;
    check_arg(row_idx && data, "null argument.");
    check_arg(nnz < (int64_t(1) << 31), "nnz must be below 2^31.");
    IRS_HIP(hipSetDevice(c->device));
    hipStream_t s = c->stream;  // (idle: the compute call that made the result has returned)
    // Scratch out of the compute call's arena: its merged-row buffers (out_idx / out_val, sized like the result)
    // and the target's column indices (t_idx: tens of MB) are dead once the call has returned - five hipMalloc /
    // hipFree pairs per learn() were 4 of this call's 13 ms.
    irs_knn_computer::Scratch &sc = c->scratch;
    const bool arena = sc.out_idx.ptr && sc.out_idx.count >= static_cast<size_t>(nnz) &&
                       sc.out_val.count >= static_cast<size_t>(nnz) && sc.t_idx.ptr;
    DeviceBuffer<int32_t> own_tidx, d_rp, d_cp;
    DeviceBuffer<double> own_tval;
    DeviceBuffer<char> tmp;
    int32_t *t_idx = nullptr;
    double *t_val = nullptr;
    if (arena) {
      t_idx = sc.out_idx.ptr;
      t_val = sc.out_val.ptr;
      tmp.view(sc.t_idx.ptr, sc.t_idx.count * sizeof(int32_t));
    } else {
      own_tidx.alloc(static_cast<size_t>(nnz));
      own_tval.alloc(static_cast<size_t>(nnz));
      t_idx = own_tidx.ptr;
      t_val = own_tval.ptr;
    }
    std::vector<int32_t> rp(static_cast<size_t>(n) + 1);
    for (int64_t i = 0; i <= n; i++) rp[i] = static_cast<int32_t>(c->res_ptr[i]);
    d_rp.upload(rp, s);
    std::vector<int32_t> t_count;
    transpose_csr_device(d_rp.ptr, c->res_idx.ptr, static_cast<const double *>(c->res_val.ptr), n, N, nnz, t_idx, t_val,
                         t_count, tmp, s);
    for (int64_t j = 0; j < N; j++) col_ptr[j + 1] = col_ptr[j] + t_count[j];
    if (zero_diagonal_row0 >= 0) {
      std::vector<int32_t> cp(static_cast<size_t>(N) + 1);
      for (int64_t j = 0; j <= N; j++) cp[j] = static_cast<int32_t>(col_ptr[j]);
      d_cp.upload(cp, s);
      hipLaunchKernelGGL(knn_zero_diagonal_csc_kernel, dim3(static_cast<unsigned>(ceil_div(N, 256))), dim3(256), 0, s,
                         static_cast<const int32_t *>(d_cp.ptr), static_cast<const int32_t *>(t_idx), N,
                         zero_diagonal_row0, t_val);
      IRS_HIP(hipGetLastError());
      IRS_HIP(hipStreamSynchronize(s));  // (cp goes out of scope)
    }
    // To the caller's (pageable, usually untouched) arrays: through the computer's page-locked staging buffer
    // when it has one of the size (one copy at the link's rate + a threaded host copy; the staged CSR copy of the
    // result is gone afterwards), else a few device-to-host copies side by side (see irs_knn_weight).
    const size_t need = static_cast<size_t>(nnz) * (sizeof(double) + sizeof(int32_t));
    if (c->stage && c->stage_bytes >= need) {
      c->staged = false;
      char *sv = c->stage, *si = c->stage + static_cast<size_t>(nnz) * sizeof(double);
      IRS_HIP(hipMemcpyAsync(sv, t_val, static_cast<size_t>(nnz) * sizeof(double), hipMemcpyDeviceToHost, s));
      IRS_HIP(hipMemcpyAsync(si, t_idx, static_cast<size_t>(nnz) * sizeof(int32_t), hipMemcpyDeviceToHost, s));
      IRS_HIP(hipStreamSynchronize(s));
      parallel_ranges(nnz, [&](int64_t lo, int64_t hi) {
        std::memcpy(data + lo, reinterpret_cast<const double *>(sv) + lo, static_cast<size_t>(hi - lo) * sizeof(double));
        std::memcpy(row_idx + lo, reinterpret_cast<const int32_t *>(si) + lo, static_cast<size_t>(hi - lo) * sizeof(int32_t));
      }, 8, 200000);
      return;
    }
    const int n_copy = static_cast<int>(std::max<int64_t>(1, std::min<int64_t>(4, nnz / (int64_t(1) << 20))));
    std::atomic<int> failed(0);
    run_on_threads(n_copy, [&](int k) {
      const int64_t q0 = nnz * k / n_copy, q1 = nnz * (k + 1) / n_copy;
      if (q1 <= q0) return;
      if (hipSetDevice(c->device) != hipSuccess ||
          hipMemcpy(data + q0, t_val + q0, static_cast<size_t>(q1 - q0) * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess ||
          hipMemcpy(row_idx + q0, t_idx + q0, static_cast<size_t>(q1 - q0) * sizeof(int32_t), hipMemcpyDeviceToHost) != hipSuccess)
        failed.store(1);
    });
    if (failed.load()) throw std::runtime_error("copying the result to the host failed.");
  });
}

irs_status irs_knn_last_walked(irs_knn_computer *c, int64_t *walked_macs, int32_t *dense_rows) {
  return guard([&] {
    check_arg(c && walked_macs && dense_rows, "null argument.");
    *walked_macs = c->last_walked;
    *dense_rows = 0;  // (the dense block of the popular items was removed in round 5)
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
