// Kernels of the iALS evaluator's fused call (irs_eval_get_metrics_ials) that never write the
// [users, items] score block: the mask as a bitmap (below) and the threshold-filtered "emit" path
// (second half of this file).  Semantics: evaluator.cpp:292-367 (rank by (score desc, index asc) among
// the scores != -inf) on scores bit-identical to user_scores_kernel's (the same MFMA sequence per
// (user, item): k in ascending steps of 16, the four 4-slices of a step in order).
// (A single-pass variant - streaming top-k lists in LDS inside the scoring kernel, `fused_topk_kernel` +
// `fused_finish_kernel`, IRSPACK_AMD_EVAL_FUSED=1 - was built in round 2, measured slower than the
// two-pass path at every K (16.5 against 11.6 - 13.0 ms at K = 64: one wave per SIMD beside 39 KB of
// LDS per wave) and then superseded by the emit path; deleted in round 5)
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <limits>

namespace irs {
namespace eval {

typedef float fz_f32x4 __attribute__((ext_vector_type(4)));

// Orders this wave's LDS traffic (cross-lane hand-over through LDS) WITHOUT touching the
// vector-memory counter: __threadfence_block() compiles to s_waitcnt vmcnt(0) lgkmcnt(0) and
// would drain the operand loads that are in flight for the next tile.
#define FZ_LDS_FENCE() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")

constexpr int FZ_MAX_CUTOFF = 32;  // largest cutoff of the fused paths
constexpr int FZ_SROW = 68;        // LDS row stride of the score tile (floats)

// mask CSR rows -> bitmap rows (the bitmap is zeroed by the host first), one wave per row
__global__ __launch_bounds__(256) void mask_bitmap_kernel(const int64_t *__restrict__ mask_ptr,
                                                          const int32_t *__restrict__ mask_idx,
                                                          int64_t rows, int64_t words,
                                                          uint64_t *__restrict__ bits) {
  const int64_t row = static_cast<int64_t>(blockIdx.x) * 4 + wave_index_in_block();
  if (row >= rows) return;
  const int ln = threadIdx.x & 63;
  unsigned long long *dst = reinterpret_cast<unsigned long long *>(bits + row * words);
  for (int64_t q = mask_ptr[row] + ln; q < mask_ptr[row + 1]; q += 64) {
    const int32_t j = mask_idx[q];
    atomicOr(dst + (j >> 6), 1ull << (j & 63));
  }
}

// masked items per row (duplicates in the CSR count once): n_rankable = n_items - this
__global__ __launch_bounds__(256) void mask_count_kernel(const uint64_t *__restrict__ bits,
                                                         int64_t rows, int64_t words,
                                                         int32_t *__restrict__ n_masked) {
  const int64_t row = static_cast<int64_t>(blockIdx.x) * 4 + wave_index_in_block();
  if (row >= rows) return;
  const int ln = threadIdx.x & 63;
  int c = 0;
  for (int64_t w = ln; w < words; w += 64) c += __popcll(bits[row * words + w]);
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) c += __shfl_xor(c, o, 64);
  if (ln == 0) n_masked[row] = c;
}

}  // namespace eval
}  // namespace irs

// ===========================================================================================
// Threshold-filtered scoring ("emit" path, the default of irs_eval_get_metrics_ials when it
// applies).  Three launches, none of which writes the [users, items] score block:
//   1. sample pass (existing kernels): the scores of the first S items, masked, and per user
//      the cutoff-th best of them = tau_u (sample_tau_kernel).  The user's final cutoff-th
//      best score is >= tau_u, so every item of the final list has a score >= tau_u.
//   2. score_emit_kernel: the whole score matrix, one wave per 64 x 64 tile exactly as
//      user_scores_kernel computes it (bit-identical scores); the tile goes to the wave's LDS
//      slab, lane l becomes user l, and the scores >= tau_l that are not masked are appended
//      to the user's candidate list in global memory (one returning atomic per user and tile
//      that has any; ~1 % of the scores pass, I / S x cutoff per user).
//   3. rank_cand_kernel: one wave per user ranks the candidates by (score desc, index asc)
//      like rank_wave_kernel ranks a score row, and finishes with the same metrics code.
// A non-finite score, a user with fewer than `cutoff` rankable sample items, or an overflowing
// candidate list abandon the path (flag) and the host runs the two-pass one.
namespace irs {
namespace eval {

constexpr int EM_CAP = 1024;     // candidate slots per user
constexpr int EM_SAMPLE = 2048;  // items of the sample pass (items in catalogue order)
constexpr int EM_SAMPLE_SORTED = 512;  // ... when the items are sorted by norm (bounded variant)
constexpr int EM_SAMPLE2 = 4096;       // second-chance sample of the rows the first left hard
constexpr int EM_HARD_CAP = 2048;      // rows the second chance handles

struct EmitParams {
  const float *user, *item;
  int64_t begin, rows, n_items;
  const uint64_t *mask_bits;  // [rows, words] or null
  int64_t words;
  const float *tau;           // [rows]
  float *cand_score;          // [rows, EM_CAP]
  int32_t *cand_item;
  int32_t *cand_cnt;          // [rows]
  int32_t *bad_flag;          // bit 0: non-finite score, bit 2: list overflow
  // bounded variant (below): sorted position -> item id / row of the call, and per 64-user
  // tile the number of leading item tiles that can still hold a candidate
  const int32_t *iperm, *uperm, *limit_tiles;
  // the work list of the bounded variant: workgroup b scores the item tiles wg_desc[b].y ..
  // + wg_desc[b].z - 1 (at most four) of user tile wg_desc[b].x - one 16-byte load instead of
  // a chain of three (only workgroups with a live tile are launched)
  const int4 *wg_desc;
  const int32_t *n_wg;  // length of the work list (device side), or null: one entry per workgroup of the grid
  // rows the path cannot finish (no threshold from the sample, candidate list overflow): set
  // to 1 here, ranked one by one from their full score rows afterwards
  int32_t *hard;
};

// tau_u = the cutoff-th best score of the user's (masked) sample block, -inf when the block
// holds fewer rankable scores (the caller then abandons the path).  One wave per user; lane l
// keeps the M best of the scores l, l + 64, ...; the list is drawn from the 64 heads and a
// lane that runs dry is simply exhausted: what it hides could only RAISE the true cutoff-th
// best, so the value found is still a lower bound.
template <int M>
__global__ __launch_bounds__(256) void sample_tau_kernel(const float *__restrict__ scores,
                                                         int64_t rows, int64_t n_sample,
                                                         int32_t cutoff, float *__restrict__ tau,
                                                         int32_t *__restrict__ bad_flag,
                                                         int32_t *__restrict__ hard,
                                                         const int32_t *__restrict__ row_list,
                                                         const int32_t *__restrict__ n_list) {
  const int ln = threadIdx.x & 63;
  const int64_t row = static_cast<int64_t>(blockIdx.x) * 4 + wave_index_in_block();
  if (row >= rows) return;
  // second chance of the hard rows: score row j belongs to row row_list[j] of the call
  if (row_list && row >= *n_list) return;
  const int64_t out_row = row_list ? row_list[row] : row;
  const float NEG_INF = -std::numeric_limits<float>::infinity();
  const float *srow = scores + row * n_sample;
  float bs[M];
#pragma unroll
  for (int t = 0; t < M; t++) bs[t] = NEG_INF;
  int n_rankable = 0;
  bool bad = false;
  for (int64_t base = 0; base < n_sample; base += 64 * 8) {
    float sv[8];
#pragma unroll
    for (int q = 0; q < 8; q++) sv[q] = srow[min<int64_t>(base + 64 * q + ln, n_sample - 1)];
#pragma unroll
    for (int q = 0; q < 8; q++) {
      float cs = base + 64 * q + ln < n_sample ? sv[q] : NEG_INF;
      bad |= cs != cs || cs == std::numeric_limits<float>::infinity();
      n_rankable += __popcll(__ballot(cs != NEG_INF));
      if (__any(cs > bs[M - 1])) {
#pragma unroll
        for (int t = 0; t < M; t++) {
          const bool gt = cs > bs[t];
          const float ts = bs[t];
          bs[t] = gt ? cs : ts;
          cs = gt ? ts : cs;
        }
      }
    }
  }
  if (__any(bad)) {
    if (ln == 0) atomicOr(bad_flag, 1);
  }
  float last = NEG_INF;
  if (n_rankable >= cutoff) {
    for (int it = 0; it < cutoff; it++) {
      float ws = bs[0];
#pragma unroll
      for (int o = 32; o >= 1; o >>= 1) ws = fmaxf(ws, __shfl_xor(ws, o, 64));
      last = ws;
      const unsigned long long who = __ballot(bs[0] == ws);
      if (ln == __ffsll(static_cast<long long>(who)) - 1) {  // one lane pops its head
#pragma unroll
        for (int t = 0; t + 1 < M; t++) bs[t] = bs[t + 1];
        bs[M - 1] = NEG_INF;
      }
    }
  } else {
    // too few rankable sample items for a threshold (a user who has seen nearly all of them)
    last = std::numeric_limits<float>::infinity();
  }
  if (ln == 0) {
    hard[out_row] = n_rankable >= cutoff ? 0 : 1;
    tau[out_row] = last;
  }
}

// The sample pass in ONE launch (round 6; bounded variant, 512 sample items, K <= 128): scores, mask and
// the cutoff-th best per user without the [users, 512] score block in memory (three launches per 32,768
// users before: user_scores_kernel, mask_entries_perm_kernel, sample_tau_kernel - 0.58 of the call's 1.5 ms).
// A 512-thread workgroup stays resident and walks tiles of 16 users.  Wave w holds the operands of the
// sample items 64 w .. 64 w + 63 in REGISTERS for the whole launch; per tile it scores 16 users x its 64
// items (the MFMA sequence of user_scores_kernel: same bits), writes them to the tile's LDS slab (two
// slabs, so one barrier per tile), and after the barrier owns the users 2 w, 2 w + 1 of the tile: lane l
// takes the scores l, l + 64, ..., the user's mask entries that fall into the sample were marked in a
// 512-byte LDS row by the same wave before the MFMAs (mask CSR -> sorted position through `iinv`), and the
// cutoff-th largest is found EXACTLY by a bit-by-bit search on order-preserving keys (32 steps of eight
// compares + ballot counts on the scalar unit) - the value sample_tau_kernel finds by popping heads.
struct SampleParams {
  const float *user;        // factors of the call's side; row r of the call = user + (begin + r) * KP
  const float *sample;      // [512, KP]: the sample items, sorted position order
  int64_t begin, rows;
  const int64_t *mask_ptr;  // mask CSR of the call's rows, or null
  const int32_t *mask_idx;
  const int32_t *iinv;      // item id -> sorted position
  int32_t cutoff;
  float *tau;               // [rows]
  int32_t *hard;            // [rows]
  int32_t *bad_flag;        // bit 0: non-finite score
};
constexpr int SF_ITEMS = 512, SF_USERS = 16, SF_SROW = 516;  // (516: rows 4 apart land 16 banks apart)
constexpr size_t SF_LDS_BYTES = 2 * SF_USERS * SF_SROW * sizeof(float) + 8 * 2 * SF_ITEMS;

__device__ __forceinline__ uint32_t sf_key(float v) {  // order-preserving: a < b <=> key(a) < key(b)
  const uint32_t b = __float_as_uint(v);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}

template <int KP>
__global__ __launch_bounds__(512, KP <= 64 ? 4 : 2) void sample_tau_fused_kernel(SampleParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char sf_smem[];
  float *slab = reinterpret_cast<float *>(sf_smem);
  const int wid = wave_index_in_block(), ln = threadIdx.x & 63;
  const int g = ln >> 4, m = ln & 15;
  unsigned char *mk = sf_smem + 2 * SF_USERS * SF_SROW * sizeof(float) + wid * (2 * SF_ITEMS);
  constexpr int NJ = KP / 16;
  const float NEG_INF = -std::numeric_limits<float>::infinity();
  // B operands: item 64 wid + 16 bb + m, dims 16 j + 4 g .. + 3
  fz_f32x4 bv[4][NJ];
#pragma unroll
  for (int bb = 0; bb < 4; bb++)
#pragma unroll
    for (int j = 0; j < NJ; j++)
      bv[bb][j] = *reinterpret_cast<const fz_f32x4 *>(p.sample + static_cast<size_t>(64 * wid + 16 * bb + m) * KP + 16 * j + 4 * g);
  const int64_t n_tiles = (p.rows + SF_USERS - 1) / SF_USERS;
  int buf = 0;
  int64_t nq[3] = {0, 0, 0};  // mask row bounds of this wave's two users in the NEXT tile
  if (p.mask_ptr) {
    const int64_t nr = static_cast<int64_t>(blockIdx.x) * SF_USERS + 2 * wid;
#pragma unroll
    for (int u = 0; u < 3; u++) nq[u] = p.mask_ptr[min(nr + u, p.rows)];
  }
  for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x, buf ^= 1) {
    const int64_t r0 = tile * SF_USERS;
    // A operands: user r0 + m, dims 16 j + 4 g .. + 3 (the eight waves load the same 16 rows)
    fz_f32x4 av[NJ];
    {
      const float *up = p.user + (p.begin + min(r0 + m, p.rows - 1)) * KP + 4 * g;
#pragma unroll
      for (int j = 0; j < NJ; j++) av[j] = *reinterpret_cast<const fz_f32x4 *>(up + 16 * j);
    }
    // this wave's two users: mark the sample positions their mask holds.  The walk is a chain of dependent
    // loads (row bounds -> item ids -> sorted positions -> LDS byte); the bounds were fetched one tile
    // ahead, and the first MK_AHEAD x 64 entries of both rows are requested HERE, before the MFMAs, with
    // clamped addresses (no load under a branch), so the chain costs two round trips per tile, not eight.
    constexpr int MK_AHEAD = 3;
    int32_t mid[2][MK_AHEAD];
    int64_t mq0[2] = {0, 0}, mq1[2] = {0, 0};
    if (p.mask_ptr) {
#pragma unroll
      for (int u = 0; u < 2; u++) {
        mq0[u] = nq[u];
        mq1[u] = nq[u + 1];
      }
      // (the next tile's bounds: rows past the end repeat the last pointer = empty rows)
      const int64_t nr = (tile + gridDim.x) * SF_USERS + 2 * wid;
#pragma unroll
      for (int u = 0; u < 3; u++) nq[u] = p.mask_ptr[min(nr + u, p.rows)];
#pragma unroll
      for (int u = 0; u < 2; u++)
#pragma unroll
        for (int k = 0; k < MK_AHEAD; k++) {
          const int64_t q = mq0[u] + 64 * k + ln;
          mid[u][k] = p.mask_idx[max(min(q, mq1[u] - 1), static_cast<int64_t>(0))];
        }
    }
    fz_f32x4 acc[4];
#pragma unroll
    for (int bb = 0; bb < 4; bb++) acc[bb] = fz_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < NJ; j++)
#pragma unroll
      for (int c = 0; c < 4; c++)
#pragma unroll
        for (int bb = 0; bb < 4; bb++)
          acc[bb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j][c], bv[bb][j][c], acc[bb], 0, 0, 0);
    float *S = slab + buf * (SF_USERS * SF_SROW);
    // acc[bb][r] of lane (g, m): user 4 g + r, item 64 wid + 16 bb + m
#pragma unroll
    for (int bb = 0; bb < 4; bb++)
#pragma unroll
      for (int r = 0; r < 4; r++) S[(4 * g + r) * SF_SROW + 64 * wid + 16 * bb + m] = acc[bb][r];
    if (p.mask_ptr) {
      reinterpret_cast<uint64_t *>(mk)[ln] = 0ull;
      reinterpret_cast<uint64_t *>(mk)[64 + ln] = 0ull;
      int32_t pos[2][MK_AHEAD];
#pragma unroll
      for (int u = 0; u < 2; u++)
#pragma unroll
        for (int k = 0; k < MK_AHEAD; k++)  // (an id fetched past the row's end is not an address)
          pos[u][k] = p.iinv[mq0[u] + 64 * k + ln < mq1[u] ? mid[u][k] : 0];
#pragma unroll
      for (int u = 0; u < 2; u++) {
#pragma unroll
        for (int k = 0; k < MK_AHEAD; k++)
          if (mq0[u] + 64 * k + ln < mq1[u] && pos[u][k] < SF_ITEMS) mk[SF_ITEMS * u + pos[u][k]] = 1;
        // rows longer than MK_AHEAD x 64 entries (wave-uniform trip count)
        for (int64_t q = mq0[u] + 64 * MK_AHEAD + ln; q - ln < mq1[u]; q += 64) {
          const int32_t ps = p.iinv[p.mask_idx[min(q, mq1[u] - 1)]];
          if (q < mq1[u] && ps < SF_ITEMS) mk[SF_ITEMS * u + ps] = 1;
        }
      }
    }
    __syncthreads();
    // ---- the two users of this wave: lane l holds the scores l + 64 q
    uint32_t key[2][8];
    int n_rank[2];
    bool bad = false;
#pragma unroll
    for (int u = 0; u < 2; u++) {
      const float *srow = S + (2 * wid + u) * SF_SROW + ln;
      n_rank[u] = 0;
#pragma unroll
      for (int q = 0; q < 8; q++) {
        float v = srow[64 * q];
        if (p.mask_ptr && mk[SF_ITEMS * u + 64 * q + ln]) v = NEG_INF;
        bad |= v != v || v == std::numeric_limits<float>::infinity();
        n_rank[u] += __popcll(__ballot(v != NEG_INF));
        key[u][q] = sf_key(v);
      }
    }
    if (__any(bad && r0 + 2 * wid < p.rows)) {
      if (ln == 0) atomicOr(p.bad_flag, 1);
    }
    // the cutoff-th largest key, bit by bit from the top (both users in step: independent chains)
    // (a prefix that exactly `cutoff` keys reach ends the search early: the answer is the smallest of them -
    // with distinct scores that happens after ~14 of the 32 bits)
    uint32_t t[2] = {0u, 0u};
    bool exact_set[2] = {false, false};
    for (int bit = 31; bit >= 0; bit--) {
#pragma unroll
      for (int u = 0; u < 2; u++) {
        if (!exact_set[u]) {  // (wave-uniform)
          const uint32_t cand = t[u] | (1u << bit);
          int cnt = 0;
#pragma unroll
          for (int q = 0; q < 8; q++) cnt += __popcll(__ballot(key[u][q] >= cand));
          if (cnt >= p.cutoff) t[u] = cand;
          exact_set[u] = cnt == p.cutoff;
        }
      }
      if (exact_set[0] && exact_set[1]) break;
    }
#pragma unroll
    for (int u = 0; u < 2; u++) {
      if (exact_set[u]) {
        uint32_t mn = 0xffffffffu;
#pragma unroll
        for (int q = 0; q < 8; q++) mn = min(mn, key[u][q] >= t[u] ? key[u][q] : 0xffffffffu);
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) mn = min(mn, static_cast<uint32_t>(__shfl_xor(static_cast<int>(mn), o, 64)));
        t[u] = mn;
      }
    }
#pragma unroll
    for (int u = 0; u < 2; u++) {
      const int64_t row = r0 + 2 * wid + u;
      if (row < p.rows && ln == 0) {
        const bool ok = n_rank[u] >= p.cutoff;
        const uint32_t b = (t[u] & 0x80000000u) ? (t[u] & 0x7fffffffu) : ~t[u];
        p.hard[row] = ok ? 0 : 1;
        // too few rankable sample items for a threshold: +inf (as sample_tau_kernel)
        p.tau[row] = ok ? __uint_as_float(b) : std::numeric_limits<float>::infinity();
      }
    }
  }
}

// BOUNDED: users and items are addressed through the two sort permutations (tile rows = 64
// users of similar pruning radius, tile columns = 64 items of similar norm), tiles at or beyond
// the user tile's limit leave at once, and the mask is probed per passing score (the bitmap is
// indexed by item id, not by sorted position).
template <int KP, bool BOUNDED>
__device__ __forceinline__ void score_emit_body(const EmitParams &p, unsigned char *em_smem, const int64_t bidx) {
  const int wid = wave_index_in_block(), ln = threadIdx.x & 63;
  const int g = ln >> 4, m = ln & 15;
  float *S = reinterpret_cast<float *>(em_smem) + wid * (64 * FZ_SROW);
  int32_t *IDS = reinterpret_cast<int32_t *>(em_smem + 4 * 64 * FZ_SROW * sizeof(float)) + wid * 64;
  const int64_t item_tiles = (p.n_items + 63) / 64;
  int64_t ut, it;
  if constexpr (BOUNDED) {
    const int4 d = p.wg_desc[bidx];
    ut = d.x;
    it = static_cast<int64_t>(d.y) + wid;
    if (wid >= d.z) return;
  } else {
    const int64_t w = bidx * 4 + wid;
    ut = w / item_tiles;
    it = w % item_tiles;
    if (ut * 64 >= p.rows) return;
  }
  // lane l as user l: threshold and mask word of this tile (requested now, used after the MFMAs)
  const int64_t my_pos = ut * 64 + ln;
  const bool my_valid = my_pos < p.rows;
  int64_t my_row = my_pos;
  if constexpr (BOUNDED) my_row = my_valid ? p.uperm[my_pos] : 0;
  const float tau = my_valid ? p.tau[my_row] : std::numeric_limits<float>::infinity();
  uint64_t mword = 0ull;
  if constexpr (!BOUNDED) mword = (p.mask_bits && my_valid) ? p.mask_bits[my_row * p.words + it] : 0ull;
  int32_t my_item = 0;
  if constexpr (BOUNDED) my_item = p.iperm[min(it * 64 + ln, p.n_items - 1)];
  // ---- the 64 x 64 score tile, as user_scores_kernel
  const float *up[4], *ip[4];
#pragma unroll
  for (int q = 0; q < 4; q++) {
    int64_t u = min(ut * 64 + q * 16 + m, p.rows - 1);
    int64_t i = min(it * 64 + 4 * m + q, p.n_items - 1);
    if constexpr (BOUNDED) {
      u = p.uperm[u];
      i = p.iperm[i];
    }
    up[q] = p.user + (p.begin + u) * KP + 4 * g;
    ip[q] = p.item + i * KP + 4 * g;
  }
  fz_f32x4 acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; a++)
#pragma unroll
    for (int b = 0; b < 4; b++) acc[a][b] = fz_f32x4{0.f, 0.f, 0.f, 0.f};
  fz_f32x4 av[2][4], bv[2][4];
  auto load_step = [&](int k, fz_f32x4 (&aa)[4], fz_f32x4 (&bb)[4]) {
#pragma unroll
    for (int q = 0; q < 4; q++) {
      aa[q] = *reinterpret_cast<const fz_f32x4 *>(up[q] + k);
      bb[q] = *reinterpret_cast<const fz_f32x4 *>(ip[q] + k);
    }
  };
  auto mfma_step = [&](const fz_f32x4 (&aa)[4], const fz_f32x4 (&bb)[4]) {
#pragma unroll
    for (int c = 0; c < 4; c++)
#pragma unroll
      for (int a = 0; a < 4; a++)
#pragma unroll
        for (int b = 0; b < 4; b++)
          acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(aa[a][c], bb[b][c], acc[a][b], 0, 0, 0);
  };
  load_step(0, av[0], bv[0]);
  __builtin_amdgcn_sched_barrier(0);
  if constexpr (KP == 16) {
    mfma_step(av[0], bv[0]);
  } else {
#pragma unroll
    for (int k = 0; k < KP; k += 32) {
      load_step(k + 16, av[1], bv[1]);
      __builtin_amdgcn_sched_barrier(0);
      mfma_step(av[0], bv[0]);
      __builtin_amdgcn_sched_barrier(0);
      if (k + 32 < KP) load_step(k + 32, av[0], bv[0]);
      __builtin_amdgcn_sched_barrier(0);
      mfma_step(av[1], bv[1]);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  // ---- tile -> LDS (row = user 16 a + 4 g + r, columns 4 m .. 4 m + 3 = items 4 m + q)
#pragma unroll
  for (int a = 0; a < 4; a++)
#pragma unroll
    for (int r = 0; r < 4; r++)
      *reinterpret_cast<fz_f32x4 *>(S + (16 * a + 4 * g + r) * FZ_SROW + 4 * m) =
          fz_f32x4{acc[a][0][r], acc[a][1][r], acc[a][2][r], acc[a][3][r]};
  if constexpr (BOUNDED) IDS[ln] = my_item;
  FZ_LDS_FENCE();
  // ---- lane l scans user l's 64 scores: which pass?
  const int valid_items = static_cast<int>(min<int64_t>(64, p.n_items - it * 64));
  uint32_t plo = 0, phi = 0;
  float fsum = 0.f;
#pragma unroll
  for (int jj = 0; jj < 16; jj++) {
    const fz_f32x4 v = *reinterpret_cast<const fz_f32x4 *>(S + ln * FZ_SROW + 4 * jj);
    // NaN / infinity anywhere poisons the sum (the bounded variant has checked the norms)
    if constexpr (!BOUNDED) fsum += (v.x + v.y) + (v.z + v.w);
#pragma unroll
    for (int c = 0; c < 4; c++) {
      const int j = 4 * jj + c;
      const uint32_t bit = v[c] >= tau ? (1u << (j & 31)) : 0u;
      if (j < 32) plo |= bit; else phi |= bit;
    }
  }
  if constexpr (!BOUNDED) {
    if (__any(my_valid && !(fabsf(fsum) < 3.0e38f))) {
      if (ln == 0) atomicOr(p.bad_flag, 1);
    }
  }
  uint64_t pm = (static_cast<uint64_t>(phi) << 32) | plo;
  pm &= ~mword;
  if (valid_items < 64) pm &= (1ull << valid_items) - 1ull;
  if (!my_valid) pm = 0ull;
  if (!__any(pm != 0ull)) return;
  if constexpr (BOUNDED) {
    if (p.mask_bits) {  // four independent probes per trip: bit (item id) of the user's bitmap row
      const uint64_t *mrow = p.mask_bits + my_row * p.words;
      uint64_t left = pm;
      while (__any(left != 0ull)) {
        int j[4];
        int32_t id[4];
        uint64_t w[4];
#pragma unroll
        for (int t = 0; t < 4; t++) {
          j[t] = -1;
          if (left != 0ull) {
            j[t] = __ffsll(static_cast<long long>(left)) - 1;
            left &= left - 1;
          }
          id[t] = j[t] >= 0 ? IDS[j[t]] : 0;
        }
#pragma unroll
        for (int t = 0; t < 4; t++) w[t] = j[t] >= 0 ? mrow[id[t] >> 6] : 0ull;
#pragma unroll
        for (int t = 0; t < 4; t++)
          if (j[t] >= 0 && ((w[t] >> (id[t] & 63)) & 1ull)) pm &= ~(1ull << j[t]);
      }
      if (!__any(pm != 0ull)) return;
    }
  }
  const int n_pass = __popcll(pm);
  int base = 0;
  if (n_pass > 0) base = atomicAdd(p.cand_cnt + my_row, n_pass);
  if (n_pass > 0 && base + n_pass > EM_CAP) {
    p.hard[my_row] = 1;
    pm = 0ull;
  }
  while (__any(pm != 0ull)) {
    if (pm != 0ull) {
      const int j = __ffsll(static_cast<long long>(pm)) - 1;
      pm &= pm - 1;
      const size_t dst = static_cast<size_t>(my_row) * EM_CAP + base;
      p.cand_score[dst] = S[ln * FZ_SROW + j];
      p.cand_item[dst] = BOUNDED ? IDS[j] : static_cast<int32_t>(it * 64 + j);
      base++;
    }
  }
}

// The bounded variant walks the work list with a grid stride: its length (`n_wg`, written by wg_scan_kernel)
// stays on the device, the host launches a grid that covers any list of the call's size class and the
// workgroups past the end leave at once - no read-back between the scan and this launch (42 us of an
// ML-20M call's 1.3 ms).  The waves of a workgroup are independent (no barrier): each walks on its own.
template <int KP, bool BOUNDED>
__global__ __launch_bounds__(256, 2) void score_emit_kernel(EmitParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char em_smem[];
  const int64_t n = (BOUNDED && p.n_wg) ? static_cast<int64_t>(*p.n_wg) : static_cast<int64_t>(gridDim.x);
  for (int64_t b = blockIdx.x; b < n; b += gridDim.x) score_emit_body<KP, BOUNDED>(p, em_smem, b);
}


// ---- bounded variant: set-up kernels -------------------------------------------------------
// A score is bounded by the product of the two factor norms (Cauchy-Schwarz), so an item whose
// norm is below r_u = tau_u / |user_u| cannot reach user u's threshold.  With the items sorted
// by norm (descending) and the users by r_u (ascending), a 64-user tile only needs the leading
// item tiles up to its smallest radius: on interaction data with a popularity skew that is a
// few percent of the score matrix, and the result is the same list (the bound is rigorous:
// norm_up >= |x| (1 + K eps-ish) covers the rounding of the fp32 dot product and of the norm).

// out[r] = upper bound of |F[begin + r, :]|: sqrt(sum x^2) * c + 3e-18 (the constant covers
// squares that underflow).  16 lanes per row.  Non-finite or > 1e18 -> bad_flag bit 0.
__global__ __launch_bounds__(256) void row_norm_up_kernel(const float *__restrict__ F, int64_t begin,
                                                          int64_t n, int32_t KP, float c,
                                                          float *__restrict__ out,
                                                          int32_t *__restrict__ bad_flag) {
  const int sub = threadIdx.x & 15;
  const int64_t r = static_cast<int64_t>(blockIdx.x) * 16 + (threadIdx.x >> 4);
  float acc = 0.f;
  if (r < n) {
    const float *row = F + (begin + r) * KP;
    for (int k = 4 * sub; k < KP; k += 64) {
      const fz_f32x4 v = *reinterpret_cast<const fz_f32x4 *>(row + k);
      acc += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
    }
  }
#pragma unroll
  for (int o = 8; o >= 1; o >>= 1) acc += __shfl_xor(acc, o, 64);
  if (r < n && sub == 0) {
    const float up = sqrtf(acc) * c + 3.0e-18f;
    out[r] = up;
    if (!(up < 1.0e18f)) atomicOr(bad_flag, 1);
  }
}

__global__ void iota_kernel(int32_t *out, int64_t n) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i < n) out[i] = static_cast<int32_t>(i);
}

__global__ void inverse_perm_kernel(const int32_t *__restrict__ perm, int64_t n,
                                    int32_t *__restrict__ inv) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i < n) inv[perm[i]] = static_cast<int32_t>(i);
}

// out[j, :] = F[perm[j], :] for j < n (the sample items, contiguous for user_scores_kernel)
// (rows at or beyond *n_valid, when given, are zero: perm holds nothing for them)
__global__ void gather_rows_kernel(const float *__restrict__ F, const int32_t *__restrict__ perm,
                                   int64_t n, int32_t KP, float *__restrict__ out,
                                   const int32_t *__restrict__ n_valid) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  const int q = KP / 4;
  if (i >= n * q) return;
  const int64_t j = i / q;
  const int k = static_cast<int>(i % q) * 4;
  fz_f32x4 v{0.f, 0.f, 0.f, 0.f};
  if (!n_valid || j < *n_valid)
    v = *reinterpret_cast<const fz_f32x4 *>(F + static_cast<int64_t>(perm[j]) * KP + k);
  *reinterpret_cast<fz_f32x4 *>(out + j * KP + k) = v;
}

// sample block [rows, n_sample] over the first n_sample SORTED items: -inf at the stored mask
// entries whose item sits in the sample (evaluator.py:426-432)
// (with a row list, score row j belongs to mask row row_list[j], j < *n_list)
__global__ void mask_rows_perm_kernel(float *scores, int64_t rows, int64_t n_sample,
                                      const int64_t *mask_ptr, const int32_t *mask_idx,
                                      const int32_t *__restrict__ inv,
                                      const int32_t *__restrict__ row_list,
                                      const int32_t *__restrict__ n_list) {
  const int64_t row = blockIdx.x;
  if (row >= rows || (row_list && row >= *n_list)) return;
  const int64_t mrow = row_list ? row_list[row] : row;
  for (int64_t q = mask_ptr[mrow] + threadIdx.x; q < mask_ptr[mrow + 1]; q += blockDim.x) {
    const int32_t j = inv[mask_idx[q]];
    if (j < n_sample) scores[row * n_sample + j] = -std::numeric_limits<float>::infinity();
  }
}

// mask_row[q] = row of mask entry q (one wave per row): lets the sample pass walk the mask by
// ENTRY - a flat, coalesced stream - instead of one small workgroup per row
__global__ __launch_bounds__(256) void mask_row_ids_kernel(const int64_t *__restrict__ mask_ptr,
                                                           int64_t rows, int32_t *__restrict__ mask_row) {
  const int64_t row = static_cast<int64_t>(blockIdx.x) * 4 + wave_index_in_block();
  if (row >= rows) return;
  for (int64_t q = mask_ptr[row] + (threadIdx.x & 63); q < mask_ptr[row + 1]; q += 64)
    mask_row[q] = static_cast<int32_t>(row);
}

// the mask walk of the sample pass by entry: entries [q0, q1) belong to the rows [row0, ...) of
// the score block [*, n_sample] over the first n_sample SORTED items
__global__ __launch_bounds__(256) void mask_entries_perm_kernel(float *__restrict__ scores, int64_t n_sample,
                                                                int64_t row0, int64_t q0, int64_t q1,
                                                                const int32_t *__restrict__ mask_row,
                                                                const int32_t *__restrict__ mask_idx,
                                                                const int32_t *__restrict__ inv) {
  const int64_t q = q0 + static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (q >= q1) return;
  const int32_t j = inv[mask_idx[q]];
  if (j < n_sample)
    scores[(static_cast<int64_t>(mask_row[q]) - row0) * n_sample + j] = -std::numeric_limits<float>::infinity();
}

// r_u = tau_u / norm_up(u), a hair low (the division's rounding); -inf (keep everything) for a
// threshold that is not positive; +inf (nothing to do; tau likewise) for a user without ground
// truth and for a hard row
__global__ void prune_radius_kernel(float *__restrict__ tau, const float *__restrict__ unorm,
                                    const int32_t *__restrict__ gt_ptr, int64_t offset, int64_t rows,
                                    const int32_t *__restrict__ hard, float *__restrict__ radius) {
  const int64_t r = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (r >= rows) return;
  const float INF = std::numeric_limits<float>::infinity();
  const float t = tau[r];
  float rad;
  if (gt_ptr[offset + r + 1] == gt_ptr[offset + r] || hard[r]) {
    rad = INF;
    tau[r] = INF;
  } else if (t > 0.f) {
    rad = (t / unorm[r]) * (1.0f - 4.0e-7f);
  } else {
    rad = -INF;
  }
  radius[r] = rad;
}

// wg_prefix[ut] = exclusive prefix sum of ceil(limit_tiles[ut] / 4) (workgroups of four item
// tiles), wg_prefix[n_ut] = their number, also written to *total.  One workgroup.
__global__ __launch_bounds__(1024) void wg_scan_kernel(const int32_t *__restrict__ limit_tiles,
                                                       int64_t n_ut, int32_t *__restrict__ wg_prefix,
                                                       int32_t *__restrict__ total) {
  __shared__ int32_t part[1024];
  const int tid = threadIdx.x;
  const int64_t per = (n_ut + 1023) / 1024;
  const int64_t b = per * tid, e = min(b + per, n_ut);
  int32_t sum = 0;
  for (int64_t i = b; i < e; i++) sum += (limit_tiles[i] + 3) / 4;
  part[tid] = sum;
  __syncthreads();
  for (int o = 1; o < 1024; o <<= 1) {  // Hillis-Steele inclusive scan of the 1024 partial sums
    const int32_t v = tid >= o ? part[tid - o] : 0;
    __syncthreads();
    part[tid] += v;
    __syncthreads();
  }
  int32_t run = part[tid] - sum;
  for (int64_t i = b; i < e; i++) {
    wg_prefix[i] = run;
    run += (limit_tiles[i] + 3) / 4;
  }
  if (tid == 1023) {
    wg_prefix[n_ut] = part[1023];
    *total = part[1023];
  }
}

// wg_desc[b] = (ut, first item tile, live tiles of the workgroup) for the workgroups b of user
// tile ut (one wave per user tile)
__global__ __launch_bounds__(256) void wg_fill_kernel(const int32_t *__restrict__ wg_prefix,
                                                      const int32_t *__restrict__ limit_tiles,
                                                      int64_t n_ut, int4 *__restrict__ wg_desc) {
  const int64_t ut = static_cast<int64_t>(blockIdx.x) * 4 + wave_index_in_block();
  if (ut >= n_ut) return;
  const int32_t b = wg_prefix[ut], e = wg_prefix[ut + 1], lim = limit_tiles[ut];
  for (int32_t i = b + (threadIdx.x & 63); i < e; i += 64) {
    const int32_t it0 = 4 * (i - b);
    wg_desc[i] = int4{static_cast<int32_t>(ut), it0, min(4, lim - it0), 0};
  }
}

// list[0 .. min(*count, cap)) = hard rows that have ground truth (any order); *count = all of them
__global__ void collect_hard_kernel(const int32_t *__restrict__ hard, const int32_t *__restrict__ gt_ptr,
                                    int64_t offset, int64_t rows, int32_t *__restrict__ list,
                                    int32_t *__restrict__ count, int32_t cap) {
  const int64_t r = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (r >= rows || !hard[r] || gt_ptr[offset + r + 1] == gt_ptr[offset + r]) return;
  const int pos = atomicAdd(count, 1);
  if (pos < cap) list[pos] = static_cast<int32_t>(r);
}

// limit_tiles[ut] = item tiles holding the sorted items whose norm_up >= the tile's smallest radius
__global__ void tile_limit_kernel(const float *__restrict__ radius_sorted, int64_t rows,
                                  const float *__restrict__ inorm_sorted, int64_t n_items,
                                  int32_t *__restrict__ limit_tiles,
                                  unsigned long long *__restrict__ tiles_scored) {
  const int64_t ut = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (ut * 64 >= rows) return;
  const float rmin = radius_sorted[ut * 64];  // ascending order: the first is the smallest
  int64_t lo = 0, hi = n_items;
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if (inorm_sorted[mid] >= rmin) lo = mid + 1; else hi = mid;
  }
  limit_tiles[ut] = static_cast<int32_t>((lo + 63) / 64);
  if (lo > 0) atomicAdd(tiles_scored, static_cast<unsigned long long>((lo + 63) / 64));
}

}  // namespace eval
}  // namespace irs
