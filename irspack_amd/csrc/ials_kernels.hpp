// iALS device kernels for gfx950 (CDNA4, wave64).  Hand-written HIP; no
// portability layer.  Replaces the per-row loops of
// /root/reference/cpp_source/als/IALSTrainer.hpp (Solver::prepare_p :78-115,
// step_cholesky :273-331, step_cg :170-271, user_scores :942-984,
// compute_loss :836-940).
//
// Data layout in HBM
//   factors   float32 [rows, KP] row-major, KP = K rounded up to 16/32/64 (or a
//             multiple of 64 above that); padded columns are kept at zero.
//   CSR       int32 indices / float32 data; rows are addressed through a task
//             list (row, begin, end, slot) sorted longest-first.
//   Gramian   "accumulator layout": the 16x16 tiles (I <= J) of the KP x KP
//             matrix exactly as v_mfma_f32_16x16x4_f32 leaves them in
//             registers, so a solve wave starts from P with 10 coalesced loads.
//
// Dimension permutation.  A wave gathers a factor row with ONE coalesced
// 16 B-per-lane load: lane (g = lane>>4, m = lane&15) holds dims T*m .. T*m+T-1
// (T = KP/16) of gathered row g.  MFMA tile I therefore consists of the dims
// {T*m + I : m = 0..15}; tile (I, J) register r of lane (g, m) is the Gramian
// element (T*(4g+r) + I, T*m + J).  The permutation is undone when the matrix
// is spilled to LDS for the solve.
#pragma once
#include <type_traits>
#include <hip/hip_runtime.h>

#include <cstdint>
#include <utility>

namespace irs {
namespace ials {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int SOLVE_WAVES = 1;  // waves per solve workgroup: waves never cooperate, so LDS per
                                // wave (not per block) sets residency

#ifndef SOLVE_MIN_WAVES_PER_SIMD
#define SOLVE_MIN_WAVES_PER_SIMD 2  // register budget hint (waves per SIMD)
#endif
#ifndef SOLVE_MIN_WAVES_PER_SIMD_K64
#define SOLVE_MIN_WAVES_PER_SIMD_K64 SOLVE_MIN_WAVES_PER_SIMD  // the same for the K <= 64 kernels (A/B builds)
#endif

struct Task {
  int32_t row;    // row of the solved side
  int32_t begin;  // [begin, end) into indices / data
  int32_t end;
  int32_t slot;   // < 0: whole row, solve inline; >= 0: partial Gramian slot
};

struct SplitRow {
  int32_t row;
  int32_t first_slot;
  int32_t n_slots;      // slots the second kernel adds: first_slot + i * slot_stride
  int32_t nnz;
  int32_t slot_stride;  // 1, or FOLD_GROUP once fold_partials_kernel has summed each group into its first slot
};

// A row cut into more than FOLD_MIN chunks has its partial Gramians summed in groups of
// FOLD_GROUP first (one thread per float4 of the partial, every group in parallel), so that the
// one wave (or workgroup) that finishes the row adds at most FOLD_MIN partials one after another.
constexpr int FOLD_GROUP = 16;
constexpr int FOLD_MIN = 32;
struct FoldGroup {
  int32_t first_slot;
  int32_t count;
};

__global__ __launch_bounds__(256) void fold_partials_kernel(float *__restrict__ partials, int partial_floats,
                                                            const FoldGroup *__restrict__ groups) {
  const FoldGroup g = groups[blockIdx.x];
  const int e = blockIdx.y * 256 + threadIdx.x;
  if (4 * e >= partial_floats) return;
  f32x4 *base = reinterpret_cast<f32x4 *>(partials + static_cast<size_t>(g.first_slot) * partial_floats) + e;
  const size_t step = static_cast<size_t>(partial_floats) / 4;
  f32x4 sum = base[0];
  int s = 1;
  for (; s + 4 <= g.count; s += 4) {  // ascending slot order, four loads in flight
    const f32x4 a = base[s * step], b = base[(s + 1) * step], c = base[(s + 2) * step], d = base[(s + 3) * step];
    sum += a;
    sum += b;
    sum += c;
    sum += d;
  }
  for (; s < g.count; s++) sum += base[s * step];
  base[0] = sum;
}

struct SolveParams {
  const Task *tasks;
  int32_t n_tasks;
  const SplitRow *split_rows;
  int32_t n_split;
  const int32_t *indices;
  const float *data;
  const float *other;   // gathered factors [n_other, KP]
  float *target;        // solved factors   [n_rows, KP]
  const float *reg;     // per-row regulariser, hpp:117-120 (host powf)
  const float *P_acc;   // alpha0 * F^T F in accumulator layout (upper tiles)
  const float *P_accL;  // the same in LOWER form: slot (a, b) = tile (row block b, column block a)
  float *partials;      // split-row scratch
  int32_t *err_flag;    // bit 0: pivot <= 0, bit 1: non-finite solution, bit 2: CG singular
  float bias;           // observation_bias, hpp:289-290
  int32_t K;            // unpadded latent dimension
  int32_t max_cg_steps; // already resolved (0 -> K), hpp:232-234
  int32_t warm_start;   // CG: start from the current row (hpp:199) or from 0 (hpp:132)
  int32_t zero_row;     // an all-zero row of `other` (UNIT kernels: entries past a row's end)
  const float *prior;   // feature prior [n_rows, KP] or null: rhs += reg_r * prior_r
                        // (step_cholesky_with_prior hpp:363, step_cg hpp:212-215)
  const float *px = nullptr;  // RESID kernels (iALS++ with one block): target @ P, [n_rows, KP]
};

// The wave's index inside its workgroup as a SCALAR: `threadIdx.x >> 6` alone is a per-lane value to
// the compiler, and everything derived from it - the wave's task, its row, begin / end, the loop
// bounds, the factor-row addresses - then lives in vector registers and is computed by vector
// instructions (the Task of ials_solve_kernel cost 4 registers and was spilled at 128 registers).
__device__ __forceinline__ int wave_in_block() {
  return __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6));
}

__device__ __forceinline__ float readlane_f(float x, int lane) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), lane));
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// Sum over the 16 lanes of a row (lanes sharing lane >> 4), in every lane of the row: four DPP adds
// (quad_perm [1,0,3,2], quad_perm [2,3,0,1], row_half_mirror, row_mirror - after the first two steps
// the quads are uniform, so the mirrors pair every lane with the other half's sum).
__device__ __forceinline__ float row16_sum(float x) {
  auto dpp = [](float v, auto ctrl) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), decltype(ctrl)::value,
                                                                  0xf, 0xf, false));
  };
  x += dpp(x, std::integral_constant<int, 0xB1>{});
  x += dpp(x, std::integral_constant<int, 0x4E>{});
  x += dpp(x, std::integral_constant<int, 0x141>{});
  x += dpp(x, std::integral_constant<int, 0x140>{});
  return x;
}

template <int T> __device__ __forceinline__ void load_dims(const float *p, float (&v)[T]) {
  if constexpr (T == 4) {
    f32x4 t = *reinterpret_cast<const f32x4 *>(p);
    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
  } else if constexpr (T == 2) {
    f32x2 t = *reinterpret_cast<const f32x2 *>(p);
    v[0] = t.x; v[1] = t.y;
  } else if constexpr (T == 1) {
    v[0] = *p;
  } else {
    static_assert(T % 4 == 0, "latent dims per lane must be 1, 2 or a multiple of 4");
#pragma unroll
    for (int q = 0; q < T / 4; q++) {
      f32x4 t = *reinterpret_cast<const f32x4 *>(p + 4 * q);
      v[4 * q] = t.x; v[4 * q + 1] = t.y; v[4 * q + 2] = t.z; v[4 * q + 3] = t.w;
    }
  }
}

// rhs += reg * prior row, in the layout of bsum (lane (g, m) holds dims T m .. T m + T - 1)
template <int T>
__device__ __forceinline__ void add_prior(const SolveParams &p, int row, float (&bsum)[T]) {
  if (p.prior == nullptr) return;
  const int m = threadIdx.x & 15;
  const float reg = p.reg[row];
  float pr[T];
  load_dims<T>(p.prior + static_cast<size_t>(row) * (16 * T) + T * m, pr);
#pragma unroll
  for (int i = 0; i < T; i++) bsum[i] = fmaf(reg, pr[i], bsum[i]);
}

// (I, J) of the t-th upper tile in row-major order of the T x T tile grid
template <int T> __host__ __device__ constexpr int tile_i(int t) {
  int i = 0;
  while (t >= T - i) { t -= T - i; i++; }
  return i;
}
template <int T> __host__ __device__ constexpr int tile_j(int t) {
  int i = 0;
  while (t >= T - i) { t -= T - i; i++; }
  return i + t;
}

template <int T> struct Geo {
  static constexpr int KP = 16 * T;
  static constexpr int NT = T * (T + 1) / 2;  // upper-triangular 16x16 tiles
  static constexpr int LD = KP + 4;           // LDS row stride (floats); keeps 16 B alignment,
                                              // conflict-free for row-per-lane ds_read_b128
  static constexpr int LDS_FLOATS = KP * LD + KP;
  static constexpr int CG_ROWS = T == 4 ? KP / 2 : KP;        // solve_row<T, 1>: rows staged per pass
  static constexpr int CG_LDS_FLOATS = CG_ROWS * LD + KP;
  static constexpr int PARTIAL_FLOATS = NT * 256 + (KP > 64 ? KP : 64);  // tiles + rhs
};

// One MFMA per owned tile; the tile coordinates are compile-time constants so that
// the operand arrays stay in registers.
template <int T, int NW, int W, int S>
__device__ __forceinline__ void mfma_tile(const float (&cv)[T], const float (&v)[T],
                                          f32x4 (&acc)[(Geo<T>::NT + NW - 1) / NW]) {
  constexpr int t = NW * S + W;
  if constexpr (t < Geo<T>::NT) {
    // forced constant evaluation: left to the optimiser, the search loops of tile_i / tile_j
    // survive for T = 16 and the operand arrays are indexed through scratch at run time
    constexpr int ti = tile_i<T>(t), tj = tile_j<T>(t);
    acc[S] = __builtin_amdgcn_mfma_f32_16x16x4f32(cv[ti], v[tj], acc[S], 0, 0, 0);
  }
}
template <int T, int NW, int W, int... S>
__device__ __forceinline__ void mfma_tiles(const float (&cv)[T], const float (&v)[T],
                                           f32x4 (&acc)[(Geo<T>::NT + NW - 1) / NW],
                                           std::integer_sequence<int, S...>) {
  (mfma_tile<T, NW, W, S>(cv, v, acc), ...);
}

// ---------------------------------------------------------------------------
// Gather + symmetric rank update on the matrix cores.
// A += sum_q c_q v_q v_q^T (upper tiles), b += sum_q (bias + c_q) v_q.
// hpp:300-308 (BatchedRankUpdater hpp:37-58 computes the same sum from
// sqrt(c) v rows; here c is applied to one operand).
// NW > 1: the tiles are dealt round-robin to NW waves of a workgroup that all walk
// the same row (tile t belongs to wave t % NW, local slot t / NW); D = gathered
// sub-steps kept in flight (a sub-step = 4 stored entries).
// UNIT: every stored confidence is exactly 1 (binary interactions).  Then c v = v feeds the
// MFMAs directly, b = (bias + 1) sum v, the value stream is never read, and an entry past
// the row's end is neutralised by gathering the all-zero row `zero_row` - 7 of the 26 vector
// instructions of a sub-step disappear (they cost the same issue cycles as the MFMAs' own,
// see DESIGN 3.1).
// LOWER: slot tix(i, j), i <= j, accumulates the tile (row block j, column block i) instead of
// its transpose (the single-wave block Cholesky of ials_chol16.hpp works on lower tiles).
// RESID (the one-block iALS++ sweep, hpp:436-502, on this kernel): the right-hand side becomes the
// NEGATIVE GRADIENT at the current row x (`xr`, in the lanes' dims), built the way the reference
// builds it - from the predictions: sum_q ((bias + c_q) - c_q (v_q . x)) v_q; the caller subtracts
// P x + reg x.  Solving A delta = that and adding delta to x is the reference's own arithmetic: a
// component of x that the row's entries do not see (fewer entries than K, alpha0 = 0) is removed by
// the ridge term alone, to rounding of ITSELF - the direct solve A^-1 b leaves kappa * 2^-24 there.
// Cost: T multiply-adds, four DPP adds and two more operations per sub-step.
template <int T, int NW, int W, int D, bool UNIT, bool LOWER, bool RESID>
__device__ __forceinline__ void syrk_gather_impl(const float *__restrict__ other,
                                                 const int32_t *__restrict__ indices,
                                                 const float *__restrict__ data, int begin,
                                                 int end, float bias,
                                                 f32x4 (&acc)[(Geo<T>::NT + NW - 1) / NW],
                                                 float (&bsum)[T], unsigned zero_row,
                                                 const float (&xr)[T]) {
  constexpr int KP = Geo<T>::KP;
  constexpr int NT = Geo<T>::NT;
  constexpr int TPW = (NT + NW - 1) / NW;
  const int lane = threadIdx.x & 63;
  const int g = lane >> 4, m = lane & 15;
  const float *col_base = other + T * m;
  const int n = end - begin;
  const int nsub = (n + 3) >> 2;

  // The row is walked in blocks of 64 stored entries: lane l holds (index, value)
  // of entry 64*blk + l, loaded two blocks ahead with ONE coalesced load each.
  // Sub-step j of a block takes entries 4j .. 4j+3 (one per 16-lane group); the
  // group's lanes fetch their entry with ds_bpermute, so the only vector-memory
  // traffic of the loop is the 1-KiB row gather, issued D sub-steps before its
  // MFMAs.  All loads are unconditional (the CSR arrays are padded): the loop body
  // is a single basic block and hipcc's vmcnt waits stay counted.  Entries past
  // the row's end are neutralised by c = w = 0.
  const int32_t *ip = indices + begin + lane;
  const float *dp = data + begin + lane;
  const int perm_base = 4 * g;  // ds_bpermute byte address of lane (4j + g) is 16 j + 4 g
  const uint32_t lane_off = static_cast<uint32_t>(T * m * sizeof(float));
  // (general confidences: an entry past the row's end carries c = -0.0 - its c v v^T adds zeros - and
  // the weight of its right-hand-side term, bias + c, is derived from c when the entry is consumed:
  // eight registers less than a second ring of weights, which is the difference between three and four
  // waves per SIMD.  Stored values are canonicalised on the host: -0.0 never comes from the data.)
  float v[D][T], vc[D];
  // (Round 5: the four adds of the UNIT right-hand side as two v_pk_add_f32 - 3 instead of 5 vector
  // instructions per sub-step beside the 10 MFMAs - changed nothing: 1.178 vs 1.175 ms for the user half.
  // The adds already issue in the shadow of the matrix instructions; the form was not kept.)
  // `all_valid` (a compile-time tag): the caller guarantees that the sub-step lies before the
  // row's last one, so no entry has to be neutralised (3 vector instructions less)
  auto fetch = [&](int k, int blk_idx, float blk_c, int j, int entry0, auto all_valid) {
    // entry0 = index (within the row) of the block's first entry
    const int src = perm_base + 16 * j;
    unsigned idx = static_cast<unsigned>(__builtin_amdgcn_ds_bpermute(src, blk_idx));
    const bool valid = decltype(all_valid)::value || entry0 + 4 * j + g < n;
    if constexpr (UNIT) {
      idx = valid ? idx : zero_row;
      // 32-bit byte offset from the (wave-uniform) table base: one v_lshl_add instead of
      // two 64-bit address operations; the host takes this path only for tables < 4 GB
      const uint32_t off = idx * static_cast<uint32_t>(KP * sizeof(float)) + lane_off;
      load_dims<T>(reinterpret_cast<const float *>(reinterpret_cast<const char *>(other) + off), v[k]);
      return;
    } else {
      const float c = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src, __builtin_bit_cast(int, blk_c)));
      vc[k] = valid ? c : -0.f;
    }
    load_dims<T>(col_base + static_cast<size_t>(idx) * KP, v[k]);
  };
  // The same fetch in two stages for the loop body.  ds_bpermute_b32 adds an immediate offset to its
  // address register, but hipcc materialises `perm_base + 16 j` in a register of its own for each of
  // the 16 sub-steps and keeps all of them live through the loop: 15 registers that decide between
  // three and four waves per SIMD.  Issued from inline assembly with `offset:16 j` the permute needs
  // the one base register; its s_waitcnt is ours to place (the compiler does not see the LDS
  // operation): after the sub-step's matrix instructions, where the compiler put its own.
  auto perm_issue = [&](unsigned &pidx, float &pc, int blk_idx, float blk_c, int j) {
    asm volatile("ds_bpermute_b32 %0, %1, %2 offset:%3" : "=v"(pidx) : "v"(perm_base), "v"(blk_idx), "i"(16 * j));
    if constexpr (!UNIT)
      asm volatile("ds_bpermute_b32 %0, %1, %2 offset:%3" : "=v"(pc) : "v"(perm_base), "v"(blk_c), "i"(16 * j));
  };
  auto fetch_finish = [&](int k, unsigned pidx, float pc, int j, int entry0, auto all_valid) {
    if constexpr (UNIT)
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(pidx));
    else
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(pidx), "+v"(pc));
    unsigned idx = pidx;
    const bool valid = decltype(all_valid)::value || entry0 + 4 * j + g < n;
    if constexpr (UNIT) {
      idx = valid ? idx : zero_row;
      const uint32_t off = idx * static_cast<uint32_t>(KP * sizeof(float)) + lane_off;
      load_dims<T>(reinterpret_cast<const float *>(reinterpret_cast<const char *>(other) + off), v[k]);
      return;
    } else {
      vc[k] = valid ? pc : -0.f;
    }
    load_dims<T>(col_base + static_cast<size_t>(idx) * KP, v[k]);
  };
  // `all_valid`: every entry in the ring was fetched from inside the row (no -0.0 marker to test)
  auto consume = [&](int k, auto all_valid) {
    float cv[T];
    float w = 0.f;
    if constexpr (RESID) {
      float d = 0.f;  // the prediction v_q . x (hpp:455-457)
#pragma unroll
      for (int i = 0; i < T; i++) d = fmaf(v[k][i], xr[i], d);
      d = row16_sum(d);
      if constexpr (UNIT) {
        w = (bias + 1.0f) - d;  // (an entry past the row's end gathered the zero row: its term vanishes)
      } else {
        w = fmaf(-vc[k], d, bias + vc[k]);
        if constexpr (!decltype(all_valid)::value)
          w = __builtin_bit_cast(unsigned, vc[k]) == 0x80000000u ? 0.f : w;
      }
    } else if constexpr (!UNIT) {
      w = bias + vc[k];
      if constexpr (!decltype(all_valid)::value)
        w = __builtin_bit_cast(unsigned, vc[k]) == 0x80000000u ? 0.f : w;
    }
#pragma unroll
    for (int i = 0; i < T; i++) {
      if constexpr (UNIT) {
        cv[i] = v[k][i];
        if constexpr (RESID) bsum[i] = fmaf(w, v[k][i], bsum[i]);
        else bsum[i] += v[k][i];
      } else {
        cv[i] = vc[k] * v[k][i];
        bsum[i] = fmaf(w, v[k][i], bsum[i]);
      }
    }
    if constexpr (NW == 1) {
      int t = 0;
#pragma unroll
      for (int i = 0; i < T; i++)
#pragma unroll
        for (int j = i; j < T; j++) {
          if constexpr (LOWER)
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(cv[j], v[k][i], acc[t], 0, 0, 0);
          else
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(cv[i], v[k][j], acc[t], 0, 0, 0);
          t++;
        }
    } else {
      float vk[T];
#pragma unroll
      for (int i = 0; i < T; i++) vk[i] = v[k][i];
      mfma_tiles<T, NW, W>(cv, vk, acc, std::make_integer_sequence<int, TPW>{});
    }
  };
  // Two block-register sets A / B used alternately (no loop-carried copies of a
  // loaded value: a copy would make hipcc drain the gather queue with vmcnt(0)).
  // While a block is consumed from one set, the other already holds the next
  // block; the set that was just exhausted is reloaded with the block after.
  int a_i = ip[0], b_i = ip[64];
  float a_c = 0.f, b_c = 0.f;
  if constexpr (!UNIT) {
    a_c = dp[0];
    b_c = dp[64];
  }
#pragma unroll
  for (int k = 0; k < D; k++) fetch(k, a_i, a_c, k, 0, std::false_type{});
  __builtin_amdgcn_sched_barrier(0);
  int s0 = 0;  // first sub-step of the current block
  static_assert(16 % D == 0 && D < 16, "the gather ring must divide the 16 sub-steps of a block");
  static_assert(!LOWER || NW == 1, "lower-form tiles: single-wave kernels only");
  auto block = [&](int &cur_i, float &cur_c, const int nxt_i, const float nxt_c, auto all_valid) {
#pragma unroll
    for (int j = 0; j < 16; j++) {  // sub-step j of this block; its gather was issued D earlier
      unsigned pidx;
      float pc = 0.f;
      if (j + D < 16)
        perm_issue(pidx, pc, cur_i, cur_c, j + D);
      else
        perm_issue(pidx, pc, nxt_i, nxt_c, j + D - 16);
      consume(j % D, all_valid);
      __builtin_amdgcn_sched_barrier(0);  // (the wait below stays behind the matrix instructions)
      if (j + D < 16)
        fetch_finish(j % D, pidx, pc, j + D, 4 * s0, all_valid);
      else
        fetch_finish(j % D, pidx, pc, j + D - 16, 4 * s0 + 64, all_valid);
      if (j == 15 - D) {  // last use of this set: reload it with block + 2
        cur_i = ip[4 * s0 + 128];
        if constexpr (!UNIT) cur_c = dp[4 * s0 + 128];
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    s0 += 16;
  };
  constexpr std::true_type all_valid_tag{};
  constexpr std::false_type masked_tag{};
  // a block fetches the sub-steps up to s0 + 15 + D; while even the second block of a round
  // stays before the row's last sub-step nothing has to be masked (long rows: most blocks)
  while (s0 + 32 + D + 1 <= nsub) {  // `block` advances s0
    block(a_i, a_c, b_i, b_c, all_valid_tag);
    block(b_i, b_c, a_i, a_c, all_valid_tag);
  }
  while (s0 + 32 <= nsub) {
    block(a_i, a_c, b_i, b_c, masked_tag);
    block(b_i, b_c, a_i, a_c, masked_tag);
  }
  if (s0 + 16 <= nsub) {
    block(a_i, a_c, b_i, b_c, masked_tag);
    a_i = b_i;  // the tail reads the current block from set A
    a_c = b_c;
  }
  const int cur_i = a_i;
  const float cur_c = a_c;
  // tail: up to 15 sub-steps of the last block; the first D are already in flight
  const int rest = nsub - s0;
  // (rounds of D: consume what is in flight, then fetch the next D together — loads
  // under per-sub-step branches would each cost a full vmcnt(0) drain)
#pragma unroll
  for (int base = 0; base < 15; base += D) {
#pragma unroll
    for (int k = 0; k < D; k++)
      if (base + k < rest) consume(k, std::false_type{});
    if (base + D < 15 && rest > base + D) {
#pragma unroll
      for (int k = 0; k < D; k++)
        if (base + D + k < rest) fetch(k, cur_i, cur_c, base + D + k, 4 * s0, std::false_type{});
    }
  }
  // fold the four gathered-row groups: every lane ends with b[T*m + i]
#pragma unroll
  for (int i = 0; i < T; i++) {
    bsum[i] += __shfl_xor(bsum[i], 16, 64);
    bsum[i] += __shfl_xor(bsum[i], 32, 64);
    if constexpr (UNIT && !RESID) bsum[i] *= bias + 1.0f;
  }
}

template <int T, int NW = 1, int W = 0, int D = 8, bool UNIT = false, bool LOWER = false>
__device__ __forceinline__ void syrk_gather(const float *__restrict__ other,
                                            const int32_t *__restrict__ indices,
                                            const float *__restrict__ data, int begin,
                                            int end, float bias,
                                            f32x4 (&acc)[(Geo<T>::NT + NW - 1) / NW],
                                            float (&bsum)[T], unsigned zero_row = 0) {
  const float none[T] = {};
  syrk_gather_impl<T, NW, W, D, UNIT, LOWER, false>(other, indices, data, begin, end, bias, acc, bsum, zero_row,
                                                    none);
}

// ---------------------------------------------------------------------------
// The same rank update on the bf16 matrix cores with fp32-equivalent accuracy ("bf16x3"):
// unit confidences only (binary interactions), lower-form tiles.  The DEFAULT rank update of binary
// interactions at K <= 64 under the Cholesky solver since round 6 (IRSPACK_AMD_IALS_BF16X3=0 keeps the
// fp32-input matrix instruction): on all 165,237 rows of the ML-20M benchmark half-steps its factors are
// CLOSER to the float64 solve than the float32 oracle's (worst row 1.1e-5 / 1.9e-6 against 3.8e-5 / 5.7e-5,
// profiles/parity_r06.json) - the six partial products are exact in fp32, only their accumulation rounds.
//
// v_mfma_f32_16x16x32_bf16 contracts 32 stored entries in 16 cycles where
// v_mfma_f32_16x16x4_f32 contracts 4 in 32 (MI355X_MICROARCH.md: the fp32-input MFMA runs at
// the vector rate, 1/16 of bf16) - and, unlike the fp32 one, leaves half of its cycles to the
// vector ALU.  Every gathered fp32 value is split EXACTLY into three bf16 terms
// x = hi + mid + lo (8 + 8 + 8 mantissa bits; each residual is exactly representable), and a
// product x y is accumulated in fp32 from the six partial products
// hi hi, hi mid, mid hi, mid mid, hi lo, lo hi; each is exact in fp32, the three dropped ones
// are below 2^-24 |x y|.  60 MFMAs (960 cycles) + ~180 conversion instructions per 32 entries
// against 80 MFMAs (2,560 cycles) + ~150 instructions for the fp32 path.
//
// Layout: a group = 32 entries; lane (g, m) gathers the entries 8 g .. 8 g + 7 of the group,
// dims T m .. T m + 3 of each (one 16 B load per entry, as in syrk_gather).  The operand of
// dims {T m + I} is then the lane's 8 values of dim I, packed two per register - the same
// registers serve as A (row block) and B (column block), so the instruction's own k-to-lane
// map does not matter.
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned pack_bf16(float a, float b) {  // v_cvt_pk_bf16_f32 (RNE)
  return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{a, b}, bf16x2_t));
}

// RESID (the one-block iALS++ sweep, see syrk_gather_impl): the right-hand side is the negative gradient at
// the current row `xr`, built from the predictions d_q = v_q . x - from the gathered fp32 values, before
// they are split: sum_q ((bias + 1) - d_q) v_q; the 16 lanes of a group share the group's eight entries, so
// a prediction is T multiply-adds and one 16-lane DPP sum.
template <int T, bool RESID = false>
__device__ __forceinline__ void syrk_gather_bf16x3(const float *__restrict__ other,
                                                   const int32_t *__restrict__ indices, int begin,
                                                   int end, float bias, f32x4 (&acc)[Geo<T>::NT],
                                                   float (&bsum)[T], unsigned zero_row, const float (&xr)[T]) {
  static_assert(T == 4 || T == 8, "dims per lane = one or two 16 B loads");
  constexpr int KP = Geo<T>::KP;
  constexpr int V4 = T / 4;  // 16-byte loads per lane and entry
  const int lane = threadIdx.x & 63;
  const int g = lane >> 4, m = lane & 15;
  const int n = end - begin;
  const int ngroups = (n + 31) >> 5;
  const uint32_t lane_off = static_cast<uint32_t>(T * m * sizeof(float));
  const int32_t *ip = indices + begin + lane;
  f32x4 raw[2][8][V4];  // two groups in flight
  // entries 32 gi + 8 g + kk of the 64-entry index block held one per lane in blk_idx
  auto issue = [&](int slot, int blk_idx, int gi, int entry0) {
#pragma unroll
    for (int kk = 0; kk < 8; kk++) {
      const int src = 4 * (32 * gi + 8 * g + kk);
      // (entries past the row's end were replaced by the zero row when their block was LOADED: one
      // compare + select per lane and 64-entry block instead of one per gathered entry - 15 vector
      // instructions less per 32-entry group)
      const unsigned idx = static_cast<unsigned>(__builtin_amdgcn_ds_bpermute(src, blk_idx));
      if constexpr (T == 8) {
        // (64-bit row addresses: configs[3] gathers a 5.1 GB user table at K = 128 - one more vector
        // instruction per entry beside ~70 of splitting)
        const float *rp = other + (static_cast<size_t>(idx) * KP + T * m);
#pragma unroll
        for (int v = 0; v < V4; v++) raw[slot][kk][v] = *reinterpret_cast<const f32x4 *>(rp + 4 * v);
      } else {
        const uint32_t off = idx * static_cast<uint32_t>(KP * sizeof(float)) + lane_off;
#pragma unroll
        for (int v = 0; v < V4; v++)
          raw[slot][kk][v] = *reinterpret_cast<const f32x4 *>(reinterpret_cast<const char *>(other) + off + 16 * v);
      }
    }
  };
  auto consume = [&](int slot) {
    u32x4_t hi[T], mid[T], lo[T];
    float w8[RESID ? 8 : 1];
    if constexpr (RESID) {
#pragma unroll
      for (int kk = 0; kk < 8; kk++) {
        float d = 0.f;  // the prediction v_q . x (hpp:455-457); a zero row (past the row's end) predicts 0 and adds 0
#pragma unroll
        for (int dd = 0; dd < T; dd++) d = fmaf(raw[slot][kk][dd >> 2][dd & 3], xr[dd], d);
        w8[kk] = (bias + 1.0f) - row16_sum(d);
      }
    }
#pragma unroll
    for (int d = 0; d < T; d++) {
#pragma unroll
      for (int pp = 0; pp < 4; pp++) {
        const float x0 = raw[slot][2 * pp][d >> 2][d & 3], x1 = raw[slot][2 * pp + 1][d >> 2][d & 3];
        if constexpr (RESID) bsum[d] = fmaf(w8[2 * pp + 1], x1, fmaf(w8[2 * pp], x0, bsum[d]));
        else bsum[d] += x0 + x1;
        const unsigned h = pack_bf16(x0, x1);
        const float r0 = x0 - __uint_as_float(h << 16), r1 = x1 - __uint_as_float(h & 0xffff0000u);
        const unsigned mi = pack_bf16(r0, r1);
        const float s0 = r0 - __uint_as_float(mi << 16), s1 = r1 - __uint_as_float(mi & 0xffff0000u);
        hi[d][pp] = h;
        mid[d][pp] = mi;
        lo[d][pp] = pack_bf16(s0, s1);
      }
    }
    int t = 0;
#pragma unroll
    for (int i = 0; i < T; i++)
#pragma unroll
      for (int j = i; j < T; j++) {  // tile (row block j, column block i)
        const bf16x8_t ah = __builtin_bit_cast(bf16x8_t, hi[j]), am = __builtin_bit_cast(bf16x8_t, mid[j]),
                       al = __builtin_bit_cast(bf16x8_t, lo[j]);
        const bf16x8_t bh = __builtin_bit_cast(bf16x8_t, hi[i]), bm = __builtin_bit_cast(bf16x8_t, mid[i]),
                       bl = __builtin_bit_cast(bf16x8_t, lo[i]);
        // smallest terms first
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bm, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bh, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bm, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, acc[t], 0, 0, 0);
        t++;
      }
  };
  // index blocks of 64 entries, one per lane, two blocks ahead (padded arrays: every load is
  // unconditional, the loop is one basic block)
  const int zr = static_cast<int>(zero_row);
  int cur = ip[0], nxt = ip[64];
  cur = lane < n ? cur : zr;
  nxt = 64 + lane < n ? nxt : zr;
  issue(0, cur, 0, 0);
  issue(1, cur, 1, 0);
  int g0 = 0;
  for (; g0 + 2 <= ngroups; g0 += 2) {
    const int e_next = 32 * (g0 + 2);       // first entry of the next block
    int nn = ip[e_next + 64];               // the block after it
    nn = e_next + 64 + lane < n ? nn : zr;
    consume(0);
    issue(0, nxt, 0, e_next);
    // (an early exit HERE for odd group counts measured slower in round 4: user half 1.19 -> 1.32 ms - a
    // branch between the loads and their use drains the gather queue)
    consume(1);
    issue(1, nxt, 1, e_next);
    cur = nxt;
    nxt = nn;
  }
  // Round 6: an odd group count ends with ONE consume behind the loop (no load follows it, so the branch
  // costs no drain) instead of a second group of all-zero rows - half a group (60 matrix + ~90 vector
  // instructions) less per row on average, a whole one for the rows of at most 32 entries.  The loads of
  // slot 1 that are in flight (zero rows) are simply never used.
  if (g0 < ngroups) consume(0);
  (void)cur;
#pragma unroll
  for (int i = 0; i < T; i++) {
    bsum[i] += __shfl_xor(bsum[i], 16, 64);
    bsum[i] += __shfl_xor(bsum[i], 32, 64);
    if constexpr (!RESID) bsum[i] *= bias + 1.0f;
  }
}

// ---------------------------------------------------------------------------
// Per-row solve by one wave.  The KP x KP system is spilled to this wave's LDS
// slab in natural coordinates, then lane i owns row i in registers.
//   SOLVER 0: Cholesky A = L L^T with forward / backward substitution
//             (Eigen::LLT + solve, hpp:316-324)
//   SOLVER 1: conjugate gradient on the explicit matrix (same iterates as the
//             matrix-free loop of hpp:199-264: A x = P x + reg x + sum c (v.x) v)
// LOWER_ACC: slot tix(i, j), i <= j, holds the tile (row block j, column block i) - what the bf16x3 rank
// update accumulates - instead of its transpose; the matrix is symmetric, only the staging indices swap.
template <int T, int SOLVER, bool LOWER_ACC = false>
__device__ __forceinline__ void solve_row(f32x4 (&acc)[Geo<T>::NT], const float (&b4)[T],
                                          float reg, float *sm, float *xrow, int K,
                                          int nnz, int max_cg_steps, int warm_start,
                                          int32_t *err_flag) {
  constexpr int KP = Geo<T>::KP;
  constexpr int LD = Geo<T>::LD;
  const int lane = threadIdx.x & 63;
  const int g = lane >> 4, m = lane & 15;

  if constexpr (SOLVER == 1) {
    if (nnz == 0) {  // hpp:207-210
      if (lane < KP) xrow[lane] = 0.f;
      return;
    }
  }
  // The regulariser stays OUT of the matrix: the product is (P + sum c v v^T) vec summed at its
  // own magnitude, + reg vec in one last fma (the reference's order, hpp:222-223 / 241-242: P x,
  // then reg x).  With reg on the diagonal every partial sum behind the diagonal term is
  // rounded at the magnitude of reg vec - visible when reg_r dominates (10^6 items: reg_r = 100)
  // and the solution is orders of magnitude smaller than the warm start.  Padded dims never
  // enter (act is false for them).
  // K = 64: the matrix crosses from the accumulator layout to one row per lane in TWO passes of 32
  // rows (round 4): 9 KB of LDS per wave instead of 17.7, so that the registers (three waves per SIMD)
  // and not the LDS (nine waves per CU) set the occupancy.  Rows < 32 are the tile rows of lane groups
  // 0 and 1 (row = 4 (4 g + r) + i), their mirrored elements come from the lanes with m < 8.
  constexpr int ROWS = Geo<T>::CG_ROWS;
  constexpr int PASSES = KP / ROWS;
  float *sb = sm + ROWS * LD;
  if (g == 0) {
#pragma unroll
    for (int i = 0; i < T; i++) sb[T * m + i] = b4[i];
  }
  const int li = lane < KP ? lane : KP - 1;
  float a[KP];
#pragma unroll
  for (int pass = 0; pass < PASSES; pass++) {
    if (pass > 0) __threadfence_block();  // (the first half's rows are in registers)
    const bool rows_here = PASSES == 1 || (g >> 1) == pass;   // this lane's tile rows are staged now
    const bool cols_here = PASSES == 1 || (m >> 3) == pass;   // ... its tile columns (mirrored part)
    int t = 0;
#pragma unroll
    for (int i = 0; i < T; i++)
#pragma unroll
      for (int j = i; j < T; j++) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const int row = T * (4 * g + r) + (LOWER_ACC ? j : i), col = T * m + (LOWER_ACC ? i : j);
          if (rows_here) sm[(row - pass * ROWS) * LD + col] = acc[t][r];
          if (i != j && cols_here) sm[(col - pass * ROWS) * LD + row] = acc[t][r];
        }
        t++;
      }
    __threadfence_block();
    if (PASSES == 1 || (li / ROWS) == pass) {
#pragma unroll
      for (int q = 0; q < KP / 4; q++) {
        f32x4 t4 = *reinterpret_cast<const f32x4 *>(sm + (li - pass * ROWS) * LD + 4 * q);
        a[4 * q] = t4.x; a[4 * q + 1] = t4.y; a[4 * q + 2] = t4.z; a[4 * q + 3] = t4.w;
      }
    }
  }
  float bv = sb[li];
  __threadfence_block();  // (the staging area becomes the vector buffer of the products)

  static_assert(SOLVER == 1, "Cholesky is solve_row_cholesky (accumulator layout)");
  {
    // conjugate gradient, hpp:199-264
    const bool act = lane < K;
    float x = (warm_start && act) ? xrow[li] : 0.f;
    // the vector is parked in LDS (the spilled matrix is in registers by now) and read back
    // four components per broadcast read: LDS instructions do not compete with the vector /
    // matrix issue budget, KP v_readlane would
    auto matvec = [&](float vec) {
      sm[lane] = vec;
      float s = 0.f;
#pragma unroll
      for (int q = 0; q < KP / 4; q++) {
        const f32x4 v4 = *reinterpret_cast<const f32x4 *>(sm + 4 * q);
        s = fmaf(a[4 * q], v4.x, s);
        s = fmaf(a[4 * q + 1], v4.y, s);
        s = fmaf(a[4 * q + 2], v4.z, s);
        s = fmaf(a[4 * q + 3], v4.w, s);
      }
      return fmaf(reg, vec, s);
    };
    float r = act ? bv - matvec(x) : 0.f;
    float p = r;
    bool singular = false;
    for (int it = 0; it < max_cg_steps; it++) {
      const float r2 = wave_sum(r * r);
      if (r2 <= 1e-20f) break;  // hpp:238
      const float Ap = act ? matvec(p) : 0.f;
      const float denom = wave_sum(p * Ap);
      if (!(denom > 0.f) || !__builtin_isfinite(denom)) {  // hpp:250-254
        singular = true;
        break;
      }
      const float alpha = r2 / denom;
      x = fmaf(alpha, p, x);
      r = fmaf(-alpha, Ap, r);
      const float r2n = wave_sum(r * r);
      if (r2n <= 1e-20f) break;  // hpp:258
      const float beta = r2n / r2;  // hpp:261
      p = fmaf(beta, p, r);
    }
    if (singular) {
      if (lane == 0) atomicOr(err_flag, 4);
    }
    if (lane < KP) xrow[lane] = act ? x : 0.f;
  }
}

// ---------------------------------------------------------------------------
// Cholesky solve on the accumulator layout (Eigen::LLT<Upper> + solve,
// hpp:316-324), in the permuted basis the rank update left the matrix in.
//
// M = R^T R is factorised by rows, 4 rows (one register quad of one 16-lane
// group) per block step:
//   panel   the 4 rows are scaled / eliminated against each other with
//           readlane'd scalars (VALU, lanes of the owning group only)
//   trail   every row below the panel gets  M[i][j] -= sum_k R[k][i] R[k][j],
//           a rank-4 update = ONE v_mfma_f32_16x16x4_f32 per 16x16 tile; the
//           operands are the panel rows re-read from LDS in operand layout
//           (lane (g', m) <- R[4q+g'][16J+m]); rows already finished are
//           protected by zeroing their A operand
// The right-hand side rides along as a 65th column (forward substitution for
// free).  R and y are then spilled once (packed upper tiles, 12.8 KB) and lane k
// back-substitutes row k.  ~1.5 k VALU + ~120 MFMA per 64 x 64 system instead of
// the ~6 k VALU of a register-resident column-by-column factorisation.
template <int T> struct CholGeo {
  static constexpr int KP = 16 * T;
  static constexpr int NT = T * (T + 1) / 2;
  static constexpr int PR = ((16 * T + 16) % 32 == 0) ? 16 * T + 32 : 16 * T + 16;  // panel row stride
#ifndef IRS_CHOL_ROW_STRIDE
#define IRS_CHOL_ROW_STRIDE 20
#endif
  static constexpr int RS = IRS_CHOL_ROW_STRIDE;  // spilled tile row stride (20: conflict-free b128 rows)
  static constexpr int TS = 16 * RS;  // spilled tile: 16 rows
  static constexpr int PAN_FLOATS = 4 * PR;
  static constexpr int SPILL_FLOATS = NT * TS + KP;
  static constexpr int LDS_FLOATS = PAN_FLOATS > SPILL_FLOATS ? PAN_FLOATS : SPILL_FLOATS;
  // solve_row_cg128 (T = 8): unpadded tile rows.  With the 20-float stride the spill is 46.6 KB per
  // wave and only THREE one-wave workgroups fit a CU's 160 KB (one SIMD idle); at 16 floats it is
  // 37.4 KB and four fit.  The price is a 4-way bank conflict on the b128 row reads that load the
  // matrix into registers once per row (IRS_CG128_ROW_STRIDE=20 restores the padded layout).
#ifndef IRS_CG128_ROW_STRIDE
#define IRS_CG128_ROW_STRIDE 16
#endif
  static constexpr int RS_CG = IRS_CG128_ROW_STRIDE;
  static constexpr int TS_CG = 16 * RS_CG;
  static constexpr int SPILL_CG_FLOATS = NT * TS_CG + KP;
  static constexpr int tix(int i, int j) { return i * T - i * (i - 1) / 2 + (j - i); }
};

template <int T>
__device__ __forceinline__ void solve_row_cholesky(f32x4 (&acc)[Geo<T>::NT], const float (&b4)[T],
                                                   float reg, float *sm, float *xrow, int K,
                                                   int32_t *err_flag) {
  using C = CholGeo<T>;
  constexpr int KP = C::KP;
  constexpr int PR = C::PR;
  const int lane = threadIdx.x & 63;
  const int g = lane >> 4, m = lane & 15;

  // diagonal: + reg (hpp:312-314); padded dims get 1 so that they decouple
#pragma unroll
  for (int i = 0; i < T; i++)
#pragma unroll
    for (int r = 0; r < 4; r++)
      if (4 * g + r == m) acc[C::tix(i, i)][r] += (T * m + i < K) ? reg : 1.0f;

  // right-hand side as an extra column in accumulator layout (replicated over m)
  f32x4 bacc[T];
#pragma unroll
  for (int i = 0; i < T; i++)
#pragma unroll
    for (int r = 0; r < 4; r++) bacc[i][r] = __shfl(b4[i], 20 * g + r, 64);

  bool bad = false;
#pragma unroll
  for (int q = 0; q < 4 * T; q++) {
    const int I = q / 4, gq = q % 4;
    const bool mine = g == gq;
    // ---- panel: rows 4q .. 4q+3 (lanes of group gq, registers 0..3)
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const float piv = readlane_f(acc[C::tix(I, I)][r], 20 * gq + r);
      bad |= !(piv > 0.f);
      const float rinv = __builtin_amdgcn_rsqf(piv);
      const float mult = mine ? rinv : 1.0f;
#pragma unroll
      for (int j = I; j < T; j++) acc[C::tix(I, j)][r] *= mult;
      bacc[I][r] *= mult;
#pragma unroll
      for (int r2 = r + 1; r2 < 4; r2++) {
        const float s = readlane_f(acc[C::tix(I, I)][r], 20 * gq + r2);  // R[k][k2]
        const float sm_ = mine ? s : 0.f;
#pragma unroll
        for (int j = I; j < T; j++)
          acc[C::tix(I, j)][r2] = fmaf(-sm_, acc[C::tix(I, j)][r], acc[C::tix(I, j)][r2]);
        bacc[I][r2] = fmaf(-sm_, bacc[I][r], bacc[I][r2]);
      }
    }
    if (q == 4 * T - 1) break;
    // ---- trail: panel rows -> LDS -> MFMA operands
    if (mine) {
#pragma unroll
      for (int r = 0; r < 4; r++) {
#pragma unroll
        for (int j = I; j < T; j++) sm[r * PR + 16 * j + m] = acc[C::tix(I, j)][r];
        if (m == 0) sm[r * PR + 16 * T] = bacc[I][r];
      }
    }
    __threadfence_block();
    float op[T];
#pragma unroll
    for (int j = I; j < T; j++) op[j] = sm[g * PR + 16 * j + m];
    const float opb = sm[g * PR + 16 * T];
    __threadfence_block();
#pragma unroll
    for (int i2 = I; i2 < T; i2++) {
      float a = op[i2];
      if (i2 == I) a = (m > 4 * gq + 3) ? a : 0.f;  // rows up to the panel are final
      a = -a;
#pragma unroll
      for (int j2 = i2; j2 < T; j2++)
        acc[C::tix(i2, j2)] =
            __builtin_amdgcn_mfma_f32_16x16x4f32(a, op[j2], acc[C::tix(i2, j2)], 0, 0, 0);
      bacc[i2] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, opb, bacc[i2], 0, 0, 0);
    }
  }
  if (__any(bad)) {  // (each group saw the pivots of its own panels)
    if (lane == 0) atomicOr(err_flag, 1);
  }
  // ---- spill R (upper tiles) and y, then lane k back-substitutes row k of R x = y
  float *ybuf = sm + C::NT * C::TS;
#pragma unroll
  for (int i = 0; i < T; i++) {
#pragma unroll
    for (int j = i; j < T; j++)
#pragma unroll
      for (int r = 0; r < 4; r++) sm[C::tix(i, j) * C::TS + (4 * g + r) * C::RS + m] = acc[C::tix(i, j)][r];
    if (m == 0) {
#pragma unroll
      for (int r = 0; r < 4; r++) ybuf[16 * i + 4 * g + r] = bacc[i][r];
    }
  }
  __threadfence_block();
  if constexpr (KP > 64) {
    // KP = 128: lane l owns the virtual rows l and 64 + l.  Row 64 + l only needs the last
    // 64 columns.  As below, whatever a row accumulates after its own step is never read.
    static_assert(KP == 128, "two rows per lane");
    const int rk = lane & 15, I0 = lane >> 4, I1 = 4 + (lane >> 4);
    auto tile_of = [&](int I, int J) { return I * T - I * (I - 1) / 2 + (J - I); };
    float row0[128], row1[64];
#pragma unroll
    for (int j = 0; j < T; j++) {
      const float *src = sm + tile_of(I0, j >= I0 ? j : I0) * C::TS + rk * C::RS;
#pragma unroll
      for (int c = 0; c < 4; c++) {
        const f32x4 v = *reinterpret_cast<const f32x4 *>(src + 4 * c);
        row0[16 * j + 4 * c] = v.x; row0[16 * j + 4 * c + 1] = v.y;
        row0[16 * j + 4 * c + 2] = v.z; row0[16 * j + 4 * c + 3] = v.w;
      }
    }
#pragma unroll
    for (int j = 4; j < T; j++) {
      const float *src = sm + tile_of(I1, j >= I1 ? j : I1) * C::TS + rk * C::RS;
#pragma unroll
      for (int c = 0; c < 4; c++) {
        const f32x4 v = *reinterpret_cast<const f32x4 *>(src + 4 * c);
        row1[16 * (j - 4) + 4 * c] = v.x; row1[16 * (j - 4) + 4 * c + 1] = v.y;
        row1[16 * (j - 4) + 4 * c + 2] = v.z; row1[16 * (j - 4) + 4 * c + 3] = v.w;
      }
    }
    const float rinv0 = 1.0f / sm[tile_of(I0, I0) * C::TS + rk * C::RS + rk];
    const float rinv1 = 1.0f / sm[tile_of(I1, I1) * C::TS + rk * C::RS + rk];
    const float y0 = ybuf[lane], y1 = ybuf[64 + lane];
    float p0 = 0.f, p1 = 0.f, x0 = 0.f, x1 = 0.f;
#pragma unroll
    for (int j = 127; j >= 64; j--) {
      const float xj = readlane_f((y1 - p1) * rinv1, j - 64);
      if (lane == j - 64) x1 = xj;
      p1 = fmaf(row1[j - 64], xj, p1);
      p0 = fmaf(row0[j], xj, p0);
    }
#pragma unroll
    for (int j = 63; j >= 0; j--) {
      const float xj = readlane_f((y0 - p0) * rinv0, j);
      if (lane == j) x0 = xj;
      p0 = fmaf(row0[j], xj, p0);
    }
    // virtual index k = 16 I + m'  <->  latent dim T m' + I
    const int dim0 = T * rk + I0, dim1 = T * rk + I1;
    const bool fin = (__builtin_isfinite(x0) || dim0 >= K) && (__builtin_isfinite(x1) || dim1 >= K);
    if (!__all(fin)) {
      if (lane == 0) atomicOr(err_flag, 2);
    }
    xrow[dim0] = dim0 < K ? x0 : 0.f;
    xrow[dim1] = dim1 < K ? x1 : 0.f;
    return;
  }
  const int k = lane < KP ? lane : KP - 1;
  const int Ik = k >> 4, rk = k & 15;
  float rowk[KP];
#pragma unroll
  for (int j = 0; j < T; j++) {
    const int jj = j >= Ik ? j : Ik;  // tiles left of the diagonal are never needed
    const float *src = sm + (Ik * T - Ik * (Ik - 1) / 2 + (jj - Ik)) * C::TS + rk * C::RS;
#pragma unroll
    for (int c = 0; c < 4; c++) {
      const f32x4 v = *reinterpret_cast<const f32x4 *>(src + 4 * c);
      rowk[16 * j + 4 * c] = v.x; rowk[16 * j + 4 * c + 1] = v.y;
      rowk[16 * j + 4 * c + 2] = v.z; rowk[16 * j + 4 * c + 3] = v.w;
    }
  }
  const float rinv_k = 1.0f / sm[(Ik * T - Ik * (Ik - 1) / 2) * C::TS + rk * C::RS + rk];
  const float ys = ybuf[k] * rinv_k;
  float partial = 0.f, xv = 0.f;
#pragma unroll
  for (int j = KP - 1; j >= 0; j--) {
    const float t = fmaf(-partial, rinv_k, ys);  // (y_k - partial) / R_kk in one instruction
    const float xj = readlane_f(t, j);
    if (lane == j) xv = xj;
    partial = fmaf(rowk[j], xj, partial);
  }
  // virtual index k = 16 I + m'  <->  latent dim T m' + I
  const int dim = T * rk + Ik;
  const bool fin = __builtin_isfinite(xv) || lane >= KP || dim >= K;
  if (!__all(fin)) {
    if (lane == 0) atomicOr(err_flag, 2);
  }
  if (lane < KP) xrow[dim] = dim < K ? xv : 0.f;
}

// Conjugate gradient for KP = 128 (hpp:199-264) with two matrix rows per lane.  The matrix
// is spilled once as packed upper tiles (the layout of solve_row_cholesky, 46 KB) and lane l
// reads the virtual rows l and 64 + l back through the symmetry: tiles right of the diagonal
// row-wise (b128), tiles left of it column-wise.  Everything runs in the virtual basis
// k = 16 I + m'  <->  latent dim T m' + I, which is a permutation and leaves CG's iterates
// unchanged.
// LOWER_ACC: slot tix(i, j) holds the tile (row block j, column block i) (the bf16x3 rank update): it is
// the transpose of the upper tile, so it is staged transposed.
template <int T, bool LOWER_ACC = false>
__device__ __forceinline__ void solve_row_cg128(f32x4 (&acc)[Geo<T>::NT], const float (&b4)[T],
                                                float reg, float *sm, float *xrow, int K, int nnz,
                                                int max_cg_steps, int warm_start,
                                                int32_t *err_flag) {
  using C = CholGeo<T>;
  static_assert(T == 8, "two rows per lane");
  constexpr int KP = 128;
  const int lane = threadIdx.x & 63;
  const int g = lane >> 4, m = lane & 15;
  if (nnz == 0) {  // hpp:207-210
    xrow[lane] = 0.f;
    xrow[64 + lane] = 0.f;
    return;
  }
  // (the regulariser is not added to the diagonal: see solve_row<T, 1>)
  float *bbuf = sm + C::NT * C::TS_CG;
#pragma unroll
  for (int i = 0; i < T; i++) {
#pragma unroll
    for (int j = i; j < T; j++)
#pragma unroll
      for (int r = 0; r < 4; r++) {
        if constexpr (LOWER_ACC) sm[C::tix(i, j) * C::TS_CG + m * C::RS_CG + (4 * g + r)] = acc[C::tix(i, j)][r];
        else sm[C::tix(i, j) * C::TS_CG + (4 * g + r) * C::RS_CG + m] = acc[C::tix(i, j)][r];
      }
    if (g == 0) bbuf[16 * i + m] = b4[i];
  }
  __threadfence_block();
  const int rk = lane & 15;
  const int Iq[2] = {lane >> 4, 4 + (lane >> 4)};
  auto tile_of = [&](int I, int J) { return I * T - I * (I - 1) / 2 + (J - I); };
  float a[2][KP];
#pragma unroll
  for (int q = 0; q < 2; q++)
#pragma unroll
    for (int J = 0; J < T; J++) {
      if (J >= Iq[q]) {  // row rk of tile (I, J)
        const float *src = sm + tile_of(Iq[q], J) * C::TS_CG + rk * C::RS_CG;
#pragma unroll
        for (int c = 0; c < 4; c++) {
          const f32x4 v = *reinterpret_cast<const f32x4 *>(src + 4 * c);
          a[q][16 * J + 4 * c] = v.x; a[q][16 * J + 4 * c + 1] = v.y;
          a[q][16 * J + 4 * c + 2] = v.z; a[q][16 * J + 4 * c + 3] = v.w;
        }
      } else {  // column rk of tile (J, I)
        const float *src = sm + tile_of(J, Iq[q]) * C::TS_CG + rk;
#pragma unroll
        for (int c = 0; c < 16; c++) a[q][16 * J + c] = src[c * C::RS_CG];
      }
    }
  // virtual row 16 I + rk is latent dim T rk + I
  const int dim[2] = {T * rk + Iq[0], T * rk + Iq[1]};
  const bool act[2] = {dim[0] < K, dim[1] < K};
  const float bv[2] = {bbuf[lane], bbuf[64 + lane]};
  float x[2], r[2], p[2];
  auto matvec = [&](const float (&vec)[2], float (&out)[2]) {
    float s0 = 0.f, s1 = 0.f;
#pragma unroll
    for (int k = 0; k < KP; k++) {
      const float vk = readlane_f(vec[k >> 6], k & 63);
      s0 = fmaf(a[0][k], vk, s0);
      s1 = fmaf(a[1][k], vk, s1);
    }
    out[0] = fmaf(reg, vec[0], s0);
    out[1] = fmaf(reg, vec[1], s1);
  };
#pragma unroll
  for (int q = 0; q < 2; q++) x[q] = (warm_start && act[q]) ? xrow[dim[q]] : 0.f;
  {
    float Ax[2];
    matvec(x, Ax);
#pragma unroll
    for (int q = 0; q < 2; q++) {
      r[q] = act[q] ? bv[q] - Ax[q] : 0.f;
      p[q] = r[q];
    }
  }
  bool singular = false;
  for (int it = 0; it < max_cg_steps; it++) {
    const float r2 = wave_sum(r[0] * r[0] + r[1] * r[1]);
    if (r2 <= 1e-20f) break;  // hpp:238
    float Ap[2];
    matvec(p, Ap);
#pragma unroll
    for (int q = 0; q < 2; q++) Ap[q] = act[q] ? Ap[q] : 0.f;
    const float denom = wave_sum(p[0] * Ap[0] + p[1] * Ap[1]);
    if (!(denom > 0.f) || !__builtin_isfinite(denom)) {  // hpp:250-254
      singular = true;
      break;
    }
    const float alpha = r2 / denom;
#pragma unroll
    for (int q = 0; q < 2; q++) {
      x[q] = fmaf(alpha, p[q], x[q]);
      r[q] = fmaf(-alpha, Ap[q], r[q]);
    }
    const float r2n = wave_sum(r[0] * r[0] + r[1] * r[1]);
    if (r2n <= 1e-20f) break;  // hpp:258
    const float beta = r2n / r2;  // hpp:261
#pragma unroll
    for (int q = 0; q < 2; q++) p[q] = fmaf(beta, p[q], r[q]);
  }
  if (singular) {
    if (lane == 0) atomicOr(err_flag, 4);
  }
#pragma unroll
  for (int q = 0; q < 2; q++) xrow[dim[q]] = act[q] ? x[q] : 0.f;
}

// The Cholesky solve of these kernels lives in ials_chol16.hpp (which includes this header).
template <int T> struct Chol16Geo;
template <int T>
__device__ __forceinline__ void solve_row_cholesky16(f32x4 (&acc)[Geo<T>::NT], const float (&b4)[T],
                                                     float reg, float *sm, float *xrow, int K,
                                                     int32_t *err_flag);

// MODE 0: one wave per task.  Whole rows are solved inline; chunks of split
//         rows store their partial Gramian / rhs.
// MODE 1: one wave per split row: sum the partials in slot order and solve.
#ifdef IRS_IALS_PHASES
// development: shader-clock sums per phase (0 gather + rank update, 1 diagonal tiles, 2 TRSM,
// 3 trailing update, 4 back substitution), read and cleared by irs_ials_profile_read
__device__ unsigned long long ials_phase_clk[8 * 4096];
#define IPHASE(i) do { if ((threadIdx.x & 63) == 0) { const unsigned long long t_ = clock64(); atomicAdd(&ials_phase_clk[(i) * 4096 + (blockIdx.x & 4095)], t_ - ph_t); ph_t = t_; } } while (0)
#define IPHASE_BEGIN unsigned long long ph_t = clock64()
#else
#define IPHASE(i)
#define IPHASE_BEGIN
#endif

// BF16X3: the rank update of syrk_gather_bf16x3 (UNIT, Cholesky, T == 4 only; the default there since round 6).
// RESID: rhs -= (P x)_row + reg x (hpp:459-466: the Gramian and ridge terms of the gradient); lane (g, m)
// holds the dims T m .. T m + T - 1
template <int T>
__device__ __forceinline__ void resid_finish_rhs(const SolveParams &p, int row, const float (&xr)[T], float (&bsum)[T]) {
  const int m = threadIdx.x & 15;
  const float reg = p.reg[row];
  float px[T];
  load_dims<T>(p.px + static_cast<size_t>(row) * (16 * T) + T * m, px);
#pragma unroll
  for (int i = 0; i < T; i++) bsum[i] -= fmaf(reg, xr[i], px[i]);
}
// RESID: the solve left the step in the row (written by the lanes of group 0): x <- x + step
template <int T> __device__ __forceinline__ void resid_add_step(float *xrow, const float (&xr)[T]) {
  const int lane = threadIdx.x & 63;
  if (lane < 16) {
#pragma unroll
    for (int i = 0; i < T; i++) xrow[T * lane + i] += xr[i];
  }
}

// RESID: the one-block iALS++ sweep (see syrk_gather_impl): right-hand side = negative gradient at the
// current row, solution = the step; SOLVER 0, T <= 4.
template <int T, int SOLVER, int MODE, bool UNIT = false, bool BF16X3 = false, bool RESID = false>
__global__ __launch_bounds__(64 * SOLVE_WAVES, (T > 4 && (SOLVER == 1 || BF16X3)) ? 1 : (T <= 4 ? (RESID && MODE == 0 ? (UNIT ? 3 : 4) : (BF16X3 && MODE == 0 ? 3 : SOLVE_MIN_WAVES_PER_SIMD_K64)) : SOLVE_MIN_WAVES_PER_SIMD)) void ials_solve_kernel(SolveParams p) {
  // bf16x3: unit confidences, K <= 64 padded to 64; Cholesky (whose split rows' second pass is the plain MODE 1
  // kernel: both work on lower tiles) and, round 6, CG on the explicit system (MODE 1 with BF16X3 = "the
  // partials and the Gramian are lower-form tiles")
  static_assert(!BF16X3 || ((T == 4 || T == 8) && (MODE == 1 ? SOLVER == 1 : UNIT)),
                "bf16x3: unit confidences at 48 < K <= 64 and 64 < K <= 128");
  static_assert(!RESID || (SOLVER == 0 && T <= 4), "the gradient form: Cholesky at K <= 64");
  constexpr int WAVES = SOLVE_WAVES;
  using G = Geo<T>;
  // Cholesky: lower-form tiles + the 16-row block solve of ials_chol16.hpp
  constexpr bool LOWER = SOLVER == 0 || BF16X3;
  constexpr int RING = (T > 4 && SOLVER == 0) ? 4 : 8;  // gathered sub-steps in flight
  constexpr int LDS_PER_WAVE = SOLVER == 0 ? Chol16Geo<T>::LDS_FLOATS
                                           : (T == 8 ? CholGeo<T>::SPILL_CG_FLOATS : G::CG_LDS_FLOATS);
  __shared__ __attribute__((aligned(16))) float lds[WAVES * LDS_PER_WAVE];
  // (K = 128: with the scalar task the register allocator spills 800 bytes per lane where the
  // per-lane form spills 12 - the two starts of the accumulators become real branches)
  const int wid = T <= 4 ? wave_in_block() : static_cast<int>(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const int w = blockIdx.x * WAVES + wid;
  float *sm = lds + wid * LDS_PER_WAVE;

  f32x4 acc[G::NT];
  float bsum[T];
#pragma unroll
  for (int i = 0; i < T; i++) bsum[i] = 0.f;
  const f32x4 *Pacc = reinterpret_cast<const f32x4 *>(LOWER ? p.P_accL : p.P_acc);

  IPHASE_BEGIN;
  if constexpr (MODE == 0) {
    if (w >= p.n_tasks) return;
    const Task task = p.tasks[w];
    if (task.slot < 0) {
#pragma unroll
      for (int t = 0; t < G::NT; t++) acc[t] = Pacc[t * 64 + lane];
    } else {
#pragma unroll
      for (int t = 0; t < G::NT; t++) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    float xr[T] = {};
    if constexpr (RESID) load_dims<T>(p.target + static_cast<size_t>(task.row) * G::KP + T * (lane & 15), xr);
    if constexpr (BF16X3)
      syrk_gather_bf16x3<T, RESID>(p.other, p.indices, task.begin, task.end, p.bias, acc, bsum,
                                   static_cast<unsigned>(p.zero_row), xr);
    else
      syrk_gather_impl<T, 1, 0, RING, UNIT, LOWER, RESID>(p.other, p.indices, p.data, task.begin, task.end, p.bias,
                                                          acc, bsum, static_cast<unsigned>(p.zero_row), xr);
    IPHASE(0);
    if (task.slot >= 0) {
      float *dst = p.partials + static_cast<size_t>(task.slot) * G::PARTIAL_FLOATS;
      f32x4 *d4 = reinterpret_cast<f32x4 *>(dst);
#pragma unroll
      for (int t = 0; t < G::NT; t++) d4[t * 64 + lane] = acc[t];
      if (lane < 16) {
#pragma unroll
        for (int i = 0; i < T; i++) dst[G::NT * 256 + T * lane + i] = bsum[i];
      }
      return;
    }
    add_prior<T>(p, task.row, bsum);
    if constexpr (RESID) resid_finish_rhs<T>(p, task.row, xr, bsum);
    // with a prior an empty row is solved like any other (hpp:207): hide nnz == 0 from CG
    const int nnz_cg = p.prior ? max(task.end - task.begin, 1) : task.end - task.begin;
    if constexpr (SOLVER == 0) {
      solve_row_cholesky16<T>(acc, bsum, p.reg[task.row], sm,
                              p.target + static_cast<size_t>(task.row) * G::KP, p.K, p.err_flag);
      if constexpr (RESID) resid_add_step<T>(p.target + static_cast<size_t>(task.row) * G::KP, xr);
    }
    else if constexpr (T == 8)
      solve_row_cg128<T, LOWER>(acc, bsum, p.reg[task.row], sm,
                         p.target + static_cast<size_t>(task.row) * G::KP, p.K, nnz_cg,
                         p.max_cg_steps, p.warm_start, p.err_flag);
    else
      solve_row<T, SOLVER, LOWER>(acc, bsum, p.reg[task.row], sm,
                           p.target + static_cast<size_t>(task.row) * G::KP, p.K, nnz_cg,
                           p.max_cg_steps, p.warm_start, p.err_flag);
  } else {
    if (w >= p.n_split) return;
    const SplitRow sr = p.split_rows[w];
#pragma unroll
    for (int t = 0; t < G::NT; t++) acc[t] = Pacc[t * 64 + lane];
    // slots are added in ascending order (deterministic); U of them are loaded before the
    // first add so that the chain is not one exposed memory latency per slot
    constexpr int U = T <= 4 ? 4 : 1;
    for (int s0 = 0; s0 < sr.n_slots; s0 += U) {
      f32x4 part[U][G::NT];
      float pb[U][T];
#pragma unroll
      for (int u = 0; u < U; u++) {
        const int s = min(s0 + u, sr.n_slots - 1);  // clamped: the duplicate is not added
        const float *src = p.partials + static_cast<size_t>(sr.first_slot + s * sr.slot_stride) * G::PARTIAL_FLOATS;
        const f32x4 *s4 = reinterpret_cast<const f32x4 *>(src);
#pragma unroll
        for (int t = 0; t < G::NT; t++) part[u][t] = s4[t * 64 + lane];
#pragma unroll
        for (int i = 0; i < T; i++) pb[u][i] = src[G::NT * 256 + T * (lane & 15) + i];
      }
#pragma unroll
      for (int u = 0; u < U; u++) {
        if (s0 + u < sr.n_slots) {
#pragma unroll
          for (int t = 0; t < G::NT; t++) acc[t] += part[u][t];
#pragma unroll
          for (int i = 0; i < T; i++) bsum[i] += pb[u][i];
        }
      }
    }
    add_prior<T>(p, sr.row, bsum);
    float xr[T] = {};
    if constexpr (RESID) {
      load_dims<T>(p.target + static_cast<size_t>(sr.row) * G::KP + T * (lane & 15), xr);
      resid_finish_rhs<T>(p, sr.row, xr, bsum);
    }
    if constexpr (SOLVER == 0) {
      solve_row_cholesky16<T>(acc, bsum, p.reg[sr.row], sm,
                              p.target + static_cast<size_t>(sr.row) * G::KP, p.K, p.err_flag);
      if constexpr (RESID) resid_add_step<T>(p.target + static_cast<size_t>(sr.row) * G::KP, xr);
    }
    else if constexpr (T == 8)
      solve_row_cg128<T, LOWER>(acc, bsum, p.reg[sr.row], sm,
                         p.target + static_cast<size_t>(sr.row) * G::KP, p.K, sr.nnz,
                         p.max_cg_steps, p.warm_start, p.err_flag);
    else
      solve_row<T, SOLVER, LOWER>(acc, bsum, p.reg[sr.row], sm,
                           p.target + static_cast<size_t>(sr.row) * G::KP, p.K, sr.nnz,
                           p.max_cg_steps, p.warm_start, p.err_flag);
  }
}

// ---------------------------------------------------------------------------
// Dense Gramian  sum_r f_r f_r^T  over rows [row_begin, row_end) on the matrix
// cores (Solver::prepare_p, hpp:78-115).  Each wave owns a slab of rows and all
// upper tiles; a block folds its four waves and gramian_reduce_kernel sums the
// per-block partials in a fixed order (bit-reproducible run to run).
template <int T>
__global__ __launch_bounds__(256) void gramian_partial_kernel(const float *__restrict__ F,
                                                              int64_t row_begin,
                                                              int64_t row_end,
                                                              int64_t rows_per_wave,
                                                              float *__restrict__ partial) {
  using G = Geo<T>;
  const int lane = threadIdx.x & 63;
  const int g = lane >> 4, m = lane & 15;
  const int64_t w = static_cast<int64_t>(blockIdx.x) * 4 + wave_in_block();
  f32x4 acc[G::NT];
#pragma unroll
  for (int t = 0; t < G::NT; t++) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int64_t b = row_begin + w * rows_per_wave;
  const int64_t e = min(b + rows_per_wave, row_end);
  for (int64_t r = b; r < e; r += 4) {
    float v[T];
    if (r + g < e) {
      load_dims<T>(F + (r + g) * G::KP + T * m, v);
    } else {
#pragma unroll
      for (int i = 0; i < T; i++) v[i] = 0.f;
    }
    int t = 0;
#pragma unroll
    for (int i = 0; i < T; i++)
#pragma unroll
      for (int j = i; j < T; j++) {
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(v[i], v[j], acc[t], 0, 0, 0);
        t++;
      }
  }
  // fold the block's four waves through LDS (fixed order), one partial per block
  __shared__ f32x4 fold[4 * G::NT * 64];
  const int wid = threadIdx.x >> 6;
#pragma unroll
  for (int t = 0; t < G::NT; t++) fold[(wid * G::NT + t) * 64 + lane] = acc[t];
  __syncthreads();
  f32x4 *dst = reinterpret_cast<f32x4 *>(partial) + static_cast<int64_t>(blockIdx.x) * (G::NT * 64);
  for (int e = threadIdx.x; e < G::NT * 64; e += 256) {
    f32x4 s = fold[e];
    s += fold[G::NT * 64 + e];
    s += fold[2 * G::NT * 64 + e];
    s += fold[3 * G::NT * 64 + e];
    dst[e] = s;
  }
}

// Sum the per-block partials (GRAM_CHAINS interleaved chains per element, combined in a fixed
// order) and write the row-major symmetric KP x KP matrix (unscaled).  Sixteen waves per element
// group (round 5; four before): 16 instead of 64 dependent adds per thread behind as many loads -
// the kernel was 12 us of latency for 2.6 MB of partials, twice per epoch.
// FINISH (the unsharded epoch, where no all-reduce sits between the two): the thread that has summed
// element e of the upper tiles also writes alpha0 times it to everything gramian_finish_kernel
// derives from P_raw - the row-major P (both triangles), the accumulator layout (its own index e) and
// the lower-form layout (the mirrored slot) - one launch and one dependent round trip less per side.
constexpr int GRAM_CHAINS = 16;
template <int T, bool FINISH = false>
__global__ __launch_bounds__(64 * GRAM_CHAINS) void gramian_reduce_kernel(const float *__restrict__ partial,
                                                                          int64_t n_parts,
                                                                          float *__restrict__ P_raw,
                                                                          float alpha0 = 0.f,
                                                                          float *__restrict__ P = nullptr,
                                                                          float *__restrict__ P_acc = nullptr,
                                                                          float *__restrict__ P_accL = nullptr) {
  using G = Geo<T>;
  __shared__ double part[GRAM_CHAINS][64];
  const int q = threadIdx.x >> 6, l = threadIdx.x & 63;
  const int e = blockIdx.x * 64 + l;  // element of [NT][64][4]
  // The partials are summed in float64 and rounded ONCE: the Gramian then carries the rounding of the
  // matrix-core partial sums only (tens of rows each), not that of another few hundred float32 adds -
  // a common-mode error of every row of the half-step (it decided the tail counts of the parity bars
  // at alpha0 = 1, K = 4).  2.6 MB of partials: the float64 adds are not what this kernel waits for.
  double sd = 0.0;
  if (e < G::NT * 256)
    for (int64_t w = q; w < n_parts; w += GRAM_CHAINS) sd += static_cast<double>(partial[w * (G::NT * 256) + e]);
  part[q][l] = sd;
  __syncthreads();
  if (q != 0 || e >= G::NT * 256) return;
  sd = 0.0;
#pragma unroll
  for (int c = 0; c < GRAM_CHAINS; c += 4)  // (a fixed order: bit-reproducible run to run)
    sd += (part[c][l] + part[c + 1][l]) + (part[c + 2][l] + part[c + 3][l]);
  const float s = static_cast<float>(sd);
  const int t = e >> 8, lane = (e >> 2) & 63, r = e & 3;
  int ti = 0, tj = 0, c = t;
  for (int i = 0; i < T; i++) {
    if (c < T - i) { ti = i; tj = i + c; break; }
    c -= T - i;
  }
  const int gg = lane >> 4, m = lane & 15;
  const int row = T * (4 * gg + r) + ti, col = T * m + tj;
  P_raw[row * G::KP + col] = s;
  if (ti != tj) P_raw[col * G::KP + row] = s;
  if constexpr (FINISH) {
    const float a = alpha0 * s;  // (the same product gramian_finish_kernel forms)
    P[row * G::KP + col] = a;
    if (ti != tj) P[col * G::KP + row] = a;
    P_acc[e] = a;
    // lower form: slot (ti, tj) holds the tile (row block tj, column block ti) = this tile transposed:
    // element (col, row) lives in lane (m >> 2, 4 gg + r), register m & 3
    P_accL[t * 256 + ((m >> 2) * 16 + (4 * gg + r)) * 4 + (m & 3)] = a;
  }
}

// P = alpha0 * P_raw (hpp:113) in both row-major and accumulator layout.
template <int T>
__global__ void gramian_finish_kernel(const float *__restrict__ P_raw, float alpha0,
                                      float *__restrict__ P, float *__restrict__ P_acc,
                                      float *__restrict__ P_accL) {
  using G = Geo<T>;
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e < G::KP * G::KP) P[e] = alpha0 * P_raw[e];
  if (e < G::NT * 256) {
    const int t = e >> 8, lane = (e >> 2) & 63, r = e & 3;
    int ti = 0, tj = 0, c = t;
    for (int i = 0; i < T; i++) {
      if (c < T - i) { ti = i; tj = i + c; break; }
      c -= T - i;
    }
    const int gg = lane >> 4, m = lane & 15;
    const int row = T * (4 * gg + r) + ti, col = T * m + tj;
    P_acc[e] = alpha0 * P_raw[row * G::KP + col];
    // lower form: slot (ti, tj) holds the tile (row block tj, column block ti)
    const int rowL = T * (4 * gg + r) + tj, colL = T * m + ti;
    P_accL[e] = alpha0 * P_raw[rowL * G::KP + colL];
  }
}

// ---------------------------------------------------------------------------
// user_scores: out[m, n_items] = user[begin:end] @ item^T (hpp:942-984).
// One wave per 64 users x 64 items (4 x 4 MFMA tiles: every item row that is loaded feeds
// four user tiles, which divides the L2 traffic of the 16-user version by ~2.5); k runs
// over the latent dims 16 at a time with each lane loading 16 B of each of its 4 user rows
// and 4 item rows.  Per (user, item) the products are added in ascending k.
template <int KP>
__global__ __launch_bounds__(256) void user_scores_kernel(const float *__restrict__ user,
                                                          const float *__restrict__ item,
                                                          int64_t begin, int64_t m_rows,
                                                          int64_t n_items,
                                                          float *__restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int g = lane >> 4, m = lane & 15;
  const int64_t item_tiles = (n_items + 63) / 64;
  const int64_t w = static_cast<int64_t>(blockIdx.x) * 4 + wave_in_block();
  const int64_t ut = w / item_tiles, it = w % item_tiles;
  if (ut * 64 >= m_rows) return;
  const float *up[4], *ip[4];
#pragma unroll
  for (int q = 0; q < 4; q++) {
    const int64_t u = min(ut * 64 + q * 16 + m, m_rows - 1);
    // item tile q holds the columns 4 m + q: lane m then owns 4 adjacent columns of a row
    // across the four tiles and a row leaves as one 256 B run of 16 B stores
    const int64_t i = min(it * 64 + 4 * m + q, n_items - 1);
    up[q] = user + (begin + u) * KP + 4 * g;
    ip[q] = item + i * KP + 4 * g;
  }
  f32x4 acc[4][4];
#pragma unroll
  for (int p = 0; p < 4; p++)
#pragma unroll
    for (int q = 0; q < 4; q++) acc[p][q] = f32x4{0.f, 0.f, 0.f, 0.f};
  // the operands of step k + 1 are requested before the 64 MFMAs of step k are issued (left to
  // itself hipcc loads each operand right in front of its first use: 16 exposed L2 latencies
  // per tile)
  f32x4 a[2][4], b[2][4];
  auto load_step = [&](int k, f32x4 (&aa)[4], f32x4 (&bb)[4]) {
#pragma unroll
    for (int q = 0; q < 4; q++) {
      aa[q] = *reinterpret_cast<const f32x4 *>(up[q] + k);
      bb[q] = *reinterpret_cast<const f32x4 *>(ip[q] + k);
    }
  };
  // (per accumulator the products are added in ascending k; the four dependent MFMAs of one
  // accumulator are issued 16 instructions apart - back to back each waits 40 cycles for its
  // predecessor instead of 32)
  auto mfma_step = [&](const f32x4 (&aa)[4], const f32x4 (&bb)[4]) {
#pragma unroll
    for (int c = 0; c < 4; c++)
#pragma unroll
      for (int p = 0; p < 4; p++)
#pragma unroll
        for (int q = 0; q < 4; q++)
          acc[p][q] = __builtin_amdgcn_mfma_f32_16x16x4f32(aa[p][c], bb[q][c], acc[p][q], 0, 0, 0);
  };
  static_assert(KP % 32 == 0 || KP == 16, "two steps per round");
  load_step(0, a[0], b[0]);
  __builtin_amdgcn_sched_barrier(0);
  if constexpr (KP == 16) {
    mfma_step(a[0], b[0]);
  } else {
#pragma unroll
    for (int k = 0; k < KP; k += 32) {
      load_step(k + 16, a[1], b[1]);
      __builtin_amdgcn_sched_barrier(0);
      mfma_step(a[0], b[0]);
      __builtin_amdgcn_sched_barrier(0);
      if (k + 32 < KP) load_step(k + 32, a[0], b[0]);
      __builtin_amdgcn_sched_barrier(0);
      mfma_step(a[1], b[1]);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  const int64_t col = it * 64 + 4 * m;
  const bool vec = (n_items & 3) == 0 && col + 3 < n_items &&
                   (reinterpret_cast<uintptr_t>(out) & 15) == 0;
#pragma unroll
  for (int p = 0; p < 4; p++)
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int64_t row = ut * 64 + p * 16 + 4 * g + r;
      if (row >= m_rows) continue;
      float *dst = out + row * n_items + col;
      if (vec) {
        *reinterpret_cast<f32x4 *>(dst) =
            f32x4{acc[p][0][r], acc[p][1][r], acc[p][2][r], acc[p][3][r]};
      } else {
#pragma unroll
        for (int q = 0; q < 4; q++)
          if (col + q < n_items) dst[q] = acc[p][q][r];
      }
    }
}

// ---------------------------------------------------------------------------
// compute_loss pieces (hpp:845-917): per-row partial losses, summed afterwards
// in row order.  One wave per row; 16 lanes share a gathered row.
template <int T>
__global__ __launch_bounds__(256) void loss_rows_kernel(const float *__restrict__ target,
                                                        const float *__restrict__ other,
                                                        const int32_t *__restrict__ indptr,
                                                        const int32_t *__restrict__ indices,
                                                        const float *__restrict__ data,
                                                        const float *__restrict__ reg,
                                                        int64_t n_rows, float bias,
                                                        int with_observed,
                                                        float *__restrict__ row_loss) {
  constexpr int KP = Geo<T>::KP;
  const int lane = threadIdx.x & 63;
  const int g = lane >> 4, m = lane & 15;
  const int64_t row = static_cast<int64_t>(blockIdx.x) * 4 + wave_in_block();
  if (row >= n_rows) return;
  float u[T];
  load_dims<T>(target + row * KP + T * m, u);
  float loss = 0.f;
  if (with_observed) {
    const int b = indptr[row], e = indptr[row + 1];
    for (int q0 = b; q0 < e; q0 += 4) {  // 4 stored entries per step, one per 16-lane group
      const int q = q0 + g;
      const bool valid = q < e;
      const int idx = valid ? indices[q] : 0;
      const float c = valid ? data[q] : 0.f;
      float v[T];
      load_dims<T>(other + static_cast<size_t>(idx) * KP + T * m, v);
      float d = 0.f;
#pragma unroll
      for (int i = 0; i < T; i++) d = fmaf(u[i], v[i], d);
      d += __shfl_xor(d, 1, 64);
      d += __shfl_xor(d, 2, 64);
      d += __shfl_xor(d, 4, 64);
      d += __shfl_xor(d, 8, 64);
      if (valid && m == 0) loss += c * d * d - 2.f * (c + bias) * d + c + bias;  // hpp:867-869
    }
  }
  float n2 = 0.f;
  if (g == 0) {
#pragma unroll
    for (int i = 0; i < T; i++) n2 = fmaf(u[i], u[i], n2);
  }
  loss += reg[row] * n2;
  loss = wave_sum(loss);
  if (lane == 0) row_loss[row] = loss;
}

// Deterministic sum of n floats in double, fixed tree.
__global__ void sum_kernel(const float *__restrict__ v, int64_t n, double *__restrict__ out) {
  __shared__ double sh[256];
  double s = 0.0;
  for (int64_t i = threadIdx.x; i < n; i += 256) s += static_cast<double>(v[i]);
  sh[threadIdx.x] = s;
  __syncthreads();
  for (int off = 128; off >= 1; off >>= 1) {
    if (static_cast<int>(threadIdx.x) < off) sh[threadIdx.x] += sh[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) *out = sh[0];
}

}  // namespace ials
}  // namespace irs
