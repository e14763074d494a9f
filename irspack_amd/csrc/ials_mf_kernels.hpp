// Matrix-free conjugate gradient for 128 < K <= 256 (step_cg, hpp:199-264).
//
// The reference never forms a row's K x K system under CG (hpp:222-247):
//     A vec = P vec + reg_r vec + sum_q c_q (v_q . vec) v_q
// costs (s + 1) (2 K^2 + 4 K n) flops for s steps over a row of n stored entries, where building
// A first costs n K^2.  Up to K = 128 the explicit build is the faster form on this machine (the
// rank update runs on the matrix cores and the system lives in one wave's registers); at K = 256 it
// executed 9 x the algorithmic flops (45 ms per epoch on the ML-20M shape).  Here:
//
//   * rows of at most MF_NCAP = 320 stored entries (mf_cg_rows_kernel): a workgroup solves
//     R = 4, 2 or 1 rows (of at most 40, 80, 160 / 320 entries) with W = 1, 2, 4 / 8 waves per row
//     and GATHERS THE ROWS' FACTOR ROWS ONCE INTO REGISTERS - 16 lanes per gathered row, four
//     entries per wave instruction, ten such groups per wave - and all s + 1 products (and the
//     right-hand side) run on those registers: the dot v_q . vec is KP / 16 fused multiply-adds
//     and one 16-lane DPP sum, the update KP / 16 more.  P vec: every wave takes its share of the
//     rows of P, lane l the columns 4 l .. 4 l + 3 (P is symmetric: one coalesced 16-byte load per
//     lane and row of P), for ALL R rows of the workgroup at once.  The partial vectors of a
//     product meet in LDS and each thread sums its dimensions in a fixed order.  With one wave per
//     row a row's dot products are wave reductions: two barriers per product instead of eight.
//   * longer rows run LEVEL-SYNCHRONOUSLY (mf_chunk_kernel + mf_row_kernel, one pair of launches
//     per product, on a second stream beside the resident kernels): the row's entries are cut
//     into chunks of <= MF_CHUNK entries, one workgroup per chunk streams its entries once per
//     product (same lane layout, loads four groups deep) and leaves one partial vector; one
//     workgroup per row then adds the row's partials in chunk order, P vec and reg vec and takes
//     the CG step.  A row of 10^5 entries is therefore spread over a hundred compute units per
//     product instead of being one workgroup's serial work.
//
// Iterates, exits and the singular-system flag are the reference's (the code of gk_cg_kernel);
// `reg vec` enters LAST in one fma (where the regulariser enters a float32 CG
// product).  Sums over a row's entries: float32 per lane (at most n / 16 terms, 64 per chunk),
// partial vectors combined in float64.
#pragma once
#include "ials_eig_kernels.hpp"

namespace irs {
namespace ials {

#ifndef IRS_MF_J
#define IRS_MF_J 10
#endif
#ifndef IRS_MF_PUNROLL
#define IRS_MF_PUNROLL 2
#endif
constexpr int MF_J = IRS_MF_J;           // groups of four gathered rows a wave keeps in registers
constexpr int MF_NCAP = 32 * MF_J;       // longest resident row: 8 waves x MF_J groups of 4 entries
// row classes by stored entries, longest first: class 0 = level-synchronous, then the resident
// classes of mf_cg_rows_kernel<KP, R, W, 10> (capacity 40 W): <1, 8>, <1, 4>, <2, 2>, <4, 1>
constexpr int MF_CHUNK = 1024;  // entries per chunk of a level-synchronous row
constexpr int MF_CLASSES = 5;
constexpr int32_t MF_CAPS[MF_CLASSES] = {INT32_MAX, MF_NCAP, 16 * MF_J, 8 * MF_J, 4 * MF_J};

struct MfLongRow {
  int32_t row;          // row of the solved side
  int32_t first_chunk;  // chunks [first_chunk, first_chunk + n_chunks) in row order
  int32_t n_chunks;
  int32_t pad;
};
struct MfChunk {
  int32_t lrow;   // index into the long-row list
  int32_t begin;  // [begin, end) into indices / data
  int32_t end;
  int32_t pad;
};

struct MfParams {
  const int32_t *rows;  // resident kernels: row = rows[row_first + blockIdx.x] (longest first)
  int32_t row_first, n_rows;
  const int32_t *indptr;
  const int32_t *indices;
  const float *data;
  const float *other;  // gathered factors [n_other, KP]
  float *target;       // solved factors   [n_rows, KP]
  const float *reg;    // per-row regulariser (hpp:117-120)
  const float *P;      // alpha0 F^T F, row-major [KP, KP]
  float bias;          // hpp:190-191
  int32_t K;
  int32_t max_cg_steps;
  int32_t *err_flag;
  // level-synchronous rows
  const MfLongRow *lrows;
  const MfChunk *chunks;
  int32_t n_lrows, n_chunks;
  float *vec;       // [n_lrows, KP]: the vector of the next product (x0, then the search direction)
  float *xs, *rs;   // [n_lrows, KP]: iterate and residual
  float *partial;   // [n_chunks, 2, KP]: per chunk sum c (v . vec) v, and (first pass) sum (bias + c) v
  float *r2;        // [n_lrows]
  int32_t *done;    // [n_lrows]: 1 once the row has left the loop (hpp:238, 250-258)
};

// ---- pieces shared by the kernels ---------------------------------------------------------

// partial[t] for t in [0, KP): (P vec)[columns 4 l ..] over this wave's rows of P
template <int KP>
__device__ __forceinline__ f32x4 mf_p_times_vec(const float *__restrict__ P, const float *vec_lds, int w,
                                                 int lane) {
  constexpr int KQ = KP / 4;  // rows of P per wave
  f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0;
  if (4 * lane < KP) {
    const float *Pw = P + static_cast<size_t>(KQ * w) * KP + 4 * lane;
#pragma unroll 2
    for (int k = 0; k < KQ; k += 4) {
      const f32x4 pk = *reinterpret_cast<const f32x4 *>(vec_lds + KQ * w + k);  // (one address: broadcast)
      const f32x4 r0 = *reinterpret_cast<const f32x4 *>(Pw + static_cast<size_t>(k) * KP);
      const f32x4 r1 = *reinterpret_cast<const f32x4 *>(Pw + static_cast<size_t>(k + 1) * KP);
      const f32x4 r2 = *reinterpret_cast<const f32x4 *>(Pw + static_cast<size_t>(k + 2) * KP);
      const f32x4 r3 = *reinterpret_cast<const f32x4 *>(Pw + static_cast<size_t>(k + 3) * KP);
#pragma unroll
      for (int i = 0; i < 4; i++) {
        a0[i] = fmaf(r0[i], pk.x, a0[i]);
        a1[i] = fmaf(r1[i], pk.y, a1[i]);
        a0[i] = fmaf(r2[i], pk.z, a0[i]);
        a1[i] = fmaf(r3[i], pk.w, a1[i]);
      }
    }
  }
  return f32x4{a0.x + a1.x, a0.y + a1.y, a0.z + a1.z, a0.w + a1.w};
}

__device__ __forceinline__ float mf_block_sum(float v, float *red, int w, int lane) {
  v = wave_sum(v);
  __syncthreads();
  if (lane == 0) red[w] = v;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}

// one group of four gathered rows held by a wave (16 lanes per row; lane m: dims 4 m + 64 i + e)
template <int NV> struct MfGroup {
  f32x4 v[NV];
  float c;
};

// (index and confidence are loaded by the caller, for ALL its groups before the first row load: an
// index load issued between two groups' row loads makes the wait for it - vmcnt counts in order -
// drain the row loads in front of it, and the groups of a row were fetched one latency after the
// other: 20 groups of a 320-entry row took 200 us from the 142 MB user table)
template <int NV>
__device__ __forceinline__ void mf_load_group(MfGroup<NV> &grp, const MfParams &p, int KP, int idx, float c,
                                              int m) {
  grp.c = c;
  // lane m takes the dims 4 m .. 4 m + 3 of every 64-dim block: each of the NV load instructions
  // reads 256 contiguous bytes per gathered row (16 B per lane at a 64 B stride touched every cache
  // line of the row from every instruction)
  const float *src = p.other + static_cast<size_t>(idx) * KP + 4 * m;
#pragma unroll
  for (int i = 0; i < NV; i++) grp.v[i] = *reinterpret_cast<const f32x4 *>(src + 64 * i);
}

// acc += c (v . pl) v for the lane's slice of one gathered row
template <int NV>
__device__ __forceinline__ void mf_dot_update(const MfGroup<NV> &grp, const f32x4 (&pl)[NV], f32x4 (&acc)[NV]) {
  float d0 = 0.f, d1 = 0.f;
#pragma unroll
  for (int i = 0; i < NV; i++) {
    d0 = fmaf(grp.v[i].x, pl[i].x, d0);
    d1 = fmaf(grp.v[i].y, pl[i].y, d1);
    d0 = fmaf(grp.v[i].z, pl[i].z, d0);
    d1 = fmaf(grp.v[i].w, pl[i].w, d1);
  }
  const float s = grp.c * row16_sum(d0 + d1);
#pragma unroll
  for (int i = 0; i < NV; i++)
#pragma unroll
    for (int e = 0; e < 4; e++) acc[i][e] = fmaf(s, grp.v[i][e], acc[i][e]);
}

template <int NV>
__device__ __forceinline__ void mf_rhs_update(const MfGroup<NV> &grp, float bias, bool ok, f32x4 (&acc)[NV]) {
  const float s = ok ? bias + grp.c : 0.f;  // hpp:212-221
#pragma unroll
  for (int i = 0; i < NV; i++)
#pragma unroll
    for (int e = 0; e < 4; e++) acc[i][e] = fmaf(s, grp.v[i][e], acc[i][e]);
}

// (P vec)[columns 4 l ..] over this wave's rows of P for the R vectors of a workgroup's rows: every
// 16-byte load of P serves all R rows
template <int KP, int R, int NW>
__device__ __forceinline__ void mf_p_times_vecs(const float *__restrict__ P, const float (*vecs)[KP], int w,
                                                int lane, f32x4 (&out)[R]) {
  constexpr int KQ = KP / NW;  // rows of P per wave
#pragma unroll
  for (int r = 0; r < R; r++) out[r] = f32x4{0.f, 0.f, 0.f, 0.f};
  if (4 * lane < KP) {
    const float *Pw = P + static_cast<size_t>(KQ * w) * KP + 4 * lane;
#pragma unroll IRS_MF_PUNROLL
    for (int k = 0; k < KQ; k += 4) {
      f32x4 rw[4];
#pragma unroll
      for (int e = 0; e < 4; e++) rw[e] = *reinterpret_cast<const f32x4 *>(Pw + static_cast<size_t>(k + e) * KP);
#pragma unroll
      for (int r = 0; r < R; r++) {
        const f32x4 pk = *reinterpret_cast<const f32x4 *>(&vecs[r][KQ * w + k]);  // (one address: broadcast)
#pragma unroll
        for (int i = 0; i < 4; i++) {
          out[r][i] = fmaf(rw[0][i], pk.x, out[r][i]);
          out[r][i] = fmaf(rw[1][i], pk.y, out[r][i]);
          out[r][i] = fmaf(rw[2][i], pk.z, out[r][i]);
          out[r][i] = fmaf(rw[3][i], pk.w, out[r][i]);
        }
      }
    }
  }
}

// ---- rows of at most 4 J W stored entries: everything on chip --------------------------------
// One workgroup of NW = R W waves solves R rows side by side, W waves per row (W = 1: every
// reduction of a row is a wave reduction, no barrier; W = 8: a 512-thread workgroup on one row).
// Wave wr of a row holds the entries q = 4 (j W + wr) + g of the row in registers: J = 10 groups,
// 160 registers, so that TWO waves fit a SIMD - with one (J = 20: 420 registers) a compute unit sat
// through every latency of its only workgroup (61 % of the wave cycles waiting, 28 % issuing).  The
// rows of a workgroup are neighbours in the longest-first list and advance in lockstep: the
// barriers of a product (vectors complete -> partial vectors complete) are shared, and so is every
// load of P (mf_p_times_vecs).  Thread tr of a row owns DPT consecutive dimensions of x, r, p.
template <int KP, int R, int W, int J>
__global__ __launch_bounds__(64 * R * W, 2) void mf_cg_rows_kernel(MfParams p) {
  constexpr int NW = R * W, DPL = KP / 16, NV = DPL / 4, DPT = W >= 4 ? 1 : 4 / W, NS = 4 * W + NW;
  static_assert(NW == 4 || NW == 8, "4 or 8 waves per workgroup");
  __shared__ __attribute__((aligned(16))) float ps[R][KP];
  __shared__ __attribute__((aligned(16))) float part[R][NS][KP];
  __shared__ float red[R][W];
  __shared__ int alive[R];
  const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, g = lane >> 4, m = lane & 15;
  const int r = w / W, wr = w % W, tr = wr * 64 + lane, d0 = tr * DPT;
  const int ridx = R * static_cast<int>(blockIdx.x) + r;
  const bool has_row = ridx < p.n_rows;
  const int row = has_row ? p.rows[p.row_first + ridx] : 0;
  const int b = has_row ? p.indptr[row] : 0, n = has_row ? p.indptr[row + 1] - b : 0;
  float *trow = p.target + static_cast<size_t>(row) * KP;
  const float reg = has_row ? p.reg[row] : 0.f;
  const bool dims_in = d0 < KP;  // (KP = 192: the last threads of a row own nothing)
  bool act[DPT];
#pragma unroll
  for (int i = 0; i < DPT; i++) act[i] = dims_in && d0 + i < p.K;
  MfGroup<NV> grp[J];
  {
    int idx[J];
    float cq[J];
#pragma unroll
    for (int j = 0; j < J; j++) {  // entries past the row's end: the row's first entry, weight 0
      const int q = 4 * (j * W + wr) + g;
      idx[j] = n > 0 ? p.indices[b + (q < n ? q : 0)] : 0;
      cq[j] = q < n ? p.data[b + q] : 0.f;
    }
#pragma unroll
    for (int j = 0; j < J; j++) {
      if (4 * (j * W + wr) < n) {  // (wave uniform)
        mf_load_group<NV>(grp[j], p, KP, idx[j], cq[j], m);
      } else {
        grp[j].c = 0.f;
#pragma unroll
        for (int i = 0; i < NV; i++) grp[j].v[i] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
  }
  auto store_gathered = [&](const f32x4 (&acc)[NV]) {
#pragma unroll
    for (int i = 0; i < NV; i++) *reinterpret_cast<f32x4 *>(&part[r][4 * wr + g][4 * m + 64 * i]) = acc[i];
  };
  // sum of the first n_slots partial vectors of this row over this thread's dimensions (fixed order)
  auto sum_slots = [&](int n_slots, float (&out)[DPT]) {
    double s[DPT];
#pragma unroll
    for (int i = 0; i < DPT; i++) s[i] = 0.0;
    if (dims_in)
      for (int sl = 0; sl < n_slots; sl++) {
        float v[DPT];
        if constexpr (DPT == 4) {
          const f32x4 t = *reinterpret_cast<const f32x4 *>(&part[r][sl][d0]);
          v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
        } else if constexpr (DPT == 2) {
          const f32x2 t = *reinterpret_cast<const f32x2 *>(&part[r][sl][d0]);
          v[0] = t.x; v[1] = t.y;
        } else {
          v[0] = part[r][sl][d0];
        }
#pragma unroll
        for (int i = 0; i < DPT; i++) s[i] += static_cast<double>(v[i]);
      }
#pragma unroll
    for (int i = 0; i < DPT; i++) out[i] = static_cast<float>(s[i]);
  };
  // sum over the threads of this row; with W > 1 every wave of the workgroup must call it
  auto row_sum = [&](float v) {
    v = wave_sum(v);
    if constexpr (W > 1) {
      __syncthreads();
      if (lane == 0) red[r][wr] = v;
      __syncthreads();
      v = red[r][0];
#pragma unroll
      for (int k = 1; k < W; k++) v += red[r][k];
    }
    return v;
  };
  auto dot = [&](const float (&a)[DPT], const float (&c)[DPT]) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < DPT; i++) s = fmaf(a[i], c[i], s);
    return row_sum(s);
  };
  // right-hand side b = sum (bias + c) v (hpp:212-221)
  float rr[DPT];
  {
    f32x4 acc[NV];
#pragma unroll
    for (int i = 0; i < NV; i++) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < J; j++)
      if (4 * (j * W + wr) < n) mf_rhs_update<NV>(grp[j], p.bias, 4 * (j * W + wr) + g < n, acc);
    store_gathered(acc);
    __syncthreads();
    sum_slots(4 * W, rr);
  }
  // A vec for the vectors in ps (hpp:222-228, 240-247), AFTER the barrier that completed them; every
  // thread gets its dimensions of its row's product
  auto matvec = [&](float (&out)[DPT]) {
    f32x4 pp[R];
    mf_p_times_vecs<KP, R, NW>(p.P, ps, w, lane, pp);
    f32x4 pl[NV], acc[NV];
#pragma unroll
    for (int i = 0; i < NV; i++) {
      pl[i] = *reinterpret_cast<const f32x4 *>(&ps[r][4 * m + 64 * i]);
      acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int j = 0; j < J; j++)
      if (4 * (j * W + wr) < n) mf_dot_update<NV>(grp[j], pl, acc);
    store_gathered(acc);
    if (4 * lane < KP) {
#pragma unroll
      for (int k = 0; k < R; k++) *reinterpret_cast<f32x4 *>(&part[k][4 * W + w][4 * lane]) = pp[k];
    }
    __syncthreads();
    sum_slots(NS, out);
#pragma unroll
    for (int i = 0; i < DPT; i++) out[i] = dims_in ? fmaf(reg, ps[r][d0 + i], out[i]) : 0.f;  // + reg vec LAST
  };
  bool active = has_row && n > 0;
  float x[DPT], pv[DPT], Ap[DPT];
#pragma unroll
  for (int i = 0; i < DPT; i++) {
    x[i] = (active && act[i]) ? trow[d0 + i] : 0.f;  // warm start (hpp:199); fold-in passes a zeroed target
    rr[i] = (active && act[i]) ? rr[i] : 0.f;
    if (dims_in) ps[r][d0 + i] = x[i];
  }
  __syncthreads();
  matvec(Ap);
#pragma unroll
  for (int i = 0; i < DPT; i++) {
    rr[i] = act[i] ? rr[i] - Ap[i] : 0.f;
    pv[i] = rr[i];
  }
  float r2 = dot(rr, rr);
  bool singular = false;
  for (int it = 0; it < p.max_cg_steps; it++) {
    if (active && r2 <= 1e-20f) active = false;  // hpp:238
    if (lane == 0 && wr == 0) alive[r] = active ? 1 : 0;
#pragma unroll
    for (int i = 0; i < DPT; i++)
      if (dims_in) ps[r][d0 + i] = pv[i];
    __syncthreads();  // vectors and flags complete; the previous product's slots have been read
    int any = 0;
#pragma unroll
    for (int k = 0; k < R; k++) any |= alive[k];
    if (!any) break;  // (uniform over the workgroup)
    matvec(Ap);
#pragma unroll
    for (int i = 0; i < DPT; i++) Ap[i] = act[i] ? Ap[i] : 0.f;
    const float denom = dot(pv, Ap);
    if (active && (!(denom > 0.f) || !__builtin_isfinite(denom))) {  // hpp:250-254
      singular = true;
      active = false;
    }
    const float alpha = active ? r2 / denom : 0.f;
#pragma unroll
    for (int i = 0; i < DPT; i++) {
      x[i] = fmaf(alpha, pv[i], x[i]);
      rr[i] = fmaf(-alpha, Ap[i], rr[i]);
    }
    const float r2n = dot(rr, rr);
    if (active) {
      if (r2n <= 1e-20f) {  // hpp:258
        active = false;
      } else {
        const float beta = r2n / r2;  // hpp:261
#pragma unroll
        for (int i = 0; i < DPT; i++) pv[i] = fmaf(beta, pv[i], rr[i]);
        r2 = r2n;
      }
    }
  }
  if (singular && lane == 0 && wr == 0) atomicOr(p.err_flag, 4);
  if (has_row && dims_in) {
#pragma unroll
    for (int i = 0; i < DPT; i++) trow[d0 + i] = (n > 0 && act[i]) ? x[i] : 0.f;  // (n == 0: hpp:207-210)
  }
}

// ---- level-synchronous rows ------------------------------------------------------------------
// One workgroup per chunk: partial[chunk][0] = sum c (v . vec) v over the chunk's entries with the
// row's current vector; with `first` also partial[chunk][1] = sum (bias + c) v (the chunk's share of
// the right-hand side).
template <int KP>
__global__ __launch_bounds__(256, 2) void mf_chunk_kernel(MfParams p, int first) {
  constexpr int DPL = KP / 16, NV = DPL / 4, DEPTH = 4;
  __shared__ __attribute__((aligned(16))) float ps[KP];
  __shared__ __attribute__((aligned(16))) float part[16][KP];
  const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, g = lane >> 4, m = lane & 15;
  const MfChunk ch = p.chunks[blockIdx.x];
  if (p.done[ch.lrow]) return;  // (uniform: the row left the loop in an earlier step)
  if (tid < KP) ps[tid] = p.vec[static_cast<size_t>(ch.lrow) * KP + tid];
  __syncthreads();
  f32x4 pl[NV], acc[NV], accb[NV];
#pragma unroll
  for (int i = 0; i < NV; i++) {
    pl[i] = *reinterpret_cast<const f32x4 *>(ps + 4 * m + 64 * i);
    acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    accb[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const int n = ch.end - ch.begin;
  // entries q = 16 j + 4 w + g of the chunk; DEPTH groups are loaded before the first is used
  for (int j0 = 0; 16 * j0 + 4 * w < n; j0 += DEPTH) {
    MfGroup<NV> grp[DEPTH];
    int idx[DEPTH];
    float cq[DEPTH];
#pragma unroll
    for (int d = 0; d < DEPTH; d++) {
      const int q = 16 * (j0 + d) + 4 * w + g;
      idx[d] = p.indices[ch.begin + (q < n ? q : 0)];
      cq[d] = q < n ? p.data[ch.begin + q] : 0.f;
    }
#pragma unroll
    for (int d = 0; d < DEPTH; d++) mf_load_group<NV>(grp[d], p, KP, idx[d], cq[d], m);
#pragma unroll
    for (int d = 0; d < DEPTH; d++) {
      mf_dot_update<NV>(grp[d], pl, acc);
      if (first) mf_rhs_update<NV>(grp[d], p.bias, 16 * (j0 + d) + 4 * w + g < n, accb);
    }
  }
  float *out = p.partial + static_cast<size_t>(blockIdx.x) * 2 * KP;
  for (int pass = 0; pass < (first ? 2 : 1); pass++) {
    if (pass) __syncthreads();
#pragma unroll
    for (int i = 0; i < NV; i++)
      *reinterpret_cast<f32x4 *>(&part[4 * w + g][4 * m + 64 * i]) = pass ? accb[i] : acc[i];
    __syncthreads();
    if (tid < KP) {
      double s = 0.0;
      for (int sl = 0; sl < 16; sl++) s += static_cast<double>(part[sl][tid]);
      out[pass * KP + tid] = static_cast<float>(s);
    }
  }
}

// One workgroup per level-synchronous row.  step == 0: x = x0, r = b - A x0, p = r (hpp:199-236);
// step >= 1: one CG step (hpp:237-263).  The iterate is written to the target row at every step, so
// a row that leaves the loop early is finished.
template <int KP>
__global__ __launch_bounds__(256) void mf_row_kernel(MfParams p, int step) {
  __shared__ __attribute__((aligned(16))) float ps[KP];
  __shared__ __attribute__((aligned(16))) float part[4][KP];
  __shared__ float red[4];
  const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
  const int lr = blockIdx.x;
  if (p.done[lr]) return;
  const MfLongRow L = p.lrows[lr];
  const bool act = tid < p.K;
  const float reg = p.reg[L.row];
  float *trow = p.target + static_cast<size_t>(L.row) * KP;
  const size_t off = static_cast<size_t>(lr) * KP + tid;
  const float vec = tid < KP ? p.vec[off] : 0.f;
  if (tid < KP) ps[tid] = vec;
  __syncthreads();
  const f32x4 pp = mf_p_times_vec<KP>(p.P, ps, w, lane);
  if (4 * lane < KP) *reinterpret_cast<f32x4 *>(&part[w][4 * lane]) = pp;
  // the row's chunk partials in chunk order (float64)
  double sa = 0.0, sb = 0.0;
  if (tid < KP) {
    const float *pa = p.partial + static_cast<size_t>(L.first_chunk) * 2 * KP + tid;
    for (int c = 0; c < L.n_chunks; c++) {
      sa += static_cast<double>(pa[static_cast<size_t>(c) * 2 * KP]);
      if (step == 0) sb += static_cast<double>(pa[static_cast<size_t>(c) * 2 * KP + KP]);
    }
  }
  __syncthreads();
  float Av = 0.f;
  if (tid < KP) {
    const double s = sa + ((static_cast<double>(part[0][tid]) + static_cast<double>(part[1][tid])) +
                           (static_cast<double>(part[2][tid]) + static_cast<double>(part[3][tid])));
    Av = fmaf(reg, vec, static_cast<float>(s));  // + reg vec LAST
  }
  Av = act ? Av : 0.f;
  if (step == 0) {
    const float x = act ? vec : 0.f;
    const float r = act ? static_cast<float>(sb) - Av : 0.f;
    const float r2 = mf_block_sum(r * r, red, w, lane);
    if (tid < KP) {
      p.xs[off] = x;
      p.rs[off] = r;
      p.vec[off] = r;
      trow[tid] = x;
    }
    if (tid == 0) {
      p.r2[lr] = r2;
      if (r2 <= 1e-20f || p.max_cg_steps <= 0) p.done[lr] = 1;  // hpp:238
    }
    return;
  }
  const float r2 = p.r2[lr];
  const float denom = mf_block_sum(vec * Av, red, w, lane);
  if (!(denom > 0.f) || !__builtin_isfinite(denom)) {  // hpp:250-254
    if (tid == 0) {
      atomicOr(p.err_flag, 4);
      p.done[lr] = 1;
    }
    return;
  }
  const float alpha = r2 / denom;
  const float x = tid < KP ? fmaf(alpha, vec, p.xs[off]) : 0.f;
  const float r = tid < KP ? fmaf(-alpha, Av, p.rs[off]) : 0.f;
  const float r2n = mf_block_sum(r * r, red, w, lane);
  if (tid < KP) {
    p.xs[off] = x;
    p.rs[off] = r;
    trow[tid] = act ? x : 0.f;
    p.vec[off] = fmaf(r2n / r2, vec, r);  // hpp:261-262
  }
  if (tid == 0) {
    p.r2[lr] = r2n;
    if (r2n <= 1e-20f || step >= p.max_cg_steps) p.done[lr] = 1;  // hpp:258
  }
}

// x0 of the level-synchronous rows -> vec; done = 0
template <int KP>
__global__ void mf_long_init_kernel(MfParams p) {
  const int lr = blockIdx.x, tid = threadIdx.x;
  const MfLongRow L = p.lrows[lr];
  if (tid < KP) p.vec[static_cast<size_t>(lr) * KP + tid] = tid < p.K ? p.target[static_cast<size_t>(L.row) * KP + tid] : 0.f;
  if (tid == 0) p.done[lr] = 0;
}

}  // namespace ials
}  // namespace irs
