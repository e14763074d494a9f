// iALS solve for 64 < K <= 256 (KP = 128 / 192 / 256): one 256-thread workgroup per
// task.  Same algorithm as ials_kernels.hpp (hpp:273-331 Cholesky, hpp:170-271 CG),
// different mapping: the KP x KP Gramian no longer fits one wave's registers, so
//   * the upper 16x16 tiles are dealt round-robin to the four waves, which all walk
//     the same row with the gather pipeline of syrk_gather (MFMA rank update),
//   * the finished matrix (+ P + reg) is written to LDS as a packed lower triangle in
//     natural coordinates (131.6 KB at K = 256) and solved there by the workgroup:
//     right-looking Cholesky with 2 barriers per column, substitutions by one wave;
//     or conjugate gradient on the explicit matrix (one thread per unknown).
// This is the correctness-first path for the K = 128 / 256 configurations; the LDS
// Cholesky is VALU-only (K^3/6 multiply-adds through LDS) and is the part to replace by
// an MFMA-blocked factorisation next.
#pragma once
#include "ials_kernels.hpp"

namespace irs {
namespace ials {

template <int T> struct WgGeo {
  static constexpr int KP = 16 * T;
  static constexpr int NT = T * (T + 1) / 2;
  static constexpr int NW = 4;
  static constexpr int TPW = (NT + NW - 1) / NW;
  static constexpr int PACKED = KP * (KP + 1) / 2;
  // matrix | b | diag | rdiag | y (also CG's p) | reduction scratch
  static constexpr int LDS_FLOATS = PACKED + 4 * KP + 16;
  // CG: upper tiles as 16 x 16 blocks | b | p | reduction scratch
  static constexpr int CG_LDS_FLOATS = NT * 256 + 2 * KP + 16;
};

__device__ __forceinline__ int pk(int r, int c) { return r * (r + 1) / 2 + c; }  // r >= c

// (I, J) of every upper tile as a constant table: indexing it with an unrolled loop
// counter folds to constants, where calling tile_i / tile_j would leave their search loops
// in the kernel for T = 16.
template <int T> struct TileTab {
  int ti[WgGeo<T>::NT], tj[WgGeo<T>::NT];
  constexpr TileTab() : ti{}, tj{} {
    int t = 0;
    for (int i = 0; i < T; i++)
      for (int j = i; j < T; j++) {
        ti[t] = i;
        tj[t] = j;
        t++;
      }
  }
};

__device__ __forceinline__ float block_sum(float v, float *red) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}

// Gather + rank update for one wave of the workgroup (tiles t % 4 == W).  Kept small on
// purpose (the per-wave MFMA work of a sub-step is 9 .. 34 instructions here, so a
// two-deep pipeline of 4-sub-step groups already covers the gather latency): group
// it+1 is gathered while group it is multiplied; indices run one group further ahead.
template <int T, int W>
__device__ __forceinline__ void syrk_gather_wg(const float *__restrict__ other,
                                               const int32_t *__restrict__ indices,
                                               const float *__restrict__ data, int begin, int end,
                                               float bias, f32x4 (&acc)[WgGeo<T>::TPW],
                                               float (&bsum)[T]) {
  using G = WgGeo<T>;
  constexpr int KP = G::KP;
  const int lane = threadIdx.x & 63;
  const int g = lane >> 4, m = lane & 15;
  const float *col_base = other + T * m;
  const int n = end - begin;
  const int nit = (n + 15) >> 4;  // groups of 4 sub-steps = 16 stored entries
  const int32_t *ip = indices + begin + g;
  const float *dp = data + begin + g;
  int ia[4], ib[4];
  float ca[4], cb[4];
  float va[4][T], vb[4][T], wa[4], wb[4], xa[4], xb[4];
  auto load_idx = [&](int it, int (&ix)[4], float (&cx)[4]) {
#pragma unroll
    for (int u = 0; u < 4; u++) {  // over-reads stay inside the padded CSR arrays
      ix[u] = ip[16 * it + 4 * u];
      cx[u] = dp[16 * it + 4 * u];
    }
  };
  auto gather = [&](int it, const int (&ix)[4], const float (&cx)[4], float (&v)[4][T],
                    float (&vc)[4], float (&vw)[4]) {
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const bool valid = 16 * it + 4 * u + g < n;
      vc[u] = valid ? cx[u] : 0.f;
      vw[u] = valid ? bias + cx[u] : 0.f;
      load_dims<T>(col_base + static_cast<size_t>(static_cast<unsigned>(ix[u])) * KP, v[u]);
    }
  };
  auto consume = [&](const float (&v)[4][T], const float (&vc)[4], const float (&vw)[4]) {
#pragma unroll
    for (int u = 0; u < 4; u++) {
      float cv[T], vk[T];
#pragma unroll
      for (int i = 0; i < T; i++) {
        vk[i] = v[u][i];
        cv[i] = vc[u] * vk[i];
        if (W == 0) bsum[i] = fmaf(vw[u], vk[i], bsum[i]);
      }
      mfma_tiles<T, G::NW, W>(cv, vk, acc, std::make_integer_sequence<int, G::TPW>{});
    }
  };
  load_idx(0, ia, ca);
  gather(0, ia, ca, va, xa, wa);
  load_idx(1, ia, ca);
  for (int it = 0; it < nit; it += 2) {
    load_idx(it + 2, ib, cb);
    gather(it + 1, ia, ca, vb, xb, wb);
    consume(va, xa, wa);
    load_idx(it + 3, ia, ca);
    gather(it + 2, ib, cb, va, xa, wa);
    consume(vb, xb, wb);  // an all-masked group when nit is odd
  }
  if (W == 0) {
#pragma unroll
    for (int i = 0; i < T; i++) {
      bsum[i] += __shfl_xor(bsum[i], 16, 64);
      bsum[i] += __shfl_xor(bsum[i], 32, 64);
    }
  }
}

// Rank update of this wave's tiles, then either the partial store (chunk of a split
// row; returns false) or the dump of the tiles into the packed LDS matrix (returns true).
template <int T, int W, int MODE, bool TILES>
__device__ __forceinline__ bool wg_accumulate(const SolveParams &p, int item, float *A, float *bvec,
                                              int &row_out, int &nnz_out) {
  using G = WgGeo<T>;
  const int lane = threadIdx.x & 63;
  const int g = lane >> 4, m = lane & 15;
  f32x4 acc[G::TPW];
  float bsum[T];
#pragma unroll
  for (int i = 0; i < T; i++) bsum[i] = 0.f;
  const f32x4 *Pacc = reinterpret_cast<const f32x4 *>(p.P_acc);
  constexpr int PARTIAL = Geo<T>::PARTIAL_FLOATS;
  int row;
  if constexpr (MODE == 0) {
    const Task task = p.tasks[item];
    row = task.row;
    nnz_out = task.end - task.begin;
#pragma unroll
    for (int s = 0; s < G::TPW; s++) acc[s] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (task.slot < 0) {  // one uniform branch around all the loads (not a select per load)
#pragma unroll
      for (int s = 0; s < G::TPW; s++) {
        const int t = G::NW * s + W;
        if (t < G::NT) acc[s] = Pacc[t * 64 + lane];
      }
    }
    syrk_gather_wg<T, W>(p.other, p.indices, p.data, task.begin, task.end, p.bias, acc, bsum);
    if (task.slot >= 0) {
      float *dst = p.partials + static_cast<size_t>(task.slot) * PARTIAL;
      f32x4 *d4 = reinterpret_cast<f32x4 *>(dst);
#pragma unroll
      for (int s = 0; s < G::TPW; s++) {
        const int t = G::NW * s + W;
        if (t < G::NT) d4[t * 64 + lane] = acc[s];
      }
      if (W == 0 && lane < 16) {
#pragma unroll
        for (int i = 0; i < T; i++) dst[G::NT * 256 + T * lane + i] = bsum[i];
      }
      return false;
    }
  } else {
    const SplitRow sr = p.split_rows[item];
    row = sr.row;
    nnz_out = sr.nnz;
#pragma unroll
    for (int s = 0; s < G::TPW; s++) {
      const int t = G::NW * s + W;
      acc[s] = t < G::NT ? Pacc[t * 64 + lane] : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    for (int sl = 0; sl < sr.n_slots; sl++) {
      const float *src = p.partials + static_cast<size_t>(sr.first_slot + sl) * PARTIAL;
      const f32x4 *s4 = reinterpret_cast<const f32x4 *>(src);
#pragma unroll
      for (int s = 0; s < G::TPW; s++) {
        const int t = G::NW * s + W;
        if (t < G::NT) acc[s] += s4[t * 64 + lane];
      }
      if (W == 0) {
#pragma unroll
        for (int i = 0; i < T; i++) bsum[i] += src[G::NT * 256 + T * m + i];
      }
    }
  }
  row_out = row;
  const float reg = p.reg[row];
  constexpr TileTab<T> tab{};
  if constexpr (TILES) {
    // dump the tiles as they are (virtual basis k = 16 I + m' <-> latent dim T m' + I):
    // tile t, element (4g + r, m); the diagonal gets reg, padded dims a unit diagonal
#pragma unroll
    for (int s = 0; s < G::TPW; s++) {
      const int t = G::NW * s + W;
      if (t < G::NT) {
        const int I = tab.ti[t], J = tab.tj[t];
#pragma unroll
        for (int r = 0; r < 4; r++) {
          float val = acc[s][r];
          if (I == J && 4 * g + r == m) val += (T * m + I < p.K) ? reg : 1.0f;
          A[t * 256 + (4 * g + r) * 16 + m] = val;
        }
      }
    }
    if (W == 0 && g == 0) {
#pragma unroll
      for (int i = 0; i < T; i++) bvec[16 * i + m] = bsum[i];
    }
    return true;
  }
  // dump: tile (I, J) register r of lane (g, m) is element (T (4g+r) + I, T m + J)
#pragma unroll
  for (int s = 0; s < G::TPW; s++) {
    const int t = G::NW * s + W;
    if (t < G::NT) {
      const int I = tab.ti[t], J = tab.tj[t];
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int vr = 4 * g + r;
        if (I == J && vr > m) continue;  // the diagonal tiles hold both triangles: keep one
        const int dr = T * vr + I, dc = T * m + J;
        float val = acc[s][r];
        if (dr == dc) val += dr < p.K ? reg : 1.0f;  // hpp:312-314; padded dims decouple
        A[dr >= dc ? pk(dr, dc) : pk(dc, dr)] = val;
      }
    }
  }
  if (W == 0 && g == 0) {
#pragma unroll
    for (int i = 0; i < T; i++) bvec[T * m + i] = bsum[i];
  }
  return true;
}

// Cholesky A = L L^T in the packed LDS matrix + both substitutions (hpp:316-324).
template <int T>
__device__ __forceinline__ void wg_cholesky(float *A, float *bvec, float *diag, float *rdiag,
                                            float *ybuf, int K, float *xrow, int32_t *err_flag) {
  constexpr int KP = WgGeo<T>::KP;
  const int tid = threadIdx.x;
  const int ty = tid >> 4, tx = tid & 15;
  bool bad = false;
  for (int j = 0; j < K; j++) {
    __syncthreads();  // trailing update of column j-1 is complete
    const float d = A[pk(j, j)];
    bad |= !(d > 0.f);
    const float rinv = __builtin_amdgcn_rsqf(d);
    if (tid == 0) {
      diag[j] = d * rinv;
      rdiag[j] = rinv;
    }
    for (int i = j + 1 + tid; i < K; i += 256) A[pk(i, j)] *= rinv;
    __syncthreads();
    for (int i = j + 1 + ty; i < K; i += 16) {
      const float lij = A[pk(i, j)];
      for (int k = j + 1 + tx; k <= i; k += 16) A[pk(i, k)] = fmaf(-lij, A[pk(k, j)], A[pk(i, k)]);
    }
  }
  __syncthreads();
  if (bad && tid == 0) atomicOr(err_flag, 1);
  // substitutions by wave 0: lane l owns unknowns l, l + 64, ...
  if (tid < 64) {
    constexpr int Q = KP / 64;
    float bl[Q];
#pragma unroll
    for (int q = 0; q < Q; q++) bl[q] = (tid + 64 * q) < K ? bvec[tid + 64 * q] : 0.f;
    // L y = b
#pragma unroll
    for (int q = 0; q < Q; q++) {
      for (int jl = 0; jl < 64; jl++) {
        const int j = 64 * q + jl;
        if (j >= K) break;
        const float yj = readlane_f(bl[q], jl) * rdiag[j];
        if (tid == jl) bl[q] = yj;
#pragma unroll
        for (int q2 = q; q2 < Q; q2++) {
          const int i = tid + 64 * q2;
          if (i > j && i < K) bl[q2] = fmaf(-A[pk(i, j)], yj, bl[q2]);
        }
      }
    }
    // L^T x = y
#pragma unroll
    for (int q = Q - 1; q >= 0; q--) {
      for (int jl = 63; jl >= 0; jl--) {
        const int j = 64 * q + jl;
        if (j >= K) continue;
        const float xj = readlane_f(bl[q], jl) * rdiag[j];
        if (tid == jl) bl[q] = xj;
#pragma unroll
        for (int q2 = 0; q2 <= q; q2++) {
          const int i = tid + 64 * q2;
          if (i < j) bl[q2] = fmaf(-A[pk(j, i)], xj, bl[q2]);
        }
      }
    }
    bool fin = true;
#pragma unroll
    for (int q = 0; q < Q; q++) {
      const int i = tid + 64 * q;
      if (i < K) fin &= __builtin_isfinite(bl[q]) != 0;
      xrow[i] = i < K ? bl[q] : 0.f;
    }
    if (!__all(fin) && tid == 0) atomicOr(err_flag, 2);
  }
  (void)ybuf;
}

// Conjugate gradient on the explicit packed matrix, same iterates as hpp:199-264.
template <int T>
__device__ __forceinline__ void wg_cg(const float *A, const float *bvec, float *pbuf, float *red,
                                      int K, int nnz, int max_cg_steps, float *xrow,
                                      int32_t *err_flag) {
  constexpr int KP = WgGeo<T>::KP;
  const int tid = threadIdx.x;
  const bool act = tid < K;
  if (nnz == 0) {  // hpp:207-210
    if (tid < KP) xrow[tid] = 0.f;
    return;
  }
  auto matvec = [&](float mine) {
    __syncthreads();
    if (tid < KP) pbuf[tid] = mine;
    __syncthreads();
    float s = 0.f;
    if (act) {
      for (int k = 0; k <= tid; k++) s = fmaf(A[pk(tid, k)], pbuf[k], s);
      for (int k = tid + 1; k < K; k++) s = fmaf(A[pk(k, tid)], pbuf[k], s);
    }
    return s;
  };
  float x = act ? xrow[tid] : 0.f;  // warm start (hpp:199); zero for fold-in (hpp:132)
  const float Ax = matvec(x);  // barriers inside: every thread calls it exactly once
  float r = act ? bvec[tid] - Ax : 0.f;
  float pv = r;
  bool singular = false;
  for (int it = 0; it < max_cg_steps; it++) {
    const float r2 = block_sum(r * r, red);
    if (r2 <= 1e-20f) break;  // hpp:238
    const float Ap = matvec(pv);
    const float denom = block_sum(pv * Ap, red);
    if (!(denom > 0.f) || !__builtin_isfinite(denom)) {  // hpp:250-254
      singular = true;
      break;
    }
    const float alpha = r2 / denom;
    x = fmaf(alpha, pv, x);
    r = fmaf(-alpha, Ap, r);
    const float r2n = block_sum(r * r, red);
    if (r2n <= 1e-20f) break;  // hpp:258
    const float beta = r2n / r2;  // hpp:261
    pv = fmaf(beta, pv, r);
  }
  if (singular && tid == 0) atomicOr(err_flag, 4);
  if (tid < KP) xrow[tid] = act ? x : 0.f;
}

// Conjugate gradient (hpp:199-264) with one matrix row per thread in registers.  The tiles
// were dumped in the virtual basis k = 16 I + m' (a permutation of the latent dims, which
// leaves the iterates unchanged); thread k reads its row back through the symmetry: tiles
// right of the diagonal row-wise, tiles left of it column-wise.  A mat-vec is then KP FMAs
// against the broadcast vector in LDS.
template <int T>
__device__ __forceinline__ void wg_cg_rows(const float *A, const float *bvec, float *pbuf, float *red,
                                           int K, int nnz, int max_cg_steps, float *xrow,
                                           int32_t *err_flag) {
  constexpr int KP = WgGeo<T>::KP;
  const int tid = threadIdx.x;
  if (nnz == 0) {  // hpp:207-210
    if (tid < KP) xrow[tid] = 0.f;
    return;
  }
  const int k = tid < KP ? tid : KP - 1;
  const int Ik = k >> 4, rk = k & 15;
  const int dim = T * rk + Ik;
  const bool act = tid < KP && dim < K;
  auto tile_of = [&](int I, int J) { return I * T - I * (I - 1) / 2 + (J - I); };
  float a[KP];
#pragma unroll
  for (int J = 0; J < T; J++) {
    if (J >= Ik) {
      const float *src = A + tile_of(Ik, J) * 256 + rk * 16;
#pragma unroll
      for (int c = 0; c < 4; c++) {
        const f32x4 v = *reinterpret_cast<const f32x4 *>(src + 4 * c);
        a[16 * J + 4 * c] = v.x; a[16 * J + 4 * c + 1] = v.y;
        a[16 * J + 4 * c + 2] = v.z; a[16 * J + 4 * c + 3] = v.w;
      }
    } else {
      const float *src = A + tile_of(J, Ik) * 256 + rk;
#pragma unroll
      for (int c = 0; c < 16; c++) a[16 * J + c] = src[c * 16];
    }
  }
  auto matvec = [&](float mine) {
    __syncthreads();
    if (tid < KP) pbuf[tid] = mine;
    __syncthreads();
    float s0 = 0.f, s1 = 0.f;
#pragma unroll
    for (int q = 0; q < KP / 4; q++) {
      const f32x4 v = *reinterpret_cast<const f32x4 *>(pbuf + 4 * q);
      s0 = fmaf(a[4 * q], v.x, s0);
      s1 = fmaf(a[4 * q + 1], v.y, s1);
      s0 = fmaf(a[4 * q + 2], v.z, s0);
      s1 = fmaf(a[4 * q + 3], v.w, s1);
    }
    return s0 + s1;
  };
  float x = act ? xrow[dim] : 0.f;  // warm start (hpp:199); zero for fold-in (hpp:132)
  const float Ax = matvec(x);  // barriers inside: every thread calls it the same number of times
  float r = act ? bvec[k] - Ax : 0.f;
  float pv = r;
  bool singular = false;
  for (int it = 0; it < max_cg_steps; it++) {
    const float r2 = block_sum(r * r, red);
    if (r2 <= 1e-20f) break;  // hpp:238
    const float Ap_all = matvec(pv);  // one call per thread: the barriers inside must match
    const float Ap = act ? Ap_all : 0.f;
    const float denom = block_sum(pv * Ap, red);
    if (!(denom > 0.f) || !__builtin_isfinite(denom)) {  // hpp:250-254
      singular = true;
      break;
    }
    const float alpha = r2 / denom;
    x = fmaf(alpha, pv, x);
    r = fmaf(-alpha, Ap, r);
    const float r2n = block_sum(r * r, red);
    if (r2n <= 1e-20f) break;  // hpp:258
    const float beta = r2n / r2;  // hpp:261
    pv = fmaf(beta, pv, r);
  }
  if (singular && tid == 0) atomicOr(err_flag, 4);
  if (tid < KP) xrow[dim] = act ? x : 0.f;
}

template <int T, int SOLVER, int MODE>
__global__ __launch_bounds__(256) void ials_wg_solve_kernel(SolveParams p) {
  using G = WgGeo<T>;
  extern __shared__ __attribute__((aligned(16))) float wg_lds[];
  constexpr bool TILES = SOLVER == 1;
  float *A = wg_lds;
  float *bvec = A + (TILES ? G::NT * 256 : G::PACKED);
  float *diag = bvec + G::KP;
  float *rdiag = diag + G::KP;
  float *ybuf = rdiag + G::KP;
  float *red = TILES ? diag + G::KP : ybuf + G::KP;  // CG: tiles | b | p | red
  const int item = blockIdx.x;
  int row = 0, nnz = 0;
  bool solve = false;
  switch (threadIdx.x >> 6) {  // the tile ownership is a compile-time property of the wave
    case 0: solve = wg_accumulate<T, 0, MODE, TILES>(p, item, A, bvec, row, nnz); break;
    case 1: solve = wg_accumulate<T, 1, MODE, TILES>(p, item, A, bvec, row, nnz); break;
    case 2: solve = wg_accumulate<T, 2, MODE, TILES>(p, item, A, bvec, row, nnz); break;
    default: solve = wg_accumulate<T, 3, MODE, TILES>(p, item, A, bvec, row, nnz); break;
  }
  if (!solve) return;  // chunk of a split row: same decision in all four waves
  __syncthreads();
  float *xrow = p.target + static_cast<size_t>(row) * G::KP;
  if constexpr (SOLVER == 0)
    wg_cholesky<T>(A, bvec, diag, rdiag, ybuf, p.K, xrow, p.err_flag);
  else
    wg_cg_rows<T>(A, bvec, diag, red, p.K, nnz, p.max_cg_steps, xrow, p.err_flag);
}

// Gramian partials for T > 4: a block owns a slab of rows, its four waves own the tiles.
template <int T, int W>
__device__ __forceinline__ void gramian_wg_body(const float *__restrict__ F, int64_t b, int64_t e,
                                                float *__restrict__ dst_block) {
  using G = WgGeo<T>;
  const int lane = threadIdx.x & 63;
  const int g = lane >> 4, m = lane & 15;
  f32x4 acc[G::TPW];
#pragma unroll
  for (int s = 0; s < G::TPW; s++) acc[s] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int64_t r = b; r < e; r += 4) {
    float v[T];
    if (r + g < e) {
      load_dims<T>(F + (r + g) * G::KP + T * m, v);
    } else {
#pragma unroll
      for (int i = 0; i < T; i++) v[i] = 0.f;
    }
    mfma_tiles<T, G::NW, W>(v, v, acc, std::make_integer_sequence<int, G::TPW>{});
  }
  f32x4 *dst = reinterpret_cast<f32x4 *>(dst_block);
#pragma unroll
  for (int s = 0; s < G::TPW; s++) {
    const int t = G::NW * s + W;
    if (t < G::NT) dst[t * 64 + lane] = acc[s];
  }
}

template <int T>
__global__ __launch_bounds__(256) void gramian_partial_wg_kernel(const float *__restrict__ F,
                                                                 int64_t row_begin, int64_t row_end,
                                                                 int64_t rows_per_block,
                                                                 float *__restrict__ partial) {
  const int64_t b = row_begin + static_cast<int64_t>(blockIdx.x) * rows_per_block;
  const int64_t e = min(b + rows_per_block, row_end);
  float *dst = partial + static_cast<int64_t>(blockIdx.x) * (Geo<T>::NT * 256);
  switch (threadIdx.x >> 6) {
    case 0: gramian_wg_body<T, 0>(F, b, e, dst); break;
    case 1: gramian_wg_body<T, 1>(F, b, e, dst); break;
    case 2: gramian_wg_body<T, 2>(F, b, e, dst); break;
    default: gramian_wg_body<T, 3>(F, b, e, dst); break;
  }
}

}  // namespace ials
}  // namespace irs
