// iALS solve for 128 < K <= 256 (KP = 192 / 256): one 256-thread workgroup per task.
// (K <= 128 runs on the one-wave-per-task kernels of ials_kernels.hpp.)  Same algorithm
// (hpp:273-331 Cholesky, hpp:170-271 CG), different mapping: the KP x KP Gramian no longer
// fits one wave's registers, so
//   * the upper 16x16 tiles are dealt round-robin to the four waves, which all walk the
//     same row with a gather pipeline (MFMA rank update on the wave's own tiles);
//   * Cholesky: the tiles stay in the accumulator registers.  Per 4-row panel the wave that
//     owns the diagonal tile factorises the 4x4 block and publishes its 10 scalars, every
//     wave replays the row operations on its tiles of that tile row, the panel rows go
//     through LDS and come back as MFMA operands for the trailing rank-4 update of every
//     tile (2 barriers per panel).  R is then dumped as 16x16 blocks and R x = y is solved
//     by blocks of 16 (a 16-lane triangular solve + a block mat-vec, 2 barriers per block);
//   * CG: the tiles are dumped as 16x16 blocks and every thread reads one matrix row back
//     into registers through the symmetry; a mat-vec is KP FMAs against the broadcast
//     vector in LDS.
// Everything after the rank update works in the virtual basis k = 16 I + m' <-> latent dim
// T m' + I (the basis the MFMA tiles are in), a permutation that changes neither the
// factorisation's result nor CG's iterates.
#pragma once
#include "ials_kernels.hpp"

namespace irs {
namespace ials {

template <int T> struct WgGeo {
  static constexpr int KP = 16 * T;
  static constexpr int NT = T * (T + 1) / 2;
  static constexpr int NW = 4;
  static constexpr int TPW = (NT + NW - 1) / NW;
  static constexpr int PR = KP + 16;  // panel row stride: KP columns + the rhs column
  // tiles as 16 x 16 blocks (the panel rows of the factorisation overlay their head: R is
  // dumped only after the last panel) | b / y | p / x | 4x4 scalars + reduction scratch
  static constexpr int LDS_FLOATS = NT * 256 + 2 * KP + 32;
  static_assert(4 * PR <= NT * 256, "panel rows must fit the tile area");
};

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
// purpose (the per-wave MFMA work of a sub-step is 20 .. 34 instructions here, so a
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

// Rank update of this wave's tiles.  Returns false after storing the partial of a chunk of
// a split row (MODE 0); otherwise the wave's tiles (+ P) and, in wave 0, the rhs are left
// in acc / bsum.
template <int T, int W, int MODE>
__device__ __forceinline__ bool wg_gather(const SolveParams &p, int item, f32x4 (&acc)[WgGeo<T>::TPW],
                                          float (&bsum)[T], int &row_out, int &nnz_out) {
  using G = WgGeo<T>;
  const int lane = threadIdx.x & 63;
  const int m = lane & 15;
#pragma unroll
  for (int i = 0; i < T; i++) bsum[i] = 0.f;
  const f32x4 *Pacc = reinterpret_cast<const f32x4 *>(p.P_acc);
  constexpr int PARTIAL = Geo<T>::PARTIAL_FLOATS;
  if constexpr (MODE == 0) {
    const Task task = p.tasks[item];
    row_out = task.row;
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
    row_out = sr.row;
    nnz_out = sr.nnz;
#pragma unroll
    for (int s = 0; s < G::TPW; s++) {
      const int t = G::NW * s + W;
      acc[s] = t < G::NT ? Pacc[t * 64 + lane] : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    for (int sl = 0; sl < sr.n_slots; sl++) {
      const float *src = p.partials + static_cast<size_t>(sr.first_slot + sl * sr.slot_stride) * PARTIAL;
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
  return true;
}

// + reg on the diagonal (hpp:312-314); padded dims get a unit diagonal so that they decouple
template <int T, int W>
__device__ __forceinline__ void wg_add_reg(f32x4 (&acc)[WgGeo<T>::TPW], float reg, int K) {
  using G = WgGeo<T>;
  constexpr TileTab<T> tab{};
  const int lane = threadIdx.x & 63;
  const int g = lane >> 4, m = lane & 15;
#pragma unroll
  for (int s = 0; s < G::TPW; s++) {
    const int t = G::NW * s + W;
    if (t < G::NT) {
      if (tab.ti[t] == tab.tj[t]) {
#pragma unroll
        for (int r = 0; r < 4; r++)
          if (4 * g + r == m) acc[s][r] += (T * m + tab.ti[t] < K) ? reg : 1.0f;
      }
    }
  }
}

// tile t, element (row 4g + r, column m) -> tiles[t * 256 + row * 16 + column]
template <int T, int W>
__device__ __forceinline__ void wg_dump_tiles(const f32x4 (&acc)[WgGeo<T>::TPW], float *tiles) {
  using G = WgGeo<T>;
  const int lane = threadIdx.x & 63;
  const int g = lane >> 4, m = lane & 15;
#pragma unroll
  for (int s = 0; s < G::TPW; s++) {
    const int t = G::NW * s + W;
    if (t < G::NT) {
#pragma unroll
      for (int r = 0; r < 4; r++) tiles[t * 256 + (4 * g + r) * 16 + m] = acc[s][r];
    }
  }
}

// Cholesky M = R^T R on the tiles in registers + both substitutions (hpp:316-324).
// LDS: tiles | ybuf | xbuf | pan | scal.
template <int T, int W>
__device__ __forceinline__ void wg_cholesky_tiles(f32x4 (&acc)[WgGeo<T>::TPW], const float (&bsum)[T],
                                                  float *tiles, float *ybuf, float *xbuf, float *pan,
                                                  float *scal, int K, float *xrow,
                                                  int32_t *err_flag) {
  using G = WgGeo<T>;
  constexpr int KP = G::KP, PR = G::PR;
  constexpr TileTab<T> tab{};
  const int tid = threadIdx.x, lane = tid & 63;
  const int g = lane >> 4, m = lane & 15;
  // rhs as an extra column in accumulator layout (wave 0): row 4g + r of tile row i
  f32x4 bacc[W == 0 ? T : 1];
  if constexpr (W == 0) {
#pragma unroll
    for (int i = 0; i < T; i++)
#pragma unroll
      for (int r = 0; r < 4; r++) bacc[i][r] = __shfl(bsum[i], 20 * g + r, 64);
  }
  bool bad = false;
  for (int I = 0; I < T; I++) {
    for (int gq = 0; gq < 4; gq++) {
      const bool mine = g == gq;
      // ---- (A) the owner of the diagonal tile factorises the 4 x 4 block
#pragma unroll
      for (int s = 0; s < G::TPW; s++) {
        const int t = G::NW * s + W;
        if (t < G::NT) {
          if (tab.ti[t] == tab.tj[t] && tab.ti[t] == I) {  // uniform
            float sc[10];
            int n_s = 4;
#pragma unroll
            for (int r = 0; r < 4; r++) {
              const float piv = readlane_f(acc[s][r], 20 * gq + r);
              bad |= !(piv > 0.f);
              const float rinv = __builtin_amdgcn_rsqf(piv);
              sc[r] = rinv;
              acc[s][r] *= mine ? rinv : 1.0f;
#pragma unroll
              for (int r2 = r + 1; r2 < 4; r2++) {
                const float sv = readlane_f(acc[s][r], 20 * gq + r2);  // R[k][k2]
                sc[n_s++] = sv;
                acc[s][r2] = fmaf(-(mine ? sv : 0.f), acc[s][r], acc[s][r2]);
              }
            }
            if (lane == 0) {
#pragma unroll
              for (int q = 0; q < 10; q++) scal[q] = sc[q];
            }
          }
        }
      }
      __syncthreads();
      // ---- (B) replay the row operations on the other tiles of tile row I and on the rhs
      float sc[10];
#pragma unroll
      for (int q = 0; q < 10; q++) sc[q] = scal[q];
      auto replay = [&](f32x4 &v) {
        int n_s = 4;
#pragma unroll
        for (int r = 0; r < 4; r++) {
          v[r] *= mine ? sc[r] : 1.0f;
#pragma unroll
          for (int r2 = r + 1; r2 < 4; r2++) {
            v[r2] = fmaf(-(mine ? sc[n_s] : 0.f), v[r], v[r2]);
            n_s++;
          }
        }
      };
#pragma unroll
      for (int s = 0; s < G::TPW; s++) {
        const int t = G::NW * s + W;
        if (t < G::NT) {
          if (tab.ti[t] == I && tab.tj[t] > I) replay(acc[s]);
        }
      }
      if constexpr (W == 0) {
#pragma unroll
        for (int i = 0; i < T; i++)
          if (i == I) replay(bacc[i]);
      }
      if (I == T - 1 && gq == 3) break;  // nothing below the last panel
      // ---- (C) panel rows -> LDS
      if (mine) {
#pragma unroll
        for (int s = 0; s < G::TPW; s++) {
          const int t = G::NW * s + W;
          if (t < G::NT) {
            if (tab.ti[t] == I) {
#pragma unroll
              for (int r = 0; r < 4; r++) pan[r * PR + 16 * tab.tj[t] + m] = acc[s][r];
            }
          }
        }
        if constexpr (W == 0) {
          if (m == 0) {
#pragma unroll
            for (int i = 0; i < T; i++)
              if (i == I) {
#pragma unroll
                for (int r = 0; r < 4; r++) pan[r * PR + KP] = bacc[i][r];
              }
          }
        }
      }
      __syncthreads();
      // ---- (D) trailing rank-4 update: one MFMA per tile at or below tile row I
      float op[T];
#pragma unroll
      for (int j = 0; j < T; j++) op[j] = pan[g * PR + 16 * j + m];
      const float opb = pan[g * PR + KP];
      const bool below = m > 4 * gq + 3;  // rows up to the panel are final
#pragma unroll
      for (int s = 0; s < G::TPW; s++) {
        const int t = G::NW * s + W;
        if (t < G::NT) {
          if (tab.ti[t] >= I) {
            float a = op[tab.ti[t]];
            if (tab.ti[t] == I) a = below ? a : 0.f;
            acc[s] = __builtin_amdgcn_mfma_f32_16x16x4f32(-a, op[tab.tj[t]], acc[s], 0, 0, 0);
          }
        }
      }
      if constexpr (W == 0) {
#pragma unroll
        for (int i = 0; i < T; i++)
          if (i >= I) {
            float a = op[i];
            if (i == I) a = below ? a : 0.f;
            bacc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(-a, opb, bacc[i], 0, 0, 0);
          }
      }
      // the next (A) writes scal and the next (C) writes pan: both after a barrier that
      // every wave reaches only when it is done with the reads above
    }
  }
  if (bad) {
    if (lane == 0) atomicOr(err_flag, 1);
  }
  // ---- dump R (16 x 16 blocks) and y, then R x = y by blocks of 16 rows, last block first
  wg_dump_tiles<T, W>(acc, tiles);
  if constexpr (W == 0) {
    if (m == 0) {
#pragma unroll
      for (int i = 0; i < T; i++)
#pragma unroll
        for (int r = 0; r < 4; r++) ybuf[16 * i + 4 * g + r] = bacc[i][r];
    }
  }
  __syncthreads();
  auto tile_of = [&](int I, int J) { return I * T - I * (I - 1) / 2 + (J - I); };
  for (int Ib = T - 1; Ib >= 0; Ib--) {
    if (tid < 64) {  // wave 0: lane c < 16 owns row c of the diagonal block
      const int rk = lane & 15;
      const float *dt = tiles + tile_of(Ib, Ib) * 256 + rk * 16;
      float rr[16];
#pragma unroll
      for (int c = 0; c < 4; c++) {
        const f32x4 v = *reinterpret_cast<const f32x4 *>(dt + 4 * c);
        rr[4 * c] = v.x; rr[4 * c + 1] = v.y; rr[4 * c + 2] = v.z; rr[4 * c + 3] = v.w;
      }
      const float rinv = 1.0f / dt[rk];
      const float yv = ybuf[16 * Ib + rk];
      float partial = 0.f, xv = 0.f;
#pragma unroll
      for (int c = 15; c >= 0; c--) {  // what a row accumulates after its own step is never read
        const float xc = readlane_f((yv - partial) * rinv, c);
        if (lane == c) xv = xc;
        partial = fmaf(rr[c], xc, partial);
      }
      if (lane < 16) xbuf[16 * Ib + lane] = xv;
    }
    __syncthreads();
    if (tid < 16 * Ib) {  // rows above the block: y -= R[:, block] x[block]
      const int Ik = tid >> 4, rk = tid & 15;
      const float *src = tiles + tile_of(Ik, Ib) * 256 + rk * 16;
      float s0 = 0.f, s1 = 0.f;
#pragma unroll
      for (int c = 0; c < 4; c++) {
        const f32x4 v = *reinterpret_cast<const f32x4 *>(src + 4 * c);
        const f32x4 x4 = *reinterpret_cast<const f32x4 *>(xbuf + 16 * Ib + 4 * c);
        s0 = fmaf(v.x, x4.x, s0);
        s1 = fmaf(v.y, x4.y, s1);
        s0 = fmaf(v.z, x4.z, s0);
        s1 = fmaf(v.w, x4.w, s1);
      }
      ybuf[tid] -= s0 + s1;
    }
    __syncthreads();
  }
  // virtual index k = 16 I + m'  <->  latent dim T m' + I
  bool fin = true;
  if (tid < KP) {
    const int dim = T * (tid & 15) + (tid >> 4);
    const float xv = xbuf[tid];
    fin = __builtin_isfinite(xv) || dim >= K;
    xrow[dim] = dim < K ? xv : 0.f;
  }
  if (!__all(fin)) {
    if (lane == 0) atomicOr(err_flag, 2);
  }
}

// Conjugate gradient (hpp:199-264) with one matrix row per thread in registers.  Thread k
// reads its row back from the dumped tiles through the symmetry: tiles right of the
// diagonal row-wise, tiles left of it column-wise.  A mat-vec is then KP FMAs against the
// broadcast vector in LDS.
template <int T>
__device__ __forceinline__ void wg_cg_rows(const float *A, const float *bvec, float *pbuf, float *red,
                                           float reg, int K, int nnz, int max_cg_steps, float *xrow,
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
    // + reg vec last: the regulariser is kept out of the matrix (see solve_row<T, 1>)
    return fmaf(reg, mine, s0 + s1);
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

// One wave's share of a task: gather, then the solve together with the other three waves.
template <int T, int W, int SOLVER, int MODE>
__device__ __forceinline__ void wg_row(const SolveParams &p, int item, float *lds) {
  using G = WgGeo<T>;
  float *tiles = lds;
  float *ybuf = tiles + G::NT * 256;  // b (CG) / y (Cholesky)
  float *xbuf = ybuf + G::KP;         // p (CG) / x (Cholesky)
  float *pan = tiles;                 // overlay, see WgGeo
  float *scal = xbuf + G::KP;         // 10 scalars of the 4 x 4 factor; red[4] at +16
  f32x4 acc[G::TPW];
  float bsum[T];
  int row = 0, nnz = 0;
  // chunk of a split row: the same decision in all four waves, nobody reaches a barrier
  if (!wg_gather<T, W, MODE>(p, item, acc, bsum, row, nnz)) return;
  float *xrow = p.target + static_cast<size_t>(row) * G::KP;
  if constexpr (SOLVER == 0) wg_add_reg<T, W>(acc, p.reg[row], p.K);
  if constexpr (W == 0) add_prior<T>(p, row, bsum);
  if (p.prior) nnz = max(nnz, 1);  // with a prior an empty row is solved like any other
  if constexpr (SOLVER == 0) {
    wg_cholesky_tiles<T, W>(acc, bsum, tiles, ybuf, xbuf, pan, scal, p.K, xrow, p.err_flag);
  } else {
    wg_dump_tiles<T, W>(acc, tiles);
    if (W == 0 && (threadIdx.x & 63) < 16) {
#pragma unroll
      for (int i = 0; i < T; i++) ybuf[16 * i + (threadIdx.x & 15)] = bsum[i];
    }
    __syncthreads();
    wg_cg_rows<T>(tiles, ybuf, xbuf, scal + 16, p.reg[row], p.K, nnz, p.max_cg_steps, xrow, p.err_flag);
  }
}

template <int T, int SOLVER, int MODE>
__global__ __launch_bounds__(256, T <= 12 ? 2 : 1) void ials_wg_solve_kernel(SolveParams p) {
  extern __shared__ __attribute__((aligned(16))) float wg_lds[];
  const int item = blockIdx.x;
  switch (threadIdx.x >> 6) {  // the tile ownership is a compile-time property of the wave
    case 0: wg_row<T, 0, SOLVER, MODE>(p, item, wg_lds); break;
    case 1: wg_row<T, 1, SOLVER, MODE>(p, item, wg_lds); break;
    case 2: wg_row<T, 2, SOLVER, MODE>(p, item, wg_lds); break;
    default: wg_row<T, 3, SOLVER, MODE>(p, item, wg_lds); break;
  }
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
