// Cholesky half-step for 64 < K <= 256 (KP = 128 / 192 / 256): one 256-thread workgroup per
// task, 16-row block steps.  (hpp:273-331: A = P + sum c v v^T + reg I, LLT, solve.)
//
// The upper 16x16 tiles are dealt round-robin to the four waves and never leave the MFMA
// accumulator registers; what the first version of this kernel (4-row panels, factor dumped
// to 136 KB of LDS, one workgroup per CU, 128 barriers per row) did with vector instructions
// is done here by the matrix cores:
//   per block row I (T = KP / 16 steps, two barriers each)
//   (1) the owner of the diagonal tile factorises it, S_II = R_II^T R_II, carrying the
//       identity along: E = R_II^-T comes out of the same row operations (4-row sub-panels
//       with v_readlane scalars, the rest of the tile by one rank-4 MFMA per sub-panel);
//   (2) TRSM on the matrix cores: every tile of block row I becomes R_IJ = E S_IJ (4 MFMAs,
//       the tile turned from accumulator into operand layout through a wave-private LDS
//       scratch), and goes to the LDS panel (16 rows x KP);
//   (3) trailing update S_I2,J2 -= R_I,I2^T R_I,J2: 4 MFMAs per tile, operands read from the
//       panel.  The right-hand side rides along as one more tile column (forward substitution).
//   Back substitution R x = y without a copy of R: x_I = R_II^-1 z_I = E^T z_I (the E tiles
//   stay in LDS, 1 KB each), and z_I -= R_IJ x_J is taken from the accumulator tiles with a
//   16-lane DPP reduction.  One barrier per block.
// LDS: 24 KB (K = 128) .. 43 KB (K = 256) instead of 138 KB, <= 256 registers: two to four
// workgroups per CU overlap each other's serial diagonal steps.
//
// Everything works in the virtual basis k = 16 I + m' <-> latent dim T m' + I (the basis the
// MFMA tiles are in), a symmetric permutation that changes neither the factorisation nor x.
#pragma once
#include "ials_chol16.hpp"
#include "ials_wg_kernels.hpp"

namespace irs {
namespace ials {

template <int T> struct WgChol {
  static constexpr int KP = 16 * T;
  static constexpr int PR = KP + 32;  // panel row stride: KP columns, rhs at column KP
  static constexpr int WS = 16 * 17;  // a 16 x 16 tile with row stride 17
  static constexpr int PAN = 0;
  static constexpr int WT = PAN + 16 * PR;      // T tiles E_I = R_II^-T
  static constexpr int SCR = WT + T * WS;       // 2 scratch tiles per wave
  static constexpr int ZB = SCR + 4 * 2 * WS;   // z (the right-hand side being reduced)
  static constexpr int XB = ZB + KP;            // x
  static constexpr int LDS_FLOATS = XB + KP + 16;
  // sub-steps (4 stored entries) of the gather kept in flight
  static constexpr int D = T >= 16 ? 2 : (T >= 12 ? 3 : 4);
};

// Gather + rank update of this wave's tiles: a ring of D gathered sub-steps (T registers
// each), indices / confidences one ring further ahead.  All loads are unconditional (the CSR
// arrays are padded); entries past the row's end are neutralised by c = 0.
template <int T, int W>
__device__ __forceinline__ void syrk_gather_ring(const float *__restrict__ other,
                                                 const int32_t *__restrict__ indices,
                                                 const float *__restrict__ data, int begin, int end,
                                                 float bias, f32x4 (&acc)[WgGeo<T>::TPW],
                                                 float (&bsum)[(T + 3) / 4]) {
  using G = WgGeo<T>;
  constexpr int KP = G::KP, D = WgChol<T>::D, NB = (T + 3) / 4;
  const int lane = threadIdx.x & 63;
  const int g = lane >> 4, m = lane & 15;
  const float *col_base = other + T * m;
  const int n = end - begin;
  const int nsub = (n + 3) >> 2;
  const int32_t *ip = indices + begin + g;
  const float *dp = data + begin + g;
  int ix[2 * D];
  float cx[2 * D];
  float v[D][T], vc[D], vw[D];
  auto load_idx = [&](int slot, int j) {  // over-reads stay inside the padded arrays
    ix[slot] = ip[4 * j];
    cx[slot] = dp[4 * j];
  };
  auto gather = [&](int k, int slot, int j) {
    const bool valid = 4 * j + g < n;
    vc[k] = valid ? cx[slot] : 0.f;
    vw[k] = valid ? bias + cx[slot] : 0.f;
    const unsigned idx = valid ? static_cast<unsigned>(ix[slot]) : 0u;
    load_dims<T>(col_base + static_cast<size_t>(idx) * KP, v[k]);
  };
  auto consume = [&](int k) {
    float cv[T], vk[T];
#pragma unroll
    for (int i = 0; i < T; i++) {
      vk[i] = v[k][i];
      cv[i] = vc[k] * vk[i];
      // the right-hand side b += (bias + c) v: wave W sums the dims T m + i with i % 4 == W
      if (i % 4 == W) bsum[i / 4] = fmaf(vw[k], vk[i], bsum[i / 4]);
    }
    mfma_tiles<T, G::NW, W>(cv, vk, acc, std::make_integer_sequence<int, G::TPW>{});
  };
#pragma unroll
  for (int k = 0; k < 2 * D; k++) load_idx(k, k);
#pragma unroll
  for (int k = 0; k < D; k++) gather(k, k, k);
  // round = 2 D sub-steps, so that ring and index slots are compile-time constants
  for (int j0 = 0; j0 < nsub; j0 += 2 * D) {
#pragma unroll
    for (int k = 0; k < 2 * D; k++) {
      // sub-step j0 + k: its gather was issued D sub-steps ago from index slot k
      consume(k % D);
      load_idx(k, j0 + k + 2 * D);                   // slot k is free again
      gather(k % D, (k + D) % (2 * D), j0 + k + D);  // (masked when past the end)
    }
  }
#pragma unroll
  for (int ii = 0; ii < NB; ii++) {
    bsum[ii] += __shfl_xor(bsum[ii], 16, 64);
    bsum[ii] += __shfl_xor(bsum[ii], 32, 64);
  }
}

// As wg_gather (ials_wg_kernels.hpp) with the ring gather.
template <int T, int W, int MODE>
__device__ __forceinline__ bool wg16_gather(const SolveParams &p, int item, f32x4 (&acc)[WgGeo<T>::TPW],
                                            float (&bsum)[(T + 3) / 4], int &row_out) {
  using G = WgGeo<T>;
  constexpr int NB = (T + 3) / 4;
  const int lane = threadIdx.x & 63;
  const int m = lane & 15;
#pragma unroll
  for (int ii = 0; ii < NB; ii++) bsum[ii] = 0.f;
  const f32x4 *Pacc = reinterpret_cast<const f32x4 *>(p.P_acc);
  constexpr int PARTIAL = Geo<T>::PARTIAL_FLOATS;
  if constexpr (MODE == 0) {
    const Task task = p.tasks[item];
    row_out = task.row;
#pragma unroll
    for (int s = 0; s < G::TPW; s++) acc[s] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (task.slot < 0) {
#pragma unroll
      for (int s = 0; s < G::TPW; s++) {
        const int t = G::NW * s + W;
        if (t < G::NT) acc[s] = Pacc[t * 64 + lane];
      }
    }
    syrk_gather_ring<T, W>(p.other, p.indices, p.data, task.begin, task.end, p.bias, acc, bsum);
    if (task.slot >= 0) {
      float *dst = p.partials + static_cast<size_t>(task.slot) * PARTIAL;
      f32x4 *d4 = reinterpret_cast<f32x4 *>(dst);
#pragma unroll
      for (int s = 0; s < G::TPW; s++) {
        const int t = G::NW * s + W;
        if (t < G::NT) d4[t * 64 + lane] = acc[s];
      }
      if (lane < 16) {
#pragma unroll
        for (int ii = 0; ii < NB; ii++)
          if (4 * ii + W < T) dst[G::NT * 256 + T * lane + 4 * ii + W] = bsum[ii];
      }
      return false;
    }
  } else {
    const SplitRow sr = p.split_rows[item];
    row_out = sr.row;
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
#pragma unroll
      for (int ii = 0; ii < NB; ii++)
        if (4 * ii + W < T) bsum[ii] += src[G::NT * 256 + T * m + 4 * ii + W];
    }
  }
  return true;
}

// sum over the 16 lanes of a group (every lane of the group ends with the total)
__device__ __forceinline__ float group16_sum(float v) {
  v += __shfl_xor(v, 1, 64);
  v += __shfl_xor(v, 2, 64);
  v += __shfl_xor(v, 4, 64);
  v += __shfl_xor(v, 8, 64);
  return v;
}

// (2) one tile of block row I: S -> R = E S, returned in accumulator layout.
//     a[s] = E[i = m][4 s + g]; scr: 16 x 17 floats, private to the wave.
__device__ __forceinline__ f32x4 trsm_tile16(const f32x4 &S, const float (&a)[4], float *scr) {
  const int lane = threadIdx.x & 63;
  const int g = lane >> 4, m = lane & 15;
#pragma unroll
  for (int r = 0; r < 4; r++) scr[(4 * g + r) * 17 + m] = S[r];
  __threadfence_block();
  float b[4];
#pragma unroll
  for (int s = 0; s < 4; s++) b[s] = scr[(4 * s + g) * 17 + m];
  __threadfence_block();
  f32x4 D = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int s = 0; s < 4; s++) D = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], b[s], D, 0, 0, 0);
  return D;
}

template <int T, int W>
__device__ __forceinline__ void wg_cholesky16(f32x4 (&acc)[WgGeo<T>::TPW],
                                              const float (&bsum)[(T + 3) / 4], float *lds, int K, float *xrow, int32_t *err_flag) {
  using G = WgGeo<T>;
  using C = WgChol<T>;
  constexpr int KP = G::KP, PR = C::PR, WS = C::WS;
  constexpr TileTab<T> tab{};
  const int tid = threadIdx.x, lane = tid & 63;
  const int g = lane >> 4, m = lane & 15;
  float *pan = lds + C::PAN, *wt = lds + C::WT, *zb = lds + C::ZB, *xb = lds + C::XB;
  float *scr0 = lds + C::SCR + W * 2 * WS, *scr1 = scr0 + WS;
  // rhs as one more tile column in accumulator layout (replicated over the column index); its
  // T tiles are dealt to the waves like the others: tile row i belongs to wave i % 4
  constexpr int NB = (T + 3) / 4;
  f32x4 bacc[NB];
  if (lane < 16) {  // (every lane group holds the same sums)
#pragma unroll
    for (int ii = 0; ii < NB; ii++)
      if (4 * ii + W < T) zb[16 * (4 * ii + W) + lane] = bsum[ii];  // virtual index 16 i + m'
  }
  __syncthreads();
#pragma unroll
  for (int ii = 0; ii < NB; ii++) {
    const int i = 4 * ii + W;
#pragma unroll
    for (int r = 0; r < 4; r++) bacc[ii][r] = i < T ? -zb[16 * i + 4 * g + r] : 0.f;
  }
  // Sign convention: the tiles not yet reached by the factorisation hold MINUS the Schur
  // complement, so that the trailing update is a plain accumulating MFMA (+ R^T R) and no
  // operand has to be negated per step; the sign is flipped back where a tile is consumed
  // (diagonal tile, TRSM through -E).  A tile of block row I holds R_IJ itself after step I.
#pragma unroll
  for (int s = 0; s < G::TPW; s++) acc[s] = -acc[s];
  bool bad = false;
  for (int I = 0; I < T; I++) {
    // ---- (1) diagonal tile (I, I): owner wave only
    const int t_diag = I * T - I * (I - 1) / 2;
    if ((t_diag & 3) == W) {
      f32x4 Cd = {0.f, 0.f, 0.f, 0.f}, E = identity_tile16();
#pragma unroll
      for (int s = 0; s < G::TPW; s++) {
        const int t = G::NW * s + W;
        if (t < G::NT) {
          if (tab.ti[t] == tab.tj[t]) {  // compile time: only the diagonal tiles of this wave
            if (tab.ti[t] == I) Cd = -acc[s];
          }
        }
      }
      diag_factor16<false>(Cd, E, scr0, scr1, bad);
      float *dst = wt + I * WS;
#pragma unroll
      for (int r = 0; r < 4; r++) dst[(4 * g + r) * 17 + m] = E[r];
    }
    __syncthreads();
    // ---- (2) TRSM of this wave's tiles of block row I (and of the rhs)
    {
      float a[4];
#pragma unroll
      for (int s = 0; s < 4; s++) a[s] = -wt[I * WS + m * 17 + 4 * s + g];  // R = (-E)(-S)
#pragma unroll
      for (int s = 0; s < G::TPW; s++) {
        const int t = G::NW * s + W;
        if (t < G::NT) {
          if (tab.ti[t] != tab.tj[t]) {
            if (tab.ti[t] == I) {
              acc[s] = trsm_tile16(acc[s], a, scr0);
#pragma unroll
              for (int r = 0; r < 4; r++) pan[(4 * g + r) * PR + 16 * tab.tj[t] + m] = acc[s][r];
            }
          }
        }
      }
#pragma unroll
      for (int ii = 0; ii < NB; ii++)
        if (4 * ii + W == I) {
          bacc[ii] = trsm_tile16(bacc[ii], a, scr1);
          if (m == 0) {
#pragma unroll
            for (int r = 0; r < 4; r++) pan[(4 * g + r) * PR + KP] = bacc[ii][r];
          }
        }
    }
    if (I == T - 1) break;
    __syncthreads();
    // ---- (3) trailing update of every tile below block row I: 4 MFMAs per tile
#pragma unroll
    for (int s4 = 0; s4 < 4; s4++) {
      float op[T];
#pragma unroll
      for (int j = 1; j < T; j++)
        op[j] = pan[(4 * s4 + g) * PR + 16 * j + m];  // (columns j <= I: stale, never used)
      op[0] = 0.f;
      const float opb = pan[(4 * s4 + g) * PR + KP];
#pragma unroll
      for (int s = 0; s < G::TPW; s++) {
        const int t = G::NW * s + W;
        if (t < G::NT) {
          if (tab.ti[t] > 0) {
            if (tab.ti[t] > I)
              acc[s] = __builtin_amdgcn_mfma_f32_16x16x4f32(op[tab.ti[t]], op[tab.tj[t]], acc[s], 0, 0, 0);
          }
        }
      }
#pragma unroll
      for (int ii = 0; ii < NB; ii++) {
        if (4 * ii + W < T && 4 * ii + W > 0) {
          if (4 * ii + W > I)
            bacc[ii] = __builtin_amdgcn_mfma_f32_16x16x4f32(op[(4 * ii + W) % T], opb, bacc[ii], 0, 0, 0);
        }
      }
    }
    // the next (1) writes wt[I + 1] (read after the next barrier only) and scratch private to
    // the wave; the next (2) writes pan after a barrier every wave reaches only when it is
    // done with the reads above
  }
  if (bad) {
    if (lane == 0) atomicOr(err_flag, 1);
  }
  // ---- back substitution R x = y.  z starts as y, block T - 1 first.
  if (m == 0) {
#pragma unroll
    for (int ii = 0; ii < NB; ii++)
      if (4 * ii + W < T) {
#pragma unroll
        for (int r = 0; r < 4; r++) zb[16 * (4 * ii + W) + 4 * g + r] = bacc[ii][r];
      }
  }
  __syncthreads();
  for (int J = T - 1; J >= 0; J--) {
    // every wave forms x_J = E_J^T z_J itself: x_J[n] = sum_k E_J[k][n] z_J[k]
    float xj = 0.f;
    {
      const float *e = wt + J * WS + m;
      const float *z = zb + 16 * J;
#pragma unroll
      for (int k = 0; k < 16; k++) xj = fmaf(e[k * 17], z[k], xj);
    }
    if (W == (J & 3) && lane < 16) xb[16 * J + lane] = xj;
    if (J == 0) break;
    // z_I -= R_IJ x_J for this wave's tiles of block column J
#pragma unroll
    for (int s = 0; s < G::TPW; s++) {
      const int t = G::NW * s + W;
      if (t < G::NT) {
        if (tab.ti[t] != tab.tj[t]) {
          if (tab.tj[t] == J) {
            float part[4];
#pragma unroll
            for (int r = 0; r < 4; r++) part[r] = group16_sum(acc[s][r] * xj);
            if (m == 0) {
#pragma unroll
              for (int r = 0; r < 4; r++) zb[16 * tab.ti[t] + 4 * g + r] -= part[r];
            }
          }
        }
      }
    }
    __syncthreads();
  }
  __syncthreads();
  // virtual index k = 16 I + m'  <->  latent dim T m' + I
  bool fin = true;
  if (tid < KP) {
    const int dim = T * (tid & 15) + (tid >> 4);
    const float xv = xb[tid];
    fin = __builtin_isfinite(xv) || dim >= K;
    xrow[dim] = dim < K ? xv : 0.f;
  }
  if (!__all(fin)) {
    if (lane == 0) atomicOr(err_flag, 2);
  }
}

template <int T, int W, int MODE>
__device__ __forceinline__ void wg16_row(const SolveParams &p, int item, float *lds) {
  using G = WgGeo<T>;
  f32x4 acc[G::TPW];
  float bsum[(T + 3) / 4];  // rhs, dims T m + i with i % 4 == W
  int row = 0;
  if (!wg16_gather<T, W, MODE>(p, item, acc, bsum, row)) return;
  wg_add_reg<T, W>(acc, p.reg[row], p.K);
  if (p.prior != nullptr) {  // rhs += reg_r * prior_r (hpp:363)
    const float reg = p.reg[row];
    const float *pr = p.prior + static_cast<size_t>(row) * G::KP + T * (threadIdx.x & 15);
#pragma unroll
    for (int ii = 0; ii < (T + 3) / 4; ii++)
      if (4 * ii + W < T) bsum[ii] = fmaf(reg, pr[4 * ii + W], bsum[ii]);
  }
  wg_cholesky16<T, W>(acc, bsum, lds, p.K, p.target + static_cast<size_t>(row) * G::KP, p.err_flag);
}

template <int T, int MODE>
__global__ __launch_bounds__(256, T <= 8 ? 3 : 2) void ials_wg16_cholesky_kernel(SolveParams p) {
  extern __shared__ __attribute__((aligned(16))) float wg16_lds[];
  const int item = blockIdx.x;
  switch (threadIdx.x >> 6) {  // the tile ownership is a compile-time property of the wave
    case 0: wg16_row<T, 0, MODE>(p, item, wg16_lds); break;
    case 1: wg16_row<T, 1, MODE>(p, item, wg16_lds); break;
    case 2: wg16_row<T, 2, MODE>(p, item, wg16_lds); break;
    default: wg16_row<T, 3, MODE>(p, item, wg16_lds); break;
  }
}

}  // namespace ials
}  // namespace irs
