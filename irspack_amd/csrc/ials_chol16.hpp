// Per-row Cholesky solve by ONE wave with 16-row block steps on the matrix cores
// (Eigen::LLT + solve, hpp:310-324), K <= 128.  Replaces the 4-row-panel version
// (solve_row_cholesky, ials_kernels.hpp: ~2.0 k vector instructions per 64 x 64 system):
// here a system costs ~0.45 k vector instructions, the rest are MFMAs and LDS moves.
//
// The tiles are held in LOWER form: storage slot tix(a, b), a <= b, is the tile
// (row block b, column block a), lane (g, m) register r = M[16 b + 4 g + r][16 a + m]
// (virtual basis k = 16 I + m' <-> latent dim T m' + I).  M = L L^T is factorised left to
// right, one block column I per step, all indices compile-time:
//   (1) diagonal tile: S_II = R^T R with R = L_II^T, and E = R^-T = L_II^-1 from the same row
//       operations (diag_factor16: v_readlane scalars on 4-row sub-panels, one rank-4 MFMA
//       per sub-panel for the rest of the tile);
//   (2) TRSM on the matrix cores: L_JI = S_JI E^T (4 MFMAs per tile; the tile goes from
//       accumulator to operand layout through a 1 KB LDS scratch) and into the LDS panel;
//   (3) trailing update S_J,J2 -= L_JI L_J2,I^T: 4 MFMAs per tile, operands from the panel.
//   The right-hand side rides along as one more tile ROW (y = L^-1 b for free).
//   Back substitution L^T x = y: x_I = E_I^T z_I, z_I -= L_JI^T x_J; with lower tiles the
//   contraction runs over the accumulator's row index (registers + lane groups: 4 FMAs and
//   two cross-group adds per tile), and z stays in registers.
// LDS per wave: 10 KB (K = 64), 18.7 KB (K = 128; the 4-row-panel version spilled 46 KB).
#pragma once
#include "ials_kernels.hpp"

namespace irs {
namespace ials {

template <int T> struct Chol16Geo {
  static constexpr int KP = 16 * T;
  static constexpr int WS = 16 * 17;                 // a 16 x 16 tile, row stride 17
  static constexpr int PAN = 0;                      // KP rows of block column I + 16 rhs rows
  static constexpr int WT = PAN + (KP + 16) * 17;    // T tiles E_I
  static constexpr int SCR = PAN;                    // sub-panel rows of the diagonal step (R, E):
                                                     // they overlay the panel, dead at that point
  static constexpr int ZX = WT + T * WS;             // z_J (16) and x_J (16), 16 B aligned
  static constexpr int LDS_FLOATS = ZX + 32;
  static_assert(WT % 4 == 0 && ZX % 4 == 0, "b128 reads of zx need 16 B alignment");
  static constexpr int tix(int i, int j) { return i * T - i * (i - 1) / 2 + (j - i); }
};

// S = R^T R of one 16 x 16 tile in accumulator layout, E = R^-T alongside.  `Cd` is
// consumed; its strictly lower triangle only ever holds rounding noise and is never read as
// a result.  scrR / scrE: 16 x 17 floats each, private to the wave.
__device__ __forceinline__ void diag_factor16(f32x4 &Cd, f32x4 &E, float *scrR, float *scrE,
                                              bool &bad) {
  const int lane = threadIdx.x & 63;
  const int g = lane >> 4, m = lane & 15;
#pragma unroll
  for (int r = 0; r < 4; r++) E[r] = (4 * g + r == m) ? 1.0f : 0.0f;
#pragma unroll
  for (int q = 0; q < 4; q++) {
#ifdef IRS_CHOL16_SELECT
    const bool mine = g == q;
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const float piv = readlane_f(Cd[r], 20 * q + r);
      bad |= !(piv > 0.f);
      const float rinv = __builtin_amdgcn_rsqf(piv);
      const float mult = mine ? rinv : 1.0f;
      Cd[r] *= mult;
      E[r] *= mult;
#pragma unroll
      for (int r2 = r + 1; r2 < 4; r2++) {
        const float sv = readlane_f(Cd[r], 20 * q + r2);  // R[k][k2]
        const float sm_ = mine ? sv : 0.f;
        Cd[r2] = fmaf(-sm_, Cd[r], Cd[r2]);
        E[r2] = fmaf(-sm_, E[r], E[r2]);
      }
    }
#else
    // The 4 x 4 elimination of the sub-panel touches the lanes of group q only: run it under that
    // execution mask (one s_and_saveexec per sub-panel) instead of selecting a neutral operand per
    // operation - the scalars come from v_readlane, which ignores the mask: 3 instead of 5 vector
    // instructions per eliminated pair, 4 instead of 5 per pivot (round 4: ~250 of the ~1100 vector
    // instructions of a 64 x 64 factorisation).
    if (g == q) {
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const float piv = readlane_f(Cd[r], 20 * q + r);
        bad |= !(piv > 0.f);
        const float rinv = __builtin_amdgcn_rsqf(piv);
        Cd[r] *= rinv;
        E[r] *= rinv;
#pragma unroll
        for (int r2 = r + 1; r2 < 4; r2++) {
          const float sv = readlane_f(Cd[r], 20 * q + r2);  // R[k][k2]
          Cd[r2] = fmaf(-sv, Cd[r], Cd[r2]);
          E[r2] = fmaf(-sv, E[r], E[r2]);
        }
      }
    }
#endif
    if (q == 3) break;
    // rows 4q .. 4q+3 are final: rank-4 update of the rows below them (and of E).  Every group
    // stores its four rows (no divergent branch: the block stays one scheduling region), the
    // read picks the sub-panel's.
#pragma unroll
    for (int r = 0; r < 4; r++) {
      scrR[(4 * g + r) * 17 + m] = Cd[r];
      scrE[(4 * g + r) * 17 + m] = E[r];
    }
    __threadfence_block();
    const float a = scrR[(4 * q + g) * 17 + m];
    const float e = scrE[(4 * q + g) * 17 + m];
    __threadfence_block();
    const float na = (m > 4 * q + 3) ? -a : 0.f;  // rows up to the sub-panel are final
    Cd = __builtin_amdgcn_mfma_f32_16x16x4f32(na, a, Cd, 0, 0, 0);
    E = __builtin_amdgcn_mfma_f32_16x16x4f32(na, e, E, 0, 0, 0);
  }
}

// acc: lower-form tiles of A = P + sum c v v^T (no regulariser yet); b4[i] in lane (g, m) =
// right-hand side at virtual index 16 i + m.
template <int T>
__device__ __forceinline__ void solve_row_cholesky16(f32x4 (&acc)[Geo<T>::NT], const float (&b4)[T],
                                                     float reg, float *sm, float *xrow, int K,
                                                     int32_t *err_flag) {
  using C = Chol16Geo<T>;
  constexpr int KP = C::KP, WS = C::WS;
  const int lane = threadIdx.x & 63;
  const int g = lane >> 4, m = lane & 15;
  float *pan = sm + C::PAN, *wt = sm + C::WT, *scr = sm + C::SCR, *zx = sm + C::ZX;

  // diagonal: + reg (hpp:312-314); padded dims get 1 so that they decouple
#pragma unroll
  for (int i = 0; i < T; i++)
#pragma unroll
    for (int r = 0; r < 4; r++)
      if (4 * g + r == m) acc[C::tix(i, i)][r] += (T * m + i < K) ? reg : 1.0f;

  // right-hand side as tile row T: every row of tile (T, I) is b_I^T
  f32x4 bacc[T];
#pragma unroll
  for (int i = 0; i < T; i++) bacc[i] = f32x4{b4[i], b4[i], b4[i], b4[i]};

  bool bad = false;
#pragma unroll
  for (int I = 0; I < T; I++) {
    float *wtI = wt + I * WS;
    // ---- (1) diagonal tile
    {
      f32x4 Cd = acc[C::tix(I, I)], E;
      diag_factor16(Cd, E, scr, scr + WS, bad);
#pragma unroll
      for (int r = 0; r < 4; r++) wtI[(4 * g + r) * 17 + m] = E[r];
    }
    __threadfence_block();
    // ---- (2) TRSM: L_JI = S_JI E^T, y_I = b_I E^T.  B[k][n] = E[n][k]
    float bE[4];
#pragma unroll
    for (int s = 0; s < 4; s++) bE[s] = wtI[m * 17 + 4 * s + g];
    // All tiles of the block column (and the rhs row, kept as 16 identical rows at panel rows
    // KP ..) are staged in the panel buffer at once - it has exactly the layout the transposed
    // operand read needs - so the step pays ONE LDS round trip and the tiles' MFMA chains
    // interleave.
#pragma unroll
    for (int J = I + 1; J < T; J++)
#pragma unroll
      for (int r = 0; r < 4; r++) pan[(16 * J + 4 * g + r) * 17 + m] = acc[C::tix(I, J)][r];
#pragma unroll
    for (int r = 0; r < 4; r++) pan[(KP + 4 * g + r) * 17 + m] = bacc[I][r];
    __threadfence_block();
    // (chunks of four tiles bound the operand registers; within a chunk the tiles' MFMA
    // chains are interleaved)
#pragma unroll
    for (int J0 = I + 1; J0 <= T; J0 += 4) {
      float aS[4][4];  // A[i][k] = S[i][k]: lane (g, i) reads row i, column 4 s + g
      f32x4 D[4];
#pragma unroll
      for (int c = 0; c < 4; c++) {
        D[c] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (J0 + c <= T) {
#pragma unroll
          for (int s = 0; s < 4; s++) aS[c][s] = pan[(16 * (J0 + c) + m) * 17 + 4 * s + g];
        }
      }
#pragma unroll
      for (int s = 0; s < 4; s++)
#pragma unroll
        for (int c = 0; c < 4; c++)
          if (J0 + c <= T) D[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(aS[c][s], bE[s], D[c], 0, 0, 0);
#pragma unroll
      for (int c = 0; c < 4; c++) {
        if (J0 + c < T) acc[C::tix(I, J0 + c < T ? J0 + c : I)] = D[c];
        if (J0 + c == T) bacc[I] = D[c];
      }
    }
    __threadfence_block();  // (all operand reads done before the results overwrite the panel)
    if (I == T - 1) break;
#pragma unroll
    for (int J = I + 1; J < T; J++)
#pragma unroll
      for (int r = 0; r < 4; r++) pan[(16 * J + 4 * g + r) * 17 + m] = acc[C::tix(I, J)][r];
    if (g == 0) pan[KP * 17 + m] = bacc[I][0];
    __threadfence_block();
    // ---- (3) trailing update of the tiles right of block column I
#pragma unroll
    for (int s = 0; s < 4; s++) {
      float op[T], nop[T];
#pragma unroll
      for (int j = I + 1; j < T; j++) {
        op[j] = pan[(16 * j + m) * 17 + 4 * s + g];  // L_jI[m][4 s + g]
        nop[j] = -op[j];
      }
      const float nopy = -pan[KP * 17 + 4 * s + g];
#pragma unroll
      for (int J2 = I + 1; J2 < T; J2++) {
#pragma unroll
        for (int J = J2; J < T; J++)
          acc[C::tix(J2, J)] =
              __builtin_amdgcn_mfma_f32_16x16x4f32(nop[J], op[J2], acc[C::tix(J2, J)], 0, 0, 0);
        bacc[J2] = __builtin_amdgcn_mfma_f32_16x16x4f32(nopy, op[J2], bacc[J2], 0, 0, 0);
      }
    }
    __threadfence_block();  // the next step overwrites the panel and the scratch
  }
  if (__any(bad)) {
    if (lane == 0) atomicOr(err_flag, 1);
  }
  // ---- back substitution L^T x = y: z in registers (lane (g, n): z_I[n])
  float z[T], x[T];
#pragma unroll
  for (int i = 0; i < T; i++) z[i] = bacc[i][0];
#pragma unroll
  for (int J = T - 1; J >= 0; J--) {
    if (lane < 16) zx[lane] = z[J];
    __threadfence_block();
    float xj;
    {
      const float *e = wt + J * WS + m;
      float xq[4];  // four independent chains
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const f32x4 zq = *reinterpret_cast<const f32x4 *>(zx + 4 * q);
        xq[q] = e[(4 * q) * 17] * zq.x;
        xq[q] = fmaf(e[(4 * q + 1) * 17], zq.y, xq[q]);
        xq[q] = fmaf(e[(4 * q + 2) * 17], zq.z, xq[q]);
        xq[q] = fmaf(e[(4 * q + 3) * 17], zq.w, xq[q]);
      }
      xj = (xq[0] + xq[1]) + (xq[2] + xq[3]);
    }
    x[J] = xj;  // x_J[n = m]
    if (J == 0) break;
    if (lane < 16) zx[16 + lane] = xj;
    __threadfence_block();
    const f32x4 x4 = *reinterpret_cast<const f32x4 *>(zx + 16 + 4 * g);  // x_J[4 g + r]
    __threadfence_block();
#pragma unroll
    for (int I = 0; I < J; I++) {
      const f32x4 t = acc[C::tix(I, J)];  // L_JI
      float c = t[0] * x4.x;
      c = fmaf(t[1], x4.y, c);
      c = fmaf(t[2], x4.z, c);
      c = fmaf(t[3], x4.w, c);
      c += __shfl_xor(c, 16, 64);
      c += __shfl_xor(c, 32, 64);
      z[I] -= c;
    }
  }
  // lane m (group 0) holds the latent dims T m .. T m + T - 1 = x[0 .. T - 1] at virtual column m
  bool fin = true;
#pragma unroll
  for (int J = 0; J < T; J++) {
    const int dim = T * m + J;
    fin = fin && (__builtin_isfinite(x[J]) || dim >= K);
    x[J] = dim < K ? x[J] : 0.f;
  }
  if (!__all(fin)) {
    if (lane == 0) atomicOr(err_flag, 2);
  }
  if (g == 0) {
    float *dst = xrow + T * m;
    if constexpr (T % 4 == 0) {
#pragma unroll
      for (int q = 0; q < T / 4; q++)
        *reinterpret_cast<f32x4 *>(dst + 4 * q) = f32x4{x[4 * q], x[4 * q + 1], x[4 * q + 2], x[4 * q + 3]};
    } else {
#pragma unroll
      for (int J = 0; J < T; J++) dst[J] = x[J];
    }
  }
}

}  // namespace ials
}  // namespace irs
