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

#ifndef IRS_CHOL16_TRSM_CHUNK
#define IRS_CHOL16_TRSM_CHUNK(T) ((T) <= 4 ? 2 : 4)
#endif
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

// DPP forms of a row operation (used by diag_factor16 and by the batched factorisations below).
// v_fmac_f32_dpp has no builtin form that the compiler folds (the DPP combiner runs before
// v_fma_f32 becomes the two-address v_fmac_f32), hence inline assembly; the wait states between a
// vector write and a DPP read of the same register (2) are carried by the `s_nop 1` of the first
// instruction of each group.
template <int N> __device__ __forceinline__ float row_bcast(float x) {  // lane N of each 16-lane row
  float r;
  asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf"
               : "=v"(r)
               : "v"(x), "n"(N));
  return r;
}
template <int N, bool FIRST>
__device__ __forceinline__ void row_elim(float &xi, float &yi, const float xk, const float yk) {
  // xi -= xk[lane N of the row] * xk;  yi -= xk[lane N of the row] * yk
  if constexpr (FIRST)
    asm volatile("s_nop 1\n\tv_fmac_f32_dpp %0, -%1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf"
                 : "+v"(xi)
                 : "v"(xk), "v"(xk), "n"(N));
  else
    asm volatile("v_fmac_f32_dpp %0, -%1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf"
                 : "+v"(xi)
                 : "v"(xk), "v"(xk), "n"(N));
  asm volatile("v_fmac_f32_dpp %0, -%1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf"
               : "+v"(yi)
               : "v"(xk), "v"(yk), "n"(N));
}

// The 4 x 4 elimination of sub-panel Q (rows 4 Q .. 4 Q + 3, all in lane group Q) of [Cd | E].
// Round 4, first: under the execution mask of group Q (one s_and_saveexec) instead of selecting a
// neutral operand per operation (-250 of ~1100 vector instructions per 64 x 64 system).  Then: the
// multiplier R[k][k2] is lane 4 Q + r2 of the 16-lane row that holds the sub-panel, so the DPP row
// broadcast of v_fmac_f32 itself delivers it - two vector instructions per eliminated pair instead
// of v_readlane + two (-96 per system; same-box A/B: user half -2.3 %).
// -DIRS_CHOL16_READLANE restores the v_readlane form.
// Sum over the 16 lanes of a row (DPP row shifts, zero fill): lane 15 of the row holds the total.
__device__ __forceinline__ float row_sum16(float v) {
  auto shr = [](float x, auto ctrl) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), decltype(ctrl)::value,
                                                                 0xf, 0xf, true));
  };
  v += shr(v, std::integral_constant<int, 0x111>{});  // row_shr:1
  v += shr(v, std::integral_constant<int, 0x112>{});  // row_shr:2
  v += shr(v, std::integral_constant<int, 0x114>{});  // row_shr:4
  v += shr(v, std::integral_constant<int, 0x118>{});  // row_shr:8
  return v;
}

#ifdef IRS_CHOL16_READLANE
constexpr bool CHOL16_DPP = false;
#else
constexpr bool CHOL16_DPP = true;
#endif
// DPP = false: the v_readlane form (the 256-register workgroup kernels of K > 128 spill 0.5 KB per
// lane around the pinned assembly of the DPP form)
// CHECK = false: the pivots are not tested here - a pivot that is not positive leaves NaN or
// infinity on the diagonal of E (1 / sqrt), which the caller looks at only when the solution came out
// non-finite (solve_row_cholesky16: 64 vector compares per 64 x 64 system off the common path).
template <int Q, bool DPP, bool CHECK>
__device__ __forceinline__ void diag_subpanel(f32x4 &Cd, f32x4 &E, bool &bad) {
  if constexpr (!DPP) {
#pragma unroll
  for (int r = 0; r < 4; r++) {
    const float piv = readlane_f(Cd[r], 20 * Q + r);
    bad |= !(piv > 0.f);
    const float rinv = __builtin_amdgcn_rsqf(piv);
    Cd[r] *= rinv;
    E[r] *= rinv;
#pragma unroll
    for (int r2 = r + 1; r2 < 4; r2++) {
      const float sv = readlane_f(Cd[r], 20 * Q + r2);  // R[k][k2]
      Cd[r2] = fmaf(-sv, Cd[r], Cd[r2]);
      E[r2] = fmaf(-sv, E[r], E[r2]);
    }
  }
  } else {
  float c[4] = {Cd[0], Cd[1], Cd[2], Cd[3]}, e[4] = {E[0], E[1], E[2], E[3]};
  auto pivot = [&](auto rc) {
    constexpr int r = decltype(rc)::value;
    const float piv = row_bcast<4 * Q + r>(c[r]);
    if constexpr (CHECK) bad |= !(piv > 0.f);
    const float rinv = __builtin_amdgcn_rsqf(piv);
    c[r] *= rinv;
    e[r] *= rinv;
    if constexpr (r < 3) row_elim<4 * Q + r + 1, true>(c[r + 1], e[r + 1], c[r], e[r]);
    if constexpr (r < 2) row_elim<4 * Q + r + 2, false>(c[r + 2], e[r + 2], c[r], e[r]);
    if constexpr (r < 1) row_elim<4 * Q + r + 3, false>(c[r + 3], e[r + 3], c[r], e[r]);
  };
  pivot(std::integral_constant<int, 0>{});
  pivot(std::integral_constant<int, 1>{});
  pivot(std::integral_constant<int, 2>{});
  pivot(std::integral_constant<int, 3>{});
  Cd = f32x4{c[0], c[1], c[2], c[3]};
  E = f32x4{e[0], e[1], e[2], e[3]};
  }
}

// S = R^T R of one 16 x 16 tile in accumulator layout, E = R^-T alongside.  `Cd` is
// consumed; its strictly lower triangle only ever holds rounding noise and is never read as
// a result.  scrR / scrE: 16 x 17 floats each, private to the wave.
template <int Q, bool DPP, bool CHECK>
__device__ __forceinline__ void diag_factor16_step(f32x4 &Cd, f32x4 &E, float *scrR, float *scrE, bool &bad) {
  const int lane = threadIdx.x & 63;
  const int g = lane >> 4, m = lane & 15;
  if (g == Q) diag_subpanel<Q, DPP, CHECK>(Cd, E, bad);
  if constexpr (Q < 3) {
    // rows 4Q .. 4Q+3 are final: rank-4 update of the rows below them (and of E).  Every group
    // stores its four rows (no divergent branch: the block stays one scheduling region), the
    // read picks the sub-panel's.
#pragma unroll
    for (int r = 0; r < 4; r++) {
      scrR[(4 * g + r) * 17 + m] = Cd[r];
      scrE[(4 * g + r) * 17 + m] = E[r];
    }
    __threadfence_block();
    const float a = scrR[(4 * Q + g) * 17 + m];
    const float e = scrE[(4 * Q + g) * 17 + m];
    __threadfence_block();
    const float na = (m > 4 * Q + 3) ? -a : 0.f;  // rows up to the sub-panel are final
    Cd = __builtin_amdgcn_mfma_f32_16x16x4f32(na, a, Cd, 0, 0, 0);
    E = __builtin_amdgcn_mfma_f32_16x16x4f32(na, e, E, 0, 0, 0);
  }
}
template <bool DPP = CHOL16_DPP, bool CHECK = true>
__device__ __forceinline__ void diag_factor16(f32x4 &Cd, f32x4 &E, float *scrR, float *scrE,
                                              bool &bad) {
  // (E arrives as the identity tile: the caller keeps it in four registers across its tiles)
  diag_factor16_step<0, DPP, CHECK>(Cd, E, scrR, scrE, bad);
  diag_factor16_step<1, DPP, CHECK>(Cd, E, scrR, scrE, bad);
  diag_factor16_step<2, DPP, CHECK>(Cd, E, scrR, scrE, bad);
  diag_factor16_step<3, DPP, CHECK>(Cd, E, scrR, scrE, bad);
}
// the identity tile in accumulator layout
__device__ __forceinline__ f32x4 identity_tile16() {
  const int lane = threadIdx.x & 63;
  const int g = lane >> 4, m = lane & 15;
  f32x4 E;
#pragma unroll
  for (int r = 0; r < 4; r++) E[r] = (4 * g + r == m) ? 1.0f : 0.0f;
  return E;
}

// acc: lower-form tiles of A = P + sum c v v^T (no regulariser yet); b4[i] in lane (g, m) =
// right-hand side at virtual index 16 i + m.
template <int T>
__device__ __forceinline__ void solve_row_cholesky16(f32x4 (&acc)[Geo<T>::NT], const float (&b4)[T],
                                                     float reg, float *sm, float *xrow, int K,
                                                     int32_t *err_flag) {
  using C = Chol16Geo<T>;
  constexpr int WS = C::WS;
  const int lane = threadIdx.x & 63;
  const int g = lane >> 4, m = lane & 15;
  float *pan = sm + C::PAN, *wt = sm + C::WT, *scr = sm + C::SCR, *zx = sm + C::ZX;

  // diagonal: + reg (hpp:312-314); padded dims get 1 so that they decouple
#pragma unroll
  for (int i = 0; i < T; i++)
#pragma unroll
    for (int r = 0; r < 4; r++)
      if (4 * g + r == m) acc[C::tix(i, i)][r] += (T * m + i < K) ? reg : 1.0f;

  // The right-hand side used to ride through the factorisation as one more tile ROW (16 identical
  // rows: y = L^-1 b for 40 of the 128 matrix instructions of a 64 x 64 system, 1,280 issue cycles for
  // 2 K multiply-adds).  Round 4: the forward substitution runs on the vector unit after the
  // factorisation (forward_substitute16 below: ~170 vector instructions).

  bool bad = false;
  const f32x4 ident = identity_tile16();
  IPHASE_BEGIN;
#pragma unroll
  for (int I = 0; I < T; I++) {
    float *wtI = wt + I * WS;
    // ---- (1) diagonal tile
    {
      f32x4 Cd = acc[C::tix(I, I)], E = ident;
      diag_factor16<CHOL16_DPP, !CHOL16_DPP>(Cd, E, scr, scr + WS, bad);
#pragma unroll
      for (int r = 0; r < 4; r++) wtI[(4 * g + r) * 17 + m] = E[r];
    }
    __threadfence_block();
    IPHASE(1);
    // ---- (2) TRSM: L_JI = S_JI E^T.  B[k][n] = E[n][k]
    float bE[4];
#pragma unroll
    for (int s = 0; s < 4; s++) bE[s] = wtI[m * 17 + 4 * s + g];
    // All tiles of the block column are staged in the panel buffer at once - it has exactly the
    // layout the transposed operand read needs - so the step pays ONE LDS round trip and the tiles'
    // MFMA chains interleave.
#pragma unroll
    for (int J = I + 1; J < T; J++)
#pragma unroll
      for (int r = 0; r < 4; r++) pan[(16 * J + 4 * g + r) * 17 + m] = acc[C::tix(I, J)][r];
    __threadfence_block();
    // (chunks of CH tiles bound the operand registers; within a chunk the tiles' MFMA chains are
    // interleaved.  K <= 64: two tiles, 16 registers less at the kernel's pressure peak - what
    // four waves per SIMD need)
    constexpr int CH = IRS_CHOL16_TRSM_CHUNK(T);
#pragma unroll
    for (int J0 = I + 1; J0 < T; J0 += CH) {
      float aS[CH][4];  // A[i][k] = S[i][k]: lane (g, i) reads row i, column 4 s + g
      f32x4 D[CH];
#pragma unroll
      for (int c = 0; c < CH; c++) {
        D[c] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (J0 + c < T) {
#pragma unroll
          for (int s = 0; s < 4; s++) aS[c][s] = pan[(16 * (J0 + c) + m) * 17 + 4 * s + g];
        }
      }
#pragma unroll
      for (int s = 0; s < 4; s++)
#pragma unroll
        for (int c = 0; c < CH; c++)
          if (J0 + c < T) D[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(aS[c][s], bE[s], D[c], 0, 0, 0);
#pragma unroll
      for (int c = 0; c < CH; c++) {
        if (J0 + c < T) acc[C::tix(I, J0 + c < T ? J0 + c : I)] = D[c];
      }
    }
    __threadfence_block();  // (all operand reads done before the results overwrite the panel)
    IPHASE(2);
    if (I == T - 1) break;
#pragma unroll
    for (int J = I + 1; J < T; J++)
#pragma unroll
      for (int r = 0; r < 4; r++) pan[(16 * J + 4 * g + r) * 17 + m] = acc[C::tix(I, J)][r];
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
#pragma unroll
      for (int J2 = I + 1; J2 < T; J2++) {
#pragma unroll
        for (int J = J2; J < T; J++)
          acc[C::tix(J2, J)] =
              __builtin_amdgcn_mfma_f32_16x16x4f32(nop[J], op[J2], acc[C::tix(J2, J)], 0, 0, 0);
      }
    }
    __threadfence_block();  // the next step overwrites the panel and the scratch
    IPHASE(3);
  }
  if (!CHOL16_DPP && __any(bad)) {
    if (lane == 0) atomicOr(err_flag, 1);
  }
  // ---- forward substitution L y = b on the vector unit: y_I = E_I (b_I - sum_{J < I} L_IJ y_J).
  // b_I[m] and y_J[m] live in lane (g, m) (every group a copy); tile L_IJ's register r of lane (g, m)
  // is L[16 I + 4 g + r][16 J + m], so the products are summed over the 16 lanes of a group (DPP row
  // shifts), the group's four sums cross to the other layout through 16 floats of LDS, and E_I
  // (natural coordinates in LDS) is applied like E_J^T in the back substitution below.
  float z[T], x[T];
#pragma unroll
  for (int I = 0; I < T; I++) {
    if (I == 0) {
      if (lane < 16) zx[lane] = -b4[0];
    } else {
      float pr[4];
#pragma unroll
      for (int r = 0; r < 4; r++) pr[r] = (4 * g + r == m) ? -b4[I] : 0.f;
#pragma unroll
      for (int J = 0; J < I; J++) {
        const f32x4 t = acc[C::tix(J, I)];  // L_IJ
#pragma unroll
        for (int r = 0; r < 4; r++) pr[r] = fmaf(t[r], z[J], pr[r]);
      }
#pragma unroll
      for (int r = 0; r < 4; r++) pr[r] = row_sum16(pr[r]);  // lane (g, 15): -w_I[4 g + r]
      if (m == 15) *reinterpret_cast<f32x4 *>(zx + 4 * g) = f32x4{pr[0], pr[1], pr[2], pr[3]};
    }
    __threadfence_block();
    {
      const float *e = wt + I * WS + m * 17;  // row n = m of E_I
      float yq[4];
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const f32x4 wq = *reinterpret_cast<const f32x4 *>(zx + 4 * q);
        yq[q] = e[4 * q] * wq.x;
        yq[q] = fmaf(e[4 * q + 1], wq.y, yq[q]);
        yq[q] = fmaf(e[4 * q + 2], wq.z, yq[q]);
        yq[q] = fmaf(e[4 * q + 3], wq.w, yq[q]);
      }
      z[I] = -((yq[0] + yq[1]) + (yq[2] + yq[3]));  // y_I[n = m]
    }
    __threadfence_block();
  }
  // ---- back substitution L^T x = y: z in registers (lane (g, n): z_I[n])
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
    // The rare path.  Was it the factorisation (hpp:317-319) or the solve (hpp:321-323)?  A pivot that
    // was not positive left NaN or infinity as its 1 / sqrt on the diagonal of an E tile.
    bool badp = false;
    if (CHOL16_DPP) {
#pragma unroll
      for (int J = 0; J < T; J++) badp |= !(wt[J * WS + m * 17 + m] < __builtin_inff());
    }
    if (lane == 0) atomicOr(err_flag, __any(badp) ? 3 : 2);
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
  IPHASE(4);
}

}  // namespace ials
}  // namespace irs
