// iALS++ / iCD subspace solver on gfx950: Solver::step_ialspp / _step_dimrange /
// _prediction / step_icd of /root/reference/cpp_source/als/IALSTrainer.hpp:387-630.
//
// Per row, per sweep (`ialspp_iteration`):  pred_q = x . v_q  for the row's stored entries
// (hpp:387-421), then for every block [c0, c0 + D) of `ialspp_subspace_dimension` latent
// dims (hpp:423-514):
//     A = P[blk, blk] + sum_q c_q v_q[blk] v_q[blk]^T + reg I
//     B = P[blk, :] x + reg x[blk] + sum_q (c_q (pred_q - 1) - bias) v_q[blk]
//     delta = A^-1 B (LLT),   x[blk] -= delta,   pred_q -= delta . v_q[blk]
// Rows are independent, so one wave owns a row for a whole sweep: the prediction cache
// lives in a global scratch array indexed like the CSR, the D x D system lives in the
// MFMA accumulator registers (same lower-form tile layout and the same 16-row block Cholesky,
// ials_chol16.hpp, as the full solve, with TS = ceil(D / 16) tiles per side), the row's
// factor vector in LDS.  Subspace dimension 1 is the iCD branch (hpp:673-677), which is the
// D = 1 case of the same arithmetic.  The reference does not test the LLT status here
// (hpp:495-497), so a failed factorisation propagates NaN instead of raising.
#pragma once
#include "ials_chol16.hpp"
#include "ials_kernels.hpp"

namespace irs {
namespace ials {

struct PpParams {
  const int32_t *indptr;
  const int32_t *indices;  // padded like SolveParams::indices
  const float *data;
  const int32_t *rows;     // rows of this launch, longest first
  int32_t n_rows;
  const float *other;      // gathered factors [n_other, KP]
  float *target;           // solved factors   [n_rows, KP]
  const float *reg;        // per-row regulariser (hpp:117-120)
  const float *P;          // alpha0 * F^T F, row-major [KP, KP], natural coordinates
  float *pred;             // prediction cache [nnz + padding]
  int32_t *ignored_flag;   // sink for the LLT status bits
  float bias;              // observation_bias (hpp:431-432)
  int32_t K, KP;
  int32_t xs_extra;        // LDS floats of the x row beyond PpGeo::XS (KP > 256: KP - 256, else 0)
  int32_t sub;             // subspace dimension, >= 1
  int32_t zero_start;      // fold-in: the row starts from 0 (hpp:132)
  int32_t chain;           // 64-dim blocks: prediction passes merged into the rank updates
  const float *P_blk;      // chain: the blocks of P in accumulator layout (pp_pack_p_kernel)
};

template <int TS> struct PpGeo {
  static constexpr int DP = 16 * TS;
  static constexpr int NT = TS * (TS + 1) / 2;
  static constexpr int XS = 256;  // x row of a padded K up to 256; PpParams::xs_extra floats more above it
  static constexpr int CHOL_FLOATS = Chol16Geo<TS>::LDS_FLOATS;  // scratch of solve_row_cholesky16
  // Cholesky scratch | x (whole row) | P-part of the rhs | delta
  static constexpr int LDS_FLOATS = CHOL_FLOATS + XS + 2 * DP;
};

// v[t] = factor[row, c0 + off[t]]; ALIGNED: c0 and off[0] are multiples of TS and the
// offsets are consecutive (one vector load).  Lanes whose dims lie past the block are given
// offset 0 by the caller (their values are masked), so nothing is read past the row.
template <int TS, bool ALIGNED>
__device__ __forceinline__ void load_sub(const float *p, const int (&off)[TS], float (&v)[TS]) {
  if constexpr (ALIGNED) {
    load_dims<TS>(p + off[0], v);
  } else {
#pragma unroll
    for (int t = 0; t < TS; t++) v[t] = p[off[t]];
  }
}

// Rank update of the block: acc += sum c v v^T (MFMA, upper tiles),
// bsum += sum (c (pred - 1) - bias) v.  Two-deep pipeline of 16-entry groups with
// unconditional loads (the CSR arrays and the prediction cache are padded).
// FUSED (single block: D == K): the block is the whole row, so pred_q = x . v_q is formed
// from the gathered row itself (16-lane butterfly) and the prediction cache is never touched;
// xs is the row's factor vector in LDS.
template <int TS, bool ALIGNED, bool FUSED = false>
__device__ __forceinline__ void pp_rank_update(const PpParams &p, int begin, int end, int c0, int D,
                                               f32x4 (&acc)[PpGeo<TS>::NT], float (&bsum)[TS],
                                               const float *xs = nullptr) {
  const int lane = threadIdx.x & 63;
  const int g = lane >> 4, m = lane & 15;
  const float *col_base = p.other + c0;
  const int n = end - begin;
  const int nit = (n + 15) >> 4;
  const int32_t *ip = p.indices + begin + g;
  const float *dp = p.data + begin + g;
  const float *pp = p.pred + begin + g;
  bool dim_ok[TS];
  int off[TS];
#pragma unroll
  for (int t = 0; t < TS; t++) {
    dim_ok[t] = TS * m + t < D;
    off[t] = ALIGNED ? (TS * m < D ? TS * m + t : t) : (dim_ok[t] ? TS * m + t : 0);
  }
  float xloc[TS];
#pragma unroll
  for (int t = 0; t < TS; t++) xloc[t] = (FUSED && dim_ok[t]) ? xs[c0 + TS * m + t] : 0.f;
  int ia[4], ib[4];
  float ca[4], cb[4], pa[4], pb[4];
  float va[4][TS], vb[4][TS], xa[4], xb[4], wa[4], wb[4];
  auto load_idx = [&](int it, int (&ix)[4], float (&cx)[4], float (&px)[4]) {
#pragma unroll
    for (int u = 0; u < 4; u++) {
      ix[u] = ip[16 * it + 4 * u];
      cx[u] = dp[16 * it + 4 * u];
      px[u] = FUSED ? 0.f : pp[16 * it + 4 * u];
    }
  };
  auto gather = [&](int it, const int (&ix)[4], const float (&cx)[4], const float (&px)[4],
                    float (&v)[4][TS], float (&vc)[4], float (&vw)[4]) {
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const bool valid = 16 * it + 4 * u + g < n;
      vc[u] = valid ? cx[u] : 0.f;
      // hpp:485-486; FUSED: the entry's validity travels in vw and the residual is formed in
      // consume() once the row is there
      vw[u] = FUSED ? (valid ? 1.f : 0.f) : (valid ? cx[u] * (px[u] - 1.0f) - p.bias : 0.f);
      load_sub<TS, ALIGNED>(col_base + static_cast<size_t>(static_cast<unsigned>(ix[u])) * p.KP, off,
                            v[u]);
    }
  };
  auto consume = [&](const float (&v)[4][TS], const float (&vc)[4], const float (&vw)[4]) {
#pragma unroll
    for (int u = 0; u < 4; u++) {
      float cv[TS], vk[TS];
#pragma unroll
      for (int i = 0; i < TS; i++) vk[i] = dim_ok[i] ? v[u][i] : 0.f;
      float w = vw[u];
      if constexpr (FUSED) {
        float dot = 0.f;
#pragma unroll
        for (int i = 0; i < TS; i++) dot = fmaf(vk[i], xloc[i], dot);
        dot += __shfl_xor(dot, 1, 64);
        dot += __shfl_xor(dot, 2, 64);
        dot += __shfl_xor(dot, 4, 64);
        dot += __shfl_xor(dot, 8, 64);
        w = vw[u] != 0.f ? vc[u] * (dot - 1.0f) - p.bias : 0.f;
      }
#pragma unroll
      for (int i = 0; i < TS; i++) {
        cv[i] = vc[u] * vk[i];
        bsum[i] = fmaf(w, vk[i], bsum[i]);
      }
      int t = 0;
#pragma unroll
      for (int i = 0; i < TS; i++)
#pragma unroll
        for (int j = i; j < TS; j++) {  // lower form: slot (i, j) = tile (row block j, column block i)
          acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(cv[j], vk[i], acc[t], 0, 0, 0);
          t++;
        }
    }
  };
  load_idx(0, ia, ca, pa);
  gather(0, ia, ca, pa, va, xa, wa);
  load_idx(1, ia, ca, pa);
  for (int it = 0; it < nit; it += 2) {
    load_idx(it + 2, ib, cb, pb);
    gather(it + 1, ia, ca, pa, vb, xb, wb);
    consume(va, xa, wa);
    load_idx(it + 3, ia, ca, pa);
    gather(it + 2, ib, cb, pb, va, xa, wa);
    consume(vb, xb, wb);  // an all-masked group when nit is odd
  }
#pragma unroll
  for (int i = 0; i < TS; i++) {
    bsum[i] += __shfl_xor(bsum[i], 16, 64);
    bsum[i] += __shfl_xor(bsum[i], 32, 64);
  }
}

// Multi-block sweep over 64-dim blocks (TS = 4) with the cache correction merged into the next
// block's rank update.  The pass that builds block b > 0 reads two sub-rows of every gathered
// row - this block's and the previous one's - and first brings the prediction up to date,
// pred_q -= delta_prev . v_q[prev block] (hpp:500-506), in the coalesced 16-lanes-per-row layout.
// Together with pp_predict16 this replaces the two lane-per-entry passes (pp_predict /
// pp_pred_update: uncoalesced row walks, a device-scope fence each), which were 47 % of the
// K = 128 user half-step.  The cache entry of an entry is written by lane m = 0 of its 16-lane
// group and read back by the whole group in the next pass of the SAME wave (workgroup-scope
// fence between the passes).  Sub-row s = 0 is the current block.
__device__ __forceinline__ void pp_chain_pass(const PpParams &p, int begin, int end, int c0, int D,
                                              const float *xs, const float *delta_prev,
                                              bool store_pred, f32x4 (&acc)[10], float (&bsum)[4]) {
  constexpr int TS = 4, NS = 2;
  constexpr bool FIRST = false;
  const int lane = threadIdx.x & 63;
  const int g = lane >> 4, m = lane & 15;
  const int n = end - begin;
  const int nit = (n + 15) >> 4;
  const int32_t *ip = p.indices + begin + g;
  const float *dp = p.data + begin + g;
  float *pp = p.pred + begin + g;
  int cs[NS];
  float wt[NS][TS];  // weights of the dot that updates the prediction
#pragma unroll
  for (int j = 0; j < NS; j++) {
    cs[j] = FIRST ? 64 * j : (j == 0 ? c0 : c0 - 64);
#pragma unroll
    for (int t = 0; t < TS; t++)
      wt[j][t] = FIRST ? xs[cs[j] + TS * m + t] : (j == 0 ? 0.f : -delta_prev[TS * m + t]);
  }
  bool dim_ok[TS];
#pragma unroll
  for (int t = 0; t < TS; t++) dim_ok[t] = TS * m + t < D;
  int ia[4], ib[4];
  float ca[4], cb[4], pa[4], pb[4];
  float va[4][NS][TS], vb[4][NS][TS];
  auto load_idx = [&](int it, int (&ix)[4], float (&cx)[4], float (&px)[4]) {
#pragma unroll
    for (int u = 0; u < 4; u++) {
      ix[u] = ip[16 * it + 4 * u];
      cx[u] = dp[16 * it + 4 * u];
      px[u] = FIRST ? 0.f : pp[16 * it + 4 * u];
    }
  };
  auto gather = [&](const int (&ix)[4], float (&v)[4][NS][TS]) {
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const float *row = p.other + static_cast<size_t>(static_cast<unsigned>(ix[u])) * p.KP + TS * m;
#pragma unroll
      for (int j = 0; j < NS; j++) load_dims<TS>(row + cs[j], v[u][j]);
    }
  };
  auto consume = [&](int it, const float (&v)[4][NS][TS], const float (&cx)[4], const float (&px)[4]) {
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const bool valid = 16 * it + 4 * u + g < n;
      float dot = 0.f;
#pragma unroll
      for (int j = FIRST ? 0 : 1; j < NS; j++)
#pragma unroll
        for (int t = 0; t < TS; t++) dot = fmaf(wt[j][t], v[u][j][t], dot);
      dot += __shfl_xor(dot, 1, 64);
      dot += __shfl_xor(dot, 2, 64);
      dot += __shfl_xor(dot, 4, 64);
      dot += __shfl_xor(dot, 8, 64);
      const float pred = FIRST ? dot : px[u] + dot;
      if (store_pred && valid && m == 0) pp[16 * it + 4 * u] = pred;
      const float c = valid ? cx[u] : 0.f;
      const float w = valid ? c * (pred - 1.0f) - p.bias : 0.f;  // hpp:485-486
      float cv[TS], vk[TS];
#pragma unroll
      for (int i = 0; i < TS; i++) {
        vk[i] = dim_ok[i] ? v[u][0][i] : 0.f;
        cv[i] = c * vk[i];
        bsum[i] = fmaf(w, vk[i], bsum[i]);
      }
      int t = 0;
#pragma unroll
      for (int i = 0; i < TS; i++)
#pragma unroll
        for (int j = i; j < TS; j++) {  // lower form: slot (i, j) = tile (row block j, column block i)
          acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(cv[j], vk[i], acc[t], 0, 0, 0);
          t++;
        }
    }
  };
  // two-deep pipeline of 16-entry groups, unconditional loads (padded arrays).  The (index,
  // confidence, prediction) triple of a group is requested one whole consume() before its
  // gather needs the index; the gather takes a copy of the two values so that the triple's
  // registers can be refilled at once.
  float xa[4], xb[4], qa[4], qb[4];
  auto keep = [&](const float (&cx)[4], const float (&px)[4], float (&xc)[4], float (&qc)[4]) {
#pragma unroll
    for (int u = 0; u < 4; u++) {
      xc[u] = cx[u];
      qc[u] = px[u];
    }
  };
  load_idx(0, ia, ca, pa);
  gather(ia, va);
  keep(ca, pa, xa, qa);
  load_idx(1, ia, ca, pa);
  for (int it = 0; it < nit; it += 2) {
    load_idx(it + 2, ib, cb, pb);
    gather(ia, vb);
    keep(ca, pa, xb, qb);
    consume(it, va, xa, qa);
    load_idx(it + 3, ia, ca, pa);
    gather(ib, va);
    keep(cb, pb, xa, qa);
    consume(it + 1, vb, xb, qb);  // an all-masked group when nit is odd
  }
#pragma unroll
  for (int i = 0; i < TS; i++) {
    bsum[i] += __shfl_xor(bsum[i], 16, 64);
    bsum[i] += __shfl_xor(bsum[i], 32, 64);
  }
}

// pred_q = x . v_q over entries [b, e) in the same coalesced layout (16 lanes per gathered row,
// lane m reads floats 64 j + 4 m .. + 3 of it), two 16-entry groups in flight (hpp:387-421)
__device__ __forceinline__ void pp_predict16(const PpParams &p, const float *xs, int begin, int end) {
  const int lane = threadIdx.x & 63;
  const int g = lane >> 4, m = lane & 15;
  const int n = end - begin;
  const int nit = (n + 7) >> 3;  // groups of 8 entries: 2 per 16-lane group (register budget)
  const int nj = p.KP >> 6;  // 64-float segments of a row (KP is a multiple of 64 when K > 64)
  const int32_t *ip = p.indices + begin + g;
  float *pp = p.pred + begin + g;
  if (nj > 4) {
    // K > 256: more than four segments per row - the same layout, four segments of an entry in
    // flight at a time (the two-deep pipeline below keeps a whole row in registers)
    for (int it = 0; it < nit; it++) {
#pragma unroll
      for (int u = 0; u < 2; u++) {
        const int e = 8 * it + 4 * u + g;
        const int ix = ip[8 * it + 4 * u];  // (padded arrays)
        const float *row = p.other + static_cast<size_t>(static_cast<unsigned>(ix)) * p.KP + 4 * m;
        float dot = 0.f;
        for (int j0 = 0; j0 < nj; j0 += 4) {
          f32x4 v[4];
#pragma unroll
          for (int j = 0; j < 4; j++) v[j] = *reinterpret_cast<const f32x4 *>(row + 64 * min(j0 + j, nj - 1));
#pragma unroll
          for (int j = 0; j < 4; j++) {
            if (j0 + j < nj) {
              const f32x4 x4 = *reinterpret_cast<const f32x4 *>(xs + 64 * (j0 + j) + 4 * m);
              dot = fmaf(x4.x, v[j].x, dot);
              dot = fmaf(x4.y, v[j].y, dot);
              dot = fmaf(x4.z, v[j].z, dot);
              dot = fmaf(x4.w, v[j].w, dot);
            }
          }
        }
        dot += __shfl_xor(dot, 1, 64);
        dot += __shfl_xor(dot, 2, 64);
        dot += __shfl_xor(dot, 4, 64);
        dot += __shfl_xor(dot, 8, 64);
        if (m == 0 && e < n) pp[8 * it + 4 * u] = dot;
      }
    }
    return;
  }
  f32x4 xw[4];
#pragma unroll
  for (int j = 0; j < 4; j++)
    xw[j] = j < nj ? *reinterpret_cast<const f32x4 *>(xs + 64 * j + 4 * m) : f32x4{0.f, 0.f, 0.f, 0.f};
  int ia[2], ib[2];
  f32x4 va[2][4], vb[2][4];
  auto load_idx = [&](int it, int (&ix)[2]) {
#pragma unroll
    for (int u = 0; u < 2; u++) ix[u] = ip[8 * it + 4 * u];
  };
  auto gather = [&](const int (&ix)[2], f32x4 (&v)[2][4]) {
#pragma unroll
    for (int u = 0; u < 2; u++) {
      const float *row = p.other + static_cast<size_t>(static_cast<unsigned>(ix[u])) * p.KP + 4 * m;
#pragma unroll
      for (int j = 0; j < 4; j++)
        v[u][j] = *reinterpret_cast<const f32x4 *>(row + 64 * min(j, nj - 1));
    }
  };
  auto consume = [&](int it, const f32x4 (&v)[2][4]) {
#pragma unroll
    for (int u = 0; u < 2; u++) {
      float dot = 0.f;
#pragma unroll
      for (int j = 0; j < 4; j++) {  // (segments past the row carry zero weights)
        dot = fmaf(xw[j].x, v[u][j].x, dot);
        dot = fmaf(xw[j].y, v[u][j].y, dot);
        dot = fmaf(xw[j].z, v[u][j].z, dot);
        dot = fmaf(xw[j].w, v[u][j].w, dot);
      }
      dot += __shfl_xor(dot, 1, 64);
      dot += __shfl_xor(dot, 2, 64);
      dot += __shfl_xor(dot, 4, 64);
      dot += __shfl_xor(dot, 8, 64);
      if (m == 0 && 8 * it + 4 * u + g < n) pp[8 * it + 4 * u] = dot;
    }
  };
  load_idx(0, ia);
  gather(ia, va);
  load_idx(1, ia);
  for (int it = 0; it < nit; it += 2) {  // indices one consume() ahead of their gather
    load_idx(it + 2, ib);
    gather(ia, vb);
    consume(it, va);
    load_idx(it + 3, ia);
    gather(ib, va);
    consume(it + 1, vb);
  }
}

// pred_q = x . v_q over entries [b, e): lane per stored entry, x broadcast from LDS (hpp:387-421)
__device__ __forceinline__ void pp_predict(const PpParams &p, const float *xs, int b, int e) {
  const int lane = threadIdx.x & 63;
  // (the row is read sixteen floats x 4 loads at a time, all issued before the first FMA: a
  // load per trip of a K / 4-trip loop is a chain of K / 4 exposed memory latencies - 48 us per
  // 64 entries at K = 128; KP is a multiple of 16.  Same FMA order as a plain loop.)
  for (int q = b + lane; q < e; q += 64) {
    const float *v = p.other + static_cast<size_t>(static_cast<unsigned>(p.indices[q])) * p.KP;
    float s = 0.f;
    for (int k = 0; k < p.KP; k += 32) {
      f32x4 t[8];
#pragma unroll
      for (int j = 0; j < 8; j++)
        t[j] = *reinterpret_cast<const f32x4 *>(v + min(k + 4 * j, p.KP - 4));
#pragma unroll
      for (int j = 0; j < 8; j++) {
        if (k + 4 * j < p.KP) {
          const f32x4 x4 = *reinterpret_cast<const f32x4 *>(xs + k + 4 * j);
          s = fmaf(t[j].x, x4.x, s);
          s = fmaf(t[j].y, x4.y, s);
          s = fmaf(t[j].z, x4.z, s);
          s = fmaf(t[j].w, x4.w, s);
        }
      }
    }
    p.pred[q] = s;
  }
}

// pred_q -= delta . v_q[blk] over entries [b, e)   (hpp:500-506)
template <bool ALIGNED>
__device__ __forceinline__ void pp_pred_update(const PpParams &p, const float *delta, int c0, int D,
                                               int b, int e) {
  const int lane = threadIdx.x & 63;
  for (int q = b + lane; q < e; q += 64) {
    const float *v =
        p.other + static_cast<size_t>(static_cast<unsigned>(p.indices[q])) * p.KP + c0;
    float s = 0.f;
    if (ALIGNED && (D & 3) == 0 && (c0 & 3) == 0) {
      for (int i = 0; i < D; i += 4) {
        const f32x4 t = *reinterpret_cast<const f32x4 *>(v + i);
        s = fmaf(delta[i], t.x, s);
        s = fmaf(delta[i + 1], t.y, s);
        s = fmaf(delta[i + 2], t.z, s);
        s = fmaf(delta[i + 3], t.w, s);
      }
    } else {
      for (int i = 0; i < D; i++) s = fmaf(delta[i], v[i], s);
    }
    p.pred[q] -= s;
  }
}

// A <- P[blk, blk] in accumulator layout, LOWER form (ials_chol16.hpp): slot (I, J), I <= J, is the
// tile (row block J, column block I): register r of lane (g, m) is element
// (TS (4g + r) + J, TS m + I) of the block
template <int TS>
__device__ __forceinline__ void pp_block_of_p(const PpParams &p, int c0, int D,
                                              f32x4 (&acc)[PpGeo<TS>::NT]) {
  const int lane = threadIdx.x & 63;
  const int g = lane >> 4, m = lane & 15;
  int t = 0;
#pragma unroll
  for (int I = 0; I < TS; I++)
#pragma unroll
    for (int J = I; J < TS; J++) {
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int rr = TS * (4 * g + r) + J, cc = TS * m + I;
        acc[t][r] = (rr < D && cc < D) ? p.P[(c0 + rr) * p.KP + c0 + cc] : 0.f;
      }
      t++;
    }
}

// B <- P[blk, :] x + reg x[blk]  (hpp:473-477); P is symmetric: column reads coalesce
template <int TS>
__device__ __forceinline__ void pp_rhs_of_p(const PpParams &p, const float *xs, float reg, int c0,
                                            int D, float *bnat) {
  const int lane = threadIdx.x & 63;
  if (lane < PpGeo<TS>::DP) {
    float s = 0.f;
    if (lane < D) {
      // four independent chains: the loads of a step do not wait for the previous FMA
      float s4[4] = {0.f, 0.f, 0.f, 0.f};
      const float *col = p.P + c0 + lane;
      int k = 0;
      for (; k + 16 <= p.K; k += 16) {  // sixteen loads in flight, same four chains
        float c[16];
#pragma unroll
        for (int q = 0; q < 16; q++) c[q] = col[(k + q) * p.KP];
#pragma unroll
        for (int q = 0; q < 16; q++) s4[q & 3] = fmaf(c[q], xs[k + q], s4[q & 3]);
      }
      for (; k + 4 <= p.K; k += 4) {
#pragma unroll
        for (int q = 0; q < 4; q++) s4[q] = fmaf(col[(k + q) * p.KP], xs[k + q], s4[q]);
      }
      for (; k < p.K; k++) s4[0] = fmaf(col[k * p.KP], xs[k], s4[0]);
      s = (s4[0] + s4[1]) + (s4[2] + s4[3]);
      s = fmaf(reg, xs[c0 + lane], s);
    }
    bnat[lane] = s;
  }
}

// ---- 64-dim blocks (chain path): P prepared once per half-step ----------------------------
// p.P_blk holds, per block b, the 10 lower-form tiles of P[blk, blk] exactly as the
// accumulators want them (64 lanes x f32x4 per tile, dims past K zero): a row starts a block
// with 10 coalesced 1 KB loads instead of 40 strided scalar loads per lane.
__global__ __launch_bounds__(64) void pp_pack_p_kernel(const float *__restrict__ P, int K, int KP,
                                                       float *__restrict__ P_blk) {
  constexpr int TS = 4;
  const int b = blockIdx.x, c0 = 64 * b, D = min(64, K - c0);
  const int lane = threadIdx.x, g = lane >> 4, m = lane & 15;
  f32x4 *dst = reinterpret_cast<f32x4 *>(P_blk) + static_cast<size_t>(b) * 10 * 64;
  int t = 0;
  for (int I = 0; I < TS; I++)
    for (int J = I; J < TS; J++) {
      f32x4 v;
      for (int r = 0; r < 4; r++) {
        const int rr = TS * (4 * g + r) + J, cc = TS * m + I;
        v[r] = (rr < D && cc < D) ? P[(c0 + rr) * KP + c0 + cc] : 0.f;
      }
      dst[t * 64 + lane] = v;
      t++;
    }
}

__device__ __forceinline__ void pp_block_of_p_packed(const PpParams &p, int c0, f32x4 (&acc)[10]) {
  const f32x4 *src = reinterpret_cast<const f32x4 *>(p.P_blk) + static_cast<size_t>(c0 >> 6) * 10 * 64 +
                     (threadIdx.x & 63);
#pragma unroll
  for (int t = 0; t < 10; t++) acc[t] = src[t * 64];
}

// b4[i] (dim 4 m + i of the block, the layout of bsum) <- P[blk, :] x + reg x[blk]  (hpp:473-477).
// Lane (g, m) reads P[k, c0 + 4 m .. + 3] for k = g, g + 4, ...: one 1 KB wave load covers four
// rows of P, eight of them in flight.  The sums are those of pp_rhs_of_p bit for bit: chain g
// adds the terms k = g (mod 4) in increasing k, the chains are combined as (s0 + s1) + (s2 + s3).
__device__ __forceinline__ void pp_rhs_of_p16(const PpParams &p, const float *xs, float reg, int c0,
                                              float (&b4)[4]) {
  const int lane = threadIdx.x & 63;
  const int g = lane >> 4, m = lane & 15;
  const float *col = p.P + c0 + 4 * m;
  float s[4] = {0.f, 0.f, 0.f, 0.f};
  for (int k0 = 0; k0 < p.KP; k0 += 32) {  // KP is a multiple of 64 here
    f32x4 pv[8];
#pragma unroll
    for (int j = 0; j < 8; j++)
      pv[j] = *reinterpret_cast<const f32x4 *>(col + static_cast<size_t>(k0 + 4 * j + g) * p.KP);
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const float xk = xs[k0 + 4 * j + g];
      s[0] = fmaf(pv[j].x, xk, s[0]);
      s[1] = fmaf(pv[j].y, xk, s[1]);
      s[2] = fmaf(pv[j].z, xk, s[2]);
      s[3] = fmaf(pv[j].w, xk, s[3]);
    }
  }
#pragma unroll
  for (int i = 0; i < 4; i++) {
    s[i] += __shfl_xor(s[i], 16, 64);
    s[i] += __shfl_xor(s[i], 32, 64);
    b4[i] = fmaf(reg, xs[c0 + 4 * m + i], s[i]);
  }
}

// One wave per row, four independent rows per workgroup.  (launch bound: without it the compiler
// takes 256 VGPRs + 96 AGPRs, i.e. ONE wave per SIMD and every latency exposed - K = 128
// epoch 15.0 ms; two waves per SIMD: 12.3 ms; forcing three spills 600 bytes per lane: 15.6 ms.)
// MODE 0: several blocks, three passes per block; 1: one block covers the row (no cache);
// 2: 64-dim blocks, chained passes.  Compile-time so that each form gets its own registers.
template <int TS, bool ALIGNED, int MODE>
__global__ __launch_bounds__(256, 2) void ialspp_kernel(PpParams p) {
  using G = PpGeo<TS>;
  extern __shared__ __attribute__((aligned(16))) float pp_lds[];
  const int wid = wave_in_block(), lane = threadIdx.x & 63;
  const int m = lane & 15;
  const int w = blockIdx.x * 4 + wid;
  if (w >= p.n_rows) return;  // the kernel uses no workgroup barrier
  float *sm = pp_lds + wid * (G::LDS_FLOATS + p.xs_extra);
  float *xs = sm + G::CHOL_FLOATS;
  float *bnat = xs + G::XS + p.xs_extra;
  float *delta = bnat + G::DP;

  const int row = p.rows[w];
  const int begin = p.indptr[row], end = p.indptr[row + 1];
  const float reg = p.reg[row];
  float *xrow = p.target + static_cast<size_t>(row) * p.KP;
  for (int i = lane; i < p.KP; i += 64) {
    const float x0 = p.zero_start ? 0.f : xrow[i];
    xs[i] = x0;
    if (p.zero_start) xrow[i] = 0.f;
  }
  __threadfence_block();
  constexpr bool single = MODE == 1;  // one block = the whole row: no prediction cache needed
  // 64-dim blocks of a longer row: the prediction passes ride in the rank-update passes
  constexpr bool chain = MODE == 2;
  static_assert(!chain || (TS == 4 && ALIGNED), "chained passes: 64-dim aligned blocks");
  if constexpr (!single && !chain) {
#ifndef IRS_PP_SKIP_PRED
    pp_predict(p, xs, begin, end);
    __threadfence();  // the cache is re-read through other lanes' addresses below
#endif
  }
  if constexpr (chain) {
#ifndef IRS_PP_SKIP_PRED
    pp_predict16(p, xs, begin, end);
#endif
    __threadfence_block();
  }

  for (int c0 = 0; c0 < p.K; c0 += p.sub) {
    const int D = min(p.sub, p.K - c0);
    f32x4 acc[G::NT];
#ifndef IRS_PP_SKIP_P
    float bp[TS];  // chain: the P part of the rhs, already in the layout of bsum
    if constexpr (chain) {
      pp_block_of_p_packed(p, c0, acc);
      pp_rhs_of_p16(p, xs, reg, c0, bp);
    }
    if constexpr (!chain) {
      pp_block_of_p<TS>(p, c0, D, acc);
      pp_rhs_of_p<TS>(p, xs, reg, c0, D, bnat);
    }
#else
#pragma unroll
    for (int t = 0; t < G::NT; t++) acc[t] = f32x4{1.f, 0.f, 0.f, 0.f};
    if (lane < G::DP) bnat[lane] = 0.f;
#endif
    float bsum[TS];
#pragma unroll
    for (int i = 0; i < TS; i++) bsum[i] = 0.f;
#ifndef IRS_PP_SKIP_RANK
    bool chained = false;
    if constexpr (chain) {
      if (c0 > 0) {
        pp_chain_pass(p, begin, end, c0, D, xs, delta, c0 + 64 < p.K, acc, bsum);
        chained = true;
      }
    }
    if constexpr (single)
      pp_rank_update<TS, ALIGNED, true>(p, begin, end, c0, D, acc, bsum, xs);
    else if (!chained)
      pp_rank_update<TS, ALIGNED>(p, begin, end, c0, D, acc, bsum);
#endif
    __threadfence_block();
    float b4[TS];
#pragma unroll
    for (int i = 0; i < TS; i++) b4[i] = bsum[i] + bnat[TS * m + i];
#ifndef IRS_PP_SKIP_P
    if constexpr (chain) {
#pragma unroll
      for (int i = 0; i < TS; i++) b4[i] = bsum[i] + bp[i];
    }
#endif
    // ---- delta = (A + reg I)^-1 B   (hpp:490-497)
#ifndef IRS_PP_SKIP_SOLVE
    solve_row_cholesky16<TS>(acc, b4, reg, sm, delta, D, p.ignored_flag);
#else
    if (lane < G::DP) delta[lane] = b4[0] + acc[0][0];
#endif
    __threadfence_block();
    if (lane < D) {  // hpp:498
      const float nx = xs[c0 + lane] - delta[lane];
      xs[c0 + lane] = nx;
      xrow[c0 + lane] = nx;
    }
    __threadfence_block();
    // the cache is rebuilt at the start of every sweep (hpp:521-523): the correction after
    // the last block would never be read
    if (c0 + p.sub < p.K && !chain) {
#ifndef IRS_PP_SKIP_PRED
      pp_pred_update<ALIGNED>(p, delta, c0, D, begin, end);
      __threadfence();
#endif
    }
  }
}

// Long rows: one workgroup of PP_LONG_WAVES waves per row.  Every pass over the row's
// entries (prediction, rank update, cache correction) is split into equal contiguous ranges,
// the partial D x D systems are summed through LDS and wave 0 solves.
constexpr int PP_LONG_WAVES = 8;

template <int TS> struct PpLongGeo {
  using G = PpGeo<TS>;
  static constexpr int PART = G::NT * 256 + 64;  // tiles + rhs of one wave
  // Cholesky scratch | x | P-part of the rhs | delta | partials of waves 1 .. 7
  static constexpr int LDS_FLOATS = G::LDS_FLOATS + (PP_LONG_WAVES - 1) * PART;
};

template <int TS, bool ALIGNED, int MODE>
__global__ __launch_bounds__(64 * PP_LONG_WAVES) void ialspp_long_kernel(PpParams p) {
  using G = PpGeo<TS>;
  using L = PpLongGeo<TS>;
  extern __shared__ __attribute__((aligned(16))) float pp_lds[];
  const int wid = wave_in_block(), lane = threadIdx.x & 63;
  const int m = lane & 15;
  float *sm = pp_lds;
  float *xs = sm + G::CHOL_FLOATS;
  float *bnat = xs + G::XS + p.xs_extra;
  float *delta = bnat + G::DP;
  float *parts = delta + G::DP;

  const int row = p.rows[blockIdx.x];
  const int begin = p.indptr[row], end = p.indptr[row + 1];
  const float reg = p.reg[row];
  float *xrow = p.target + static_cast<size_t>(row) * p.KP;
  // this wave's entries: equal ranges, multiples of 16 (the rank update's group size)
  const int per = (((end - begin + PP_LONG_WAVES - 1) / PP_LONG_WAVES) + 15) & ~15;
  const int wb = min(begin + wid * per, end), we = min(wb + per, end);
  for (int i = threadIdx.x; i < p.KP; i += 64 * PP_LONG_WAVES) {
    const float x0 = p.zero_start ? 0.f : xrow[i];
    xs[i] = x0;
    if (p.zero_start) xrow[i] = 0.f;
  }
  __syncthreads();
  constexpr bool single = MODE == 1;  // see ialspp_kernel
  constexpr bool chain = MODE == 2;
  if constexpr (!single && !chain) {
    pp_predict(p, xs, wb, we);
    __threadfence();
    __syncthreads();
  }
  if constexpr (chain) {
    pp_predict16(p, xs, wb, we);
    __threadfence_block();
  }

  for (int c0 = 0; c0 < p.K; c0 += p.sub) {
    const int D = min(p.sub, p.K - c0);
    f32x4 acc[G::NT];
    if (wid == 0) {
      bool packed = false;
      if constexpr (chain) {
        pp_block_of_p_packed(p, c0, acc);
        packed = true;
      }
      if (!packed) pp_block_of_p<TS>(p, c0, D, acc);
      pp_rhs_of_p<TS>(p, xs, reg, c0, D, bnat);
    } else {
#pragma unroll
      for (int t = 0; t < G::NT; t++) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    float bsum[TS];
#pragma unroll
    for (int i = 0; i < TS; i++) bsum[i] = 0.f;
    bool chained = false;
    if constexpr (chain) {
      if (c0 > 0) {
        pp_chain_pass(p, wb, we, c0, D, xs, delta, c0 + 64 < p.K, acc, bsum);
        chained = true;
      }
    }
    if constexpr (single)
      pp_rank_update<TS, ALIGNED, true>(p, wb, we, c0, D, acc, bsum, xs);
    else if (!chained)
      pp_rank_update<TS, ALIGNED>(p, wb, we, c0, D, acc, bsum);
    if (wid > 0) {
      float *dst = parts + (wid - 1) * L::PART;
      f32x4 *d4 = reinterpret_cast<f32x4 *>(dst);
#pragma unroll
      for (int t = 0; t < G::NT; t++) d4[t * 64 + lane] = acc[t];
      if (lane < 16) {
#pragma unroll
        for (int i = 0; i < TS; i++) dst[G::NT * 256 + TS * lane + i] = bsum[i];
      }
    }
    __syncthreads();
    if (wid == 0) {
      for (int w2 = 0; w2 < PP_LONG_WAVES - 1; w2++) {  // fixed order: reproducible sums
        const float *src = parts + w2 * L::PART;
        const f32x4 *s4 = reinterpret_cast<const f32x4 *>(src);
#pragma unroll
        for (int t = 0; t < G::NT; t++) acc[t] += s4[t * 64 + lane];
#pragma unroll
        for (int i = 0; i < TS; i++) bsum[i] += src[G::NT * 256 + TS * m + i];
      }
      float b4[TS];
#pragma unroll
      for (int i = 0; i < TS; i++) b4[i] = bsum[i] + bnat[TS * m + i];
      solve_row_cholesky16<TS>(acc, b4, reg, sm, delta, D, p.ignored_flag);
      __threadfence_block();
      if (lane < D) {  // hpp:498
        const float nx = xs[c0 + lane] - delta[lane];
        xs[c0 + lane] = nx;
        xrow[c0 + lane] = nx;
      }
    }
    __syncthreads();
    if (c0 + p.sub < p.K && !chain) {  // see ialspp_kernel: the last correction is never read
      pp_pred_update<ALIGNED>(p, delta, c0, D, wb, we);
      __threadfence();
      __syncthreads();
    }
  }
}

}  // namespace ials
}  // namespace irs
