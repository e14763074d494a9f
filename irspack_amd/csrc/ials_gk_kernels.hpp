// General-size iALS kernels for gfx950: any latent dimension (K > 256) and any iALS++ block
// width (ialspp_subspace_dimension > 64), with every size a RUN-TIME value.  The reference
// has no limit on either (IALSLearningConfig.hpp:119, 139-141; ials.py:358 tunes
// n_components up to 300 and tune_doubling_dimension keeps doubling, ials.py:657-749), so a
// drop-in must compute there too.  The tuned kernels of ials_kernels.hpp / ials_wg16_kernels.hpp
// keep a row's whole system in registers / LDS, which stops at 256 x 256; here a system lives
// in HBM scratch (served from L2 / Infinity Cache) as 16 x 16 tiles and two launches per batch
// of rows do the work of Solver::step_cholesky (hpp:273-331):
//
//   gk_syrk_kernel   A = P[slice] + sum c v v^T (+ reg I), b = sum w v (+ b0)     on the matrix
//                    cores: one wave per (row, 64 x 64 block pair), 16 accumulator tiles,
//                    one 16-byte gather per lane and operand block per 4 stored entries
//   gk_chol_kernel   A = L L^T left-looking by 16-wide block columns, the products
//                    L_Jk L_Ik^T on the matrix cores with operands read straight from the
//                    tile scratch, the diagonal tile by diag_factor16 (ials_chol16.hpp), then
//                    both substitutions; one 256-thread workgroup per system
//
// A "slice" is the dim range [c0, c0 + D) of the factors (the whole row for Cholesky, one
// block for iALS++: _step_dimrange, hpp:436-502).  Np = D rounded up to 64.
//
// Virtual basis (as in ials_kernels.hpp, per 64-dim block): a lane's 16-byte gather holds the
// dims 4 m .. 4 m + 3 of a block, MFMA tile ti of the block is therefore the dim set
// {4 i + ti}; virtual index a = 64 blk + 16 ti + i  <->  slice dim 64 blk + 4 i + ti.  The
// system is built, factorised and solved in that (permuted) basis and x is written back in
// natural order - a symmetric permutation of an SPD system.
//
// Scratch system of a row: lower tiles (R >= C) row-major 16 x 16 at ((R (R + 1) / 2 + C) * 256,
// then the right-hand side (Np floats).
//
// Also here, for K > 256: the Gramian (prepare_p, hpp:78-115), the matrix-free CG of
// hpp:170-271 (one workgroup per row, the vectors in LDS), user_scores (hpp:942-984) and the
// per-row loss terms (hpp:845-917) with a run-time KP (a multiple of 64).
#pragma once
#include "ials_chol16.hpp"

namespace irs {
namespace ials {

struct GkParams {
  // rows of this launch: row = rows[row_first + i], i < n_rows (longest first)
  const int32_t *rows;
  int32_t row_first, n_rows;
  const int32_t *indptr;
  const int32_t *indices;
  const float *data;      // matrix weights c_q
  const float *pred;      // iALS++: prediction cache (CSR-indexed); rhs weight c (pred - 1) - bias
  const float *other;     // gathered factors [n_other, ld_other]
  int32_t ld_other;
  float *target;          // [n_rows_total, ld_target]
  int32_t ld_target;
  const float *reg;       // per-row regulariser
  const float *P;         // alpha0 F^T F row-major [ldP, ldP]
  int32_t ldP;
  const float *prior;     // feature prior [rows, ld_target] or null
  float *sys;             // scratch: n_rows systems of sys_floats floats
  int64_t sys_floats;
  int32_t c0, D, Np;      // slice [c0, c0 + D), padded width
  int32_t K;              // unpadded latent dimension of the target rows
  float bias;
  int32_t mode;           // 0: Cholesky half-step (x -> target row); 1: iALS++ block (target[c0 ..] -= x)
  int32_t *err_flag;
};

__device__ __forceinline__ int gk_tile_off(int R, int C) { return (R * (R + 1) / 2 + C) * 256; }
// slice dim of virtual index a
__device__ __forceinline__ int gk_dim_of(int a) { return (a & ~63) + 4 * (a & 15) + ((a >> 4) & 3); }

// ---------------------------------------------------------------------------------------
// Rank update.  Grid: ceil(n_rows * nbp / 4) workgroups of 4 waves; unit u = (row, block pair).
template <bool DIAG>
__device__ __forceinline__ void gk_syrk_body(const GkParams &p, int row, int BI, int BJ, float *sys) {
  const int lane = threadIdx.x & 63;
  const int g = lane >> 4, m = lane & 15;
  const int b = p.indptr[row], e = p.indptr[row + 1];
  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = 0; j < 4; j++) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  float bsum[4] = {0.f, 0.f, 0.f, 0.f};
  // dims of the two operand blocks this lane loads; dims past the slice read as zero
  const int da = 64 * BI + 4 * m, db = 64 * BJ + 4 * m;
  const bool aligned = ((p.c0 | p.ld_other) & 3) == 0;
  auto load4 = [&](const float *src, int d0, float (&v)[4]) {
    if (aligned && d0 + 3 < p.D) {
      const f32x4 t = *reinterpret_cast<const f32x4 *>(src + d0);
      v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    } else {
#pragma unroll
      for (int t = 0; t < 4; t++) v[t] = d0 + t < p.D ? src[d0 + t] : 0.f;
    }
  };
  auto fetch = [&](int q0, float (&va)[4], float (&vb)[4], float &c, float &w) {
    const int q = q0 + g;
    const bool valid = q < e;
    const int qq = valid ? q : b;  // (b < e here)
    const int idx = p.indices[qq];
    c = valid ? p.data[qq] : 0.f;
    // rhs weight: bias + c (hpp:301-306) or, for an iALS++ block, c (pred - 1) - bias (hpp:476-482)
    w = !valid ? 0.f : (p.pred ? c * (p.pred[qq] - 1.0f) - p.bias : p.bias + c);
    const float *src = p.other + static_cast<size_t>(idx) * p.ld_other + p.c0;
    load4(src, da, va);
    if constexpr (!DIAG) load4(src, db, vb);
  };
  if (b < e) {
    float va[2][4], vb[2][4], c[2], w[2];
    fetch(b, va[0], vb[0], c[0], w[0]);
    int cur = 0;
    for (int q0 = b; q0 < e; q0 += 4) {
      const int nxt = cur ^ 1;
      if (q0 + 4 < e) fetch(q0 + 4, va[nxt], vb[nxt], c[nxt], w[nxt]);
      float cv[4];
#pragma unroll
      for (int t = 0; t < 4; t++) cv[t] = c[cur] * va[cur][t];
#pragma unroll
      for (int ti = 0; ti < 4; ti++)
#pragma unroll
        for (int tj = 0; tj < 4; tj++) {
          if constexpr (DIAG) {
            if (ti >= tj)
              acc[ti][tj] = __builtin_amdgcn_mfma_f32_16x16x4f32(cv[ti], va[cur][tj], acc[ti][tj], 0, 0, 0);
          } else {
            acc[ti][tj] = __builtin_amdgcn_mfma_f32_16x16x4f32(cv[ti], vb[cur][tj], acc[ti][tj], 0, 0, 0);
          }
        }
      if constexpr (DIAG) {
#pragma unroll
        for (int t = 0; t < 4; t++) bsum[t] = fmaf(w[cur], va[cur][t], bsum[t]);
      }
      cur = nxt;
    }
  }
  // + P[slice] (+ reg on the diagonal; padded dims get a unit diagonal so that they decouple),
  // then the tiles go to the scratch system row-major
  const float reg = p.reg[row];
#pragma unroll
  for (int ti = 0; ti < 4; ti++)
#pragma unroll
    for (int tj = 0; tj < 4; tj++) {
      if (DIAG && ti < tj) continue;
      float *dst = sys + gk_tile_off(4 * BI + ti, 4 * BJ + tj);
      const int dc = 64 * BJ + 4 * m + tj;  // slice dim of column n = m
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int dr = 64 * BI + 4 * (4 * g + r) + ti;
        float v = acc[ti][tj][r];
        if (dr < p.D && dc < p.D) {
          v += p.P[static_cast<size_t>(p.c0 + dr) * p.ldP + p.c0 + dc];
          if (dr == dc) v += reg;
        } else if (DIAG && dr == dc) {
          v = 1.0f;
        }
        dst[(4 * g + r) * 16 + m] = v;
      }
    }
  if constexpr (DIAG) {
    float *rhs = sys + p.sys_floats - p.Np;
#pragma unroll
    for (int t = 0; t < 4; t++) {
      float s = bsum[t];
      s += __shfl_xor(s, 16, 64);
      s += __shfl_xor(s, 32, 64);
      const int d = 64 * BI + 4 * m + t;  // slice dim; virtual index 64 BI + 16 t + m
      if (g == 0) {
        float v = 0.f;
        if (d < p.D) {
          v = s;
          if (p.mode == 0 && p.prior)  // hpp:363 / 212-215: rhs += reg_r * prior_r
            v = fmaf(reg, p.prior[static_cast<size_t>(row) * p.ld_target + p.c0 + d], v);
          if (p.mode == 1) v += rhs[64 * BI + 16 * t + m];  // b0 = P[blk, :] x + reg x_blk, set before
        }
        rhs[64 * BI + 16 * t + m] = v;
      }
    }
  }
}

__global__ __launch_bounds__(256, 2) void gk_syrk_kernel(GkParams p) {
  const int nb = p.Np / 64, nbp = nb * (nb + 1) / 2;
  const int64_t u = static_cast<int64_t>(blockIdx.x) * 4 + wave_in_block();
  if (u >= static_cast<int64_t>(p.n_rows) * nbp) return;
  const int ri = static_cast<int>(u / nbp);
  int bp = static_cast<int>(u % nbp);
  // the diagonal pairs first (they also sum the right-hand side): bp < nb -> (bp, bp)
  int BI, BJ;
  if (bp < nb) {
    BI = BJ = bp;
  } else {
    bp -= nb;  // strictly lower pairs, row-major: BI = 1: (1,0); BI = 2: (2,0), (2,1); ...
    BI = 1;
    while (bp >= BI) { bp -= BI; BI++; }
    BJ = bp;
  }
  const int row = p.rows[p.row_first + ri];
  float *sys = p.sys + static_cast<int64_t>(ri) * p.sys_floats;
  if (BI == BJ) gk_syrk_body<true>(p, row, BI, BJ, sys);
  else gk_syrk_body<false>(p, row, BI, BJ, sys);
}

// iALS++ block: b0 = P[blk, :] x + reg_r x_blk (hpp:473-475) into the rhs slots of the scratch
// systems (virtual order), one workgroup per row.
__global__ __launch_bounds__(256) void gk_block_rhs0_kernel(GkParams p) {
  extern __shared__ float gk_x[];  // the row's current factor
  const int ri = blockIdx.x;
  const int row = p.rows[p.row_first + ri];
  const float *x = p.target + static_cast<size_t>(row) * p.ld_target;
  for (int k = threadIdx.x; k < p.ldP; k += 256) gk_x[k] = k < p.K ? x[k] : 0.f;
  __syncthreads();
  float *rhs = p.sys + static_cast<int64_t>(ri) * p.sys_floats + p.sys_floats - p.Np;
  const float reg = p.reg[row];
  for (int a = threadIdx.x; a < p.Np; a += 256) {
    const int d = gk_dim_of(a);
    float s = 0.f;
    if (d < p.D) {
      const float *prow = p.P + static_cast<size_t>(p.c0 + d) * p.ldP;
      float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
      for (int k = 0; k < p.ldP; k += 4) {
        const f32x4 pv = *reinterpret_cast<const f32x4 *>(prow + k);
        s0 = fmaf(pv.x, gk_x[k], s0);
        s1 = fmaf(pv.y, gk_x[k + 1], s1);
        s2 = fmaf(pv.z, gk_x[k + 2], s2);
        s3 = fmaf(pv.w, gk_x[k + 3], s3);
      }
      s = fmaf(reg, gk_x[p.c0 + d], (s0 + s1) + (s2 + s3));
    }
    rhs[a] = s;
  }
}

// ---------------------------------------------------------------------------------------
// Blocked Cholesky + substitutions on the scratch system of one row.  256 threads.
struct GkCholLds {
  float scr[4][2 * 16 * 17];  // per wave: tile transposition / diag_factor16 scratch
};

__global__ __launch_bounds__(256, 2) void gk_chol_kernel(GkParams p) {
  extern __shared__ __attribute__((aligned(16))) float gk_lds[];
  const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
  const int g = lane >> 4, m = lane & 15;
  const int ri = blockIdx.x;
  const int row = p.rows[p.row_first + ri];
  float *sys = p.sys + static_cast<int64_t>(ri) * p.sys_floats;
  const int nt = p.Np / 16;
  float *scr = gk_lds + wv * (2 * 16 * 17);
  float *vec = gk_lds + 4 * (2 * 16 * 17);  // Np floats: b -> y -> z
  float *xs = vec + p.Np;                   // 16 floats: the block just solved
  bool bad = false;
  // operand form of a row-major tile: lane (g, i) holds tile[i][4 g .. 4 g + 3]
  auto op4 = [&](const float *tile) { return *reinterpret_cast<const f32x4 *>(tile + m * 16 + 4 * g); };
  for (int I = 0; I < nt; I++) {
    // S_JI = A_JI - sum_{k < I} L_Jk L_Ik^T for the tiles J = I + wv, I + wv + 4, ...; wave 0
    // starts with the diagonal tile
    auto accumulate = [&](int J) {
      const float *src = sys + gk_tile_off(J, I);
      f32x4 acc;
#pragma unroll
      for (int r = 0; r < 4; r++) acc[r] = src[(4 * g + r) * 16 + m];
      f32x4 a = f32x4{0.f, 0.f, 0.f, 0.f}, bq = a;
      if (I > 0) {
        a = op4(sys + gk_tile_off(J, 0));
        bq = op4(sys + gk_tile_off(I, 0));
      }
      for (int k = 0; k < I; k++) {
        f32x4 an = a, bn = bq;
        if (k + 1 < I) {
          an = op4(sys + gk_tile_off(J, k + 1));
          bn = op4(sys + gk_tile_off(I, k + 1));
        }
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(-a.x, bq.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(-a.y, bq.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(-a.z, bq.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(-a.w, bq.w, acc, 0, 0, 0);
        a = an;
        bq = bn;
      }
      return acc;
    };
    if (wv == 0) {
      f32x4 Cd = accumulate(I), E = identity_tile16();
      diag_factor16<false>(Cd, E, scr, scr + 16 * 17, bad);
      float *dst = sys + gk_tile_off(I, I);  // the slot of L_II keeps E = L_II^-1 (L_II is not needed again)
#pragma unroll
      for (int r = 0; r < 4; r++) dst[(4 * g + r) * 16 + m] = E[r];
    }
    __syncthreads();
    {
      const f32x4 e4 = op4(sys + gk_tile_off(I, I));  // B[c][n] = E[n][c], c = 4 g + s
      for (int J = I + 1 + wv; J < nt; J += 4) {
        const f32x4 S = accumulate(J);
        // accumulator -> operand layout through the wave's LDS scratch
#pragma unroll
        for (int r = 0; r < 4; r++) scr[(4 * g + r) * 17 + m] = S[r];
        __threadfence_block();
        float a[4];
#pragma unroll
        for (int s = 0; s < 4; s++) a[s] = scr[m * 17 + 4 * g + s];
        __threadfence_block();
        f32x4 L = f32x4{0.f, 0.f, 0.f, 0.f};
        L = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], e4.x, L, 0, 0, 0);
        L = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1], e4.y, L, 0, 0, 0);
        L = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2], e4.z, L, 0, 0, 0);
        L = __builtin_amdgcn_mfma_f32_16x16x4f32(a[3], e4.w, L, 0, 0, 0);
        float *dst = sys + gk_tile_off(J, I);
#pragma unroll
        for (int r = 0; r < 4; r++) dst[(4 * g + r) * 16 + m] = L[r];
      }
    }
    __syncthreads();  // column I is final before column I + 1 reads it
  }
  if (__any(bad)) {
    if (lane == 0) atomicOr(p.err_flag, 1);  // hpp:317-318
  }
  // ---- forward substitution y = L^-1 b (right-looking over block columns)
  const float *rhs = sys + p.sys_floats - p.Np;
  for (int a = tid; a < p.Np; a += 256) vec[a] = rhs[a];
  __syncthreads();
  for (int I = 0; I < nt; I++) {
    if (tid < 16) {  // y_I = E_I b_I
      const float *E = sys + gk_tile_off(I, I) + tid * 16;
      float s = 0.f;
#pragma unroll
      for (int c = 0; c < 16; c++) s = fmaf(E[c], vec[16 * I + c], s);
      xs[tid] = s;
    }
    __syncthreads();
    if (tid < 16) vec[16 * I + tid] = xs[tid];
    for (int j = 16 * (I + 1) + tid; j < p.Np; j += 256) {  // b_J -= L_JI y_I
      const float *Lr = sys + gk_tile_off(j >> 4, I) + (j & 15) * 16;
      float s = 0.f;
#pragma unroll
      for (int c = 0; c < 16; c++) s = fmaf(Lr[c], xs[c], s);
      vec[j] -= s;
    }
    __syncthreads();
  }
  // ---- back substitution L^T x = y
  for (int I = nt - 1; I >= 0; I--) {
    if (tid < 16) {  // x_I = E_I^T z_I
      const float *E = sys + gk_tile_off(I, I) + tid;
      float s = 0.f;
#pragma unroll
      for (int i = 0; i < 16; i++) s = fmaf(E[i * 16], vec[16 * I + i], s);
      xs[tid] = s;
    }
    __syncthreads();
    if (tid < 16) vec[16 * I + tid] = xs[tid];
    for (int a = tid; a < 16 * I; a += 256) {  // z_I2 -= L_{I,I2}^T x_I
      const float *Lc = sys + gk_tile_off(I, a >> 4) + (a & 15);
      float s = 0.f;
#pragma unroll
      for (int j = 0; j < 16; j++) s = fmaf(Lc[j * 16], xs[j], s);
      vec[a] -= s;
    }
    __syncthreads();
  }
  // ---- x back to natural order
  bool fin = true;
  float *trow = p.target + static_cast<size_t>(row) * p.ld_target;
  for (int a = tid; a < p.Np; a += 256) {
    const int d = gk_dim_of(a);
    if (d >= p.D) continue;
    const float x = vec[a];
    fin = fin && __builtin_isfinite(x);
    if (p.mode == 0) trow[p.c0 + d] = x;
    else trow[p.c0 + d] -= x;  // hpp:498
  }
  if (!fin) atomicOr(p.err_flag, 2);  // hpp:321-323
}

// ---------------------------------------------------------------------------------------
// iALS++ with wide blocks: the prediction cache (hpp:410-413) and its correction (hpp:500-506).
// One wave per row; 16 lanes per stored entry, every lane a strided share of the dims.
// mode 0: pred_q = x . v_q over all K dims; mode 1: pred_q -= delta . v_q[c0 .. c0 + D) where
// delta = x_before - x_now of the block is read from `delta` ([rows, D] scratch, natural order).
__global__ __launch_bounds__(256) void gk_pred_kernel(GkParams p, const float *__restrict__ delta,
                                                      float *__restrict__ pred, int mode) {
  const int lane = threadIdx.x & 63, g = lane >> 4, m = lane & 15;
  const int64_t ri = static_cast<int64_t>(blockIdx.x) * 4 + wave_in_block();
  if (ri >= p.n_rows) return;
  const int row = p.rows[p.row_first + ri];
  const int b = p.indptr[row], e = p.indptr[row + 1];
  const float *x = mode == 0 ? p.target + static_cast<size_t>(row) * p.ld_target
                             : delta + static_cast<size_t>(ri) * p.D;
  const int c0 = mode == 0 ? 0 : p.c0, n = mode == 0 ? p.K : p.D;
  for (int q0 = b; q0 < e; q0 += 4) {
    const int q = q0 + g;
    const bool valid = q < e;
    const float *v = p.other + static_cast<size_t>(p.indices[valid ? q : b]) * p.ld_other + c0;
    float s = 0.f;
    for (int k = m; k < n; k += 16) s = fmaf(x[k], v[k], s);
    s += __shfl_xor(s, 1, 64);
    s += __shfl_xor(s, 2, 64);
    s += __shfl_xor(s, 4, 64);
    s += __shfl_xor(s, 8, 64);
    if (valid && m == 0) pred[q] = mode == 0 ? s : pred[q] - s;
  }
}

// saves / differences the block of the target rows around a block solve: mode 0 copies
// x[c0 .. c0 + D) into `delta`, mode 1 turns it into (saved - now) = the step taken
__global__ void gk_block_delta_kernel(GkParams p, float *__restrict__ delta, int mode) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= static_cast<int64_t>(p.n_rows) * p.D) return;
  const int ri = static_cast<int>(i / p.D), d = static_cast<int>(i % p.D);
  const int row = p.rows[p.row_first + ri];
  const float now = p.target[static_cast<size_t>(row) * p.ld_target + p.c0 + d];
  delta[i] = mode == 0 ? now : delta[i] - now;
}

// ---------------------------------------------------------------------------------------
// Conjugate gradient, matrix free (hpp:199-264), any K: one 256-thread workgroup per row, the
// vectors in LDS.  A vec = P vec (column-wise through the symmetry of P, coalesced), + reg vec,
// + sum c (v . vec) v with the stored entries dealt to the four waves.
__global__ __launch_bounds__(256, 2) void gk_cg_kernel(GkParams p, int max_cg_steps, int warm_start) {
  extern __shared__ __attribute__((aligned(16))) float gk_cg_lds[];
  const int KP = p.ldP, tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
  float *x = gk_cg_lds, *r = x + KP, *pv = r + KP, *Ap = pv + KP;
  // the gathered sums of a row (up to 10^5 terms per dim on this path: rows are not split) are
  // accumulated in float64, one partial per wave: the float32 form was 7e-4 from the float64
  // iterate on ML-20M item rows of 10^4 entries at K = 320 (the oracle's sequential float32 sums: 1.2e-3)
  double *wacc = reinterpret_cast<double *>(Ap + KP);             // 4 x KP doubles
  float *red = reinterpret_cast<float *>(wacc + 4 * KP);          // 8 floats
  const int row = p.rows[p.row_first + blockIdx.x];
  const int b = p.indptr[row], e = p.indptr[row + 1];
  float *trow = p.target + static_cast<size_t>(row) * p.ld_target;
  if (e == b && p.prior == nullptr) {  // hpp:207-210
    for (int k = tid; k < KP; k += 256) trow[k] = 0.f;
    return;
  }
  const float reg = p.reg[row];
  auto block_sum = [&](float v) {
    v = wave_sum(v);
    __syncthreads();
    if (lane == 0) red[wv] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
  };
  // out (+)= sum_q wgt_q (v_q . vec) v_q over the row; wgt = c.  `vec` may be null: out = sum w v
  // with the rhs weights (the right-hand side b, hpp:212-221)
  auto gather = [&](const float *vec, float *out, bool add) {
    double *mine = wacc + wv * KP;
    for (int k = lane; k < KP; k += 64) mine[k] = 0.0;
    // four stored entries of this wave at a time: their loads are issued together (one entry
    // after the other is a chain of exposed L2 latencies - the 116 k-entry item row of the
    // ML-20M shape alone took 60 ms per half-step that way)
    for (int q0 = b + wv; q0 < e; q0 += 16) {
      const float *v[4];
      float c[4], s[4];
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const int q = q0 + 4 * j;
        const bool ok = q < e;
        v[j] = p.other + static_cast<size_t>(p.indices[ok ? q : b]) * p.ld_other;
        c[j] = ok ? p.data[q] : 0.f;
        s[j] = 0.f;
      }
      if (vec) {
        for (int k = lane; k < KP; k += 64) {
          const float x = vec[k];
#pragma unroll
          for (int j = 0; j < 4; j++) s[j] = fmaf(v[j][k], x, s[j]);
        }
#pragma unroll
        for (int j = 0; j < 4; j++) s[j] = c[j] * wave_sum(s[j]);
      } else {
#pragma unroll
        for (int j = 0; j < 4; j++) s[j] = q0 + 4 * j < e ? p.bias + c[j] : 0.f;
      }
      for (int k = lane; k < KP; k += 64) {
        double a = mine[k];
#pragma unroll
        for (int j = 0; j < 4; j++)  // ascending entry order
          a = fma(static_cast<double>(s[j]), static_cast<double>(v[j][k]), a);
        mine[k] = a;
      }
    }
    __syncthreads();
    for (int k = tid; k < KP; k += 256) {
      const double s = (wacc[k] + wacc[KP + k]) + (wacc[2 * KP + k] + wacc[3 * KP + k]);
      // (`add`: out holds P vec; + reg vec comes last, see ials_short_kernels.hpp)
      out[k] = add ? fmaf(reg, vec[k], static_cast<float>(static_cast<double>(out[k]) + s))
                   : static_cast<float>(s);
    }
    __syncthreads();
  };
  auto matvec = [&](const float *vec, float *out) {  // hpp:222-228, 240-247
    for (int t = tid; t < KP; t += 256) {
      float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
      const float *col = p.P + t;
      for (int k = 0; k < KP; k += 4) {
        s0 = fmaf(col[static_cast<size_t>(k) * KP], vec[k], s0);
        s1 = fmaf(col[static_cast<size_t>(k + 1) * KP], vec[k + 1], s1);
        s2 = fmaf(col[static_cast<size_t>(k + 2) * KP], vec[k + 2], s2);
        s3 = fmaf(col[static_cast<size_t>(k + 3) * KP], vec[k + 3], s3);
      }
      out[t] = (s0 + s1) + (s2 + s3);
    }
    __syncthreads();
    gather(vec, out, true);
  };
  for (int k = tid; k < KP; k += 256) x[k] = (warm_start && k < p.K) ? trow[k] : 0.f;
  __syncthreads();
  gather(nullptr, r, false);  // r = b for now
  if (p.prior) {
    for (int k = tid; k < p.K; k += 256)
      r[k] = fmaf(reg, p.prior[static_cast<size_t>(row) * p.ld_target + k], r[k]);
    __syncthreads();
  }
  if (warm_start) {
    matvec(x, Ap);
    for (int k = tid; k < KP; k += 256) r[k] -= Ap[k];
  }
  for (int k = tid; k < KP; k += 256) {
    if (k >= p.K) r[k] = 0.f;
    pv[k] = r[k];
  }
  __syncthreads();
  auto dot = [&](const float *a, const float *c) {
    float s = 0.f;
    for (int k = tid; k < KP; k += 256) s = fmaf(a[k], c[k], s);
    return block_sum(s);
  };
  float r2 = dot(r, r);
  bool singular = false;
  for (int it = 0; it < max_cg_steps; it++) {
    if (r2 <= 1e-20f) break;  // hpp:238
    matvec(pv, Ap);
    for (int k = tid; k < KP; k += 256)
      if (k >= p.K) Ap[k] = 0.f;
    __syncthreads();
    const float denom = dot(pv, Ap);
    if (!(denom > 0.f) || !__builtin_isfinite(denom)) {  // hpp:250-254
      singular = true;
      break;
    }
    const float alpha = r2 / denom;
    for (int k = tid; k < KP; k += 256) {
      x[k] = fmaf(alpha, pv[k], x[k]);
      r[k] = fmaf(-alpha, Ap[k], r[k]);
    }
    __syncthreads();
    const float r2n = dot(r, r);
    if (r2n <= 1e-20f) break;  // hpp:258
    const float beta = r2n / r2;  // hpp:261
    for (int k = tid; k < KP; k += 256) pv[k] = fmaf(beta, pv[k], r[k]);
    __syncthreads();
    r2 = r2n;
  }
  if (singular && tid == 0) atomicOr(p.err_flag, 4);
  for (int k = tid; k < KP; k += 256) trow[k] = k < p.K ? x[k] : 0.f;
}

// ---------------------------------------------------------------------------------------
// Gramian F^T F for KP > 256 (prepare_p, hpp:78-115): unit = (64 x 64 block pair, slab of rows),
// one wave each; partial blocks in natural coordinates, summed over the slabs in a fixed order.
__global__ __launch_bounds__(256) void gk_gramian_partial_kernel(const float *__restrict__ F, int KP,
                                                                 int64_t row_begin, int64_t row_end,
                                                                 int64_t rows_per_slab, int n_slabs,
                                                                 float *__restrict__ partial) {
  const int lane = threadIdx.x & 63, g = lane >> 4, m = lane & 15;
  const int nb = KP / 64, nbp = nb * (nb + 1) / 2;
  const int64_t u = static_cast<int64_t>(blockIdx.x) * 4 + wave_in_block();
  if (u >= static_cast<int64_t>(n_slabs) * nbp) return;
  const int slab = static_cast<int>(u / nbp);
  int bp = static_cast<int>(u % nbp), BI = 0;
  while (bp > BI) { bp -= BI + 1; BI++; }
  const int BJ = bp;
  const int64_t rb = row_begin + slab * rows_per_slab, re = min(rb + rows_per_slab, row_end);
  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = 0; j < 4; j++) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int64_t r0 = rb; r0 < re; r0 += 4) {
    f32x4 a = f32x4{0.f, 0.f, 0.f, 0.f}, b = a;
    if (r0 + g < re) {
      a = *reinterpret_cast<const f32x4 *>(F + (r0 + g) * KP + 64 * BI + 4 * m);
      b = *reinterpret_cast<const f32x4 *>(F + (r0 + g) * KP + 64 * BJ + 4 * m);
    }
#pragma unroll
    for (int ti = 0; ti < 4; ti++)
#pragma unroll
      for (int tj = 0; tj < 4; tj++)
        acc[ti][tj] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ti], b[tj], acc[ti][tj], 0, 0, 0);
  }
  float *dst = partial + static_cast<size_t>(slab) * KP * KP;
#pragma unroll
  for (int ti = 0; ti < 4; ti++)
#pragma unroll
    for (int tj = 0; tj < 4; tj++)
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int dr = 64 * BI + 4 * (4 * g + r) + ti, dc = 64 * BJ + 4 * m + tj;
        dst[static_cast<size_t>(dr) * KP + dc] = acc[ti][tj][r];
      }
}

// P_raw[i][j] = sum over the slabs (ascending) of the lower-block partials, mirrored
__global__ void gk_gramian_reduce_kernel(const float *__restrict__ partial, int KP, int n_slabs,
                                         float *__restrict__ P_raw) {
  const int64_t idx = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (idx >= static_cast<int64_t>(KP) * KP) return;
  const int i = static_cast<int>(idx / KP), j = static_cast<int>(idx % KP);
  const bool lower = (i >> 6) >= (j >> 6);
  const size_t src = lower ? static_cast<size_t>(i) * KP + j : static_cast<size_t>(j) * KP + i;
  float s = 0.f;
  for (int sl = 0; sl < n_slabs; sl++) s += partial[static_cast<size_t>(sl) * KP * KP + src];
  P_raw[idx] = s;
}

__global__ void gk_gramian_finish_kernel(const float *__restrict__ P_raw, float alpha0, int64_t n,
                                         float *__restrict__ P) {
  const int64_t idx = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (idx < n) P[idx] = alpha0 * P_raw[idx];
}

// ---------------------------------------------------------------------------------------
// user_scores (hpp:942-984) with a run-time KP (multiple of 32): the tile shape and the order of
// the products are those of user_scores_kernel.
__global__ __launch_bounds__(256) void gk_user_scores_kernel(const float *__restrict__ user,
                                                             const float *__restrict__ item, int KP,
                                                             int64_t begin, int64_t m_rows,
                                                             int64_t n_items, float *__restrict__ out) {
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
    const int64_t i = min(it * 64 + 4 * m + q, n_items - 1);
    up[q] = user + (begin + u) * KP + 4 * g;
    ip[q] = item + i * KP + 4 * g;
  }
  f32x4 acc[4][4];
#pragma unroll
  for (int p = 0; p < 4; p++)
#pragma unroll
    for (int q = 0; q < 4; q++) acc[p][q] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 a[4], b[4], an[4], bn[4];
#pragma unroll
  for (int q = 0; q < 4; q++) {
    a[q] = *reinterpret_cast<const f32x4 *>(up[q]);
    b[q] = *reinterpret_cast<const f32x4 *>(ip[q]);
  }
  for (int k = 0; k < KP; k += 16) {
    const int kn = k + 16 < KP ? k + 16 : k;
#pragma unroll
    for (int q = 0; q < 4; q++) {
      an[q] = *reinterpret_cast<const f32x4 *>(up[q] + kn);
      bn[q] = *reinterpret_cast<const f32x4 *>(ip[q] + kn);
    }
#pragma unroll
    for (int c = 0; c < 4; c++)
#pragma unroll
      for (int p = 0; p < 4; p++)
#pragma unroll
        for (int q = 0; q < 4; q++)
          acc[p][q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[p][c], b[q][c], acc[p][q], 0, 0, 0);
#pragma unroll
    for (int q = 0; q < 4; q++) {
      a[q] = an[q];
      b[q] = bn[q];
    }
  }
  const int64_t col = it * 64 + 4 * m;
#pragma unroll
  for (int p = 0; p < 4; p++)
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int64_t row = ut * 64 + p * 16 + 4 * g + r;
      if (row >= m_rows) continue;
      float *dst = out + row * n_items + col;
#pragma unroll
      for (int q = 0; q < 4; q++)
        if (col + q < n_items) dst[q] = acc[p][q][r];
    }
}

// per-row loss terms (hpp:845-917) with a run-time KP: one wave per row, 16 lanes per entry
__global__ __launch_bounds__(256) void gk_loss_rows_kernel(const float *__restrict__ target,
                                                           const float *__restrict__ other, int KP,
                                                           const int32_t *__restrict__ indptr,
                                                           const int32_t *__restrict__ indices,
                                                           const float *__restrict__ data,
                                                           const float *__restrict__ reg,
                                                           int64_t n_rows, float bias, int with_observed,
                                                           float *__restrict__ row_loss) {
  const int lane = threadIdx.x & 63;
  const int g = lane >> 4, m = lane & 15;
  const int64_t row = static_cast<int64_t>(blockIdx.x) * 4 + wave_in_block();
  if (row >= n_rows) return;
  const float *u = target + row * KP;
  float loss = 0.f;
  if (with_observed) {
    const int b = indptr[row], e = indptr[row + 1];
    for (int q0 = b; q0 < e; q0 += 4) {
      const int q = q0 + g;
      const bool valid = q < e;
      const int idx = valid ? indices[q] : 0;
      const float c = valid ? data[q] : 0.f;
      const float *v = other + static_cast<size_t>(idx) * KP;
      float d = 0.f;
      for (int k = 4 * m; k < KP; k += 64) {
        const f32x4 uu = *reinterpret_cast<const f32x4 *>(u + k);
        const f32x4 vv = *reinterpret_cast<const f32x4 *>(v + k);
        d = fmaf(uu.x, vv.x, d);
        d = fmaf(uu.y, vv.y, d);
        d = fmaf(uu.z, vv.z, d);
        d = fmaf(uu.w, vv.w, d);
      }
      d += __shfl_xor(d, 1, 64);
      d += __shfl_xor(d, 2, 64);
      d += __shfl_xor(d, 4, 64);
      d += __shfl_xor(d, 8, 64);
      if (valid && m == 0) loss += c * d * d - 2.f * (c + bias) * d + c + bias;  // hpp:867-869
    }
  }
  float n2 = 0.f;
  if (g == 0)
    for (int k = m; k < KP; k += 16) n2 = fmaf(u[k], u[k], n2);
  loss += reg[row] * n2;
  loss = wave_sum(loss);
  if (lane == 0) row_loss[row] = loss;
}

}  // namespace ials
}  // namespace irs
