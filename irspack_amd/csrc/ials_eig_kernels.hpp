// Short rows in the eigenbasis of the Gramian (K <= 128).
//
// For a row with n stored entries the reference factorises the K x K matrix
// A = P + reg_r I + sum c v v^T (step_cholesky, hpp:273-331) or multiplies by it (step_cg,
// hpp:222-247).  With P = Q diag(lambda) Q^T (one symmetric eigen-decomposition per half-step,
// shared by every row) and everything expressed in the basis Q - the gathered table
// V~ = V Q once per half-step, the row x~ = Q^T x - the K x K part becomes DIAGONAL:
//     M = P + reg_r I  ->  diag(lambda_k + reg_r),  whatever the row's regulariser is.
//   * Cholesky, n <= 32 entries (the low-rank / Woodbury form of the same solve):
//       A^-1 b = Z y,  Z = M^-1 V,  (C^-1 + V^T Z) y = C^-1 w   (b = V w, w = bias + c)
//     i.e. with d_k = 1 / (lambda_k + reg_r), G = C^1/2:
//       S = V~ D V~^T (n x n, on the matrix cores),  (I + G S G) s = G^-1 w,  y = G s,
//       x~ = D V~^T y,  x = Q x~ (one batched product over all short rows afterwards).
//     ~n^2 K / 2 + n K + K^2 multiply-adds per row instead of n K^2 / 2 + K^3 / 6: at K = 128,
//     n = 10 that is 30 k instead of 430 k (configs[3]: 10 M such rows per epoch).
//   * CG, n <= 32 entries: the product A p = (lambda + reg) * p + sum c (v~ . p) v~ needs no
//     K x K matrix at all (the reference's matrix-free loop with P diagonal); CG's iterates are
//     the same vectors in another orthonormal basis.  x~0 = Q^T x0 before, x = Q x~ after.
// The mathematics is the reference's; the rounding path is not (like the reference's own CG vs
// Cholesky): parity is checked per row against the oracle on the full configs[3] matrix
// (tests/test_gpu_fullsize.py) and the path is only taken when M is well conditioned
// (lambda_max + reg_min <= 10^4 (lambda_min + reg_min), decided from the eigenvalues).
//
// eig_jacobi_kernel: one-sided (Hestenes) Jacobi in float64 on W = P (columns orthogonalised:
// W = P Q), W in LDS, the accumulated Q in global scratch; round-robin pair schedule, one
// 1024-thread workgroup.  ~1 ms for K = 128: once per half-step, next to 10^7 row solves.
#pragma once
#include "ials_chol16.hpp"

namespace irs {
namespace ials {

// sum over the 16 lanes of a DPP row, result in every lane (quad_perm xor 1, xor 2, then
// row_half_mirror and row_mirror: vector-ALU moves, no LDS crossbar)
__device__ __forceinline__ double row16_sum(double v) {
  auto step = [](double x, auto ctrl) {
    constexpr int C = decltype(ctrl)::value;
    const int lo = __double2loint(x), hi = __double2hiint(x);
    const int lo2 = __builtin_amdgcn_update_dpp(0, lo, C, 0xf, 0xf, false);
    const int hi2 = __builtin_amdgcn_update_dpp(0, hi, C, 0xf, 0xf, false);
    return x + __hiloint2double(hi2, lo2);
  };
  v = step(v, std::integral_constant<int, 0xB1>{});   // quad_perm [1, 0, 3, 2]
  v = step(v, std::integral_constant<int, 0x4E>{});   // quad_perm [2, 3, 0, 1]
  v = step(v, std::integral_constant<int, 0x141>{});  // row_half_mirror
  v = step(v, std::integral_constant<int, 0x140>{});  // row_mirror
  return v;
}
// (row16_sum(float): ials_kernels.hpp)

struct EigOut {
  float *Qrows;   // [KP, KP]: row k = eigenvector k (natural dims)
  float *Qcols;   // [KP, KP]: Qcols[d][k] = eigenvector k at dim d
  float *lam;     // [KP] eigenvalues (0 for padded dims)
  float *stats;   // [0] = largest, [1] = smallest eigenvalue over the K real dims, [2] = sweeps
  double *Qd;     // [KP, KP] column-major, float64: the eigenvectors of this call; on entry (when
                  // `warm` is set) those of the previous call for the same side - the Gramian moves
                  // little between epochs, so the sweeps start from an almost diagonal problem
  int *resident;  // host-pinned word (or null): set to `token` by the first thread once the workgroup runs
  int token;
};

// One-sided (Hestenes) Jacobi, everything on chip.  W (float64, LDS, column-major by column id)
// is orthogonalised column pair by column pair; the KP / 2 pairs of a round are disjoint (circle
// method: positions top[0 .. NP), bot[0 .. NP), pair p = (top[p], bot[p]); top[0] stays, the
// others rotate one position per round).  The accumulated rotations Q live in REGISTERS, owned
// by rows and indexed by POSITION: thread (row r, part t') holds Q[r][top[p]], Q[r][bot[p]] of
// its PPT pairs, so applying a round's rotations is thread-local and the rotation of the
// positions is a register shift plus one lane exchange per row.  Phase 1 of a round (16 / 32
// threads per pair: dots, rotation, W columns) and phase 2 (Q) are separated by one barrier.
template <int KP>
__global__ __launch_bounds__(1024) void eig_jacobi_kernel(const float *__restrict__ P, int K, EigOut o,
                                                          int warm) {
  extern __shared__ __attribute__((aligned(16))) unsigned char eig_lds[];
  double *W = reinterpret_cast<double *>(eig_lds);  // W[col * KP + row]
  constexpr int NP = KP / 2, G = 1024 / NP, RPT = KP / G;  // phase 1: G threads per pair, RPT rows each
  constexpr int TPR = 1024 / KP, PPT = NP / TPR;            // phase 2: TPR threads per row, PPT pairs each
  static_assert(G <= 64 && (G & (G - 1)) == 0 && TPR * PPT == NP, "thread mapping");
  __shared__ double sh_cs[2][NP], sh_sn[2][NP];
  __shared__ unsigned int sh_off;   // the sweep's largest remaining coupling (float bits), see phase 1
  // largest squared column norm of W seen so far (float bits), in three slots used in rotation:
  // round n reads slot (n - 1) % 3 - complete since the barrier of round n - 1 - and folds it, with
  // its own norms, into slot n % 3; slot (n + 1) % 3 is the one round n - 2 wrote and nobody touches.
  // (One word read while other pairs' leaders atomicMax it made the convergence measure, and with it
  // the sweep count and the factors, depend on wave timing.)
  __shared__ unsigned int sh_amax[3];
  __shared__ float sh_lam[KP];
  const int tid = threadIdx.x, pr = tid / G, l = tid % G;
  const int qr = tid / TPR, qt = tid % TPR;  // phase 2: row, part
  // the workgroup holds KP x KP doubles of LDS - a whole CU at KP = 128: kernels launched beside it
  // before it is resident keep every CU partly occupied and it starts when THEY end (measured:
  // 4.4 ms alone, 8.4 ms "beside" a 4.1 ms kernel).  The host launches them once this word is set.
  if (tid == 0 && o.resident) __hip_atomic_store(o.resident, o.token, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  auto top_col = [](int p, int rd) { return p == 0 ? KP - 1 : (rd + p) % (KP - 1); };
  auto bot_col = [](int p, int rd) { return (rd - p + (KP - 1)) % (KP - 1); };
  double *Qd = o.Qd;
  // ---- start: Q = previous eigenvectors (or I), W = P Q
  double qtop[PPT], qbot[PPT];
#pragma unroll
  for (int k = 0; k < PPT; k++) {
    const int p = PPT * qt + k, ct = top_col(p, 0), cb = bot_col(p, 0);
    qtop[k] = warm ? Qd[ct * KP + qr] : (ct == qr ? 1.0 : 0.0);
    qbot[k] = warm ? Qd[cb * KP + qr] : (cb == qr ? 1.0 : 0.0);
  }
  for (int i = tid; i < KP * KP; i += 1024) {
    const int col = i / KP, row = i % KP;
    double s;
    if (warm) {
      // W = P Q_prev.  P[d][row] for P[row][d] (P is symmetric): the lanes of a wave are
      // consecutive rows, so the load is one 256-byte line instead of 64 lines 512 bytes apart;
      // Q_prev[col][d] is the same word for the whole wave.  Four partial sums, eight loads in
      // flight: the product took 1.5 of the decomposition's 4 ms as a chain of 128 exposed loads.
      double s4[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll 2
      for (int d = 0; d < KP; d += 4) {
#pragma unroll
        for (int e = 0; e < 4; e++)
          s4[e] = fma(static_cast<double>(P[static_cast<size_t>(d + e) * KP + row]), Qd[col * KP + d + e], s4[e]);
      }
      s = (s4[0] + s4[1]) + (s4[2] + s4[3]);
    } else {
      s = static_cast<double>(P[static_cast<size_t>(row) * KP + col]);  // (P is symmetric)
    }
    W[i] = s;
  }
  if (tid == 0) {
    sh_off = 0u;
    sh_amax[0] = sh_amax[1] = sh_amax[2] = 0u;
  }
  __syncthreads();
  {  // the first round's "previous" slot: the largest squared column norm of the starting W
    constexpr int TPC = 1024 / KP;  // threads per column
    const int col = tid / TPC, part = tid % TPC;
    double s = 0;
    for (int r = part; r < KP; r += TPC) s = fma(W[col * KP + r], W[col * KP + r], s);
#pragma unroll
    for (int off = 1; off < TPC; off <<= 1) s += __shfl_xor(s, off, 64);
    if (part == 0) atomicMax(&sh_amax[2], __float_as_uint(static_cast<float>(s)));
  }
  __syncthreads();
  int sweeps = 0, slot = 0;  // slot = (rounds so far) % 3
  for (int sweep = 0; sweep < 16; sweep++) {
    sweeps = sweep + 1;
    for (int rd = 0; rd < KP - 1; rd++, slot = slot == 2 ? 0 : slot + 1) {
      // ---- phase 1: rotation of pair `pr`, applied to its two columns of W
      {
        const int i = top_col(pr, rd), j = bot_col(pr, rd);
        // rows r G + l: neighbouring lanes read neighbouring doubles (no LDS bank conflicts)
        double *wi = W + i * KP + l, *wj = W + j * KP + l;
        double a = 0, b = 0, c = 0, vi[RPT], vj[RPT];
#pragma unroll
        for (int r = 0; r < RPT; r++) {
          vi[r] = wi[r * G];
          vj[r] = wj[r * G];
          a = fma(vi[r], vi[r], a);
          b = fma(vj[r], vj[r], b);
          c = fma(vi[r], vj[r], c);
        }
        a = row16_sum(a);
        b = row16_sum(b);
        c = row16_sum(c);
        if constexpr (G == 32) {
          a += __shfl_xor(a, 16, 64);
          b += __shfl_xor(b, 16, 64);
          c += __shfl_xor(c, 16, 64);
        }
        double cs = 1.0, sn = 0.0;
        const double ab = a * b;
        // (running maximum of the squared column norms over the earlier rounds: never above
        // lambda_max^2; every pair folds the previous slot into this round's, so the slots only grow)
        const float mab = fmaxf(static_cast<float>(a), static_cast<float>(b));
        const float amax = fmaxf(__uint_as_float(sh_amax[slot == 0 ? 2 : slot - 1]), mab);
        if (l == 0) atomicMax(&sh_amax[slot], __float_as_uint(amax));
        if (ab > 0.0 && c != 0.0) {
          // |c| / sqrt(a b): only its size matters (convergence test), float arithmetic.  The
          // ANGLE may be approximate too - any (cs, sn) with cs^2 + sn^2 = 1 to float64 accuracy
          // is an exact rotation, a float-accurate angle only leaves 1e-7 of the off-diagonal
          // for the next sweep - so t comes from float division / square root (float64 ones are
          // ~30-instruction sequences at a quarter rate, a chain of seven per round) and only
          // cs = 1 / sqrt(1 + t^2) is refined to float64 by two Newton steps.
          // The test is on the COUPLING the pair leaves in P, relative to the largest eigenvalue:
          // c = q_i^T P^2 q_j ~ e_ij (lambda_i + lambda_j) with e_ij = q_i^T P q_j, and the
          // column norms are |lambda|, so e_ij / lambda_max ~ |c| / (sqrt(max(a, b)) sqrt(a_max)).
          // (|c| / sqrt(a b) alone never settles for a pair of near-null columns - a Gramian of
          // rank-deficient factors has eigenvalues at 1e-7 of the largest, whose columns of W are
          // rounding noise - and every call then ran all 16 sweeps.)
          const float cf = static_cast<float>(c);
          const float rel = fabsf(cf) * __builtin_amdgcn_rsqf(fmaxf(mab, 1e-37f)) *
                            __builtin_amdgcn_rsqf(fmaxf(amax, 1e-37f));
          if (l == 0) atomicMax(&sh_off, __float_as_uint(rel));
          if (c * c > 1e-30 * ab) {
            const float zeta = static_cast<float>(b - a) / (2.0f * cf);
            const float tf = (zeta >= 0.f ? 1.0f : -1.0f) / (fabsf(zeta) + sqrtf(1.0f + zeta * zeta));
            const double t = static_cast<double>(tf), s1 = fma(t, t, 1.0);
            double y = static_cast<double>(__builtin_amdgcn_rsqf(static_cast<float>(s1)));
            y = y * fma(-0.5 * s1, y * y, 1.5);
            y = y * fma(-0.5 * s1, y * y, 1.5);
            cs = y;
            sn = cs * t;
#pragma unroll
            for (int r = 0; r < RPT; r++) {
              wi[r * G] = cs * vi[r] - sn * vj[r];
              wj[r * G] = sn * vi[r] + cs * vj[r];
            }
          }
        }
        if (l == 0) {
          sh_cs[rd & 1][pr] = cs;
          sh_sn[rd & 1][pr] = sn;
        }
      }
      __syncthreads();
      // ---- phase 2: the same rotations on this thread's Q entries, then the positions move:
      //      top[p] <- top[p + 1], top[NP - 1] <- bot[NP - 1], bot[p] <- bot[p - 1], bot[0] <- top[1]
      {
#pragma unroll
        for (int k = 0; k < PPT; k++) {
          const double cs = sh_cs[rd & 1][PPT * qt + k], sn = sh_sn[rd & 1][PPT * qt + k];
          const double x = qtop[k], y = qbot[k];
          qtop[k] = cs * x - sn * y;
          qbot[k] = sn * x + cs * y;
        }
        // values that cross a thread boundary (lanes of one row are contiguous)
        const double top_from_next = __shfl_down(qtop[0], 1, 64);      // next part's top[first]
        const double bot_from_prev = __shfl_up(qbot[PPT - 1], 1, 64);  // previous part's bot[last]
        const double top1 = PPT > 1 ? qtop[1] : top_from_next;         // (part 0) the row's top[1]
        const double old_top0 = qtop[0], old_bot_last = qbot[PPT - 1];
        // top: shift left
#pragma unroll
        for (int k = 0; k + 1 < PPT; k++) qtop[k] = qtop[k + 1];
        qtop[PPT - 1] = qt == TPR - 1 ? old_bot_last : top_from_next;
        // bot: shift right
#pragma unroll
        for (int k = PPT - 1; k > 0; k--) qbot[k] = qbot[k - 1];
        qbot[0] = qt == 0 ? top1 : bot_from_prev;
        if (qt == 0) qtop[0] = old_top0;  // top[0] never moves
      }
    }
    const float off_max = __uint_as_float(sh_off);
    __syncthreads();
    if (tid == 0) sh_off = 0u;
    __syncthreads();
    if (off_max < 1e-10f) break;  // (eigenvector error ~1e-10: far below the float32 the results are rounded to)
  }
  // ---- Q back to memory (whole sweeps: the positions are those of round 0 again)
#pragma unroll
  for (int k = 0; k < PPT; k++) {
    const int p = PPT * qt + k;
    Qd[top_col(p, 0) * KP + qr] = qtop[k];
    Qd[bot_col(p, 0) * KP + qr] = qbot[k];
  }
  __syncthreads();
  // eigenvalue k = q_k . (P q_k) = q_k . w_k; outputs in float
  for (int k = tid / 16; k < KP; k += 64) {  // 16 lanes per column
    const int ll = tid % 16;
    double s = 0;
    for (int r = ll; r < KP; r += 16) s = fma(Qd[k * KP + r], W[k * KP + r], s);
#pragma unroll
    for (int off = 1; off < 16; off <<= 1) s += __shfl_xor(s, off, 64);
    if (ll == 0) sh_lam[k] = k < K ? static_cast<float>(s) : 0.f;
  }
  __syncthreads();
  for (int i = tid; i < KP * KP; i += 1024) {
    const int k = i / KP, d = i % KP;
    const float q = static_cast<float>(Qd[i]);
    o.Qrows[static_cast<size_t>(k) * KP + d] = q;
    o.Qcols[static_cast<size_t>(d) * KP + k] = q;
  }
  if (tid < KP) o.lam[tid] = sh_lam[tid];
  if (tid == 0) {
    float hi = -3.0e38f, lo = 3.0e38f;
    for (int k = 0; k < K; k++) {
      hi = fmaxf(hi, sh_lam[k]);
      lo = fminf(lo, sh_lam[k]);
    }
    o.stats[0] = hi;
    o.stats[1] = lo;
    o.stats[2] = static_cast<float>(sweeps);
  }
}

// dst[r'][a] = sum_b src[r][b] * Mat[a][b] over rows of a list: src row = src_rows ? src_rows[i].row
// : i, dst row likewise (Task lists: the short rows of a side).  One wave per 64 rows (four 16-row
// tiles share every operand load of Mat: with one tile per wave the kernel was bound by re-reading
// the 64 KB of Mat from L2 for every 16 rows), all KP output columns: 4 x KP / 16 accumulator tiles.
template <int KP>
__global__ __launch_bounds__(256, 2) void rows_times_matT_kernel(const float *__restrict__ src,
                                                                 const Task *__restrict__ src_rows,
                                                                 float *__restrict__ dst,
                                                                 const Task *__restrict__ dst_rows,
                                                                 const float *__restrict__ Mat, int n_rows) {
  constexpr int NTL = KP / 16, RT = 4;
  const int lane = threadIdx.x & 63, g = lane >> 4, m = lane & 15;
  const int w = blockIdx.x * 4 + wave_in_block();
  const int r0 = w * 16 * RT;
  if (r0 >= n_rows) return;
  const float *srow[RT];
#pragma unroll
  for (int q = 0; q < RT; q++) {
    const int ri = min(r0 + 16 * q + m, n_rows - 1);
    srow[q] = src + static_cast<size_t>(src_rows ? src_rows[ri].row : ri) * KP + 4 * g;
  }
  f32x4 acc[RT][NTL];
#pragma unroll
  for (int q = 0; q < RT; q++)
#pragma unroll
    for (int t = 0; t < NTL; t++) acc[q][t] = f32x4{0.f, 0.f, 0.f, 0.f};
  // contraction index b = 16 j + 4 g + s: lane (g, i) holds src[i][16 j + 4 g .. + 3] (A operand)
  // and Mat[16 t + i][16 j + 4 g .. + 3] (B operand of output tile t, n = i)
  for (int j = 0; j < KP / 16; j++) {
    f32x4 a[RT];
#pragma unroll
    for (int q = 0; q < RT; q++) a[q] = *reinterpret_cast<const f32x4 *>(srow[q] + 16 * j);
#pragma unroll
    for (int t = 0; t < NTL; t++) {
      const f32x4 b = *reinterpret_cast<const f32x4 *>(Mat + static_cast<size_t>(16 * t + m) * KP + 16 * j + 4 * g);
#pragma unroll
      for (int q = 0; q < RT; q++) {
        acc[q][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q].x, b.x, acc[q][t], 0, 0, 0);
        acc[q][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q].y, b.y, acc[q][t], 0, 0, 0);
        acc[q][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q].z, b.z, acc[q][t], 0, 0, 0);
        acc[q][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q].w, b.w, acc[q][t], 0, 0, 0);
      }
    }
  }
  // acc[q][t] reg r of lane (g, n): out[r0 + 16 q + 4 g + r][16 t + n]
#pragma unroll
  for (int q = 0; q < RT; q++)
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int row = r0 + 16 * q + 4 * g + r;
      if (row >= n_rows) continue;
      float *drow = dst + static_cast<size_t>(dst_rows ? dst_rows[row].row : row) * KP;
#pragma unroll
      for (int t = 0; t < NTL; t++) drow[16 * t + m] = acc[q][t][r];
    }
}

struct EigShortParams {
  const Task *tasks;        // the short rows (a slice of the side's task list)
  int32_t n_tasks;
  const int32_t *indices;
  const float *data;
  const float *table;       // V~ = other Q: [n_other, KP]
  const float *lam;         // [KP]
  const float *reg;         // per row
  float *xt;                // [n_tasks, KP]: x~ of task i (in: the rotated warm start for CG)
  float bias;
  int32_t K;
  int32_t max_cg_steps, warm_start;
  int32_t *err_flag;
};

// ---------------------------------------------------------------------------------------
// Cholesky of a short row in the eigenbasis (see the header comment).  One wave per row.
// TT = 1: up to 16 stored entries, TT = 2: up to 32.  The n x n system lives in the
// accumulator layout of ials_chol16.hpp in its virtual basis: entry e = TT m' + I is row m' of
// tile row I, so lane (g, m) of gather set I loads entry TT m + I.
template <int KP, int TT>
__global__ __launch_bounds__(256, 2) void ials_wb_short_kernel(EigShortParams p) {
  using C = Chol16Geo<TT>;
  constexpr int NJ = KP / 16;  // 16-dim segments: lane (g, m) holds dims 16 j + 4 g .. + 3
  extern __shared__ __attribute__((aligned(16))) float wb_lds[];
  const int wv = wave_in_block(), lane = threadIdx.x & 63, g = lane >> 4, m = lane & 15;
  const int ti = blockIdx.x * 4 + wv;
  if (ti >= p.n_tasks) return;
  float *sm = wb_lds + wv * (C::LDS_FLOATS + 32 * TT);
  float *ybuf = sm + C::LDS_FLOATS;  // 16 TT floats: the solution of the small system
  const Task task = p.tasks[ti];
  const int n = task.end - task.begin;
  const float reg = p.reg[task.row];
  // d = 1 / (lambda + reg) for this lane's dims
  f32x4 d[NJ];
#pragma unroll
  for (int j = 0; j < NJ; j++) {
    const f32x4 lm = *reinterpret_cast<const f32x4 *>(p.lam + 16 * j + 4 * g);
    d[j] = f32x4{1.0f / (lm.x + reg), 1.0f / (lm.y + reg), 1.0f / (lm.z + reg), 1.0f / (lm.w + reg)};
  }
  // gather: set I holds the entries TT m + I
  f32x4 v[TT][NJ];
  float cw[TT], sq[TT];  // confidence of the lane's entry and its square root (0 past the end)
#pragma unroll
  for (int I = 0; I < TT; I++) {
    const int e = TT * m + I;
    const bool ok = e < n;
    const int q = task.begin + (ok ? e : 0);
    const int idx = n > 0 ? p.indices[q] : 0;
    cw[I] = ok ? p.data[q] : 0.f;
    sq[I] = sqrtf(cw[I]);
    const float *row = p.table + static_cast<size_t>(static_cast<unsigned>(idx)) * KP + 4 * g;
#pragma unroll
    for (int j = 0; j < NJ; j++) {
      v[I][j] = *reinterpret_cast<const f32x4 *>(row + 16 * j);
      if (!ok) v[I][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  }
  // G S G (lower tiles: slot tix(a, b), a <= b, = tile (row block b, column block a)): operand
  // A = sq * d * v of the row block, operand B = sq * v of the column block
  f32x4 acc[Geo<TT>::NT];
#pragma unroll
  for (int t = 0; t < Geo<TT>::NT; t++) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int j = 0; j < NJ; j++) {
    f32x4 av[TT], bv[TT];
#pragma unroll
    for (int I = 0; I < TT; I++) {
      bv[I] = v[I][j] * sq[I];
      av[I] = bv[I] * d[j];
    }
#pragma unroll
    for (int a = 0; a < TT; a++)
#pragma unroll
      for (int b = a; b < TT; b++) {
        f32x4 &t = acc[C::tix(a, b)];
        t = __builtin_amdgcn_mfma_f32_16x16x4f32(av[b].x, bv[a].x, t, 0, 0, 0);
        t = __builtin_amdgcn_mfma_f32_16x16x4f32(av[b].y, bv[a].y, t, 0, 0, 0);
        t = __builtin_amdgcn_mfma_f32_16x16x4f32(av[b].z, bv[a].z, t, 0, 0, 0);
        t = __builtin_amdgcn_mfma_f32_16x16x4f32(av[b].w, bv[a].w, t, 0, 0, 0);
      }
  }
  // right-hand side G^-1 w = (bias + c) / sqrt(c) of entry TT m + I (0 past the end); the solve
  // adds 1 to the diagonal (its `reg`) - entries past n get the unit diagonal of padded dims
  float b4[TT];
#pragma unroll
  for (int I = 0; I < TT; I++) b4[I] = cw[I] > 0.f ? (p.bias + cw[I]) / sq[I] : 0.f;
  solve_row_cholesky16<TT>(acc, b4, 1.0f, sm, ybuf, max(n, 1), p.err_flag);
  __threadfence_block();
  // y = G s of the lane's entries, then x~ = D V~^T y: sum over the 16 lanes m of a group
  float ye[TT];
#pragma unroll
  for (int I = 0; I < TT; I++) ye[I] = sq[I] * ybuf[TT * m + I];
  float *xrow = p.xt + static_cast<size_t>(ti) * KP;
#pragma unroll
  for (int j = 0; j < NJ; j++) {
    f32x4 s = v[0][j] * ye[0];
#pragma unroll
    for (int I = 1; I < TT; I++) s += v[I][j] * ye[I];
#pragma unroll
    for (int off = 1; off < 16; off <<= 1) {
      s.x += __shfl_xor(s.x, off, 64);
      s.y += __shfl_xor(s.y, off, 64);
      s.z += __shfl_xor(s.z, off, 64);
      s.w += __shfl_xor(s.w, off, 64);
    }
    if (m == 0) *reinterpret_cast<f32x4 *>(xrow + 16 * j + 4 * g) = s * d[j];
  }
}

// ---------------------------------------------------------------------------------------
// CG of a short row in the eigenbasis (hpp:199-264 with P diagonal).  One wave per row, up to 32
// entries in registers; lane l holds dims DPL l .. (as ials_cg_short_kernel).
template <int KP>
__global__ __launch_bounds__(256, 2) void ials_cg_eig_short_kernel(EigShortParams p) {
  constexpr int DPL = KP / 64;
  static_assert(KP % 64 == 0, "lanes over the dims");
  const int wv = wave_in_block(), lane = threadIdx.x & 63;
  const int ti = blockIdx.x * 4 + wv;
  if (ti >= p.n_tasks) return;
  const Task task = p.tasks[ti];
  const int n = task.end - task.begin;
  float *xrow = p.xt + static_cast<size_t>(ti) * KP + DPL * lane;
  if (n == 0) {  // hpp:207-210
#pragma unroll
    for (int c = 0; c < DPL; c++) xrow[c] = 0.f;
    return;
  }
  const float reg = p.reg[task.row];
  float dm[DPL], x[DPL], r[DPL], pv[DPL], Ap[DPL], bvec[DPL];
#pragma unroll
  for (int c = 0; c < DPL; c++) {
    dm[c] = p.lam[DPL * lane + c];  // P is diag(lambda) here; reg is added last (hpp:222-223)
    x[c] = p.warm_start ? xrow[c] : 0.f;
    bvec[c] = 0.f;
  }
  // the row's entries: index / confidence of entry `lane` (n <= 32), gathered 8 at a time
  const int qe = task.begin + min(lane, n - 1);
  const int my_idx = p.indices[qe];
  const float my_c = lane < n ? p.data[qe] : 0.f;
  float v[32][DPL], cj[32];
#pragma unroll
  for (int j0 = 0; j0 < 32; j0 += 8) {
    if (j0 < n) {
#pragma unroll
      for (int j = j0; j < j0 + 8; j++) {
        const unsigned idx = static_cast<unsigned>(__builtin_amdgcn_readlane(my_idx, j));
        const float *src = p.table + static_cast<size_t>(idx) * KP + DPL * lane;
#pragma unroll
        for (int c = 0; c < DPL; c++) v[j][c] = src[c];
      }
    } else {
#pragma unroll
      for (int j = j0; j < j0 + 8; j++)
#pragma unroll
        for (int c = 0; c < DPL; c++) v[j][c] = 0.f;
    }
  }
#pragma unroll
  for (int j = 0; j < 32; j++) {
    cj[j] = readlane_f(my_c, j);
    const float w = j < n ? p.bias + cj[j] : 0.f;
#pragma unroll
    for (int c = 0; c < DPL; c++) bvec[c] = fmaf(w, v[j][c], bvec[c]);
  }
  auto matvec = [&](const float (&vec)[DPL], float (&out)[DPL]) {
#pragma unroll
    for (int c = 0; c < DPL; c++) out[c] = dm[c] * vec[c];  // (+ reg * vec last, see ials_cg_eig16_kernel)
#pragma unroll
    for (int j0 = 0; j0 < 32; j0 += 8) {
      if (j0 >= n) continue;  // (wave-uniform)
      float dot[8];
#pragma unroll
      for (int j = 0; j < 8; j++) {
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < DPL; c++) s = fmaf(v[j0 + j][c], vec[c], s);
        dot[j] = s;
      }
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1)
#pragma unroll
        for (int j = 0; j < 8; j++) dot[j] += __shfl_xor(dot[j], off, 64);
#pragma unroll
      for (int j = 0; j < 8; j++) {
        const float w = cj[j0 + j] * dot[j];
#pragma unroll
        for (int c = 0; c < DPL; c++) out[c] = fmaf(w, v[j0 + j][c], out[c]);
      }
    }
#pragma unroll
    for (int c = 0; c < DPL; c++) out[c] = fmaf(reg, vec[c], out[c]);
  };
  auto dotw = [&](const float (&a)[DPL], const float (&b)[DPL]) {
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < DPL; c++) s = fmaf(a[c], b[c], s);
    return wave_sum(s);
  };
  if (p.warm_start) {
    matvec(x, Ap);
#pragma unroll
    for (int c = 0; c < DPL; c++) r[c] = bvec[c] - Ap[c];
  } else {
#pragma unroll
    for (int c = 0; c < DPL; c++) r[c] = bvec[c];
  }
#pragma unroll
  for (int c = 0; c < DPL; c++) pv[c] = r[c];
  float r2 = dotw(r, r);
  bool singular = false;
  for (int it = 0; it < p.max_cg_steps; it++) {
    if (r2 <= 1e-20f) break;  // hpp:238
    matvec(pv, Ap);
    const float denom = dotw(pv, Ap);
    if (!(denom > 0.f) || !__builtin_isfinite(denom)) {  // hpp:250-254
      singular = true;
      break;
    }
    const float alpha = r2 / denom;
#pragma unroll
    for (int c = 0; c < DPL; c++) {
      x[c] = fmaf(alpha, pv[c], x[c]);
      r[c] = fmaf(-alpha, Ap[c], r[c]);
    }
    const float r2n = dotw(r, r);
    if (r2n <= 1e-20f) break;  // hpp:258
    const float beta = r2n / r2;  // hpp:261
#pragma unroll
    for (int c = 0; c < DPL; c++) pv[c] = fmaf(beta, pv[c], r[c]);
    r2 = r2n;
  }
  if (singular && lane == 0) atomicOr(p.err_flag, 4);
#pragma unroll
  for (int c = 0; c < DPL; c++) xrow[c] = x[c];
}


// ---------------------------------------------------------------------------------------
// Rows of at most 16 stored entries, FOUR per wave: the 16 lanes (g, m) of a DPP row own row g of
// the wave; lane m holds the dims DPL m .. DPL m + DPL - 1 of every gathered (eigenbasis) row and
// of the CG vectors.  A dot product over K is DPL FMAs + four DPP adds (row16_sum: no LDS
// crossbar, no 6-step 64-lane butterfly), and four rows' dependency chains interleave in one
// instruction stream - the one-row-per-wave forms above spent their time waiting on both.
// LPR = lanes per row: 16 (four rows per wave) or 32 (two); NEV = most entries a row may have
// (LPR, or 8 with LPR = 16: the gathered rows are NEV x DPL registers per lane - 128 of the ~200
// at KP = 128 - and rows of <= 8 entries, more than half of configs[3]'s users, run at twice
// the waves per SIMD with the 8-entry form; the 9..16 class on 32 lanes per row - half the
// registers, three waves of two rows - was measured: CG equal, Cholesky 49.1 -> 51.1 ms).
template <int KP, int LPR = 16, int NEV = LPR> struct Eig16 {
  static constexpr int DPL = KP / LPR, NE = NEV;
  static_assert((LPR == 16 || LPR == 32) && DPL >= 2 && (NEV == LPR || (NEV == 8 && LPR == 16)), "lanes per row");
  // entry j of this lane's row: broadcast from lane (g, j)
  static __device__ __forceinline__ int bcast_i(int v, int j) {
    return __builtin_amdgcn_ds_bpermute(((threadIdx.x & (64 - LPR)) | j) << 2, v);
  }
  // sum over the lanes of the row, result in every lane
  static __device__ __forceinline__ float row_sum(float v) {
    v = row16_sum(v);
    if constexpr (LPR == 32) v += __shfl_xor(v, 16, 64);
    return v;
  }
  // DPL consecutive floats (16-byte loads where the width allows)
  static __device__ __forceinline__ void load_dpl(const float *src, float (&out)[DPL]) {
    if constexpr (DPL % 4 == 0) {
#pragma unroll
      for (int c = 0; c < DPL; c += 4) {
        const f32x4 t4 = *reinterpret_cast<const f32x4 *>(src + c);
        out[c] = t4.x; out[c + 1] = t4.y; out[c + 2] = t4.z; out[c + 3] = t4.w;
      }
    } else {
      static_assert(DPL == 2, "two floats per lane");
      const f32x2 t2 = *reinterpret_cast<const f32x2 *>(src);
      out[0] = t2.x; out[1] = t2.y;
    }
  }
  static __device__ __forceinline__ void store_dpl(float *dst, const float (&in)[DPL]) {
    if constexpr (DPL % 4 == 0) {
#pragma unroll
      for (int c = 0; c < DPL; c += 4) *reinterpret_cast<f32x4 *>(dst + c) = f32x4{in[c], in[c + 1], in[c + 2], in[c + 3]};
    } else {
      *reinterpret_cast<f32x2 *>(dst) = f32x2{in[0], in[1]};
    }
  }
  static __device__ __forceinline__ float bcast_f(float v, int j) {
    return __builtin_bit_cast(float, bcast_i(__builtin_bit_cast(int, v), j));
  }
  // v[j][:] = table[idx_j][DPL m ..] for j < nmax (rows past the group's own n read as zero)
  static __device__ __forceinline__ void gather(const EigShortParams &p, const Task &task, int n, int nmax,
                                                float (&v)[NE][DPL], float (&cj)[NE]) {
    const int m = threadIdx.x & (LPR - 1);
    const int qe = task.begin + min(m, max(n, 1) - 1);
    const int my_idx = n > 0 ? p.indices[qe] : 0;  // (lanes past the end repeat the last entry: a valid row)
    const float my_c = m < n ? p.data[qe] : 0.f;
    // All loads of a wave are issued together (eight entries unconditionally, the other eight
    // under ONE wave-uniform branch) and masked afterwards: a load under its own branch makes
    // the wave wait for it before the next one is issued - sixteen exposed cache / HBM latencies
    // per row (40 us per wave measured that way).
    const float *src[NE];
#pragma unroll
    for (int j = 0; j < NE; j++) {
      cj[j] = bcast_f(my_c, j);
      const unsigned idx = static_cast<unsigned>(bcast_i(my_idx, j));
      src[j] = p.table + static_cast<size_t>(idx) * KP + DPL * m;
    }
    auto load8 = [&](int j0) {
#pragma unroll
      for (int j = j0; j < j0 + 8; j++) load_dpl(src[j], v[j]);
    };
    load8(0);
#pragma unroll
    for (int j0 = 8; j0 < NE; j0 += 8)
      if (nmax > j0) load8(j0);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < NE; j++) {
      const bool ok = j < n && (j < 8 || nmax > (j & ~7));
#pragma unroll
      for (int c = 0; c < DPL; c++) v[j][c] = ok ? v[j][c] : 0.f;
    }
  }
};

template <int KP, int LPR = 16, int NEV = LPR>
__global__ __launch_bounds__(256, NEV == 8 ? 4 : 2) void ials_cg_eig16_kernel(EigShortParams p) {
  using E = Eig16<KP, LPR, NEV>;
  constexpr int DPL = E::DPL, NE = E::NE;
  const int lane = threadIdx.x & 63, g = lane / LPR, m = lane % LPR;
  const int ti = (blockIdx.x * 4 + wave_in_block()) * (64 / LPR) + g;
  const bool exists = ti < p.n_tasks;
  const Task task = p.tasks[min(ti, p.n_tasks - 1)];
  const int n = exists ? task.end - task.begin : 0;
  int nmax = n;
#pragma unroll
  for (int off = LPR; off < 64; off <<= 1) nmax = max(nmax, __shfl_xor(nmax, off, 64));
  nmax = __builtin_amdgcn_readfirstlane(nmax);
  float *xrow = p.xt + static_cast<size_t>(min(ti, p.n_tasks - 1)) * KP + DPL * m;
  const float reg = p.reg[task.row];
  float v[NE][DPL], cj[NE];
  E::gather(p, task, n, nmax, v, cj);
  float dm[DPL], x[DPL], r[DPL], pv[DPL], Ap[DPL];
#pragma unroll
  for (int c = 0; c < DPL; c++) {
    dm[c] = p.lam[DPL * m + c];  // P is diag(lambda) in this basis; + reg last (hpp:222-223)
    x[c] = (p.warm_start && n > 0) ? xrow[c] : 0.f;
    r[c] = 0.f;
  }
  // b = sum (bias + c) v   (hpp:212-219)
#pragma unroll
  for (int j = 0; j < NE; j++) {
    if (j < nmax) {  // (wave-uniform guard, no `break`: the loop must unroll fully or v[][] goes to scratch)
      const float w = j < n ? p.bias + cj[j] : 0.f;
#pragma unroll
      for (int c = 0; c < DPL; c++) r[c] = fmaf(w, v[j][c], r[c]);
    }
  }
  auto matvec = [&](const float (&vec)[DPL], float (&out)[DPL]) {  // hpp:222-228, 240-247
    // (lambda * vec and the gathered terms are summed at their own magnitude, reg * vec comes
    // LAST in one fma: with reg_r = 100 and a solution thousands of times smaller than the warm
    // start, every term added after reg * vec is rounded at that magnitude and the error of the
    // first residual survives into x - see the comment in ials_short_kernels.hpp)
#pragma unroll
    for (int c = 0; c < DPL; c++) out[c] = dm[c] * vec[c];
#pragma unroll
    for (int j = 0; j < NE; j++) {
      if (j < nmax) {
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < DPL; c++) s = fmaf(v[j][c], vec[c], s);
        const float w = cj[j] * E::row_sum(s);
#pragma unroll
        for (int c = 0; c < DPL; c++) out[c] = fmaf(w, v[j][c], out[c]);
      }
    }
#pragma unroll
    for (int c = 0; c < DPL; c++) out[c] = fmaf(reg, vec[c], out[c]);
  };
  auto dotw = [&](const float (&a)[DPL], const float (&b)[DPL]) {
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < DPL; c++) s = fmaf(a[c], b[c], s);
    return E::row_sum(s);
  };
  if (p.warm_start) {
    matvec(x, Ap);
#pragma unroll
    for (int c = 0; c < DPL; c++) r[c] -= Ap[c];
  }
#pragma unroll
  for (int c = 0; c < DPL; c++) pv[c] = r[c];
  float r2 = dotw(r, r);
  bool run = n > 0, singular = false;  // per row (lane-group uniform)
  for (int it = 0; it < p.max_cg_steps; it++) {
    run = run && !(r2 <= 1e-20f);  // hpp:238
    if (!__any(run)) break;
    matvec(pv, Ap);
    const float denom = dotw(pv, Ap);
    const bool bad = run && (!(denom > 0.f) || !__builtin_isfinite(denom));  // hpp:250-254
    singular = singular || bad;
    run = run && !bad;
    const float alpha = run ? r2 / denom : 0.f;
#pragma unroll
    for (int c = 0; c < DPL; c++) {
      x[c] = fmaf(alpha, pv[c], x[c]);
      r[c] = fmaf(-alpha, Ap[c], r[c]);
    }
    const float r2n = dotw(r, r);
    run = run && !(r2n <= 1e-20f);  // hpp:258
    const float beta = run ? r2n / r2 : 0.f;  // hpp:261
    if (run) {
#pragma unroll
      for (int c = 0; c < DPL; c++) pv[c] = fmaf(beta, pv[c], r[c]);
      r2 = r2n;
    }
  }
  if (__any(singular)) {
    if (lane == 0) atomicOr(p.err_flag, 4);
  }
  if (exists) {
    if (n == 0) {  // hpp:207-210
#pragma unroll
      for (int c = 0; c < DPL; c++) x[c] = 0.f;
    }
    E::store_dpl(xrow, x);
  }
}

// Cholesky (low-rank form, see the header comment) of rows with at most 16 entries, four per
// wave.  The n x n system I + G S G lives in registers, lane m of the row's 16 owning matrix row m;
// it is factorised column by column with one lane broadcast per (column, row) pair.
template <int KP, int LPR = 16, int NEV = LPR>
__global__ __launch_bounds__(256, NEV == 8 ? 4 : 2) void ials_wb_eig16_kernel(EigShortParams p) {
  using E = Eig16<KP, LPR, NEV>;
  constexpr int DPL = E::DPL, NE = E::NE;
  const int lane = threadIdx.x & 63, g = lane / LPR, m = lane % LPR;
  const int ti = (blockIdx.x * 4 + wave_in_block()) * (64 / LPR) + g;
  const bool exists = ti < p.n_tasks;
  const Task task = p.tasks[min(ti, p.n_tasks - 1)];
  const int n = exists ? task.end - task.begin : 0;
  int nmax = n;
#pragma unroll
  for (int off = LPR; off < 64; off <<= 1) nmax = max(nmax, __shfl_xor(nmax, off, 64));
  nmax = __builtin_amdgcn_readfirstlane(nmax);
  float *xrow = p.xt + static_cast<size_t>(min(ti, p.n_tasks - 1)) * KP + DPL * m;
  const float reg = p.reg[task.row];
  float v[NE][DPL], cj[NE];
  E::gather(p, task, n, nmax, v, cj);
  float d[DPL];
#pragma unroll
  for (int c = 0; c < DPL; c++) d[c] = 1.0f / (p.lam[DPL * m + c] + reg);
  // A = I + G S G (G = C^1/2), lane m keeps row m; rhs G^-1 w of entry m
  const float my_c = m < n ? p.data[task.begin + m] : 0.f;  // confidence of entry m (this lane's matrix row)
  const float my_sq = sqrtf(my_c);
  float sqj[NE];
#pragma unroll
  for (int j = 0; j < NE; j++) sqj[j] = E::bcast_f(my_sq, j);
  float A[NE];
#pragma unroll
  for (int q = 0; q < NE; q++) A[q] = q == m ? 1.0f : 0.f;
#pragma unroll
  for (int pp = 0; pp < NE; pp++) {
    if (pp < nmax) {  // (wave-uniform guards, no `break`: the loops must unroll fully)
      float dv[DPL];
#pragma unroll
      for (int c = 0; c < DPL; c++) dv[c] = d[c] * v[pp][c] * sqj[pp];
#pragma unroll
      for (int q = pp; q < NE; q++) {
        if (q < nmax) {
          float s = 0.f;
#pragma unroll
          for (int c = 0; c < DPL; c++) s = fmaf(dv[c], v[q][c], s);
          s = E::row_sum(s) * sqj[q];  // (G S G)[pp][q], in every lane of the row
          A[q] += m == pp ? s : 0.f;
          if (q != pp) A[pp] += m == q ? s : 0.f;
        }
      }
    }
  }
  float rhs = my_c > 0.f ? (p.bias + my_c) / my_sq : 0.f;
  // ---- Cholesky A = L L^T by columns; lane m ends with row m of L in A[0 .. m]
  bool bad = false;
#pragma unroll
  for (int k = 0; k < NE; k++) {
    if (k < nmax) {
      const float piv = E::bcast_f(A[k], k);
      bad = bad || !(piv > 0.f);
      const float rinv = __builtin_amdgcn_rsqf(piv);
      const float lmk = A[k] * rinv;  // L[m][k] for m >= k (lane k: sqrt(piv))
      A[k] = lmk;
#pragma unroll
      for (int q = k + 1; q < NE; q++) {
        if (q < nmax) {
          const float lqk = E::bcast_f(lmk, q);  // L[q][k]
          A[q] = fmaf(-lmk, lqk, A[q]);          // (only the part q <= m of row m is used below)
        }
      }
    }
  }
  // ---- L z = rhs, then L^T s = z
#pragma unroll
  for (int k = 0; k < NE; k++) {
    if (k < nmax) {
      const float zk = E::bcast_f(rhs * __builtin_amdgcn_rcpf(A[k]), k);  // lane k holds L[k][k] in A[k]
      rhs = m == k ? zk : (m > k ? fmaf(-A[k], zk, rhs) : rhs);
    }
  }
  // s_k = (z_k - sum_{q > k} L[q][k] s_q) / L[k][k]: lane q holds L[q][k] and (once final) s_q, so
  // the sum is one product per lane and a row sum - no per-element hand-over between lanes
#pragma unroll
  for (int k = NE - 1; k >= 0; k--) {
    if (k < nmax) {
      const float part = E::row_sum(m > k && m < nmax ? A[k] * rhs : 0.f);
      rhs = m == k ? (rhs - part) * __builtin_amdgcn_rcpf(A[k]) : rhs;
    }
  }
  if (__any(bad)) {
    if (lane == 0) atomicOr(p.err_flag, 1);
  }
  // y = G s of entry m; x~ = D V~^T y
  const float ym = m < n ? my_sq * rhs : 0.f;
  float acc[DPL];
#pragma unroll
  for (int c = 0; c < DPL; c++) acc[c] = 0.f;
#pragma unroll
  for (int j = 0; j < NE; j++) {
    if (j < nmax) {
      const float yj = E::bcast_f(ym, j);
#pragma unroll
      for (int c = 0; c < DPL; c++) acc[c] = fmaf(yj, v[j][c], acc[c]);
    }
  }
  if (exists) {
#pragma unroll
    for (int c = 0; c < DPL; c++) acc[c] *= d[c];
    E::store_dpl(xrow, acc);
  }
}

}  // namespace ials
}  // namespace irs
