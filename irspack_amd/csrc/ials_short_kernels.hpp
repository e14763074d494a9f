// Conjugate gradient for SHORT rows (at most SHORT_MAX stored entries), matrix free -
// Solver::step_cg of /root/reference/cpp_source/als/IALSTrainer.hpp:170-271 evaluated the way
// the reference evaluates it:  A p = P p + reg p + sum_q c_q (v_q . p) v_q  without ever
// forming A.  The general kernel builds the K x K system in MFMA accumulators; for a row with
// ten entries at K = 128 that is 36 KB of Gramian loads, a 46 KB LDS spill and a 463-register
// wave (three waves per CU) to add ten outer products - 15 us per row.  Here a wave keeps the
// row's gathered factor rows in registers (lane l holds dims DPL l .. DPL l + DPL - 1 of each),
// P sits in LDS once per workgroup, and a step costs KP readlane + KP / DPL LDS reads of P
// plus a 16-value butterfly for the dots v_q . p: 16 resident waves per CU, ~2 us per row.
#pragma once
#include "ials_kernels.hpp"

namespace irs {
namespace ials {

constexpr int SHORT_MAX = 32;    // rows up to this many stored entries take a short kernel
constexpr int SHORT_WAVES = 8;   // waves per workgroup (they share P in LDS)

template <int KP> struct ShortGeo {
  static constexpr int DPL = KP >= 128 ? KP / 64 : 1;  // dims per lane
  static constexpr int NL = KP / DPL;                  // lanes that hold dims
  static constexpr int G = 4 / DPL;                    // rows of P per 16-byte LDS read
  // P + the parked vectors of the waves (up to two rows each)
  static constexpr size_t LDS_BYTES = (static_cast<size_t>(KP) * KP + SHORT_WAVES * 2 * KP) * sizeof(float);
};

// sum over the wave of 16 per-lane values at once: after the call out[j] (same in every
// lane) is the wave total of val[j].  Halving butterfly: each step a lane keeps half of its
// values and hands the other half to its partner (34 shuffles instead of 96).
__device__ __forceinline__ void wave_sum16(const float (&val)[16], float (&out)[16]) {
  const int lane = threadIdx.x & 63;
  float a[8];
  {
    const bool hi = (lane & 32) != 0;
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const float send = hi ? val[j] : val[j + 8];
      const float keep = hi ? val[j + 8] : val[j];
      a[j] = keep + __shfl_xor(send, 32, 64);
    }
  }
  float b[4];
  {
    const bool hi = (lane & 16) != 0;
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const float send = hi ? a[j] : a[j + 4];
      const float keep = hi ? a[j + 4] : a[j];
      b[j] = keep + __shfl_xor(send, 16, 64);
    }
  }
  float c[2];
  {
    const bool hi = (lane & 8) != 0;
#pragma unroll
    for (int j = 0; j < 2; j++) {
      const float send = hi ? b[j] : b[j + 2];
      const float keep = hi ? b[j + 2] : b[j];
      c[j] = keep + __shfl_xor(send, 8, 64);
    }
  }
  float d;
  {
    const bool hi = (lane & 4) != 0;
    const float send = hi ? c[0] : c[1];
    const float keep = hi ? c[1] : c[0];
    d = keep + __shfl_xor(send, 4, 64);
  }
  d += __shfl_xor(d, 2, 64);
  d += __shfl_xor(d, 1, 64);
  // lane l now holds the total of value j(l) = 8 b5 + 4 b4 + 2 b3 + b2 (b_i = bit i of l)
#pragma unroll
  for (int j = 0; j < 16; j++) {
    const int src = ((j >> 3) & 1) * 32 + ((j >> 2) & 1) * 16 + ((j >> 1) & 1) * 8 + (j & 1) * 4;
    out[j] = readlane_f(d, src);
  }
}

// CAP: entries a row may have (16: registers for 16 gathered rows, 4 waves per SIMD; 32: 3).
// NR: rows a wave solves side by side.  One row per wave reads all of P from LDS for every
// product (64 KB at K = 128, 512 LDS clocks: the kernel is bound by that); with two rows each
// ds_read_b128 of P feeds both rows' FMAs, so the LDS traffic per row halves while the vector
// work per row stays.  The rows of a wave are neighbours in the longest-first task list (about
// the same length); their scalars (alpha, beta, exit tests) are wave-uniform per row.
// (Tried and dropped: warming the cache with the next rows' gather lines while a row is solved -
// C4 shape 42 -> 46 ms; the rows in flight already exceed an XCD's 4 MB of L2.)
// P lives in LDS as 16-byte groups [k / G][lane][G rows x DPL columns] so that a lane reads
// its columns of G consecutive rows with one ds_read_b128, and the vector of a product is
// parked in LDS and read back four values at a time (a broadcast read) instead of KP
// v_readlane: ~100 LDS reads + 2 KP / DPL... FMAs per product.
template <int KP, int CAP, int NR>
__global__ __launch_bounds__(64 * SHORT_WAVES, CAP == 16 ? 4 : (NR == 1 ? 3 : 2)) void ials_cg_short_kernel(
    SolveParams p, const float *__restrict__ P_nat, int first_task, int n_short) {
  using SG = ShortGeo<KP>;
  constexpr int DPL = SG::DPL, NL = SG::NL, G = SG::G;
  static_assert((KP % 16) == 0 && NL <= 64 && (CAP == 16 || CAP == 32) && (NR == 1 || NR == 2),
                "unsupported shape");
  extern __shared__ __attribute__((aligned(16))) float short_smem[];
  float *Psh = short_smem;                       // KP * KP
  const int lane = threadIdx.x & 63, wv = wave_in_block();
  float *vsh = short_smem + KP * KP + wv * (NR * KP);   // this wave's parked vectors
  for (int i = threadIdx.x; i < KP * KP; i += 64 * SHORT_WAVES) {
    const int k = i / KP, d = i % KP;
    Psh[((k / G) * NL + d / DPL) * 4 + (k % G) * DPL + d % DPL] = P_nat[i];
  }
  __syncthreads();
  const int dl = lane < NL ? lane : NL - 1;  // idle lanes (KP < 64) shadow the last one, masked
  const bool act = lane < NL;
  const int waves_total = gridDim.x * SHORT_WAVES;
  for (int ti = (blockIdx.x * SHORT_WAVES + wv) * NR; ti < n_short; ti += waves_total * NR) {
    int n[NR];
    float *xrow[NR];
    float reg[NR];
    float x[NR][DPL], v[NR][CAP][DPL], cj[NR][CAP], bvec[NR][DPL];
    bool live[NR];  // the row exists and has entries
#pragma unroll
    for (int rr = 0; rr < NR; rr++) {
      const bool exists = ti + rr < n_short;
      const Task task = p.tasks[first_task + min(ti + rr, n_short - 1)];
      n[rr] = exists ? task.end - task.begin : 0;
      xrow[rr] = p.target + static_cast<size_t>(task.row) * KP + DPL * dl;
      live[rr] = n[rr] > 0;
      if (exists && n[rr] == 0 && act) {  // hpp:207-210
#pragma unroll
        for (int c = 0; c < DPL; c++) xrow[rr][c] = 0.f;
      }
      reg[rr] = p.reg[task.row];
      // (index, value) of entry `lane`; entries past the end repeat the last index (a valid
      // address) with weight 0
      const int qe = task.begin + min(lane, max(n[rr], 1) - 1);
      const int my_idx = live[rr] ? p.indices[qe] : 0;
      const float my_c = lane < n[rr] ? p.data[qe] : 0.f;
      auto gather8 = [&](int j0) {
#pragma unroll
        for (int j = j0; j < j0 + 8; j++) {
          const unsigned idx = static_cast<unsigned>(__builtin_amdgcn_readlane(my_idx, j));
          const float *src = p.other + static_cast<size_t>(idx) * KP + DPL * dl;
          if constexpr (DPL == 2) {
            const float2 t = *reinterpret_cast<const float2 *>(src);
            v[rr][j][0] = t.x;
            v[rr][j][1] = t.y;
          } else {
#pragma unroll
            for (int c = 0; c < DPL; c++) v[rr][j][c] = src[c];
          }
        }
      };
      gather8(0);
#pragma unroll
      for (int j0 = 8; j0 < CAP; j0 += 8) {
        if (n[rr] > j0) {
          gather8(j0);
        } else {
#pragma unroll
          for (int j = j0; j < j0 + 8; j++)
#pragma unroll
            for (int c = 0; c < DPL; c++) v[rr][j][c] = 0.f;
        }
      }
#pragma unroll
      for (int c = 0; c < DPL; c++) x[rr][c] = (p.warm_start && act && live[rr]) ? xrow[rr][c] : 0.f;
#pragma unroll
      for (int j = 0; j < CAP; j++) cj[rr][j] = readlane_f(my_c, j);  // wave-uniform, 0 past the end
      if (!act || !live[rr]) {
#pragma unroll
        for (int j = 0; j < CAP; j++)
#pragma unroll
          for (int c = 0; c < DPL; c++) v[rr][j][c] = 0.f;
      }
      // b = sum (bias + c) v   (hpp:212-219)
#pragma unroll
      for (int c = 0; c < DPL; c++) bvec[rr][c] = 0.f;
#pragma unroll
      for (int j = 0; j < CAP; j++) {
        const float w = j < n[rr] ? p.bias + cj[rr][j] : 0.f;
#pragma unroll
        for (int c = 0; c < DPL; c++) bvec[rr][c] = fmaf(w, v[rr][j][c], bvec[rr][c]);
      }
    }
    bool any_live = false;
#pragma unroll
    for (int rr = 0; rr < NR; rr++) any_live = any_live || live[rr];
    if (!any_live) continue;
    // out = P vec + reg vec + sum c_q (v_q . vec) v_q   (hpp:222-228, 240-247), all rows at once
    auto matvec = [&](const float (&vec)[NR][DPL], float (&out)[NR][DPL]) {
      if (act) {
#pragma unroll
        for (int rr = 0; rr < NR; rr++)
#pragma unroll
          for (int c = 0; c < DPL; c++) vsh[rr * KP + DPL * dl + c] = vec[rr][c];
      }
      // P vec and the gathered terms are summed from zero, at their own magnitude, and reg vec
      // is added LAST in one fma (the reference adds it between the two, hpp:222-228, 240-247).
      // Starting the accumulator at reg vec rounds every one of the KP products of P at the
      // magnitude of reg vec: with reg_r = 100 (10^6 items) and a solution 5,000 times smaller
      // than the warm start that was 8 x the oracle's error on the full configs[3] matrix
      // (2.4e-3 against 3e-4 of a 2e-7-norm row; tests/test_gpu_fullsize.py::
      // test_ials_k128_c4_full_matrix_vs_oracle): the error of the FIRST residual r0 = b - A x0
      // is never seen by the recursively updated residual and survives into x as A^-1 e(r0).
#pragma unroll
      for (int rr = 0; rr < NR; rr++)
#pragma unroll
        for (int c = 0; c < DPL; c++) out[rr][c] = 0.f;
      const f32x4 *pg = reinterpret_cast<const f32x4 *>(Psh) + dl;  // group (k / G, lane)
      const f32x4 *vg = reinterpret_cast<const f32x4 *>(vsh);
#pragma unroll 8
      for (int i = 0; i < KP / 4; i++) {
        f32x4 s4[NR];  // vec[4 i .. 4 i + 3], the same address in every lane
#pragma unroll
        for (int rr = 0; rr < NR; rr++) s4[rr] = vg[rr * (KP / 4) + i];
        if constexpr (DPL == 2) {
          // v_pk_fma_f32: the lane's two columns advance together (same products, same order
          // per column as the scalar form)
          const f32x4 a = pg[(2 * i) * NL], b = pg[(2 * i + 1) * NL];
#pragma unroll
          for (int rr = 0; rr < NR; rr++) {
            f32x2 o2{out[rr][0], out[rr][1]};
            o2 = __builtin_elementwise_fma(f32x2{a.x, a.y}, f32x2{s4[rr].x, s4[rr].x}, o2);
            o2 = __builtin_elementwise_fma(f32x2{a.z, a.w}, f32x2{s4[rr].y, s4[rr].y}, o2);
            o2 = __builtin_elementwise_fma(f32x2{b.x, b.y}, f32x2{s4[rr].z, s4[rr].z}, o2);
            o2 = __builtin_elementwise_fma(f32x2{b.z, b.w}, f32x2{s4[rr].w, s4[rr].w}, o2);
            out[rr][0] = o2.x;
            out[rr][1] = o2.y;
          }
        } else {
          const f32x4 a = pg[i * NL];
#pragma unroll
          for (int rr = 0; rr < NR; rr++) {
            out[rr][0] = fmaf(a.x, s4[rr].x, out[rr][0]);
            out[rr][0] = fmaf(a.y, s4[rr].y, out[rr][0]);
            out[rr][0] = fmaf(a.z, s4[rr].z, out[rr][0]);
            out[rr][0] = fmaf(a.w, s4[rr].w, out[rr][0]);
          }
        }
      }
#pragma unroll
      for (int rr = 0; rr < NR; rr++) {
#pragma unroll
        for (int j0 = 0; j0 < CAP; j0 += 16) {
          if (n[rr] <= j0) continue;  // (wave-uniform)
          float part[16], dot[16];
#pragma unroll
          for (int j = 0; j < 16; j++) {
            float s = 0.f;
#pragma unroll
            for (int c = 0; c < DPL; c++) s = fmaf(v[rr][j0 + j][c], vec[rr][c], s);
            part[j] = s;
          }
          wave_sum16(part, dot);
#pragma unroll
          for (int j = 0; j < 16; j++) {
            const float w = cj[rr][j0 + j] * dot[j];
            if constexpr (DPL == 2) {
              const f32x2 o2 = __builtin_elementwise_fma(
                  f32x2{w, w}, f32x2{v[rr][j0 + j][0], v[rr][j0 + j][1]}, f32x2{out[rr][0], out[rr][1]});
              out[rr][0] = o2.x;
              out[rr][1] = o2.y;
            } else {
#pragma unroll
              for (int c = 0; c < DPL; c++) out[rr][c] = fmaf(w, v[rr][j0 + j][c], out[rr][c]);
            }
          }
        }
      }
#pragma unroll
      for (int rr = 0; rr < NR; rr++)
#pragma unroll
        for (int c = 0; c < DPL; c++) out[rr][c] = fmaf(reg[rr], vec[rr][c], out[rr][c]);
    };
    auto dotw = [&](const float (&a)[DPL], const float (&b)[DPL]) {
      float s = 0.f;
#pragma unroll
      for (int c = 0; c < DPL; c++) s = fmaf(a[c], b[c], s);
      return wave_sum(act ? s : 0.f);
    };
    float r[NR][DPL], pv[NR][DPL], Ap[NR][DPL];
    if (p.warm_start) {
      matvec(x, Ap);
#pragma unroll
      for (int rr = 0; rr < NR; rr++)
#pragma unroll
        for (int c = 0; c < DPL; c++) r[rr][c] = act ? bvec[rr][c] - Ap[rr][c] : 0.f;
    } else {
#pragma unroll
      for (int rr = 0; rr < NR; rr++)
#pragma unroll
        for (int c = 0; c < DPL; c++) r[rr][c] = act ? bvec[rr][c] : 0.f;
    }
    float r2[NR];
    bool run[NR];  // the row's iteration is still going (wave-uniform)
    bool singular = false;
#pragma unroll
    for (int rr = 0; rr < NR; rr++) {
#pragma unroll
      for (int c = 0; c < DPL; c++) pv[rr][c] = r[rr][c];
      r2[rr] = dotw(r[rr], r[rr]);
      run[rr] = live[rr];
    }
    for (int it = 0; it < p.max_cg_steps; it++) {
      bool any = false;
#pragma unroll
      for (int rr = 0; rr < NR; rr++) {
        if (run[rr] && r2[rr] <= 1e-20f) run[rr] = false;  // hpp:238
        any = any || run[rr];
      }
      if (!any) break;
      matvec(pv, Ap);
#pragma unroll
      for (int rr = 0; rr < NR; rr++) {
        if (!run[rr]) continue;
        const float denom = dotw(pv[rr], Ap[rr]);
        if (!(denom > 0.f) || !__builtin_isfinite(denom)) {  // hpp:250-254
          singular = true;
          run[rr] = false;
          continue;
        }
        const float alpha = r2[rr] / denom;
#pragma unroll
        for (int c = 0; c < DPL; c++) {
          x[rr][c] = fmaf(alpha, pv[rr][c], x[rr][c]);
          r[rr][c] = fmaf(-alpha, Ap[rr][c], r[rr][c]);
        }
        const float r2n = dotw(r[rr], r[rr]);
        if (r2n <= 1e-20f) {  // hpp:258
          run[rr] = false;
          continue;
        }
        const float beta = r2n / r2[rr];  // hpp:261
#pragma unroll
        for (int c = 0; c < DPL; c++) pv[rr][c] = fmaf(beta, pv[rr][c], r[rr][c]);
        r2[rr] = r2n;
      }
    }
    if (singular) {
      if (lane == 0) atomicOr(p.err_flag, 4);
    }
    if (act) {
#pragma unroll
      for (int rr = 0; rr < NR; rr++) {
        if (!live[rr]) continue;
#pragma unroll
        for (int c = 0; c < DPL; c++) xrow[rr][c] = (DPL * dl + c < p.K) ? x[rr][c] : 0.f;
      }
    }
  }
}

}  // namespace ials
}  // namespace irs
