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
  static constexpr size_t LDS_BYTES = (static_cast<size_t>(KP) * KP + SHORT_WAVES * KP) * sizeof(float);
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
// P lives in LDS as 16-byte groups [k / G][lane][G rows x DPL columns] so that a lane reads
// its columns of G consecutive rows with one ds_read_b128, and the vector of a product is
// parked in LDS and read back four values at a time (a broadcast read) instead of KP
// v_readlane: ~100 LDS reads + 2 KP / DPL... FMAs per product.
template <int KP, int CAP>
__global__ __launch_bounds__(64 * SHORT_WAVES, CAP <= 16 ? 4 : 3) void ials_cg_short_kernel(
    SolveParams p, const float *__restrict__ P_nat, int first_task, int n_short) {
  using SG = ShortGeo<KP>;
  constexpr int DPL = SG::DPL, NL = SG::NL, G = SG::G;
  static_assert((KP % 16) == 0 && NL <= 64 && (CAP == 16 || CAP == 32), "unsupported shape");
  extern __shared__ __attribute__((aligned(16))) float short_smem[];
  float *Psh = short_smem;                       // KP * KP
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  float *vsh = short_smem + KP * KP + wv * KP;   // this wave's parked vector
  for (int i = threadIdx.x; i < KP * KP; i += 64 * SHORT_WAVES) {
    const int k = i / KP, d = i % KP;
    Psh[((k / G) * NL + d / DPL) * 4 + (k % G) * DPL + d % DPL] = P_nat[i];
  }
  __syncthreads();
  const int dl = lane < NL ? lane : NL - 1;  // idle lanes (KP < 64) shadow the last one, masked
  const bool act = lane < NL;
  const int waves_total = gridDim.x * SHORT_WAVES;
  for (int ti = blockIdx.x * SHORT_WAVES + wv; ti < n_short; ti += waves_total) {
    const Task task = p.tasks[first_task + ti];
    const int n = task.end - task.begin;
    float *xrow = p.target + static_cast<size_t>(task.row) * KP + DPL * dl;
    float x[DPL];
    if (n == 0) {  // hpp:207-210
      if (act) {
#pragma unroll
        for (int c = 0; c < DPL; c++) xrow[c] = 0.f;
      }
      continue;
    }
    const float reg = p.reg[task.row];
    // (index, value) of entry `lane`; entries past the end repeat the last index (a valid
    // address) with weight 0
    const int qe = task.begin + min(lane, n - 1);
    const int my_idx = p.indices[qe];
    const float my_c = lane < n ? p.data[qe] : 0.f;
    float v[CAP][DPL];
    auto gather8 = [&](int j0) {
#pragma unroll
      for (int j = j0; j < j0 + 8; j++) {
        const unsigned idx = static_cast<unsigned>(__builtin_amdgcn_readlane(my_idx, j));
        const float *src = p.other + static_cast<size_t>(idx) * KP + DPL * dl;
        if constexpr (DPL == 2) {
          const float2 t = *reinterpret_cast<const float2 *>(src);
          v[j][0] = t.x;
          v[j][1] = t.y;
        } else {
#pragma unroll
          for (int c = 0; c < DPL; c++) v[j][c] = src[c];
        }
      }
    };
    gather8(0);
#pragma unroll
    for (int j0 = 8; j0 < CAP; j0 += 8) {
      if (n > j0) {
        gather8(j0);
      } else {
#pragma unroll
        for (int j = j0; j < j0 + 8; j++)
#pragma unroll
          for (int c = 0; c < DPL; c++) v[j][c] = 0.f;
      }
    }
#pragma unroll
    for (int c = 0; c < DPL; c++) x[c] = (p.warm_start && act) ? xrow[c] : 0.f;
    float cj[CAP];  // confidences (wave-uniform), 0 past the end
#pragma unroll
    for (int j = 0; j < CAP; j++) cj[j] = readlane_f(my_c, j);
    if (!act) {
#pragma unroll
      for (int j = 0; j < CAP; j++)
#pragma unroll
        for (int c = 0; c < DPL; c++) v[j][c] = 0.f;
    }
    // b = sum (bias + c) v   (hpp:212-219)
    float bvec[DPL];
#pragma unroll
    for (int c = 0; c < DPL; c++) bvec[c] = 0.f;
#pragma unroll
    for (int j = 0; j < CAP; j++) {
      const float w = j < n ? p.bias + cj[j] : 0.f;
#pragma unroll
      for (int c = 0; c < DPL; c++) bvec[c] = fmaf(w, v[j][c], bvec[c]);
    }
    // out = P vec + reg vec + sum c_q (v_q . vec) v_q   (hpp:222-228, 240-247)
    auto matvec = [&](const float (&vec)[DPL], float (&out)[DPL]) {
      if (act) {
#pragma unroll
        for (int c = 0; c < DPL; c++) vsh[DPL * dl + c] = vec[c];
      }
#pragma unroll
      for (int c = 0; c < DPL; c++) out[c] = reg * vec[c];
      const f32x4 *pg = reinterpret_cast<const f32x4 *>(Psh) + dl;  // group (k / G, lane)
      const f32x4 *vg = reinterpret_cast<const f32x4 *>(vsh);
#pragma unroll 8
      for (int i = 0; i < KP / 4; i++) {
        const f32x4 s4 = vg[i];  // vec[4 i .. 4 i + 3], the same address in every lane
        if constexpr (DPL == 2) {
          const f32x4 a = pg[(2 * i) * NL], b = pg[(2 * i + 1) * NL];
          out[0] = fmaf(a.x, s4.x, out[0]);
          out[1] = fmaf(a.y, s4.x, out[1]);
          out[0] = fmaf(a.z, s4.y, out[0]);
          out[1] = fmaf(a.w, s4.y, out[1]);
          out[0] = fmaf(b.x, s4.z, out[0]);
          out[1] = fmaf(b.y, s4.z, out[1]);
          out[0] = fmaf(b.z, s4.w, out[0]);
          out[1] = fmaf(b.w, s4.w, out[1]);
        } else {
          const f32x4 a = pg[i * NL];
          out[0] = fmaf(a.x, s4.x, out[0]);
          out[0] = fmaf(a.y, s4.y, out[0]);
          out[0] = fmaf(a.z, s4.z, out[0]);
          out[0] = fmaf(a.w, s4.w, out[0]);
        }
      }
#pragma unroll
      for (int j0 = 0; j0 < CAP; j0 += 16) {
        float part[16], dot[16];
#pragma unroll
        for (int j = 0; j < 16; j++) {
          float s = 0.f;
#pragma unroll
          for (int c = 0; c < DPL; c++) s = fmaf(v[j0 + j][c], vec[c], s);
          part[j] = s;
        }
        wave_sum16(part, dot);
#pragma unroll
        for (int j = 0; j < 16; j++) {
          const float w = cj[j0 + j] * dot[j];
#pragma unroll
          for (int c = 0; c < DPL; c++) out[c] = fmaf(w, v[j0 + j][c], out[c]);
        }
      }
    };
    auto dotw = [&](const float (&a)[DPL], const float (&b)[DPL]) {
      float s = 0.f;
#pragma unroll
      for (int c = 0; c < DPL; c++) s = fmaf(a[c], b[c], s);
      return wave_sum(act ? s : 0.f);
    };
    float r[DPL], pv[DPL], Ap[DPL];
    if (p.warm_start) {
      matvec(x, Ap);
#pragma unroll
      for (int c = 0; c < DPL; c++) r[c] = act ? bvec[c] - Ap[c] : 0.f;
    } else {
#pragma unroll
      for (int c = 0; c < DPL; c++) r[c] = act ? bvec[c] : 0.f;
    }
#pragma unroll
    for (int c = 0; c < DPL; c++) pv[c] = r[c];
    bool singular = false;
    float r2 = dotw(r, r);
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
    if (singular) {
      if (lane == 0) atomicOr(p.err_flag, 4);
    }
    if (act) {
#pragma unroll
      for (int c = 0; c < DPL; c++) xrow[c] = (DPL * dl + c < p.K) ? x[c] : 0.f;
    }
  }
}

}  // namespace ials
}  // namespace irs
