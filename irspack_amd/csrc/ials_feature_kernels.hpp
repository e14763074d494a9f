// Feature-aware iALS, device side of the two feature products (IALSTrainer.hpp:702-708,
// 1134-1171): the prior  features @ W  that enters the per-row solves, and the right-hand side
// features^T (D factor)  of the feature-weight ridge system (D = diag of the per-row
// regularisers).  The F x F ridge solve itself stays on the host (see _ials_core.py).
#pragma once
#include "ials_kernels.hpp"

namespace irs {
namespace ials {

// prior[r, :] = sum_q val_q * W[col_q, :]   (one wave per row, lanes over the latent dims;
// stored entries are added in column order like Eigen's / scipy's CSR product)
__global__ __launch_bounds__(256) void feature_prior_kernel(const int32_t *__restrict__ indptr,
                                                            const int32_t *__restrict__ indices,
                                                            const float *__restrict__ data,
                                                            const float *__restrict__ W,  // [F, KP]
                                                            int64_t n_rows, int KP,
                                                            float *__restrict__ prior) {  // [rows, KP]
  const int lane = threadIdx.x & 63;
  const int64_t row = static_cast<int64_t>(blockIdx.x) * 4 + wave_in_block();
  if (row >= n_rows) return;
  const int k0 = 256 * blockIdx.y + lane;  // blockIdx.y: block of 256 latent dims (K > 256)
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  for (int q = indptr[row]; q < indptr[row + 1]; q++) {
    const float v = data[q];
    const float *w = W + static_cast<size_t>(indices[q]) * KP;
#pragma unroll
    for (int c = 0; c < 4; c++)
      if (k0 + 64 * c < KP) acc[c] = fmaf(v, w[k0 + 64 * c], acc[c]);
  }
#pragma unroll
  for (int c = 0; c < 4; c++)
    if (k0 + 64 * c < KP) prior[row * KP + k0 + 64 * c] = acc[c];
}

// part[c, f, :] = sum over chunk c of the rows r that store feature f of  val * reg_r *
// factor[r, :].  One 256-thread workgroup per (feature, chunk of `chunk` stored rows) of the
// CSR of features^T: the four waves take the rows round-robin, lanes cover the latent dims,
// the four partial sums are added in wave order.  feature_rhs_reduce_kernel then adds the
// chunks in order (reproducible).  Dense features (every row stores every feature) get
// rows / chunk workgroups per feature instead of one.
__global__ __launch_bounds__(256) void feature_rhs_kernel(const int32_t *__restrict__ t_indptr,
                                                          const int32_t *__restrict__ t_indices,
                                                          const float *__restrict__ t_data,
                                                          const float *__restrict__ reg,
                                                          const float *__restrict__ factor,  // [rows, KP]
                                                          int KP, int chunk, int n_feat,
                                                          float *__restrict__ part) {  // [chunks, F, KP]
  __shared__ float sh[4][256];
  const int lane = threadIdx.x & 63, wv = wave_in_block();
  const int f = blockIdx.x, c = blockIdx.y;
  const int kb = 256 * blockIdx.z;  // block of 256 latent dims (K > 256)
  const int qb = t_indptr[f] + c * chunk;
  const int qe = min(qb + chunk, t_indptr[f + 1]);
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  for (int q = qb + wv; q < qe; q += 4) {
    const int r = t_indices[q];
    const float s = t_data[q] * reg[r];
    const float *x = factor + static_cast<size_t>(r) * KP;
#pragma unroll
    for (int cc = 0; cc < 4; cc++)
      if (kb + lane + 64 * cc < KP) acc[cc] = fmaf(s, x[kb + lane + 64 * cc], acc[cc]);
  }
#pragma unroll
  for (int cc = 0; cc < 4; cc++) sh[wv][lane + 64 * cc] = acc[cc];
  __syncthreads();
  float *dst = part + (static_cast<size_t>(c) * n_feat + f) * KP;
  {
    const int k = threadIdx.x;
    if (kb + k < KP) dst[kb + k] = ((sh[0][k] + sh[1][k]) + sh[2][k]) + sh[3][k];
  }
}

__global__ void feature_rhs_reduce_kernel(const float *__restrict__ part, int n_chunks,
                                          int64_t n,  // n_feat * KP
                                          float *__restrict__ rhs) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float s = 0.f;
  for (int c = 0; c < n_chunks; c++) s += part[static_cast<size_t>(c) * n + i];
  rhs[i] = s;
}

}  // namespace ials
}  // namespace irs
