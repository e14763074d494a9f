// Development harness: eig_jacobi_kernel<128> alone - cold start, then warm starts on a Gramian that moves a
// little between calls (what an epoch does) - time, sweeps, ||P Q - Q diag(lam)|| / lam_max, ||Q^T Q - I||.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../irspack_amd/csrc -o eig_jacobi_bench eig_jacobi_bench.hip
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <random>
#include <vector>

#include "common.hpp"
#include "ials_kernels.hpp"
#include "ials_eig_kernels.hpp"

#define CK(x)                                                                          \
  do {                                                                                 \
    hipError_t e_ = (x);                                                               \
    if (e_ != hipSuccess) {                                                            \
      std::fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));   \
      std::exit(1);                                                                    \
    }                                                                                  \
  } while (0)

using namespace irs::ials;

int main(int argc, char **argv) {
  constexpr int KP = 128;
  const int K = argc > 1 ? std::atoi(argv[1]) : 128;
  const int n = 20000;
  std::mt19937_64 rng(3);
  std::normal_distribution<double> nd(0.0, 1.0);
  // factors with a decaying spectrum (like trained factors), Gramian P = V^T V
  std::vector<double> V(static_cast<size_t>(n) * KP, 0.0);
  for (int i = 0; i < n; i++)
    for (int d = 0; d < K; d++) V[static_cast<size_t>(i) * KP + d] = nd(rng) * 0.1 / std::sqrt(1.0 + 0.3 * d) + 0.02;
  auto gramian = [&](std::vector<float> &P) {
    std::vector<double> G(KP * KP, 0.0);
    for (int i = 0; i < n; i++) {
      const double *v = &V[static_cast<size_t>(i) * KP];
      for (int a = 0; a < K; a++)
        for (int b = 0; b <= a; b++) G[a * KP + b] += v[a] * v[b];
    }
    P.assign(KP * KP, 0.f);
    for (int a = 0; a < K; a++)
      for (int b = 0; b <= a; b++) P[a * KP + b] = P[b * KP + a] = static_cast<float>(G[a * KP + b]);
  };
  float *dP, *dQr, *dQc, *dlam, *dstats;
  double *dQd;
  CK(hipMalloc(&dP, KP * KP * 4)); CK(hipMalloc(&dQr, KP * KP * 4)); CK(hipMalloc(&dQc, KP * KP * 4));
  CK(hipMalloc(&dlam, KP * 4)); CK(hipMalloc(&dstats, 16)); CK(hipMalloc(&dQd, KP * KP * 8));
  const size_t lds = static_cast<size_t>(KP) * KP * sizeof(double);
  CK(hipFuncSetAttribute(reinterpret_cast<const void *>(eig_jacobi_kernel<KP>), hipFuncAttributeMaxDynamicSharedMemorySize,
                         static_cast<int>(lds)));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  std::vector<float> P, Q(KP * KP), lam(KP);
  float stats[4];
  for (int call = 0; call < 6; call++) {
    gramian(P);
    CK(hipMemcpy(dP, P.data(), KP * KP * 4, hipMemcpyHostToDevice));
    EigOut o{dQr, dQc, dlam, dstats, dQd, nullptr, 0};
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(eig_jacobi_kernel<KP>, dim3(1), dim3(1024), lds, 0, static_cast<const float *>(dP), K, o, call > 0 ? 1 : 0);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipMemcpy(Q.data(), dQr, KP * KP * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(lam.data(), dlam, KP * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(stats, dstats, 12, hipMemcpyDeviceToHost));
    // residuals in float64 from the float outputs (rows of Q = eigenvectors)
    double res = 0, orth = 0;
    for (int k = 0; k < K; k++) {
      for (int a = 0; a < K; a++) {
        double s = 0;
        for (int b = 0; b < K; b++) s += static_cast<double>(P[a * KP + b]) * Q[k * KP + b];
        res = std::max(res, std::fabs(s - static_cast<double>(lam[k]) * Q[k * KP + a]));
      }
      for (int k2 = 0; k2 <= k; k2++) {
        double s = 0;
        for (int b = 0; b < K; b++) s += static_cast<double>(Q[k * KP + b]) * Q[k2 * KP + b];
        orth = std::max(orth, std::fabs(s - (k == k2 ? 1.0 : 0.0)));
      }
    }
    std::printf("call %d (%s): %.3f ms, %d sweeps, lam %.4g .. %.4g, residual / lam_max %.2e, |Q^T Q - I| %.2e\n", call,
                call ? "warm" : "cold", ms, static_cast<int>(stats[2]), stats[0], stats[1], res / stats[0], orth);
    // the factors move ~3 % (an epoch)
    for (auto &v : V) v *= 1.0 + 0.03 * nd(rng);
  }
  return 0;
}
